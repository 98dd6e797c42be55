// Register-resident measurement model of the FastSLAM landmark EKF (gfx950).
//
// Everything here is a __device__ inline on scalars: the linear algebra of the
// path is 2x2 (landmark xy) and 3x3 (landmark rgb), far too small for MFMA, so
// it lives in VGPRs and is written out by hand.  Each function cites the lines
// of /root/reference/src/prkt_core_v2.py it stands for, including the quirks a
// textbook EKF would not have.
#pragma once
#include <hip/hip_runtime.h>

namespace pk {

template <typename T>
struct Consts;
template <>
struct Consts<double> {
  static constexpr double two_pi = 6.283185307179586476925286766559;
  static constexpr double half_pi = 1.5707963267948966192313216916398;
  static constexpr double log_two_pi = 1.8378770664093454835606594728112;
  static constexpr double log_no_match = -2.3025850929940456840179914546844;  // log(0.1), :857
};

// One landmark EKF in the compact layout: mean (x, y, r, g, b), the symmetric
// 2x2 position block and the symmetric 3x3 colour block of its covariance, and
// Feature.update_count (prkt_core_v2.py:881-895).
template <typename T>
struct Landmark {
  T mx, my, mr, mg, mb;
  T pxx, pxy, pyy;
  T crr, crg, crb, cgg, cgb, cbb;
  int count;  // update_count | kPotentialBit
};

// A POTENTIAL feature (the reference's negative ids, prkt_core_v2.py:109-118): matched and updated like any landmark, but
// the particle's weight takes the unmatched-blob factor instead of the importance factor, until its update count passes
// 5 and it joins the full feature set (:113-117).  The flag rides in the count word (PK_LANDMARK_POTENTIAL in the header):
// every kernel only ever adds 2 to that word, so it travels through all of them unchanged.
constexpr int kPotentialBit = 0x40000000;
__device__ __forceinline__ void count_update(int& count) {
  count += 2;  // :914 and :930
  if ((count & kPotentialBit) && (count & ~kPotentialBit) > 5) count &= ~kPotentialBit;  // :113-117 promotion
}

// Measurement noise Qt (prkt_core_v2.py:50-53) in the same block structure.
template <typename T>
struct Noise {
  T q00;
  T rr, rg, rb, gg, gb, bb;
  int diag;  // colour block diagonal (rg = rb = gb = 0, the reference's Qt = 0.1 I, :50-53): short forms below
};
template <typename T>
__host__ __device__ inline Noise<T> make_noise(T q00, T rr, T rg, T rb, T gg, T gb, T bb) {
  return Noise<T>{q00, rr, rg, rb, gg, gb, bb, (rg == T(0) && rb == T(0) && gb == T(0)) ? 1 : 0};
}

// One observed blob (matrix.blob_to_matrix, matrix.py:35-39).
template <typename T>
struct BlobT {
  T bearing, r, g, b;
};

// Inverse (6 unique entries) and determinant of a symmetric 3x3.
template <typename T>
struct Sym3 {
  T a, b, c, d, e, f;  // [[a,b,c],[b,d,e],[c,e,f]]
};

// Adjugate (the symmetric cofactor matrix) and determinant of a symmetric 3x3: inverse = adj / det.
template <typename T>
__device__ __forceinline__ Sym3<T> sym3_adjugate(const Sym3<T>& m, T& det) {
  T c00 = m.d * m.f - m.e * m.e;
  T c01 = m.c * m.e - m.b * m.f;
  T c02 = m.b * m.e - m.c * m.d;
  T c11 = m.a * m.f - m.c * m.c;
  T c12 = m.b * m.c - m.a * m.e;
  T c22 = m.a * m.d - m.b * m.b;
  det = m.a * c00 + m.b * c01 + m.c * c02;
  return Sym3<T>{c00, c01, c02, c11, c12, c22};
}

template <typename T>
__device__ __forceinline__ Sym3<T> sym3_inverse(const Sym3<T>& m, T& det) {
  const Sym3<T> c = sym3_adjugate(m, det);
  T inv = T(1) / det;
  return Sym3<T>{c.a * inv, c.b * inv, c.c * inv, c.d * inv, c.e * inv, c.f * inv};
}

template <typename T>
__device__ __forceinline__ T sym3_quad(const Sym3<T>& m, T x, T y, T z) {
  return m.a * x * x + m.d * y * y + m.f * z * z + T(2) * (m.b * x * y + m.c * x * z + m.e * y * z);
}

// Heading after heading_to_quaternion -> quaternion_to_heading (utils.py:8-35):
// wrapped to (-pi, pi].
template <typename T>
__device__ __forceinline__ T wrap_heading(T h) {
  T s, c;
  sincos(h, &s, &c);
  return atan2(s, c);
}

// a * b + K, one rounding, with the CONSTANT K in a scalar register pair (VOP3 v_fma_f64 takes one scalar operand).  Left to
// itself the compiler writes a Horner step `p = p * z + K` as v_fmac_f64 (VOP2: the addend is the destination), which wants
// K in vector registers first: two v_mov_b32 of 32-bit literals and the multiply-add -- three VALU issue slots (twelve
// cycles of a wave64 SIMD) where one does, in kernels whose gates / keys phases are float64-issue-bound.
// The two s_mov_b32 that load K issue on the scalar unit, beside the other wave's vector instructions.  Same value as the
// contracted expression (one fused multiply-add either way).
__device__ __forceinline__ double fma_k(double a, double b, double k) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PK_NO_SCALAR_FMA)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
  return r;
#else
  return __builtin_fma(a, b, k);
#endif
}

// atan2 for the EXPECTED BEARING of a landmark (prkt_core_v2.py:408, :473, :871: math.atan2(fy - sy, fx - sx)), within
// 2 ulp, in ~47 float64 instructions against the library's ~100 + 40 constant moves (every particle x landmark pays
// one, in the issue-bound gates phase of the one-pass kernels).  ONE division for both range reductions:
//   t = min(|x|, |y|) / max(|x|, |y|) in [0, 1];  beyond tan(pi / 8):  atan t = pi / 4 + atan((t - 1) / (t + 1)), and
//   (t - 1) / (t + 1) = (mn - mx) / (mn + mx) -- the same quotient with another numerator and denominator --
// so r = num / den lies in [-tan(pi / 8), tan(pi / 8)] and atan r = r + r s Q(s), s = r^2, Q of degree 10 (coefficients:
// scripts/fit_atan_poly.py, polynomial error 9e-18 relative, 0.64 ulp evaluated in float64).  The quotient by v_rcp_f64, two
// Newton steps and one residual correction (< 1 ulp).  Then the octant: pi / 2 - a where |y| > |x|, pi - a where x < 0
// (hi + lo constants), the sign of y.  atan2(0, 0) = 0 like the library's for +0; NaN in gives NaN out (v_max / v_min drop
// NaNs: restored by hand -- the kernels rely on a NaN state failing every comparison).  Infinite arguments are not handled (a
// state is finite or NaN).  EVERY observe kernel takes its expected bearings here, so the routes stay bit-identical to each
// other; the motion model's heading wrap and summary keep the library's atan2 (O(P) work, utils.py:8-35).
__device__ __forceinline__ double pk_atan2(double y, double x) {
#if defined(PK_LIBM_ATAN2)  // diagnostic build: the library's
  return atan2(y, x);
#else
  const double ax = fabs(x), ay = fabs(y);
  const double mx = fmax(ax, ay), mn = fmin(ax, ay);
  const bool upper = mn > 0.41421356237309503 * mx;  // tan(pi / 8)
  const double num = upper ? mn - mx : mn;
  double den = upper ? mn + mx : mx;
  den = mx == 0.0 ? 1.0 : den;  // atan2(0, 0) = 0
  double rc = __builtin_amdgcn_rcp(den);
  rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
  rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
  double r = num * rc;
  r = __builtin_fma(__builtin_fma(-den, r, num), rc, r);
  const double s = r * r;
  double q = -0x1.3a25edd998780p-6;
  q = fma_k(q, s, 0x1.415d4935ee1ffp-5);
  q = fma_k(q, s, -0x1.a09769c825878p-5);
  q = fma_k(q, s, 0x1.dfe5c951ef847p-5);
  q = fma_k(q, s, -0x1.10fa6e9b77db9p-4);
  q = fma_k(q, s, 0x1.3b126231f5ee5p-4);
  q = fma_k(q, s, -0x1.745d0b1c2ae99p-4);
  q = fma_k(q, s, 0x1.c71c7184c78c8p-4);
  q = fma_k(q, s, -0x1.2492492434d72p-3);
  q = fma_k(q, s, 0x1.9999999999310p-3);
  q = fma_k(q, s, -0x1.5555555555555p-2);
  double a = __builtin_fma(r * s, q, r);  // atan r
  a = upper ? 0x1.921fb54442d18p-1 + (a + 0x1.1a62633145c07p-55) : a;        // + pi / 4
  a = ay > ax ? (0x1.921fb54442d18p+0 - a) + 0x1.1a62633145c07p-54 : a;      // pi / 2 - a
  a = x < 0.0 ? (0x1.921fb54442d18p+1 - a) + 0x1.1a62633145c07p-53 : a;      // pi - a
  a += 0.0 * (x + y);  // NaN in, NaN out
  return __builtin_copysign(a, y);
#endif
}

// closest_point, prkt_core_v2.py:496-522.  (ux, uy) = unit((cos b, sin b, 0)),
// utils.py:68-76, precomputed per blob.
template <typename T>
__device__ __forceinline__ void closest_point(T fx, T fy, T sx, T sy, T ux, T uy, T& nx, T& ny) {
  T ox = fx - sx, oy = fy - sy;
  T mag = ox * ux + oy * uy;  // dot_product, utils.py:37-43
  if (mag < T(0)) {           // :515-516 behind the ray: the robot position itself
    nx = sx;
    ny = sy;
  } else {
    nx = sx + ux * mag;
    ny = sy + uy * mag;
  }
}

// prob_position_match, prkt_core_v2.py:457-494 (scipy multivariate_normal.pdf, k = 2,
// in closed form).  pse = atan2(fy - sy, fx - sx).
template <typename T>
__device__ __forceinline__ T prob_position_match(const Landmark<T>& f, T sx, T sy, T pse, T bearing,
                                                 T ux, T uy, T* near_xy = nullptr) {
  T nx, ny;
  closest_point(f.mx, f.my, sx, sy, ux, uy, nx, ny);
  if (near_xy) {
    near_xy[0] = nx;
    near_xy[1] = ny;
  }
  if (fabs(pse - bearing) > Consts<T>::half_pi) return T(0);  // :473-475, frames mixed as in the reference
  T ex = nx - f.mx, ey = ny - f.my;
  T det = f.pxx * f.pyy - f.pxy * f.pxy;
  T maha = (f.pyy * ex * ex - T(2) * f.pxy * ex * ey + f.pxx * ey * ey) / det;
  return exp(T(-0.5) * (T(2) * Consts<T>::log_two_pi + log(det) + maha));
}

// prob_color_match, prkt_core_v2.py:524-544 (scipy pdf, k = 3).
template <typename T>
__device__ __forceinline__ T prob_color_match(const Landmark<T>& f, T r, T g, T b) {
  T det;
  Sym3<T> inv = sym3_inverse(Sym3<T>{f.crr, f.crg, f.crb, f.cgg, f.cgb, f.cbb}, det);
  T maha = sym3_quad(inv, r - f.mr, g - f.mg, b - f.mb);
  return exp(T(-0.5) * (T(3) * Consts<T>::log_two_pi + log(det) + maha));
}

// The two gates of probability_of_match (prkt_core_v2.py:433, :441) need only the mean.
template <typename T>
__device__ __forceinline__ T color_distance2(T mr, T mg, T mb, T r, T g, T b) {
  T dr = r - mr, dg = g - mg, db = b - mb;
  return dr * dr + dg * dg + db * db;  // :425-427
}

// probability_of_match, prkt_core_v2.py:383-455.
template <typename T>
__device__ __forceinline__ T probability_of_match(const Landmark<T>& f, T sx, T sy, T heading,
                                                  const BlobT<T>& z, T ux, T uy) {
  T pse = pk_atan2(f.my - sy, f.mx - sx);
  T delb = z.bearing - (pse - heading);  // :408-415, never wrapped (:416-423 commented out)
  if (fabs(delb) > T(0.5)) return T(0);  // :433
  T cd = color_distance2(f.mr, f.mg, f.mb, z.r, z.g, z.b);
  if (fabs(cd) > T(300)) return T(0);  // :441
  T bp = T(500) * prob_position_match(f, sx, sy, pse, z.bearing, ux, uy);  // :439
  T cp = T(500) * prob_color_match(f, z.r, z.g, z.b);                     // :446
  return bp * cp / T(250000);                                            // :455
}

// Natural logarithm accurate to 2 ulp (checked on the host against logl over 2e7 arguments from
// 1e-304 to 1e304 and around 1): log(x) = e ln 2 + 2 atanh(s), s = (m - 1) / (m + 1), with x = m 2^e
// and m in [sqrt(1/2), sqrt(2)) so that s^2 <= 0.0295 and eleven terms of the atanh series leave
// < 1e-18.  About 45 float64 instructions against ~95 for the library log.  Used for the importance
// factor (:844-849) and for the probabilities of contested pairs (:439, :446) -- quantities whose
// tolerance is 1e-5 relative (north star) and which the parity tests hold to 1e-9; a tie between
// identical landmarks stays a tie (same function, same inputs).  Anything that is not a positive
// normal number takes the library path.
__device__ __forceinline__ double log_few_ulp(double x) {
  if (!(x > 2.3e-308 && x < 1.7e308)) return log(x);
  int e = 0;
  double m = frexp(x, &e);  // m in [0.5, 1)
  if (m < 0.70710678118654752440) {
    m *= 2.0;
    e -= 1;
  }
  const double s = (m - 1.0) / (m + 1.0);
  const double z = s * s;
  double p = 1.0 / 21.0;
  p = fma_k(p, z, 1.0 / 19.0);
  p = fma_k(p, z, 1.0 / 17.0);
  p = fma_k(p, z, 1.0 / 15.0);
  p = fma_k(p, z, 1.0 / 13.0);
  p = fma_k(p, z, 1.0 / 11.0);
  p = fma_k(p, z, 1.0 / 9.0);
  p = fma_k(p, z, 1.0 / 7.0);
  p = fma_k(p, z, 1.0 / 5.0);
  p = fma_k(p, z, 1.0 / 3.0);
  const double lm = 2.0 * s + 2.0 * s * (z * p);  // 2 atanh(s)
  const double ed = (double)e;
  return ed * 0.69314718055994528623 + (ed * 2.3190468138462995584e-17 + lm);  // ln 2 = hi + lo
}

// probability_of_match (:439-455) from the determinants and the Mahalanobis terms' numerators
// (maha = num / det: e' adj(P) e and d' adj(C) d -- the divisions are made here, for the few pairs
// whose probability VALUE is needed, not in the code every lane runs)
__device__ __forceinline__ double pr_from_parts(double det2, double det3, double num2, double num3) {
  const double maha2 = num2 / det2, maha3 = num3 / det3;
  const double bp = 500.0 * exp(-0.5 * (2.0 * Consts<double>::log_two_pi + log_few_ulp(det2) + maha2));  // :439
  const double cp = 500.0 * exp(-0.5 * (3.0 * Consts<double>::log_two_pi + log_few_ulp(det3) + maha3));  // :446
  return bp * cp / 250000.0;                                                                              // :455
}

// Intermediate quantities of one EKF update, for the probe entry point.
template <typename T>
struct EkfAux {
  T zhat0, h0, h1;
  T q00;
  Sym3<T> qc;     // Q[1:,1:]
  T k0, k1;       // K[0:2, 0]
  T kc[9];        // K[2:5, 1:4] row-major
};

// One blob applied to one landmark: generate_measurement :859-877,
// measurement_jacobian :748-802, measurement_covariance :804-819, inverse matrix.py:11,
// kalman_gain :821-833, Feature.update_mean :897-914, Feature.update_covar :916-930,
// importance_factor :835-849.  Returns log(importance factor); the state is updated in
// place unless the landmark is immutable (:909, :926).
//
// With Sigma = Pxy (+) C and Qt = q00 (+) Qc the 4x4 / 5x4 algebra of the reference
// factors exactly into a scalar and a 3x3 problem:
//   Q   = [h' Pxy h + q00] (+) [C + Qc]
//   K   = [Pxy h / Q00]    (+) [C (C + Qc)^-1]
//   Sigma' = [Pxy - (Pxy h)(Pxy h)'/Q00] (+) [C - C (C + Qc)^-1 C]
// With a diagonal Qc the colour block is written C' = Qc (C + Qc)^-1 C (the same matrix:
// I - C (C + Qc)^-1 = Qc (C + Qc)^-1), six products instead of eighteen and without the
// subtraction; the Mahalanobis term is d'v with v = (C + Qc)^-1 d in both forms.
template <typename T>
__device__ __forceinline__ T ekf_update(Landmark<T>& f, T sx, T sy, const BlobT<T>& z,
                                        const Noise<T>& qt, bool immutable,
                                        EkfAux<T>* aux = nullptr, const T* zhat0_known = nullptr, T* fro_prod = nullptr) {
  T dx = f.mx - sx, dy = f.my - sy;
  // :871 world frame: the heading is NOT subtracted here.  A caller that already holds
  // atan2(dy, dx) for this very state passes it in (one float64 atan2 saved).
  T zhat0 = zhat0_known ? *zhat0_known : pk_atan2(dy, dx);
  T q = dx * dx + dy * dy;  // :785
  T h0, h1;                 // :789/:795 -- (dy/q, dx/q): the reference's signs, not the textbook's
  if (q == T(0)) {          // ZeroDivisionError branch :790,:796
    h0 = T(0);
    h1 = T(0);
  } else {
    T iq = T(1) / q;
    h0 = dy * iq;
    h1 = dx * iq;
  }
  T a0 = f.pxx * h0 + f.pxy * h1;  // Pxy h
  T a1 = f.pxy * h0 + f.pyy * h1;
  T q00 = h0 * a0 + h1 * a1 + qt.q00;  // :817-818
  Sym3<T> qc{f.crr + qt.rr, f.crg + qt.rg, f.crb + qt.rb, f.cgg + qt.gg, f.cgb + qt.gb, f.cbb + qt.bb};
  T detc;
  Sym3<T> qci = sym3_inverse(qc, detc);  // matrix.py:11-12 on the 3x3 block
  T iq00 = T(1) / q00;

  T d0 = z.bearing - zhat0;  // :846/:911 innovation, not wrapped
  T d1 = z.r - f.mr, d2 = z.g - f.mg, d3 = z.b - f.mb;

  // importance_factor :844-849: (2 pi ||Q||_F)^-1/2 exp(-1/2 d' Q^-1 d) with the
  // Frobenius norm of Q (matrix.py:31-33), not its determinant.
  T fro2 = q00 * q00 + qc.a * qc.a + qc.d * qc.d + qc.f * qc.f +
           T(2) * (qc.b * qc.b + qc.c * qc.c + qc.e * qc.e);
  // v = (C + Qc)^-1 d_c ;  M = (C + Qc)^-1 C  (3x3, rows r,g,b)
  T v0 = qci.a * d1 + qci.b * d2 + qci.c * d3;
  T v1 = qci.b * d1 + qci.d * d2 + qci.e * d3;
  T v2 = qci.c * d1 + qci.e * d2 + qci.f * d3;
  T maha = d0 * d0 * iq00 + (d1 * v0 + d2 * v1 + d3 * v2);
  // fro_prod (the one-pass kernels): the caller takes ONE logarithm, of the product of the norms of all the updates of its lane,
  // -1/4 log(prod) -- the logarithm is 45 of an update's 300 instructions; here the factor is multiplied in and the term left out
  T logw;
  if (fro_prod) {
    logw = T(-0.5) * Consts<T>::log_two_pi - T(0.5) * maha;
    if (f.count & kPotentialBit)
      logw = (T)Consts<double>::log_no_match;  // :111-112 "update as if the feature not seen"
    else
      *fro_prod *= fro2;
  } else {
    logw = T(-0.5) * (Consts<T>::log_two_pi + T(0.5) * log_few_ulp(fro2)) - T(0.5) * maha;
    if (f.count & kPotentialBit) logw = (T)Consts<double>::log_no_match;  // :111-112 "update as if the feature not seen"
  }
  T k0 = a0 * iq00, k1 = a1 * iq00;

  if (aux) {
    aux->zhat0 = zhat0;
    aux->h0 = h0;
    aux->h1 = h1;
    aux->q00 = q00;
    aux->qc = qc;
    aux->k0 = k0;
    aux->k1 = k1;
    // Kc = C Qci
    const T C[3][3] = {{f.crr, f.crg, f.crb}, {f.crg, f.cgg, f.cgb}, {f.crb, f.cgb, f.cbb}};
    const T I[3][3] = {{qci.a, qci.b, qci.c}, {qci.b, qci.d, qci.e}, {qci.c, qci.e, qci.f}};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) aux->kc[i * 3 + j] = C[i][0] * I[0][j] + C[i][1] * I[1][j] + C[i][2] * I[2][j];
  }

  if (!immutable) {
    // update_mean :912-913: mu += K d
    f.mx += k0 * d0;
    f.my += k1 * d0;
    T nr = f.mr + (f.crr * v0 + f.crg * v1 + f.crb * v2);
    T ng = f.mg + (f.crg * v0 + f.cgg * v1 + f.cgb * v2);
    T nb = f.mb + (f.crb * v0 + f.cgb * v1 + f.cbb * v2);
    // update_covar :928-929: Sigma' = (I - K H) Sigma
    T nxx = f.pxx - k0 * a0;
    T nxy = f.pxy - T(0.5) * (k0 * a1 + k1 * a0);
    T nyy = f.pyy - k1 * a1;
    T nrr, nrg, nrb, ngg, ngb, nbb;
    // M = Qci C (rows r, g, b)
    T m00 = qci.a * f.crr + qci.b * f.crg + qci.c * f.crb;
    T m01 = qci.a * f.crg + qci.b * f.cgg + qci.c * f.cgb;
    T m02 = qci.a * f.crb + qci.b * f.cgb + qci.c * f.cbb;
    T m11 = qci.b * f.crg + qci.d * f.cgg + qci.e * f.cgb;
    T m12 = qci.b * f.crb + qci.d * f.cgb + qci.e * f.cbb;
    T m22 = qci.c * f.crb + qci.e * f.cgb + qci.f * f.cbb;
    if (qt.diag) {  // uniform.  C' = Qc M: the upper triangle of M is all it takes
      nrr = qt.rr * m00;
      nrg = qt.rr * m01;
      nrb = qt.rr * m02;
      ngg = qt.gg * m11;
      ngb = qt.gg * m12;
      nbb = qt.bb * m22;
    } else {  // C' = C - C M (symmetric)
      T m10 = qci.b * f.crr + qci.d * f.crg + qci.e * f.crb;
      T m20 = qci.c * f.crr + qci.e * f.crg + qci.f * f.crb;
      T m21 = qci.c * f.crg + qci.e * f.cgg + qci.f * f.cgb;
      nrr = f.crr - (f.crr * m00 + f.crg * m10 + f.crb * m20);
      nrg = f.crg - (f.crr * m01 + f.crg * m11 + f.crb * m21);
      nrb = f.crb - (f.crr * m02 + f.crg * m12 + f.crb * m22);
      ngg = f.cgg - (f.crg * m01 + f.cgg * m11 + f.cgb * m21);
      ngb = f.cgb - (f.crg * m02 + f.cgg * m12 + f.cgb * m22);
      nbb = f.cbb - (f.crb * m02 + f.cgb * m12 + f.cbb * m22);
    }
    f.mr = nr;
    f.mg = ng;
    f.mb = nb;
    f.pxx = nxx;
    f.pxy = nxy;
    f.pyy = nyy;
    f.crr = nrr;
    f.crg = nrg;
    f.crb = nrb;
    f.cgg = ngg;
    f.cgb = ngb;
    f.cbb = nbb;
    count_update(f.count);
  }
  return logw;
}

// motion_model, prkt_core_v2.py:168-208, on one pose.  n0..n2 are the three draws
// normal(0, sigma, 1) already scaled (:185, :190, :193).
template <typename T>
__device__ __forceinline__ void motion_model(T& x, T& y, T& h, T v, T w, T dt, T n0, T n1, T n2) {
  T dheading = w * dt;         // :183
  T ds = v * dt + n0;          // :186
  T h1 = h + dheading / T(2) + n1;   // :191
  T h2 = h1 + dheading / T(2) + n2;  // :194
  T s, c;
  sincos(h1, &s, &c);
  x += ds * c;  // :198,:203
  y += ds * s;  // :199,:204
  h = wrap_heading(h2);  // :206 quaternion round trip
}

}  // namespace pk
