// Arithmetic shared by the publish / subscribe kernels (pk_k_step_pub.hip) and the per-scan candidate lists (k_candidates,
// pk_k_assoc.hip): the KEY of a (landmark, blob) pair -- -2 log of the reference's match probability (prkt_core_v2.py:439-455) up
// to rounding -- and the bound that says when a key is certainly beyond the float64 underflow edge (probability 0, :369).
#pragma once
#include "pk_math.hpp"

namespace pk {

// 1 / x to a few ulp (v_rcp_f64 and two Newton steps: the full division sequence is twice as long; the keys are compared
// with each other only, all made the same way)
__device__ __forceinline__ double pub_recip(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
// log(x) for the keys: x a positive normal number (the product of two determinants inside 1e-20 ... 1e60 each -- anything
// else is flagged), absolute error below 1e-12: keys are only compared with each other and with thresholds a unit wide, two
// keys within 1e-7 relative are handed to the exact route anyway, and identical landmarks still give identical keys (same
// function, same inputs).  log_few_ulp's form -- e ln 2 + 2 atanh((m - 1) / (m + 1)) -- with eight terms instead of eleven,
// a Newton reciprocal instead of the division and no special cases: 35 instructions against 65.
__device__ __forceinline__ double pub_log(double x) {
  int e = 0;
  double m = frexp(x, &e);  // m in [0.5, 1)
  const bool low = m < 0.70710678118654752440;
  m = low ? m * 2.0 : m;
  e = low ? e - 1 : e;
  const double s = (m - 1.0) * pub_recip(m + 1.0);
  const double z = s * s;  // <= 0.0295
  double p = 1.0 / 15.0;  // (fma_k: the coefficients through scalar registers, pk_math.hpp)
  p = fma_k(p, z, 1.0 / 13.0);
  p = fma_k(p, z, 1.0 / 11.0);
  p = fma_k(p, z, 1.0 / 9.0);
  p = fma_k(p, z, 1.0 / 7.0);
  p = fma_k(p, z, 1.0 / 5.0);
  p = fma_k(p, z, 1.0 / 3.0);
  const double lm = 2.0 * s + 2.0 * s * (z * p);  // 2 atanh(s); the series' remainder: 2 s z^8 / 17 < 3e-14
  return (double)e * 0.69314718055994530942 + lm;
}
// The lower bound of a landmark's keys that pub_keysN calls "far": key >= fk + fi |colour difference|^2 for a colour block that
// is certainly positive definite (fi > 0; fi = 0: no bound) -- fk = 5 log 2 pi + log(det2 det3), the key's constant term, and
// fi = 1 / (largest absolute row sum of the colour block) <= 1 / lmax(C) (Gershgorin; d' C^-1 d >= |d|^2 / lmax, the position
// term >= 0).  The same expressions as in pub_keysN -- the same values.
__device__ __forceinline__ void pub_far_bound(const Landmark<double>& lm, double& fk, double& fi) {
  const double det2 = lm.pxx * lm.pyy - lm.pxy * lm.pxy;
  double det3;
  const Sym3<double> adj3 = sym3_adjugate(Sym3<double>{lm.crr, lm.crg, lm.crb, lm.cgg, lm.cgb, lm.cbb}, det3);
  const bool sane = det2 > 1e-20 && det2 < 1e60 && det3 > 1e-20 && det3 < 1e60;  // NaN: false
  const bool pd3 = sane && lm.crr > 0.0 && adj3.f > 0.0 && lm.pxx > 0.0;
  const double rowmax = fmax(fmax(lm.crr + (fabs(lm.crg) + fabs(lm.crb)), lm.cgg + (fabs(lm.crg) + fabs(lm.cgb))), lm.cbb + (fabs(lm.crb) + fabs(lm.cgb)));
  fk = 5.0 * Consts<double>::log_two_pi + pub_log(det2 * det3);  // (pub_keysN's kbase, whatever the blocks are like)
  fi = pd3 ? pub_recip(rowmax) : 0.0;                            // (0: no bound)
}
// A key beyond this is a probability of exactly 0 in the reference's arithmetic: pr = fl(fl(bp cp) / 250000) with
// bp cp = 250000 exp(-key / 2) rounds to 0 from key > 1490.27 on (exp(-745.13) = 2^-1075, half the smallest subnormal).
constexpr double kPubFarKey = 1492.0;

// ---- look-alikes taken off the candidate lists once per scan (round 5) -----------------------------------------------------
// A blob inside a landmark's gates whose KEY is certainly beyond the underflow edge has probability 0: it is no contender for
// anything, yet it cost every particle a gate test and -- passing -- a verdict round, and a round costs a whole wave its arithmetic.
// Once a landmark has been seen a few times its colour block is tight (Qt = 0.1 I: C_n = 1 / (4 + 10 n) from 0.25 I) and nearly every
// look-alike inside the colour gate (radius^2 300) is such a blob.  The colour block's update does not depend on the particle
// (C' = Q (C + Q)^-1 C, prkt_core_v2.py:916-930), so all particles that saw the landmark equally often hold the SAME block: the
// verdict "far for everybody" can be reached ONCE per scan, by k_candidates on the reference particle, with margins:
//   bound of the reference (kb_ref, ib_ref) -> Kb = kb_ref - kFarKeySlack, Ib = ib_ref / kFarVarFactor, both rounded DOWN to float;
//   a candidate blob is FAR when Kb + dmin^2 Ib > kPubFarKey + 0.5, dmin_c = max(0, |z_c - ref_c| - kCandColour) per channel:
//   for every particle inside the candidate margins (|mean_c - ref_c| <= kCandColour, checked per landmark anyway) whose own
//   bound satisfies fk >= Kb and fi >= Ib, its key >= fk + |d|^2 fi >= Kb + dmin^2 Ib > kPubFarKey: probability 0.
// Far candidates leave the landmark's list (and never enter the blobs' inverse lists: no publish entry, no contested blob); they
// are kept in a FAR list beside it.  A particle's landmark whose own bound is NOT that good (it missed most of the updates the
// reference made: a colour block kFarVarFactor times wider -- with C_n = 1 / (4 + 10 n) more than half of them --, or determinants
// e^kFarKeySlack times smaller) walks the far list as
// well, in a rare wave-uniform branch: a far-listed blob that passes its gates and is not certainly far by its OWN bound sends the
// particle to the fall-back kernels (exact as ever).
#ifndef PK_FAR_VAR_FACTOR  // (tuning builds only)
#define PK_FAR_VAR_FACTOR 2.0
#endif
// (kFarVarFactor 4 / 2 / 1.5 at 100 000 x 2 000, two interleaved repetitions on one box: 8.18 / 8.04 / 8.02 ms per launch, nobody
// flagged in any of them -- profiles/r05/ab_far_var_factor.log)
constexpr double kFarKeySlack = 8.0;
constexpr double kFarVarFactor = PK_FAR_VAR_FACTOR;
__device__ __forceinline__ float pub_round_down_to_float(double x) {
  float f = (float)x;
  if ((double)f > x) f = __uint_as_float(f > 0.f ? __float_as_uint(f) - 1u : f < 0.f ? __float_as_uint(f) + 1u : 0x80000001u);
  return f;
}

}  // namespace pk
