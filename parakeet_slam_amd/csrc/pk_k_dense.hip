// The general dense path: landmark covariances that couple position and colour (full 5x5), any 4x4 Qt.
//
// Hand-written gfx950 (CDNA4, wave64) kernels of the FastSLAM particle update; see DESIGN.md section 4.
// The reference takes any Feature(mean, covar) (prkt_core_v2.py:882-895) and updates it with dense 4x4 / 5x4 / 5x5
// algebra (:804-833, :897-930, matrix.py:11-12); from block-diagonal inputs that algebra stays block diagonal, which
// is what the compact 14-row layout and every fast kernel rely on.  Inputs WITH xy-rgb coupling take this path
// instead: 30 rows per landmark (pk_layout.hpp), the reference's formulas entry by entry -- H Sigma H' + Qt, a 4x4
// inverse, K = Sigma H' Q^-1, Sigma' = (I - K H) Sigma kept unsymmetrised like the reference keeps it -- one
// workgroup per particle, brute-force maximum-likelihood association in the same kernel.  Correct, not fast:
// 480 B per particle.landmark and O(B L) gate tests per particle.
#include "pk_device.hpp"

namespace pk {

struct DenseLm {
  double m[5];
  double S[25];
  int count;
};

__device__ __forceinline__ DenseLm dense_load(const double* f, const int* cnt, int Lp, int l) {
  DenseLm d;
#pragma unroll
  for (int i = 0; i < 5; ++i) d.m[i] = f[(size_t)i * Lp + l];
#pragma unroll
  for (int i = 0; i < 25; ++i) d.S[i] = f[(size_t)(5 + i) * Lp + l];
  d.count = cnt[l];
  return d;
}
__device__ __forceinline__ void dense_store(double* f, int* cnt, int Lp, int l, const DenseLm& d) {
#pragma unroll
  for (int i = 0; i < 5; ++i) f[(size_t)i * Lp + l] = d.m[i];
#pragma unroll
  for (int i = 0; i < 25; ++i) f[(size_t)(5 + i) * Lp + l] = d.S[i];
  cnt[l] = d.count;
}

// probability_of_match (:383-455) reads only the two diagonal blocks, covar[0:2, 0:2] and covar[2:, 2:] (:489, :543);
// scipy's pdf takes the lower triangle of each.
__device__ __forceinline__ Landmark<double> dense_blocks(const DenseLm& d) {
  return Landmark<double>{d.m[0], d.m[1], d.m[2], d.m[3], d.m[4], d.S[0], d.S[5], d.S[6],
                          d.S[12], d.S[17], d.S[22], d.S[18], d.S[23], d.S[24], d.count};
}

// inverse of a general 4x4 (matrix.py:11-12; numpy.linalg.inv): cofactors over 2x2 sub-determinants
__device__ __forceinline__ void inverse4(const double* a, double* inv) {
  const double s0 = a[0] * a[5] - a[4] * a[1], s1 = a[0] * a[6] - a[4] * a[2], s2 = a[0] * a[7] - a[4] * a[3];
  const double s3 = a[1] * a[6] - a[5] * a[2], s4 = a[1] * a[7] - a[5] * a[3], s5 = a[2] * a[7] - a[6] * a[3];
  const double c5 = a[10] * a[15] - a[14] * a[11], c4 = a[9] * a[15] - a[13] * a[11], c3 = a[9] * a[14] - a[13] * a[10];
  const double c2 = a[8] * a[15] - a[12] * a[11], c1 = a[8] * a[14] - a[12] * a[10], c0 = a[8] * a[13] - a[12] * a[9];
  const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
  const double id = 1.0 / det;
  inv[0] = (a[5] * c5 - a[6] * c4 + a[7] * c3) * id;
  inv[1] = (-a[1] * c5 + a[2] * c4 - a[3] * c3) * id;
  inv[2] = (a[13] * s5 - a[14] * s4 + a[15] * s3) * id;
  inv[3] = (-a[9] * s5 + a[10] * s4 - a[11] * s3) * id;
  inv[4] = (-a[4] * c5 + a[6] * c2 - a[7] * c1) * id;
  inv[5] = (a[0] * c5 - a[2] * c2 + a[3] * c1) * id;
  inv[6] = (-a[12] * s5 + a[14] * s2 - a[15] * s1) * id;
  inv[7] = (a[8] * s5 - a[10] * s2 + a[11] * s1) * id;
  inv[8] = (a[4] * c4 - a[5] * c2 + a[7] * c0) * id;
  inv[9] = (-a[0] * c4 + a[1] * c2 - a[3] * c0) * id;
  inv[10] = (a[12] * s4 - a[13] * s2 + a[15] * s0) * id;
  inv[11] = (-a[8] * s4 + a[9] * s2 - a[11] * s0) * id;
  inv[12] = (-a[4] * c3 + a[5] * c1 - a[6] * c0) * id;
  inv[13] = (a[0] * c3 - a[1] * c1 + a[2] * c0) * id;
  inv[14] = (-a[12] * s3 + a[13] * s1 - a[14] * s0) * id;
  inv[15] = (a[8] * s3 - a[9] * s1 + a[10] * s0) * id;
}

struct DenseAux {
  double zhat0, h0, h1;
  double Q[16];
  double K[20];
};

// One blob applied to one dense landmark, formula by formula as the reference: generate_measurement :859-877,
// measurement_jacobian :748-802 (the reference's (dy/q, dx/q), q == 0 branch), measurement_covariance :804-819,
// matrix.inverse, kalman_gain :821-833, update_mean :897-914, update_covar :916-930, importance_factor :835-849.
// Returns log(importance factor).
__device__ __forceinline__ double dense_ekf_update(DenseLm& f, double sx, double sy, const BlobT<double>& z, const double* Qt,
                                                   bool immutable, DenseAux* aux = nullptr) {
  const double dx = f.m[0] - sx, dy = f.m[1] - sy;
  const double zhat0 = pk_atan2(dy, dx);  // :871 world frame
  const double q = dx * dx + dy * dy;
  double h0 = 0.0, h1 = 0.0;
  if (q != 0.0) {
    h0 = dy / q;
    h1 = dx / q;
  }
  // H Sigma (4x5): row 0 = h0 Sigma[0,:] + h1 Sigma[1,:], rows 1-3 = Sigma[2..4,:]
  double HS[20];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    HS[j] = h0 * f.S[j] + h1 * f.S[5 + j];
    HS[5 + j] = f.S[10 + j];
    HS[10 + j] = f.S[15 + j];
    HS[15 + j] = f.S[20 + j];
  }
  // Q = (H Sigma) H' + Qt: column 0 of H' is (h0, h1, 0, 0, 0), column k is e_{k+1}
  double Q[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    Q[4 * i] = HS[5 * i] * h0 + HS[5 * i + 1] * h1 + Qt[4 * i];
    Q[4 * i + 1] = HS[5 * i + 2] + Qt[4 * i + 1];
    Q[4 * i + 2] = HS[5 * i + 3] + Qt[4 * i + 2];
    Q[4 * i + 3] = HS[5 * i + 4] + Qt[4 * i + 3];
  }
  double Qi[16];
  inverse4(Q, Qi);
  // Sigma H' (5x4): column 0 = h0 Sigma[:,0] + h1 Sigma[:,1], columns 1-3 = Sigma[:,2..4]
  double SH[20];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    SH[4 * i] = f.S[5 * i] * h0 + f.S[5 * i + 1] * h1;
    SH[4 * i + 1] = f.S[5 * i + 2];
    SH[4 * i + 2] = f.S[5 * i + 3];
    SH[4 * i + 3] = f.S[5 * i + 4];
  }
  double K[20];  // 5x4
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      K[4 * i + j] = SH[4 * i] * Qi[j] + SH[4 * i + 1] * Qi[4 + j] + SH[4 * i + 2] * Qi[8 + j] + SH[4 * i + 3] * Qi[12 + j];
  const double d[4] = {z.bearing - zhat0, z.r - f.m[2], z.g - f.m[3], z.b - f.m[4]};  // :846/:911, not wrapped
  double fro2 = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) fro2 += Q[i] * Q[i];  // matrix.magnitude, matrix.py:31-33
  double maha = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) maha += d[i] * (Qi[4 * i] * d[0] + Qi[4 * i + 1] * d[1] + Qi[4 * i + 2] * d[2] + Qi[4 * i + 3] * d[3]);
  double logw = -0.5 * (Consts<double>::log_two_pi + 0.5 * log(fro2)) - 0.5 * maha;
  if (f.count & kPotentialBit) logw = Consts<double>::log_no_match;  // :111-112
  if (aux) {
    aux->zhat0 = zhat0;
    aux->h0 = h0;
    aux->h1 = h1;
    for (int i = 0; i < 16; ++i) aux->Q[i] = Q[i];
    for (int i = 0; i < 20; ++i) aux->K[i] = K[i];
  }
  if (!immutable) {
#pragma unroll
    for (int i = 0; i < 5; ++i) f.m[i] += K[4 * i] * d[0] + K[4 * i + 1] * d[1] + K[4 * i + 2] * d[2] + K[4 * i + 3] * d[3];
    double N[25];  // Sigma' = Sigma - K (H Sigma)
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
        N[5 * i + j] = f.S[5 * i + j] - (K[4 * i] * HS[j] + K[4 * i + 1] * HS[5 + j] + K[4 * i + 2] * HS[10 + j] + K[4 * i + 3] * HS[15 + j]);
#pragma unroll
    for (int i = 0; i < 25; ++i) f.S[i] = N[i];
    count_update(f.count);
  }
  return logw;
}

// ------------------------------------------------------------------ the dense observe kernel
struct DenseArgs {
  SlotSource ss;
  unsigned char* map_dst;  // NULL: association only (pk_associate), nothing is updated
  size_t count_off;
  int32_t* src;
  const double *x, *y, *h;
  double* logw;
  const double* blobs;    // [B][4]
  const double* blobdir;  // [B][2]
  const int32_t* ids_in;  // [B] supplied ids shared by all particles, or NULL: maximum likelihood
  int32_t* ids_out;       // [P][B] or NULL
  const unsigned char* immutable;
  double Qt[16];
  int L, Lp, B;
  int reset;
  unsigned long long* gmax_key;
};

size_t dense_lds_bytes(int Lp, int B) { return (size_t)B * 20 + (size_t)Lp * 4 + 32; }

__global__ void __launch_bounds__(256) k_observe_dense(DenseArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double red[256 / kWave];
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);  // [B]
  int* bid = reinterpret_cast<int*>(best + a.B);                             // [B] landmark index or INT_MAX
  int* s_ids = bid + a.B;                                                    // [B] 1-based id, 0 = unmatched
  int* s_next = s_ids + a.B;                                                 // [B]
  int* s_first = s_next + a.B;                                               // [Lp]
  const int64_t p = blockIdx.x;
  const int tid = threadIdx.x;
  const int Lp = a.Lp, B = a.B;
  const unsigned char* sslot = a.ss.at(a.src[p]);
  const double* sf = reinterpret_cast<const double*>(sslot);
  const int* sc = reinterpret_cast<const int*>(sslot + a.count_off);
  const double sx = a.x[p], sy = a.y[p], sh = a.h[p];
  for (int b = tid; b < B; b += 256) {
    best[b] = 0ull;
    bid[b] = INT_MAX;
  }
  __syncthreads();
  if (a.ids_in) {
    for (int b = tid; b < B; b += 256) s_ids[b] = a.ids_in[b];
  } else {
    // match_features_to_scan / match_one (:317-381): the largest probability per blob, strict '>' from 0.0, the
    // earliest landmark on a tie -- atomicMax on the probability bits, then atomicMin on the landmark index
    for (int pass = 0; pass < 2; ++pass) {
      for (int l = tid; l < a.L; l += 256) {
        const DenseLm d = dense_load(sf, sc, Lp, l);
        const Landmark<double> lm = dense_blocks(d);
        const double pse = pk_atan2(lm.my - sy, lm.mx - sx);
        const double eb = pse - sh;  // :408
        for (int b = 0; b < B; ++b) {
          const BlobT<double> z{a.blobs[4 * b], a.blobs[4 * b + 1], a.blobs[4 * b + 2], a.blobs[4 * b + 3]};
          if (fabs(z.bearing - eb) > 0.5) continue;                                            // :433
          if (fabs(color_distance2(lm.mr, lm.mg, lm.mb, z.r, z.g, z.b)) > 300.0) continue;  // :441
          const double bp = 500.0 * prob_position_match(lm, sx, sy, pse, z.bearing, a.blobdir[2 * b], a.blobdir[2 * b + 1]);
          const double cp = 500.0 * prob_color_match(lm, z.r, z.g, z.b);
          const double pr = bp * cp / 250000.0;  // :455
          if (!(pr > 0.0)) continue;
          const unsigned long long bits = (unsigned long long)__double_as_longlong(pr);
          if (pass == 0)
            atomicMax(&best[b], bits);
          else if (bits == best[b])
            atomicMin(&bid[b], l);
        }
      }
      __syncthreads();
    }
    for (int b = tid; b < B; b += 256) s_ids[b] = best[b] != 0ull ? bid[b] + 1 : 0;
  }
  __syncthreads();
  if (a.ids_out)
    for (int b = tid; b < B; b += 256) a.ids_out[(size_t)p * B + b] = s_ids[b];
  if (!a.map_dst) return;
  // landmark -> blob chains in scan order (:88): first[l] the lowest blob matched to l, next[b] the following one
  for (int l = tid; l < Lp; l += 256) s_first[l] = INT_MAX;
  for (int b = tid; b < B; b += 256) s_next[b] = -1;
  __syncthreads();
  int cnt0 = 0;
  for (int b = tid; b < B; b += 256) {
    const int id = s_ids[b];
    if (id > 0)
      atomicMin(&s_first[id - 1], b);
    else
      ++cnt0;
  }
  __syncthreads();
  for (int b = tid; b < B; b += 256) {
    const int id = s_ids[b];
    if (id > 0 && s_first[id - 1] != b) {
      int q = b - 1;
      while (s_ids[q] != id) --q;
      s_next[q] = b;
    }
  }
  __syncthreads();
  unsigned char* dslot = a.map_dst + (size_t)p * a.ss.slot_bytes;
  double* df = reinterpret_cast<double*>(dslot);
  int* dc = reinterpret_cast<int*>(dslot + a.count_off);
  double acc = (double)cnt0 * Consts<double>::log_no_match;  // unseen features: weight *= 0.1 each (:94-95)
  for (int l = tid; l < Lp; l += 256) {
    DenseLm d = dense_load(sf, sc, Lp, l);
    if (l < a.L) {
      const bool imm = a.immutable[l] != 0;
      for (int b = s_first[l] == INT_MAX ? -1 : s_first[l]; b >= 0; b = s_next[b]) {
        const BlobT<double> z{a.blobs[4 * b], a.blobs[4 * b + 1], a.blobs[4 * b + 2], a.blobs[4 * b + 3]};
        acc += dense_ekf_update(d, sx, sy, z, a.Qt, imm);
      }
    }
    dense_store(df, dc, Lp, l, d);
  }
  const double tot = block_sum<256 / kWave>(acc, red);
  if (tid == 0) {
    const double v = (a.reset ? 0.0 : a.logw[p]) + tot;
    a.logw[p] = v;
    if (a.gmax_key) atomicMax(a.gmax_key + (p & (kGmaxKeys - 1)), double_to_key(v));
    a.src[p] = (int32_t)p;
  }
}

void launch_observe_dense(hipStream_t s, DeviceState& d, const double* blobs_dev, const double* blobdir_dev, int B,
                          const int32_t* ids_in_dev, int32_t* ids_out_dev, const double Qt[16], bool update,
                          const ObserveExtras& ex) {
  if (d.P == 0) return;
  static bool attr_set[kMaxDevices] = {false};
  if (first_time_on_this_device(attr_set)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_observe_dense), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kMaxDynLds) != hipSuccess)
      (void)hipGetLastError();
  }
  DenseArgs a;
  a.ss = slot_source(d);
  a.map_dst = update ? d.map[d.mcur ^ 1] : nullptr;
  a.count_off = d.lay.count_off;
  a.src = d.src[d.cur];
  a.x = d.x[d.cur];
  a.y = d.y[d.cur];
  a.h = d.h[d.cur];
  a.logw = d.logw[d.cur];
  a.blobs = blobs_dev;
  a.blobdir = blobdir_dev;
  a.ids_in = ids_in_dev;
  a.ids_out = ids_out_dev;
  a.immutable = d.immutable;
  for (int i = 0; i < 16; ++i) a.Qt[i] = Qt[i];
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  a.reset = ex.reset ? 1 : 0;
  a.gmax_key = ex.gmax_key;
  hipLaunchKernelGGL(k_observe_dense, dim3((unsigned)d.P), dim3(256), dense_lds_bytes(d.lay.Lp, B), s, a);
  if (update) {
    d.mcur ^= 1;
    d.alt = nullptr;
  }
}

// ------------------------------------------------------------------ probe, dense inputs
// in: pose[3] mean[5] cov[25] blob[4] Qt[16] dir[2] (55 doubles); out: PK_PROBE_LEN doubles (see parakeet_slam.h)
__global__ void k_probe_dense(const double* __restrict__ in, double* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double sx = in[0], sy = in[1], sh = in[2];
  DenseLm f;
  for (int i = 0; i < 5; ++i) f.m[i] = in[3 + i];
  for (int i = 0; i < 25; ++i) f.S[i] = in[8 + i];
  f.count = 0;
  const BlobT<double> z{in[33], in[34], in[35], in[36]};
  const double ux = in[53], uy = in[54];
  for (int i = 0; i < 79; ++i) out[i] = 0.0;
  const Landmark<double> lm = dense_blocks(f);
  out[0] = probability_of_match(lm, sx, sy, sh, z, ux, uy);
  const double pse = pk_atan2(lm.my - sy, lm.mx - sx);
  out[1] = prob_position_match(lm, sx, sy, pse, z.bearing, ux, uy, out + 2);
  out[4] = prob_color_match(lm, z.r, z.g, z.b);
  DenseAux aux;
  DenseLm g = f;
  const double lw = dense_ekf_update(g, sx, sy, z, in + 37, false, &aux);
  out[5] = aux.zhat0;
  out[6] = f.m[2];
  out[7] = f.m[3];
  out[8] = f.m[4];
  out[9] = aux.h0;
  out[10] = aux.h1;
  for (int i = 0; i < 16; ++i) out[11 + i] = aux.Q[i];
  for (int i = 0; i < 20; ++i) out[27 + i] = aux.K[i];
  out[47] = exp(lw);
  for (int i = 0; i < 5; ++i) out[48 + i] = g.m[i];
  for (int i = 0; i < 25; ++i) out[53 + i] = g.S[i];
  out[78] = lw;
}
void launch_probe_dense(hipStream_t s, const double* in_dev, double* out_dev) {
  hipLaunchKernelGGL(k_probe_dense, dim3(1), dim3(64), 0, s, in_dev, out_dev);
}

}  // namespace pk
