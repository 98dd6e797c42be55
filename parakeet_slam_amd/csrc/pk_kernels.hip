// Hand-written gfx950 (CDNA4, wave64) kernels of the FastSLAM particle update.
//
// The path is HBM-bound streaming over per-particle landmark maps with O(100) fp64
// flops per 232 bytes, so the rules that matter are coalescing (landmark index fastest,
// 16-byte loads of two adjacent landmarks), enough bytes in flight per CU, one pass over
// the map per filter step, and deterministic wave-shuffle reductions.  No MFMA: the
// algebra is 2x2 / 3x3 and register resident (pk_math.hpp).
#include "pk_kernels.hpp"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cstring>
#include <type_traits>

#include "pk_math.hpp"
#include "pk_philox.hpp"

namespace pk {

constexpr int kWave = 64;
constexpr int kObsThreads = 256;

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;  // identical in every lane; butterfly order is fixed => deterministic
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, kWave));
  return v;
}

// Sum over a workgroup of NW waves; result valid in every thread.  Fixed order.
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* lds /* >= NW doubles */) {
  v = wave_sum(v);
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  __syncthreads();
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double t = lds[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t += lds[i];
  return t;
}
template <int NW>
__device__ __forceinline__ double block_max(double v, double* lds) {
  v = wave_max(v);
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  __syncthreads();
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double t = lds[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmax(t, lds[i]);
  return t;
}

// Order-preserving map double -> uint64 (max of keys == max of doubles), for the running max
// of the log-weights that the observe kernels keep with one atomicMax per particle.
__host__ __device__ inline unsigned long long double_to_key(double x) {
  unsigned long long b;
  memcpy(&b, &x, 8);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double key_to_double(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double x;
  memcpy(&x, &b, 8);
  return x;
}

// ------------------------------------------------------------------ K1 motion
__global__ void __launch_bounds__(256) k_motion(double* __restrict__ x, double* __restrict__ y,
                                                double* __restrict__ h, int64_t P, double v, double w,
                                                double dt, double sd, double sh,
                                                const double* __restrict__ z, uint64_t seed,
                                                uint64_t draw, int64_t goff) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  double z0, z1, z2;
  if (z) {
    z0 = z[3 * i];
    z1 = z[3 * i + 1];
    z2 = z[3 * i + 2];
  } else {
    uint64_t g = (uint64_t)(i + goff);
    Philox4 a = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)draw,
                              (uint32_t)(draw >> 32) & 0x7fffffffu, (uint32_t)seed, (uint32_t)(seed >> 32));
    Philox4 b = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)draw,
                              ((uint32_t)(draw >> 32) & 0x7fffffffu) | 0x80000000u, (uint32_t)seed,
                              (uint32_t)(seed >> 32));
    double u1 = u53(a.v[0], a.v[1]), u2 = u53(a.v[2], a.v[3]);
    double u3 = u53(b.v[0], b.v[1]), u4 = u53(b.v[2], b.v[3]);
    double r1 = sqrt(-2.0 * log(u1)), r2 = sqrt(-2.0 * log(u3));
    double s1, c1, s2, c2;
    sincos(Consts<double>::two_pi * u2, &s1, &c1);
    sincos(Consts<double>::two_pi * u4, &s2, &c2);
    z0 = r1 * c1;
    z1 = r1 * s1;
    z2 = r2 * c2;
    (void)s2;
  }
  double xi = x[i], yi = y[i], hi = h[i];
  // normal(0, s, 1) == 0 + s * gauss  (numpy legacy), prkt_core_v2.py:185,190,193
  motion_model(xi, yi, hi, v, w, dt, 0.0 + sd * z0, 0.0 + sh * z1, 0.0 + sh * z2);
  x[i] = xi;
  y[i] = yi;
  h[i] = hi;
}

void launch_motion(hipStream_t s, DeviceState& d, double v, double w, double dt, const double* z_dev,
                   uint64_t seed, uint64_t draw, int64_t global_offset) {
  if (d.P == 0) return;
  double sd = fabs(.05 * v) + fabs(.005 * w) + .0005;  // :185
  double sh = fabs(.025 * w) + fabs(.005 * v) + .0005;  // :190,:193
  int blocks = (int)((d.P + 255) / 256);
  hipLaunchKernelGGL(k_motion, dim3(blocks), dim3(256), 0, s, d.x[d.cur], d.y[d.cur], d.h[d.cur], d.P, v, w,
                     dt, sd, sh, z_dev, seed, draw, global_offset + d.global_offset);
}

__global__ void k_fill(double* p, int64_t n, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_reset_weights(hipStream_t s, DeviceState& d) {
  if (d.P == 0) return;
  int blocks = (int)((d.P + 255) / 256);
  hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, s, d.logw[d.cur], d.P, 0.0);
}

__global__ void k_iota(int32_t* p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int32_t)i;
}
void launch_iota(hipStream_t s, int32_t* p, int64_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
}

// ------------------------------------------------------------------ landmark slot access
__device__ __forceinline__ Landmark<double> load_landmark(const double* f, const int* cnt, int Lp, int l) {
  Landmark<double> m;
  m.mx = f[F_MX * Lp + l];
  m.my = f[F_MY * Lp + l];
  m.mr = f[F_MR * Lp + l];
  m.mg = f[F_MG * Lp + l];
  m.mb = f[F_MB * Lp + l];
  m.pxx = f[F_PXX * Lp + l];
  m.pxy = f[F_PXY * Lp + l];
  m.pyy = f[F_PYY * Lp + l];
  m.crr = f[F_CRR * Lp + l];
  m.crg = f[F_CRG * Lp + l];
  m.crb = f[F_CRB * Lp + l];
  m.cgg = f[F_CGG * Lp + l];
  m.cgb = f[F_CGB * Lp + l];
  m.cbb = f[F_CBB * Lp + l];
  m.count = cnt[l];
  return m;
}

// ------------------------------------------------------------------ K2 association
// One workgroup per particle, lanes over landmarks, uniform loop over blobs (scalar
// loads).  Two passes: (1) atomicMax of the probability per blob in LDS, (2) the lowest
// landmark index that attains it -- the reference's strict '>' scan keeps the earliest
// (prkt_core_v2.py:369-381).  Probability 0 never matches.
struct AssocArgs {
  SlotSource ss;
  size_t count_off;
  const int32_t* src;
  const double *x, *y, *h;
  const double* blobs;    // B x 4
  const double* blobdir;  // B x 2 unit ray direction (closest_point :510)
  int32_t* ids;           // P x B
  int L, Lp, B;
};

__device__ __forceinline__ double match_probability_lazy(const double* f, const int* cnt, int Lp, int l,
                                                         Landmark<double>& lm, bool& have_cov, double sx,
                                                         double sy, double pse, const BlobT<double>& z,
                                                         double ux, double uy) {
  if (!have_cov) {
    lm.pxx = f[F_PXX * Lp + l];
    lm.pxy = f[F_PXY * Lp + l];
    lm.pyy = f[F_PYY * Lp + l];
    lm.crr = f[F_CRR * Lp + l];
    lm.crg = f[F_CRG * Lp + l];
    lm.crb = f[F_CRB * Lp + l];
    lm.cgg = f[F_CGG * Lp + l];
    lm.cgb = f[F_CGB * Lp + l];
    lm.cbb = f[F_CBB * Lp + l];
    have_cov = true;
  }
  double bp = 500.0 * prob_position_match(lm, sx, sy, pse, z.bearing, ux, uy);
  double cp = 500.0 * prob_color_match(lm, z.r, z.g, z.b);
  return bp * cp / 250000.0;
}

template <int PASS>
__device__ __forceinline__ void assoc_pass(const AssocArgs& a, const double* f, const int* cnt, double sx,
                                           double sy, double sh, unsigned long long* best, int* bid) {
  for (int l = threadIdx.x; l < a.L; l += blockDim.x) {
    Landmark<double> lm;
    lm.mx = f[F_MX * a.Lp + l];
    lm.my = f[F_MY * a.Lp + l];
    lm.mr = f[F_MR * a.Lp + l];
    lm.mg = f[F_MG * a.Lp + l];
    lm.mb = f[F_MB * a.Lp + l];
    bool have_cov = false;
    double pse = atan2(lm.my - sy, lm.mx - sx);
    double eb = pse - sh;  // :408
    for (int b = 0; b < a.B; ++b) {
      BlobT<double> z{a.blobs[4 * b], a.blobs[4 * b + 1], a.blobs[4 * b + 2], a.blobs[4 * b + 3]};
      if (fabs(z.bearing - eb) > 0.5) continue;                                    // :433
      if (fabs(color_distance2(lm.mr, lm.mg, lm.mb, z.r, z.g, z.b)) > 300.0) continue;  // :441
      double pr = match_probability_lazy(f, cnt, a.Lp, l, lm, have_cov, sx, sy, pse, z, a.blobdir[2 * b],
                                         a.blobdir[2 * b + 1]);
      if (!(pr > 0.0)) continue;
      unsigned long long bits = (unsigned long long)__double_as_longlong(pr);
      if (PASS == 0) {
        atomicMax(&best[b], bits);
      } else if (bits == best[b]) {
        atomicMin(&bid[b], l);
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_assoc_brute(AssocArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);
  int* bid = reinterpret_cast<int*>(best + a.B);
  const int64_t p = blockIdx.x;
  const unsigned char* slot = a.ss.at(a.src[p]);
  const double* f = reinterpret_cast<const double*>(slot);
  const int* cnt = reinterpret_cast<const int*>(slot + a.count_off);
  const double sx = a.x[p], sy = a.y[p], sh = a.h[p];
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
    best[b] = 0ull;
    bid[b] = INT_MAX;
  }
  __syncthreads();
  assoc_pass<0>(a, f, cnt, sx, sy, sh, best, bid);
  __syncthreads();
  assoc_pass<1>(a, f, cnt, sx, sy, sh, best, bid);
  __syncthreads();
  for (int b = threadIdx.x; b < a.B; b += blockDim.x)
    a.ids[(size_t)p * a.B + b] = best[b] != 0ull ? bid[b] + 1 : 0;
}

void launch_assoc_brute(hipStream_t s, DeviceState& d, const double* blobs_dev, const double* blobdir_dev, int B,
                        int32_t* ids_dev) {
  if (d.P == 0 || B == 0) return;
  AssocArgs a;
  a.ss = slot_source(d);
  a.count_off = d.lay.count_off;
  a.src = d.src[d.cur];
  a.x = d.x[d.cur];
  a.y = d.y[d.cur];
  a.h = d.h[d.cur];
  a.blobs = blobs_dev;
  a.blobdir = blobdir_dev;
  a.ids = ids_dev;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  size_t lds = (size_t)B * 12;
  hipLaunchKernelGGL(k_assoc_brute, dim3((unsigned)d.P), dim3(256), lds, s, a);
}


// ------------------------------------------------------------------ K2 (grid)
// Persistent workgroups: the scan tables are staged in LDS once per workgroup and reused for
// every particle the workgroup processes.  Blobs are handled in cell order (index t); the
// scan order b = order[t] only matters on write-out.
//
// Per particle:
//   S1 (lanes over landmarks, means only = 40 B/landmark of HBM):
//      phase 1 (LDS only): the blobs that can pass the colour gate of a landmark lie in the
//        <= 27 colour cells around it.  With the 9x column-duplicated index list (DUP) that
//        neighbourhood is ONE contiguous range; otherwise it is walked as nine ranges.  Each
//        blob there is tested against both gates in fp32 with conservative thresholds (one
//        ds_read_b128: r, g, b, bearing); the few survivors are kept in registers.
//      phase 2 (global, convergent): the exact float64 records of all survivors of all lanes
//        are loaded together, then the exact gates (:433, :441) decide.  A landmark that
//        passes both gates of blob t is appended to t's candidate list (4 slots + a count).
//   S2 (lanes over blobs): 0 candidates -> id 0.  1 candidate -> that landmark, TENTATIVELY:
//      the match stands iff its probability is > 0, which k_observe decides with the landmark
//      state it has in registers anyway (the reference's strict '>' from 0.0, :369-381).
//      2..4 candidates -> queued for S3; more -> queued for S4.
//   S3 (lanes over contested blobs): evaluate their probabilities exactly in landmark order,
//      keep the largest, the earliest landmark on a tie, none if all are 0.
//   S4 (rare; lanes over blobs with > 4 gate-passers): the reference's own loop -- every
//      landmark in order, strict '>' -- so any number of contenders and ties come out right.
struct AssocGridArgs {
  AssocArgs a;  // a.blobs / a.blobdir unused here
  BlobGrid g;
  const unsigned char* tables;  // see blob_grid_table_bytes
  const double* exact;          // [B][6] in cell order: bearing, r, g, b, ux, uy
  int64_t P;
  int finalize;  // 1: also settle single-candidate blobs here (pk_associate), not in k_observe
  int n9;        // DUP: entries of the duplicated index list (padded to 8)
  // Fast hand-off to k_observe_fast (L <= 512): per landmark the (<= 4) blobs that pass its
  // gates, per blob how many landmarks pass; a particle where some landmark passes more than
  // four blobs is flagged and settled the general way (S2..S4 + ids) instead.
  uint4* lmpass;          // [P][Lp]  x,y: four 16-bit fields = blob (cell order) or 0xFFFF; z,w: atan2(my-sy, mx-sx)
  unsigned char* bcount;  // [P][B]   saturating count
  unsigned char* pflag;   // [P]      1 = general path
  const unsigned char* only_flagged;  // GENERAL instance: skip particles whose flag is 0
  unsigned* n_flagged;                // count of flagged particles (zeroed by the scan upload)
};

// tables: start u16[ncell+1] (16-byte padded) | rec32 float4[B] | idx9 u16[n9] (DUP only) | order u16[B]
// `start` is cell_start (offsets into rec32) or, with DUP, col_start (offsets into idx9).
__host__ __device__ inline size_t grid_cs_bytes(int ncell) { return ((size_t)(ncell + 1) * 2 + 15) & ~(size_t)15; }
size_t blob_grid_table_bytes(int ncell, int B, int n9) {
  return grid_cs_bytes(ncell) + (size_t)B * 16 + (size_t)n9 * 2 + (size_t)B * 2;
}

__device__ __forceinline__ Landmark<double> load_landmark_nocount(const double* f, int Lp, int l) {
  Landmark<double> m;
  m.mx = f[F_MX * Lp + l];
  m.my = f[F_MY * Lp + l];
  m.mr = f[F_MR * Lp + l];
  m.mg = f[F_MG * Lp + l];
  m.mb = f[F_MB * Lp + l];
  m.pxx = f[F_PXX * Lp + l];
  m.pxy = f[F_PXY * Lp + l];
  m.pyy = f[F_PYY * Lp + l];
  m.crr = f[F_CRR * Lp + l];
  m.crg = f[F_CRG * Lp + l];
  m.crb = f[F_CRB * Lp + l];
  m.cgg = f[F_CGG * Lp + l];
  m.cgb = f[F_CGB * Lp + l];
  m.cbb = f[F_CBB * Lp + l];
  m.count = 0;
  return m;
}

// probability_of_match (:383-455) of landmark l for the blob record rec = (bearing, r, g, b, ux, uy)
__device__ __forceinline__ double full_match_probability(const double* f, int Lp, int l, double sx, double sy,
                                                         double sh, const double* rec) {
  const Landmark<double> lm = load_landmark_nocount(f, Lp, l);
  BlobT<double> z{rec[0], rec[1], rec[2], rec[3]};
  return probability_of_match(lm, sx, sy, sh, z, rec[4], rec[5]);
}

constexpr int kCand = 4;

// Diagnostic build only (-DPK_STAMPS, never shipped): per-phase cycle sums of k_assoc_grid.
#ifdef PK_STAMPS
__device__ unsigned long long pk_stamp_acc[16];
#define PK_STAMP(var) \
  unsigned long long var; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");
#define PK_STAMP_ADD(slot, a, b) \
  if ((threadIdx.x & 63) == 0) atomicAdd(&pk_stamp_acc[slot], (b) - (a));
#else
#define PK_STAMP(var)
#define PK_STAMP_ADD(slot, a, b)
#endif

// GENERAL = false: S1 + hand-off only (light on registers); particles it flags are redone by
// the GENERAL = true instance launched with only_flagged.
template <int THREADS, bool DUP, bool GENERAL>
__global__ void __launch_bounds__(THREADS) k_assoc_grid(AssocGridArgs ga) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int n_few, n_many, wg_flag;
  __shared__ unsigned long long s3_best[THREADS / 4];
  __shared__ int s3_win[THREADS / 4];
  const AssocArgs& a = ga.a;
  const BlobGrid& g = ga.g;
  const int B = a.B;
  if (GENERAL && ga.only_flagged && *ga.n_flagged == 0u) return;  // nothing was flagged
  const size_t cs_bytes = grid_cs_bytes(g.ncell);
  const size_t tab_bytes = cs_bytes + (size_t)B * 16 + (DUP ? (size_t)ga.n9 * 2 : 0);  // the part kept in LDS
  const unsigned short* start = reinterpret_cast<const unsigned short*>(smem);
  const float4* rec32 = reinterpret_cast<const float4*>(smem + cs_bytes);
  const unsigned short* idx9 = reinterpret_cast<const unsigned short*>(smem + cs_bytes + (size_t)B * 16);
  int* ccount = reinterpret_cast<int*>(smem + tab_bytes);  // [B] gate-passing landmarks of blob t; then the result
  unsigned short* cand = reinterpret_cast<unsigned short*>(ccount + B);  // [B][4] first four of them (arrival order)
  unsigned short* queue = cand + 4 * (size_t)B;  // [B] contested blobs (2..4 from the front, > 4 from the end)
  int* result = ccount;
  const unsigned short* order =
      reinterpret_cast<const unsigned short*>(ga.tables + cs_bytes + (size_t)B * 16 + (size_t)ga.n9 * 2);  // global
  {
    const uint4* src = reinterpret_cast<const uint4*>(ga.tables);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (size_t i = threadIdx.x; i < tab_bytes / 16; i += THREADS) dst[i] = src[i];
  }
  for (int64_t p = blockIdx.x; p < ga.P; p += gridDim.x) {
    if (GENERAL && ga.only_flagged && !ga.only_flagged[p]) continue;  // workgroup-uniform
    const unsigned char* slot = a.ss.at(a.src[p]);
    const double* f = reinterpret_cast<const double*>(slot);
    const double sx = a.x[p], sy = a.y[p], sh = a.h[p];
    PK_STAMP(ts0)
    for (int t = threadIdx.x; t < B; t += THREADS) ccount[t] = 0;
    if (threadIdx.x == 0) {
      n_few = 0;
      n_many = 0;
      wg_flag = 0;
    }
    __syncthreads();
    PK_STAMP(ts1)
    PK_STAMP_ADD(0, ts0, ts1)
    // ---- S1 ------------------------------------------------------------------------------
    // software pipeline: the means of the NEXT landmark are in flight while this one is searched
    int l = threadIdx.x;
    double nmx = 0, nmy = 0, nmr = 0, nmg = 0, nmb = 0;
    if (l < a.L) {
      nmx = f[F_MX * a.Lp + l];
      nmy = f[F_MY * a.Lp + l];
      nmr = f[F_MR * a.Lp + l];
      nmg = f[F_MG * a.Lp + l];
      nmb = f[F_MB * a.Lp + l];
    }
    for (; l < a.L; l += THREADS) {
      const double mx = nmx, my = nmy, mr = nmr, mg = nmg, mb = nmb;
      const int ln = l + THREADS;
      if (ln < a.L) {
        nmx = f[F_MX * a.Lp + ln];
        nmy = f[F_MY * a.Lp + ln];
        nmr = f[F_MR * a.Lp + ln];
        nmg = f[F_MG * a.Lp + ln];
        nmb = f[F_MB * a.Lp + ln];
      }
      PK_STAMP(ta0)
      const double pse = atan2(my - sy, mx - sx);
      const double eb = pse - sh;  // :408
      const float mr32 = (float)mr, mg32 = (float)mg, mb32 = (float)mb, eb32 = (float)eb;
      // same cell function as the host (floor((v - lo) * inv_h)): inside the colour gate
      // |dv| <= 17.3205 < 17.5, so the cell indices of blob and landmark differ by at most
      // one.  -1 / G mean "outside the grid": only the edge cell can hold a neighbour.
      int c[3];
      const double m3[3] = {mr, mg, mb};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double q = floor(__dmul_rn(__dsub_rn(m3[k], g.lo[k]), g.inv_h));
        q = fmin(fmax(q, -1.0), (double)g.G[k]);
        c[k] = (int)q;
      }
      const int k0 = max(c[2] - 1, 0), k1 = min(c[2] + 1, g.G[2] - 1);
      int pc[kCand];
      int npc = 0;
      PK_STAMP(ta1)
      PK_STAMP_ADD(1, ta0, ta1)
      auto prefilter = [&](int t) {
        const float4 q = rec32[t];
        const float d0 = q.x - mr32, d1 = q.y - mg32, d2 = q.z - mb32;
        const float cd32 = d0 * d0 + d1 * d1 + d2 * d2;
        // conservative fp32 gates; NaN/inf fall through to the exact float64 tests
        return !(cd32 > g.thr32) && !(fabsf(q.w - eb32) > g.thrb32);
      };
      unsigned pass01 = 0xFFFFFFFFu, pass23 = 0xFFFFFFFFu;  // the blobs that pass this landmark's gates (first four)
      int npass = 0;
      auto exact_gates = [&](int tt, const double2& z01, const double2& z23) {
        if (!(fabs(z01.x - eb) > 0.5) && !(fabs(color_distance2(mr, mg, mb, z01.y, z23.x, z23.y)) > 300.0)) {
          const int n = atomicAdd(&ccount[tt], 1);
          if (n < 4) cand[4 * tt + n] = (unsigned short)l;
          if (npass == 0) pass01 = (pass01 & 0xFFFF0000u) | (unsigned)tt;
          if (npass == 1) pass01 = (pass01 & 0x0000FFFFu) | ((unsigned)tt << 16);
          if (npass == 2) pass23 = (pass23 & 0xFFFF0000u) | (unsigned)tt;
          if (npass == 3) pass23 = (pass23 & 0x0000FFFFu) | ((unsigned)tt << 16);
          ++npass;
        }
      };
      // ---- phase 1 (LDS only) ---------------------------------------------------------
      if (DUP) {
        // column (r, g) clamped into the grid: its list holds every blob within one cell in r
        // and g, ordered by the b cell, so [k0, k1] is one contiguous range
        const int r = min(max(c[0], 0), g.G[0] - 1), gg = min(max(c[1], 0), g.G[1] - 1);
        const int base = (r * g.G[1] + gg) * g.G[2];
        int i = 0, i1 = 0;
        if (k0 <= k1) {
          i = start[base + k0];
          i1 = start[base + k1 + 1];
        }
        int tnext = i < i1 ? (int)idx9[i] : 0;
        for (; i < i1; ++i) {
          const int t = tnext;
          if (i + 1 < i1) tnext = idx9[i + 1];
          if (prefilter(t)) {
#pragma unroll
            for (int k = 0; k < kCand; ++k)
              if (npc == k) pc[k] = t;
            ++npc;
          }
        }
      } else {
        // flattened walk over the 9 (r, g) columns x [k0, k1]; j = next column, [t, t1) = open range
        int j = (k0 <= k1) ? 0 : 9, t = 0, t1 = 0;
        for (;;) {
          if (t >= t1) {
            if (j >= 9) break;
            const int jr = (j * 11) >> 5;  // j / 3 for j < 9
            const int r = c[0] - 1 + jr, gg = c[1] - 1 + (j - 3 * jr);
            ++j;
            if ((unsigned)r >= (unsigned)g.G[0] || (unsigned)gg >= (unsigned)g.G[1]) continue;
            const int base = (r * g.G[1] + gg) * g.G[2];
            t = start[base + k0];
            t1 = start[base + k1 + 1];
            if (t >= t1) continue;
          }
          if (prefilter(t)) {
#pragma unroll
            for (int k = 0; k < kCand; ++k)
              if (npc == k) pc[k] = t;
            ++npc;
          }
          ++t;
        }
      }
      PK_STAMP(ta2)
      PK_STAMP_ADD(2, ta1, ta2)
      // ---- phase 2 (global, convergent): all survivors' exact records in one batch -----
      if (__any(npc > 0)) {
        double2 z01[kCand], z23[kCand];
#pragma unroll
        for (int k = 0; k < kCand; ++k)
          if (npc > k) {
            const double* rec = ga.exact + 6 * (size_t)pc[k];
            z01[k] = *reinterpret_cast<const double2*>(rec);
            z23[k] = *reinterpret_cast<const double2*>(rec + 2);
          }
#pragma unroll
        for (int k = 0; k < kCand; ++k)
          if (npc > k) exact_gates(pc[k], z01[k], z23[k]);
      }
      PK_STAMP(ta3)
      PK_STAMP_ADD(3, ta2, ta3)
      if (npc > kCand) {
        // more fp32 survivors than register slots (dense colour clusters): walk again and take
        // the ones beyond the first kCand as they come (nine-range walk works for both layouts
        // only without DUP; with DUP repeat the single range)
        int seen = 0;
        auto late = [&](int t) {
          if (prefilter(t)) {
            if (seen >= kCand) {
              const double* rec = ga.exact + 6 * (size_t)t;
              exact_gates(t, *reinterpret_cast<const double2*>(rec), *reinterpret_cast<const double2*>(rec + 2));
            }
            ++seen;
          }
        };
        if (DUP) {
          const int r = min(max(c[0], 0), g.G[0] - 1), gg = min(max(c[1], 0), g.G[1] - 1);
          const int base = (r * g.G[1] + gg) * g.G[2];
          for (int i = start[base + k0], i1 = start[base + k1 + 1]; i < i1; ++i) late(idx9[i]);
        } else {
          for (int j = 0; j < 9; ++j) {
            const int jr = (j * 11) >> 5;
            const int r = c[0] - 1 + jr, gg = c[1] - 1 + (j - 3 * jr);
            if ((unsigned)r >= (unsigned)g.G[0] || (unsigned)gg >= (unsigned)g.G[1]) continue;
            const int base = (r * g.G[1] + gg) * g.G[2];
            for (int t = start[base + k0], t1 = start[base + k1 + 1]; t < t1; ++t) late(t);
          }
        }
      }
      if (!GENERAL) {
        const unsigned long long pb = (unsigned long long)__double_as_longlong(pse);
        ga.lmpass[(size_t)p * a.Lp + l] = make_uint4(pass01, pass23, (unsigned)pb, (unsigned)(pb >> 32));
        if (npass > kFastSlots) wg_flag = 1;
      }
    }
    PK_STAMP(ts2)
    __syncthreads();
    PK_STAMP(ts3)
    PK_STAMP_ADD(4, ts1, ts2)
    PK_STAMP_ADD(5, ts2, ts3)
    if (!GENERAL) {
      for (int t = threadIdx.x; t < B; t += THREADS) {
        const int n = ccount[t];
        ga.bcount[(size_t)p * B + t] = (unsigned char)(n > 255 ? 255 : n);
      }
      if (threadIdx.x == 0) {
        ga.pflag[p] = (unsigned char)(wg_flag != 0);
        if (wg_flag) atomicAdd(ga.n_flagged, 1u);
      }
      __syncthreads();
      continue;  // k_observe_fast (or, if flagged, the GENERAL instance) takes it from here
    }
    // ---- S2 ------------------------------------------------------------------------------
    for (int t = threadIdx.x; t < B; t += THREADS) {
      const int n = ccount[t];
      if ((n == 1 && ga.finalize) || (n >= 2 && n <= 4)) queue[atomicAdd(&n_few, 1)] = (unsigned short)t;
      if (n > 4) queue[B - 1 - atomicAdd(&n_many, 1)] = (unsigned short)t;
    }
    __syncthreads();
    // blobs with exactly one gate-passer keep it (tentatively); none -> -1
    for (int t = threadIdx.x; t < B; t += THREADS) {
      const int n = ccount[t];
      result[t] = n == 1 ? (int)cand[4 * t] : (n == 0 ? -1 : -(n + 1));  // contested: -(n+1) until settled
    }
    __syncthreads();
    PK_STAMP(ts4)
    PK_STAMP_ADD(6, ts3, ts4)
    // ---- S3: 2..4 contenders (or 1 when finalising): four lanes per blob, one candidate each.
    // atomicMax on the probability bits, then atomicMin on the landmark index among the lanes
    // that attain it: the largest probability wins, the earliest landmark on a tie (:377),
    // nobody if all are 0.
    for (int base = 0; base < n_few; base += THREADS / 4) {
      const int slot = threadIdx.x >> 2, k = threadIdx.x & 3;
      const int qi = base + slot;
      if (k == 0) {
        s3_best[slot] = 0ull;
        s3_win[slot] = INT_MAX;
      }
      __syncthreads();
      int t = 0, lcand = 0;
      unsigned long long bits = 0ull;
      bool valid = false;
      if (qi < n_few) {
        t = queue[qi];
        const int n = result[t] >= 0 ? 1 : -result[t] - 1;
        valid = k < n;
      }
      if (valid) {
        lcand = cand[4 * t + k];
        const double pr = full_match_probability(f, a.Lp, lcand, sx, sy, sh, ga.exact + 6 * (size_t)t);
        if (pr > 0.0) {
          bits = (unsigned long long)__double_as_longlong(pr);
          atomicMax(&s3_best[slot], bits);
        }
      }
      __syncthreads();
      if (valid && bits != 0ull && bits == s3_best[slot]) atomicMin(&s3_win[slot], lcand);
      __syncthreads();
      if (k == 0 && qi < n_few) result[t] = s3_win[slot] == INT_MAX ? -1 : s3_win[slot];
    }
    PK_STAMP(ts5)
    PK_STAMP_ADD(7, ts4, ts5)
    // ---- S4: more than four contenders: the reference's sequential scan ------------------------
    for (int i = threadIdx.x; i < n_many; i += THREADS) {
      const int t = queue[B - 1 - i];
      const double* rec = ga.exact + 6 * (size_t)t;
      const double zb = rec[0], zr = rec[1], zg = rec[2], zbl = rec[3];
      int best = -1;
      double pm = 0.0;
      for (int l2 = 0; l2 < a.L; ++l2) {
        if (fabs(color_distance2(f[F_MR * a.Lp + l2], f[F_MG * a.Lp + l2], f[F_MB * a.Lp + l2], zr, zg, zbl)) > 300.0)
          continue;
        const double mx = f[F_MX * a.Lp + l2], my = f[F_MY * a.Lp + l2];
        const double eb = atan2(my - sy, mx - sx) - sh;
        if (fabs(zb - eb) > 0.5) continue;
        const double pr = full_match_probability(f, a.Lp, l2, sx, sy, sh, rec);
        if (pr > pm) {
          pm = pr;
          best = l2;
        }
      }
      result[t] = best;
    }
    PK_STAMP(ts6)
    PK_STAMP_ADD(8, ts5, ts6)
    __syncthreads();
    for (int t = threadIdx.x; t < B; t += THREADS) a.ids[(size_t)p * B + order[t]] = result[t] + 1;
    __syncthreads();
    PK_STAMP(ts7)
    PK_STAMP_ADD(9, ts6, ts7)
  }
}

#ifdef PK_STAMPS
void debug_read_stamps(unsigned long long* out, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pk_stamp_acc), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(pk_stamp_acc), z, sizeof(z));
  }
}
#endif

size_t assoc_grid_lds_bytes(int ncell, int B, int n9) {
  // start | rec32 16 B | idx9 | count/result 4 B | 4 candidates 8 B | queue 2 B   per blob
  return grid_cs_bytes(ncell) + (size_t)B * 30 + (size_t)n9 * 2 + 16;
}

template <int THREADS, bool DUP, bool GENERAL>
static void launch_assoc_grid_t(hipStream_t s, const AssocGridArgs& ga, size_t lds, int64_t P) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_assoc_grid<THREADS, DUP, GENERAL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds) != hipSuccess)
      (void)hipGetLastError();  // leave no sticky error behind for other users of the runtime
    attr_set = true;
  }
  // persistent grid: as many workgroups as LDS and the 2048-thread CU limit allow
  int per_cu = (int)((160 * 1024) / (lds + 64));
  per_cu = per_cu < 1 ? 1 : per_cu;
  const int by_threads = 2048 / THREADS;
  per_cu = per_cu > by_threads ? by_threads : per_cu;
  int64_t blocks = 256 * (int64_t)per_cu;
  if (blocks > P) blocks = P;
  hipLaunchKernelGGL((k_assoc_grid<THREADS, DUP, GENERAL>), dim3((unsigned)blocks), dim3(THREADS), lds, s, ga);
}

void launch_assoc_grid(hipStream_t s, DeviceState& d, int B, const BlobGrid& grid, int n9,
                       const unsigned char* tables_dev, const double* exact_dev, int32_t* ids_dev, bool finalize,
                       const FastHandoff& fh) {
  if (d.P == 0 || B == 0) return;
  AssocGridArgs ga;
  AssocArgs& a = ga.a;
  a.ss = slot_source(d);
  a.count_off = d.lay.count_off;
  a.src = d.src[d.cur];
  a.x = d.x[d.cur];
  a.y = d.y[d.cur];
  a.h = d.h[d.cur];
  a.blobs = nullptr;
  a.blobdir = nullptr;
  a.ids = ids_dev;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  ga.g = grid;
  ga.tables = tables_dev;
  ga.exact = exact_dev;
  ga.P = d.P;
  ga.finalize = finalize ? 1 : 0;
  ga.n9 = n9;
  ga.lmpass = fh.lmpass;
  ga.bcount = fh.bcount;
  ga.pflag = fh.pflag;
  ga.n_flagged = fh.n_flagged;
  ga.only_flagged = nullptr;
  const size_t lds = assoc_grid_lds_bytes(grid.ncell, B, n9);
  // bigger workgroups when the LDS tables are large, so that a CU still holds >= 16 waves
  const bool big = lds > 40 * 1024;
  auto go = [&](auto general) {
    constexpr bool G = decltype(general)::value;
    if (n9 > 0) {
      if (big)
        launch_assoc_grid_t<512, true, G>(s, ga, lds, d.P);
      else
        launch_assoc_grid_t<256, true, G>(s, ga, lds, d.P);
    } else {
      if (big)
        launch_assoc_grid_t<512, false, G>(s, ga, lds, d.P);
      else
        launch_assoc_grid_t<256, false, G>(s, ga, lds, d.P);
    }
  };
  if (fh.lmpass) {
    go(std::false_type{});  // S1 + hand-off for every particle
    ga.only_flagged = fh.pflag;
    go(std::true_type{});   // the flagged ones (a landmark with > 2 gate-passing blobs) the general way
  } else {
    go(std::true_type{});
  }
}

// ------------------------------------------------------------------ K3 observe (EKF + weight)
struct ObserveArgs {
  SlotSource ss;
  unsigned char* map_dst;
  size_t slot_bytes, count_off;
  int32_t* src;  // in: slot of particle p in map_src; out: identity
  const double *x, *y;
  double* logw;
  const double* blobs;   // B x 4
  const double* blobdir; // ML: B x 2 unit ray directions (closest_point :510)
  const int32_t* first;  // KNOWN: [L] first blob matched to landmark l, or -1
  const int32_t* next;   // KNOWN: [B] next blob matched to the same landmark, or -1
  int32_t* ids;          // ML: [P x B]; a tentative id whose probability is 0 is reset to 0
  const unsigned char* immutable;
  int n_unmatched;  // KNOWN: blobs with id 0
  const unsigned char* only_flagged;  // when set: skip particles whose flag is 0 (k_observe_fast did them)
  const unsigned* n_flagged;          // with only_flagged: number of flagged particles (0 -> nothing to do)
  int reset;                          // 1: the weight restarts from 1 (prkt_core_v2.py:73) instead of accumulating
  unsigned long long* gmax_key;       // running max of the new log-weights (double_to_key), or NULL
  int L, Lp, B;
  Noise<double> qt;
};

// Is probability_of_match(...) > 0 for a pair that already passed both gates (:433, :441)?
// The association kernel leaves this to us for blobs with a single gate-passing landmark,
// because the landmark's covariance is in registers here.  pr = (500 exp(a1)) (500 exp(a2))
// / 250000 with a1, a2 the two log-pdfs; whenever a1 + a2 is far from the float64 underflow
// edge the answer is known without evaluating a single exp/log; otherwise evaluate it
// exactly as the reference does.  pse = atan2(f.my - sy, f.mx - sx).
__device__ __forceinline__ bool match_is_positive(const Landmark<double>& f, double sx, double sy, double pse,
                                                  const BlobT<double>& z, double ux, double uy) {
  if (fabs(pse - z.bearing) > Consts<double>::half_pi) return false;  // :473-475 -> bp = 0
  double nx, ny;
  closest_point(f.mx, f.my, sx, sy, ux, uy, nx, ny);
  const double ex = nx - f.mx, ey = ny - f.my;
  const double det2 = f.pxx * f.pyy - f.pxy * f.pxy;
  const double maha2 = (f.pyy * ex * ex - 2.0 * f.pxy * ex * ey + f.pxx * ey * ey) / det2;
  double det3;
  const Sym3<double> inv = sym3_inverse(Sym3<double>{f.crr, f.crg, f.crb, f.cgg, f.cgb, f.cbb}, det3);
  const double maha3 = sym3_quad(inv, z.r - f.mr, z.g - f.mg, z.b - f.mb);
  // log det <= 138.2 for det <= 1e60, so a1 + a2 >= -0.5 (9.2 + 276.4 + 800) > -543: no underflow
  if (det2 > 0.0 && det2 < 1e60 && det3 > 0.0 && det3 < 1e60 && maha2 >= 0.0 && maha3 >= 0.0 &&
      maha2 + maha3 < 800.0)
    return true;
  const double bp = 500.0 * exp(-0.5 * (2.0 * Consts<double>::log_two_pi + log(det2) + maha2));
  const double cp = 500.0 * exp(-0.5 * (3.0 * Consts<double>::log_two_pi + log(det3) + maha3));
  return bp * cp / 250000.0 > 0.0;
}

__device__ __forceinline__ BlobT<double> load_blob(const double* blobs, int b) {
  const double2 z01 = *reinterpret_cast<const double2*>(blobs + 4 * (size_t)b);
  const double2 z23 = *reinterpret_cast<const double2*>(blobs + 4 * (size_t)b + 2);
  return BlobT<double>{z01.x, z01.y, z23.x, z23.y};
}

// All blobs matched to landmark l, in scan order (prkt_core_v2.py:88).
// ML: the ids are tentative.  Association saw the state BEFORE any update (:84), so first
// settle every blob of the chain against the untouched state (s_ids[b] = 0 drops it), then
// apply the surviving ones sequentially.
template <bool KNOWN>
__device__ __forceinline__ double apply_blobs(Landmark<double>& lm, int l, double sx, double sy,
                                              const ObserveArgs& a, const int32_t* first,
                                              const int32_t* next, int32_t* s_ids, int32_t* gid) {
  double acc = 0.0;
  const int b0 = first[l];
  if (b0 < 0) return acc;
  const bool imm = a.immutable[l] != 0;
  const double pse = atan2(lm.my - sy, lm.mx - sx);
  if (!KNOWN) {
    for (int b = b0; b >= 0; b = next[b]) {
      const BlobT<double> z = load_blob(a.blobs, b);
      const double2 dir = *reinterpret_cast<const double2*>(a.blobdir + 2 * (size_t)b);
      if (!match_is_positive(lm, sx, sy, pse, z, dir.x, dir.y)) {
        s_ids[b] = 0;
        gid[b] = 0;
      }
    }
  }
  bool fresh = true;  // lm still equals the state pse was computed from
  for (int b = b0; b >= 0; b = next[b]) {
    if (!KNOWN && s_ids[b] == 0) {
      acc += Consts<double>::log_no_match;  // unseen feature: weight *= 0.1 (:94-95)
      continue;
    }
    const BlobT<double> z = load_blob(a.blobs, b);
    acc += ekf_update(lm, sx, sy, z, a.qt, imm, (EkfAux<double>*)nullptr, fresh ? &pse : (const double*)nullptr);
    fresh = imm;
  }
  return acc;
}

template <bool KNOWN, int NV>
__global__ void __launch_bounds__(kObsThreads) k_observe(ObserveArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double red[kObsThreads / kWave];
  const int64_t p = blockIdx.x;
  if (a.only_flagged && (*a.n_flagged == 0u || !a.only_flagged[p])) return;  // workgroup-uniform
  const int tid = threadIdx.x;
  const int32_t sp = a.src[p];
  const unsigned char* sslot = a.ss.at(sp);
  unsigned char* dslot = a.map_dst + (size_t)p * a.slot_bytes;
  const double* sf = reinterpret_cast<const double*>(sslot);
  double* df = reinterpret_cast<double*>(dslot);
  const int* sc = reinterpret_cast<const int*>(sslot + a.count_off);
  int* dc = reinterpret_cast<int*>(dslot + a.count_off);
  const double sx = a.x[p], sy = a.y[p];
  const int Lp = a.Lp;

  const int32_t* first = a.first;
  const int32_t* next = a.next;
  int n_unmatched = a.n_unmatched;
  int32_t* gid_mut = KNOWN ? nullptr : a.ids + (size_t)p * a.B;
  int32_t* s_ids_mut = nullptr;
  if (!KNOWN) {
    // Build the per-particle landmark -> blob chains in LDS from this particle's ids.
    // Blobs are applied in scan order (prkt_core_v2.py:88): first[l] is the lowest blob
    // index matched to l, next[b] the following blob matched to the same landmark.
    int32_t* s_first = reinterpret_cast<int32_t*>(smem);
    int32_t* s_next = s_first + Lp;
    int32_t* s_ids = s_next + a.B;
    s_ids_mut = s_ids;
    const int32_t* gid = gid_mut;
    for (int l = tid; l < Lp; l += blockDim.x) s_first[l] = INT_MAX;
    for (int b = tid; b < a.B; b += blockDim.x) {
      s_ids[b] = gid[b];
      s_next[b] = -1;
    }
    __syncthreads();
    int cnt0 = 0;
    for (int b = tid; b < a.B; b += blockDim.x) {
      int id = s_ids[b];
      if (id > 0)
        atomicMin(&s_first[id - 1], b);
      else
        ++cnt0;
    }
    __syncthreads();
    for (int b = tid; b < a.B; b += blockDim.x) {
      int id = s_ids[b];
      if (id > 0 && s_first[id - 1] != b) {  // not the first sighting: link from my predecessor
        int q = b - 1;
        while (s_ids[q] != id) --q;  // terminates: first[id-1] < b has this id
        s_next[q] = b;
      }
    }
    for (int l = tid; l < Lp; l += blockDim.x)
      if (s_first[l] == INT_MAX) s_first[l] = -1;
    // number of unmatched blobs of this particle (weight *= 0.1 each, :94-95)
    double c = block_sum<kObsThreads / kWave>((double)cnt0, red);
    n_unmatched = (int)c;
    __syncthreads();
    first = s_first;
    next = s_next;
  }

  double acc = 0.0;
  if (NV == 2) {
    // two adjacent landmarks per lane: 16-byte loads/stores, 14 rows x 1 KiB per wave instruction
    for (int l0 = 2 * tid; l0 < Lp; l0 += 2 * kObsThreads) {
      double2 v[F_COUNT_FIELDS];
#pragma unroll
      for (int f = 0; f < F_COUNT_FIELDS; ++f) v[f] = *reinterpret_cast<const double2*>(sf + (size_t)f * Lp + l0);
      int2 c = *reinterpret_cast<const int2*>(sc + l0);
      Landmark<double> A{v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x,
                         v[8].x, v[9].x, v[10].x, v[11].x, v[12].x, v[13].x, c.x};
      Landmark<double> Bq{v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y,
                          v[8].y, v[9].y, v[10].y, v[11].y, v[12].y, v[13].y, c.y};
      if (l0 < a.L) acc += apply_blobs<KNOWN>(A, l0, sx, sy, a, first, next, s_ids_mut, gid_mut);
      if (l0 + 1 < a.L) acc += apply_blobs<KNOWN>(Bq, l0 + 1, sx, sy, a, first, next, s_ids_mut, gid_mut);
      *reinterpret_cast<double2*>(df + (size_t)F_MX * Lp + l0) = make_double2(A.mx, Bq.mx);
      *reinterpret_cast<double2*>(df + (size_t)F_MY * Lp + l0) = make_double2(A.my, Bq.my);
      *reinterpret_cast<double2*>(df + (size_t)F_MR * Lp + l0) = make_double2(A.mr, Bq.mr);
      *reinterpret_cast<double2*>(df + (size_t)F_MG * Lp + l0) = make_double2(A.mg, Bq.mg);
      *reinterpret_cast<double2*>(df + (size_t)F_MB * Lp + l0) = make_double2(A.mb, Bq.mb);
      *reinterpret_cast<double2*>(df + (size_t)F_PXX * Lp + l0) = make_double2(A.pxx, Bq.pxx);
      *reinterpret_cast<double2*>(df + (size_t)F_PXY * Lp + l0) = make_double2(A.pxy, Bq.pxy);
      *reinterpret_cast<double2*>(df + (size_t)F_PYY * Lp + l0) = make_double2(A.pyy, Bq.pyy);
      *reinterpret_cast<double2*>(df + (size_t)F_CRR * Lp + l0) = make_double2(A.crr, Bq.crr);
      *reinterpret_cast<double2*>(df + (size_t)F_CRG * Lp + l0) = make_double2(A.crg, Bq.crg);
      *reinterpret_cast<double2*>(df + (size_t)F_CRB * Lp + l0) = make_double2(A.crb, Bq.crb);
      *reinterpret_cast<double2*>(df + (size_t)F_CGG * Lp + l0) = make_double2(A.cgg, Bq.cgg);
      *reinterpret_cast<double2*>(df + (size_t)F_CGB * Lp + l0) = make_double2(A.cgb, Bq.cgb);
      *reinterpret_cast<double2*>(df + (size_t)F_CBB * Lp + l0) = make_double2(A.cbb, Bq.cbb);
      *reinterpret_cast<int2*>(dc + l0) = make_int2(A.count, Bq.count);
    }
  } else {
    // one landmark per lane: half the registers, twice the waves in flight
    for (int l = tid; l < Lp; l += kObsThreads) {
      Landmark<double> A = load_landmark(sf, sc, Lp, l);
      if (l < a.L) acc += apply_blobs<KNOWN>(A, l, sx, sy, a, first, next, s_ids_mut, gid_mut);
      df[(size_t)F_MX * Lp + l] = A.mx;
      df[(size_t)F_MY * Lp + l] = A.my;
      df[(size_t)F_MR * Lp + l] = A.mr;
      df[(size_t)F_MG * Lp + l] = A.mg;
      df[(size_t)F_MB * Lp + l] = A.mb;
      df[(size_t)F_PXX * Lp + l] = A.pxx;
      df[(size_t)F_PXY * Lp + l] = A.pxy;
      df[(size_t)F_PYY * Lp + l] = A.pyy;
      df[(size_t)F_CRR * Lp + l] = A.crr;
      df[(size_t)F_CRG * Lp + l] = A.crg;
      df[(size_t)F_CRB * Lp + l] = A.crb;
      df[(size_t)F_CGG * Lp + l] = A.cgg;
      df[(size_t)F_CGB * Lp + l] = A.cgb;
      df[(size_t)F_CBB * Lp + l] = A.cbb;
      dc[l] = A.count;
    }
  }
  double tot = block_sum<kObsThreads / kWave>(acc, red);
  if (tid == 0) {
    const double v = (a.reset ? 0.0 : a.logw[p]) + tot + (double)n_unmatched * Consts<double>::log_no_match;
    a.logw[p] = v;
    if (a.gmax_key) atomicMax(a.gmax_key + (p & (kGmaxKeys - 1)), double_to_key(v));  // sharded: same-address atomics serialise
    a.src[p] = (int32_t)p;
  }
}


// ------------------------------------------------------------------ K3 (fast ML variant, L <= 512)
// One workgroup per particle, two adjacent landmarks per lane, the particle's whole map in
// registers from the single coalesced load to the single coalesced store.  Input is the
// association kernel's hand-off: per landmark the (<= 2) blobs that pass its gates, per blob
// the number of landmarks that pass.  A blob passed by one landmark is matched iff its
// probability is > 0 (strict '>' from 0.0, :369-381); a blob passed by several is given to
// the landmark with the largest probability, the earliest on a tie -- LDS atomicMax on the
// probability bits, then atomicMin on the landmark index among those that attain it --
// evaluated here because the covariances are already in registers.  Blobs nobody passes or
// wins multiply the weight by 0.1 (:94-95).  Updates of one landmark are applied in scan
// order (:88) and every probability refers to the state before any update (:84).
struct FastArgs {
  SlotSource ss;
  unsigned char* map_dst;
  size_t count_off;
  int32_t* src;
  const double *x, *y;
  double* logw;
  const double* exact;          // [B][6] cell order: bearing, r, g, b, ux, uy
  const unsigned short* order;  // [B] cell order -> scan order
  const uint4* lmpass;
  const unsigned char* bcount;
  const unsigned char* pflag;
  const unsigned char* immutable;
  int L, Lp, B;
  int reset;
  unsigned long long* gmax_key;
  Noise<double> qt;
};

struct FastSlot {
  int t;                    // blob (cell order) or -1
  int b;                    // its scan index
  unsigned long long bits;  // contested candidate: probability bits (0: not positive)
  unsigned flags;           // bit 0 contested, bit 1 apply the update, bit 2 unmatched (single, probability 0)
};

// What probability_of_match needs from the landmark alone, computed once per landmark instead
// of once per blob: determinant / inverse of the 2x2 position block and of the 3x3 colour block.
struct FastLm {
  double det2, idet2, det3;
  Sym3<double> inv3;
};

__device__ __forceinline__ void fast_prepare(const FastArgs& a, const Landmark<double>& lm, double sx, double sy,
                                             double pse, uint2 packed, const unsigned char* bc,
                                             unsigned long long* best, FastSlot (&sl)[kFastSlots]) {
  FastLm q;
  q.det2 = lm.pxx * lm.pyy - lm.pxy * lm.pxy;
  q.idet2 = 1.0 / q.det2;
  q.inv3 = sym3_inverse(Sym3<double>{lm.crr, lm.crg, lm.crb, lm.cgg, lm.cgb, lm.cbb}, q.det3);
  const bool dets_sane = q.det2 > 0.0 && q.det2 < 1e60 && q.det3 > 0.0 && q.det3 < 1e60;
  double ldet2 = 0.0, ldet3 = 0.0;  // log determinants, evaluated once per landmark on first use
  bool have_logs = false;
  const unsigned w[2] = {packed.x, packed.y};
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    const int t = (int)((w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
    sl[k].t = t == 0xFFFF ? -1 : t;
    sl[k].b = INT_MAX;
    sl[k].bits = 0ull;
    sl[k].flags = 0u;
    if (sl[k].t < 0) continue;
    sl[k].b = a.order[t];
    const double* rec = a.exact + 6 * (size_t)t;
    const double2 z01 = *reinterpret_cast<const double2*>(rec);
    const double2 z23 = *reinterpret_cast<const double2*>(rec + 2);
    const double2 dir = *reinterpret_cast<const double2*>(rec + 4);
    // the quantities both branches need (prob_position_match :457-494, prob_color_match :524-544)
    const bool angle_ok = !(fabs(pse - z01.x) > Consts<double>::half_pi);  // :473-475
    double nx, ny;
    closest_point(lm.mx, lm.my, sx, sy, dir.x, dir.y, nx, ny);
    const double ex = nx - lm.mx, ey = ny - lm.my;
    const double maha2 = (lm.pyy * ex * ex - 2.0 * lm.pxy * ex * ey + lm.pxx * ey * ey) / q.det2;
    const double maha3 = sym3_quad(q.inv3, z01.y - lm.mr, z23.x - lm.mg, z23.y - lm.mb);
    const bool contested = bc[t] >= 2;
    // pr = (500 exp(a1)) (500 exp(a2)) / 250000 is certainly > 0 when a1 + a2 is far from the
    // float64 underflow edge: log det <= 138.2 for det <= 1e60, so a1 + a2 > -543 here
    const bool surely_positive = angle_ok && dets_sane && maha2 >= 0.0 && maha3 >= 0.0 && maha2 + maha3 < 800.0;
    double pr = 0.0;
    if (contested || (angle_ok && !surely_positive)) {
      if (angle_ok) {
        if (!have_logs) {
          ldet2 = log(q.det2);
          ldet3 = log(q.det3);
          have_logs = true;
        }
        const double bp = 500.0 * exp(-0.5 * (2.0 * Consts<double>::log_two_pi + ldet2 + maha2));  // :439
        const double cp = 500.0 * exp(-0.5 * (3.0 * Consts<double>::log_two_pi + ldet3 + maha3));  // :446
        pr = bp * cp / 250000.0;                                                                 // :455
      }
    }
    if (contested) {
      sl[k].flags = 1u;
      if (pr > 0.0) {
        sl[k].bits = (unsigned long long)__double_as_longlong(pr);
        atomicMax(&best[t], sl[k].bits);
      }
    } else {
      sl[k].flags = (surely_positive || pr > 0.0) ? 2u : 4u;
    }
  }
}

__device__ __forceinline__ double fast_apply(const FastArgs& a, Landmark<double>& lm, int l, double sx, double sy,
                                             double pse, FastSlot (&sl)[kFastSlots], const int* win) {
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    if (sl[k].t < 0) continue;
    if ((sl[k].flags & 1u) && sl[k].bits != 0ull && win[sl[k].t] == l) sl[k].flags |= 2u;
    if (sl[k].flags & 4u) acc += Consts<double>::log_no_match;  // single candidate, probability 0 (:94-95)
    if (!(sl[k].flags & 2u)) sl[k].b = INT_MAX;                 // not applied: sorts to the back
  }
  // the blobs to apply first, in scan order (:88) -- so that nearly every lane of the wave
  // applies its (usually only) update in the same iteration (5-comparator network)
  auto cswap = [&](FastSlot& u, FastSlot& v) {
    if (u.b > v.b) {
      const FastSlot tmp = u;
      u = v;
      v = tmp;
    }
  };
  cswap(sl[0], sl[1]);
  cswap(sl[2], sl[3]);
  cswap(sl[0], sl[2]);
  cswap(sl[1], sl[3]);
  cswap(sl[1], sl[2]);
  const bool imm = a.immutable[l] != 0;
  bool fresh = true;
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    if (sl[k].b == INT_MAX) continue;
    const double* rec = a.exact + 6 * (size_t)sl[k].t;
    const double2 z01 = *reinterpret_cast<const double2*>(rec);
    const double2 z23 = *reinterpret_cast<const double2*>(rec + 2);
    BlobT<double> z{z01.x, z01.y, z23.x, z23.y};
    acc += ekf_update(lm, sx, sy, z, a.qt, imm, (EkfAux<double>*)nullptr, fresh ? &pse : (const double*)nullptr);
    fresh = imm;
  }
  return acc;
}

constexpr int kFastThreads = 512;  // one landmark per lane: L <= 512 in one pass

__global__ void __launch_bounds__(kFastThreads) k_observe_fast(FastArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double red[kFastThreads / kWave];
  const int64_t p = blockIdx.x;
  if (a.pflag[p]) return;  // workgroup-uniform: the general kernel takes this particle
  const int tid = threadIdx.x;
  const int B = a.B, Lp = a.Lp;
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);
  int* win = reinterpret_cast<int*>(best + B);
  unsigned char* bc = reinterpret_cast<unsigned char*>(win + B);
  const unsigned char* sslot = a.ss.at(a.src[p]);
  unsigned char* dslot = a.map_dst + (size_t)p * a.ss.slot_bytes;
  const double* sf = reinterpret_cast<const double*>(sslot);
  double* df = reinterpret_cast<double*>(dslot);
  const int* sc = reinterpret_cast<const int*>(sslot + a.count_off);
  int* dc = reinterpret_cast<int*>(dslot + a.count_off);
  const double sx = a.x[p], sy = a.y[p];
  const int l = tid;
  const bool active = l < Lp, has = l < a.L;
  Landmark<double> A{};
  uint4 lp = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
  if (active) {
    A = load_landmark(sf, sc, Lp, l);
    lp = a.lmpass[(size_t)p * Lp + l];
  }
  for (int t = tid; t < B; t += kFastThreads) {
    best[t] = 0ull;
    win[t] = INT_MAX;
    bc[t] = a.bcount[(size_t)p * B + t];
  }
  __syncthreads();
  int nun = 0;  // blobs no landmark passes
  for (int t = tid; t < B; t += kFastThreads) nun += bc[t] == 0;
  FastSlot sa[kFastSlots];
  // atan2(my - sy, mx - sx) of the untouched state, handed over by the association kernel
  const double pseA = __longlong_as_double((long long)(((unsigned long long)lp.w << 32) | lp.z));
  fast_prepare(a, A, sx, sy, pseA, has ? make_uint2(lp.x, lp.y) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu), bc, best, sa);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k)
    if (sa[k].t >= 0 && sa[k].bits != 0ull && sa[k].bits == best[sa[k].t]) atomicMin(&win[sa[k].t], l);
  __syncthreads();
  for (int t = tid; t < B; t += kFastThreads) nun += (bc[t] >= 2 && best[t] == 0ull);  // contested, all 0
  double acc = (double)nun * Consts<double>::log_no_match;
  if (has) acc += fast_apply(a, A, l, sx, sy, pseA, sa, win);
  if (active) {
    df[(size_t)F_MX * Lp + l] = A.mx;
    df[(size_t)F_MY * Lp + l] = A.my;
    df[(size_t)F_MR * Lp + l] = A.mr;
    df[(size_t)F_MG * Lp + l] = A.mg;
    df[(size_t)F_MB * Lp + l] = A.mb;
    df[(size_t)F_PXX * Lp + l] = A.pxx;
    df[(size_t)F_PXY * Lp + l] = A.pxy;
    df[(size_t)F_PYY * Lp + l] = A.pyy;
    df[(size_t)F_CRR * Lp + l] = A.crr;
    df[(size_t)F_CRG * Lp + l] = A.crg;
    df[(size_t)F_CRB * Lp + l] = A.crb;
    df[(size_t)F_CGG * Lp + l] = A.cgg;
    df[(size_t)F_CGB * Lp + l] = A.cgb;
    df[(size_t)F_CBB * Lp + l] = A.cbb;
    dc[l] = A.count;
  }
  const double tot = block_sum<kFastThreads / kWave>(acc, red);
  if (tid == 0) {
    const double v = (a.reset ? 0.0 : a.logw[p]) + tot;
    a.logw[p] = v;
    if (a.gmax_key) atomicMax(a.gmax_key + (p & (kGmaxKeys - 1)), double_to_key(v));  // sharded: same-address atomics serialise
    a.src[p] = (int32_t)p;
  }
}

void launch_observe_fast(hipStream_t s, DeviceState& d, int B, const double* exact_dev,
                         const unsigned short* order_dev, const FastHandoff& fh, const NoiseD& qt,
                         const ObserveExtras& ex) {
  if (d.P == 0) return;
  FastArgs a;
  a.ss = slot_source(d);
  a.map_dst = d.map[d.mcur ^ 1];
  a.count_off = d.lay.count_off;
  a.src = d.src[d.cur];
  a.x = d.x[d.cur];
  a.y = d.y[d.cur];
  a.logw = d.logw[d.cur];
  a.exact = exact_dev;
  a.order = order_dev;
  a.lmpass = fh.lmpass;
  a.bcount = fh.bcount;
  a.pflag = fh.pflag;
  a.immutable = d.immutable;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  a.reset = ex.reset ? 1 : 0;
  a.gmax_key = ex.gmax_key;
  a.qt = Noise<double>{qt.q00, qt.rr, qt.rg, qt.rb, qt.gg, qt.gb, qt.bb};
  const size_t lds = (size_t)B * 13 + 16;
  hipLaunchKernelGGL(k_observe_fast, dim3((unsigned)d.P), dim3(kFastThreads), lds, s, a);
}

int g_observe_nv = 0;  // tuning: 0 = default per variant, 1 / 2 = landmarks per lane

void launch_observe(hipStream_t s, DeviceState& d, const double* blobs_dev, const double* blobdir_dev, int B,
                    const int32_t* first_dev, const int32_t* next_dev, int n_unmatched, int32_t* ids_dev,
                    const NoiseD& qt, const ObserveExtras& ex) {
  if (d.P == 0) return;
  ObserveArgs a;
  a.ss = slot_source(d);
  a.map_dst = d.map[d.mcur ^ 1];
  a.slot_bytes = d.lay.slot_bytes;
  a.count_off = d.lay.count_off;
  a.src = d.src[d.cur];
  a.x = d.x[d.cur];
  a.y = d.y[d.cur];
  a.logw = d.logw[d.cur];
  a.blobs = blobs_dev;
  a.blobdir = blobdir_dev;
  a.first = first_dev;
  a.next = next_dev;
  a.ids = ids_dev;
  a.immutable = d.immutable;
  a.n_unmatched = n_unmatched;
  a.only_flagged = ex.only_flagged;
  a.n_flagged = ex.n_flagged;
  a.reset = ex.reset ? 1 : 0;
  a.gmax_key = ex.gmax_key;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  a.qt = Noise<double>{qt.q00, qt.rr, qt.rg, qt.rb, qt.gg, qt.gb, qt.bb};
  if (ids_dev == nullptr) {
    if (g_observe_nv == 1)
      hipLaunchKernelGGL((k_observe<true, 1>), dim3((unsigned)d.P), dim3(kObsThreads), 0, s, a);
    else
      hipLaunchKernelGGL((k_observe<true, 2>), dim3((unsigned)d.P), dim3(kObsThreads), 0, s, a);
  } else {
    size_t lds = sizeof(int32_t) * ((size_t)d.lay.Lp + 2 * (size_t)B);
    if (g_observe_nv == 2)
      hipLaunchKernelGGL((k_observe<false, 2>), dim3((unsigned)d.P), dim3(kObsThreads), lds, s, a);
    else
      hipLaunchKernelGGL((k_observe<false, 1>), dim3((unsigned)d.P), dim3(kObsThreads), lds, s, a);
  }
  if (ex.flip) {
    d.mcur ^= 1;
    d.alt = nullptr;  // every slot was rewritten into the particle's own map buffer
  }
}

// ------------------------------------------------------------------ K4 weights
__global__ void __launch_bounds__(256) k_block_max(const double* __restrict__ logw, int64_t P,
                                                   double* __restrict__ partial) {
  __shared__ double red[4];
  double m = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x)
    m = fmax(m, logw[i]);
  m = block_max<4>(m, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = m;
}
__global__ void __launch_bounds__(256) k_final_max(const double* __restrict__ partial, int n,
                                                   double* __restrict__ out) {
  __shared__ double red[4];
  double m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmax(m, partial[i]);
  m = block_max<4>(m, red);
  if (threadIdx.x == 0) out[0] = m;
}
constexpr int kRedBlocks = 1024;
void launch_block_max(hipStream_t s, DeviceState& d, double* partial_dev, double* gmax_dev) {
  int nb = (int)((d.P + 255) / 256);
  if (nb > kRedBlocks) nb = kRedBlocks;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_block_max, dim3(nb), dim3(256), 0, s, d.logw[d.cur], d.P, partial_dev);
  hipLaunchKernelGGL(k_final_max, dim3(1), dim3(256), 0, s, partial_dev, nb, gmax_dev);
}

// Block-local inclusive scan of w = exp(logw - shift) over kScanBlock particles:
// 4 consecutive particles per thread (sequential), Kogge-Stone over the wave with
// __shfl_up, sequential over the 4 waves.  The association order is fixed by the block
// size alone, so shards that are multiples of kScanBlock reproduce the 1-GPU bits.
__global__ void __launch_bounds__(256) k_scan_local(const double* __restrict__ logw, int64_t P,
                                                    const double* __restrict__ gmax, int domain,
                                                    double* __restrict__ clocal, double* __restrict__ totals,
                                                    const unsigned long long* __restrict__ gmax_key) {
  __shared__ double wtot[4];
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  double shift = 0.0;
  if (domain == 1) {
    if (gmax_key) {  // max over the sharded running-max keys (kGmaxKeys == blockDim.x)
      const double mine = key_to_double(gmax_key[tid]);
      shift = block_max<4>(mine, wtot);
      __syncthreads();
    } else {
      shift = gmax[0];
    }
    if (!(shift > -INFINITY)) shift = 0.0;  // all weights zero: keep exp(-inf) = 0, not NaN
  }
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + 4 * tid;
  double w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (base + i < P) ? exp(logw[base + i] - shift) : 0.0;
  double s0 = w[0], s1 = s0 + w[1], s2 = s1 + w[2], s3 = s2 + w[3];
  double val = s3;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    double t = __shfl_up(val, off, kWave);
    if (lane >= off) val += t;
  }
  if (lane == kWave - 1) wtot[wave] = val;
  double prev = __shfl_up(val, 1, kWave);  // exclusive prefix inside the wave
  if (lane == 0) prev = 0.0;
  __syncthreads();
  double woff = 0.0;
  for (int i = 0; i < wave; ++i) woff += wtot[i];
  const double excl = woff + prev;
  if (base < P) clocal[base] = excl + s0;
  if (base + 1 < P) clocal[base + 1] = excl + s1;
  if (base + 2 < P) clocal[base + 2] = excl + s2;
  if (base + 3 < P) clocal[base + 3] = excl + s3;
  // the block total IS the inclusive value of its last particle (bit for bit): shards hand
  // over at block boundaries and both sides must see the same cumulative weight there
  if (tid == 255) totals[blockIdx.x] = excl + s3;
}
void launch_scan_local(hipStream_t s, DeviceState& d, const double* gmax_dev, int domain, double* clocal_dev,
                       double* totals_dev, const unsigned long long* gmax_key_dev) {
  if (d.P == 0) return;
  int nb = (int)((d.P + kScanBlock - 1) / kScanBlock);
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, s, d.logw[d.cur], d.P, gmax_dev, domain,
                     clocal_dev, totals_dev, gmax_key_dev);
}

// Exclusive scan of the block totals in block order by ONE thread: the canonical
// (shard-count independent) association of the global prefix sum.
__global__ void __launch_bounds__(256) k_scan_blocks(const double* __restrict__ totals, int64_t nb,
                                                     double* __restrict__ offsets, double* __restrict__ sum) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* t = reinterpret_cast<double*>(smem);
  double run = 0.0;
  for (int64_t c0 = 0; c0 < nb; c0 += 2048) {
    int n = (int)((nb - c0 < 2048) ? (nb - c0) : 2048);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) t[i] = totals[c0 + i];
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 0; i < n; ++i) {
        double v = t[i];
        t[i] = run;
        run += v;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) offsets[c0 + i] = t[i];
  }
  if (threadIdx.x == 0) sum[0] = run;
}
void launch_scan_blocks(hipStream_t s, const double* totals_dev, int64_t nb, double* offsets_dev,
                        double* sum_dev) {
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 2048 * sizeof(double), s, totals_dev, nb, offsets_dev,
                     sum_dev);
}

// Ancestor of output slot k: first particle j whose inclusive cumulative weight C_j is
// >= u*r + k*r, r = sum/P  (equivalent to the walk at prkt_core_v2.py:233-250, '<=' at :239).
__global__ void __launch_bounds__(256) k_ancestors(const double* __restrict__ clocal,
                                                   const double* __restrict__ totals,
                                                   const double* __restrict__ offsets,
                                                   const double* __restrict__ sum, int64_t nb, int64_t Pg,
                                                   int64_t Pscan, double u, int64_t slot0, int64_t n,
                                                   int32_t* __restrict__ anc, const double* __restrict__ gx,
                                                   const double* __restrict__ gy, const double* __restrict__ gh,
                                                   const double* __restrict__ glw, const int32_t* __restrict__ gsrc,
                                                   double* __restrict__ gx2, double* __restrict__ gy2,
                                                   double* __restrict__ gh2, double* __restrict__ glw2,
                                                   int32_t* __restrict__ gsrc2) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double r = __ddiv_rn(sum[0], (double)Pg);                  // range_ :225
  const double t = __dadd_rn(__dmul_rn(u, r), __dmul_rn((double)(slot0 + k), r));  // step :226 + k*range_
  // block: first b with offsets[b] + totals[b] >= t
  int64_t lo = 0, hi = nb - 1;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (__dadd_rn(offsets[mid], totals[mid]) >= t)
      hi = mid;
    else
      lo = mid + 1;
  }
  const int64_t b = lo;
  const double off = offsets[b];
  int64_t j0 = b * kScanBlock, j1 = j0 + kScanBlock - 1;
  if (j1 > Pscan - 1) j1 = Pscan - 1;
  while (j0 < j1) {
    int64_t mid = (j0 + j1) >> 1;
    if (__dadd_rn(off, clocal[mid]) >= t)
      j1 = mid;
    else
      j0 = mid + 1;
  }
  anc[k] = (int32_t)j0;
  if (gx) {  // fused k_gather_poses (single-GPU resample)
    const int32_t a = (int32_t)j0;
    gx2[k] = gx[a];
    gy2[k] = gy[a];
    gh2[k] = gh[a];
    glw2[k] = glw[a];
    gsrc2[k] = gsrc[a];
  }
}
void launch_ancestors(hipStream_t s, const double* clocal_dev, const double* totals_dev,
                      const double* offsets_dev, const double* sum_dev, int64_t nb, int64_t P_global,
                      int64_t P_scan, double u, int64_t slot0, int64_t n, int32_t* anc_dev, DeviceState* gather) {
  if (n == 0) return;
  if (gather) {
    DeviceState& d = *gather;
    const int c = d.cur, m = c ^ 1;
    hipLaunchKernelGGL(k_ancestors, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, clocal_dev, totals_dev,
                       offsets_dev, sum_dev, nb, P_global, P_scan, u, slot0, n, anc_dev, d.x[c], d.y[c], d.h[c],
                       d.logw[c], d.src[c], d.x[m], d.y[m], d.h[m], d.logw[m], d.src[m]);
    d.cur = m;
    return;
  }
  hipLaunchKernelGGL(k_ancestors, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, clocal_dev, totals_dev,
                     offsets_dev, sum_dev, nb, P_global, P_scan, u, slot0, n, anc_dev, (const double*)nullptr,
                     (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (const int32_t*)nullptr,
                     (double*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr, (int32_t*)nullptr);
}

// ------------------------------------------------------------------ K6 summary
__global__ void __launch_bounds__(256) k_summary_partials(const double* __restrict__ x,
                                                          const double* __restrict__ y,
                                                          const double* __restrict__ h, int64_t P,
                                                          double* __restrict__ partial) {
  __shared__ double red[4];
  double sx = 0, sy = 0, ss = 0, sc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    sx += x[i];
    sy += y[i];
    double s, c;
    sincos(h[i], &s, &c);
    ss += s;
    sc += c;
  }
  sx = block_sum<4>(sx, red);
  sy = block_sum<4>(sy, red);
  ss = block_sum<4>(ss, red);
  sc = block_sum<4>(sc, red);
  if (threadIdx.x == 0) {
    partial[4 * blockIdx.x + 0] = sx;
    partial[4 * blockIdx.x + 1] = sy;
    partial[4 * blockIdx.x + 2] = ss;
    partial[4 * blockIdx.x + 3] = sc;
  }
}
__global__ void __launch_bounds__(256) k_summary_final(const double* __restrict__ partial, int n,
                                                       double* __restrict__ out4) {
  __shared__ double red[4];
  double v[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    for (int c = 0; c < 4; ++c) v[c] += partial[4 * i + c];
  for (int c = 0; c < 4; ++c) {
    double t = block_sum<4>(v[c], red);
    if (threadIdx.x == 0) out4[c] = t;
  }
}
void launch_summary_partials(hipStream_t s, DeviceState& d, double* partial_dev, double* out4_dev) {
  int nb = (int)((d.P + 255) / 256);
  if (nb > kRedBlocks) nb = kRedBlocks;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_summary_partials, dim3(nb), dim3(256), 0, s, d.x[d.cur], d.y[d.cur], d.h[d.cur], d.P,
                     partial_dev);
  hipLaunchKernelGGL(k_summary_final, dim3(1), dim3(256), 0, s, partial_dev, nb, out4_dev);
}

// ------------------------------------------------------------------ map maintenance
// dst slot p <- src slot src[p]; then src <- identity.  Streaming 16-byte copy.
__global__ void __launch_bounds__(256) k_copy_slots(SlotSource ss, unsigned char* __restrict__ mdst,
                                                    int32_t* __restrict__ src, int fixed_src) {
  const int64_t p = blockIdx.x;
  const uint4* s = reinterpret_cast<const uint4*>(fixed_src ? ss.map : ss.at(src[p]));
  uint4* d = reinterpret_cast<uint4*>(mdst + (size_t)p * ss.slot_bytes);
  const size_t n = ss.slot_bytes / 16;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
  if (!fixed_src) {
    __syncthreads();
    if (threadIdx.x == 0) src[p] = (int32_t)p;
  }
}
void launch_materialise(hipStream_t s, DeviceState& d) {
  if (d.P == 0) return;
  hipLaunchKernelGGL(k_copy_slots, dim3((unsigned)d.P), dim3(256), 0, s, slot_source(d), d.map[d.mcur ^ 1],
                     d.src[d.cur], 0);
  d.mcur ^= 1;
  d.alt = nullptr;
}
void launch_broadcast_slot(hipStream_t s, DeviceState& d, const unsigned char* slot_dev) {
  if (d.P == 0) return;
  SlotSource one{slot_dev, d.lay.slot_bytes, nullptr, 0, 0};
  hipLaunchKernelGGL(k_copy_slots, dim3((unsigned)d.P), dim3(256), 0, s, one, d.map[d.mcur], d.src[d.cur], 1);
  launch_iota(s, d.src[d.cur], d.P);
  d.alt = nullptr;
}


// ------------------------------------------------------------------ sharded resample
// Owner-computes offspring: with C_j the global inclusive cumulative weight of local
// particle j and t_k = u r + k r the comb (same expressions as k_ancestors), particle j fills
// the output slots [hi_{j-1}, hi_j), hi_j = #{k : t_k <= C_j}.  hi[0] is the count at the
// shard's lower boundary, hi[1 + j] that of particle j.  Every shard evaluates the same
// formula on the same block offsets, so the slot ranges tile [0, P) without communication.
__global__ void __launch_bounds__(256) k_offspring(const double* __restrict__ clocal,
                                                   const double* __restrict__ offsets,
                                                   const double* __restrict__ sum, int64_t first_block,
                                                   int64_t Pl, int64_t Pg, double u, int last_shard,
                                                   int64_t* __restrict__ hi) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. Pl
  if (i > Pl) return;
  const double r = __ddiv_rn(sum[0], (double)Pg);
  const double ur = __dmul_rn(u, r);
  double C;
  if (i == 0) {
    if (first_block == 0) {
      hi[0] = 0;
      return;
    }
    C = offsets[first_block];
  } else {
    const int64_t j = i - 1;
    if (last_shard && j == Pl - 1) {  // the tail is clamped to the last particle, as k_ancestors does
      hi[i] = Pg;
      return;
    }
    C = __dadd_rn(offsets[first_block + j / kScanBlock], clocal[j]);
  }
  int64_t lo = 0, up = Pg;  // first k with t_k > C
  while (lo < up) {
    const int64_t mid = (lo + up) >> 1;
    const double t = __dadd_rn(ur, __dmul_rn((double)mid, r));
    if (t > C)
      up = mid;
    else
      lo = mid + 1;
  }
  hi[i] = lo;
}
void launch_offspring(hipStream_t s, const double* clocal_dev, const double* offsets_dev, const double* sum_dev,
                      int64_t first_block, int64_t P_local, int64_t P_global, double u, int last_shard,
                      int64_t* hi_dev) {
  hipLaunchKernelGGL(k_offspring, dim3((unsigned)((P_local + 1 + 255) / 256)), dim3(256), 0, s, clocal_dev,
                     offsets_dev, sum_dev, first_block, P_local, P_global, u, last_shard, hi_dev);
}

// record = (x, y, h, logw) + map slot
__global__ void __launch_bounds__(256) k_pack(SlotSource ss, const int32_t* __restrict__ src,
                                              const double* __restrict__ x, const double* __restrict__ y,
                                              const double* __restrict__ h, const double* __restrict__ lw,
                                              const int64_t* __restrict__ idx, unsigned char* __restrict__ buf) {
  const int64_t i = blockIdx.x;
  const int64_t j = idx[i];
  unsigned char* rec = buf + (size_t)i * (kPoseRecordBytes + ss.slot_bytes);
  if (threadIdx.x == 0) {
    double* hd = reinterpret_cast<double*>(rec);
    hd[0] = x[j];
    hd[1] = y[j];
    hd[2] = h[j];
    hd[3] = lw[j];
    hd[4] = 0.0;
    hd[5] = 0.0;
  }
  const uint4* s = reinterpret_cast<const uint4*>(ss.at(src[j]));
  uint4* d = reinterpret_cast<uint4*>(rec + kPoseRecordBytes);
  const size_t n = ss.slot_bytes / 16;
  for (size_t k = threadIdx.x; k < n; k += blockDim.x) d[k] = s[k];
}
void launch_pack(hipStream_t s, DeviceState& d, const int64_t* idx_dev, int64_t n, unsigned char* buf_dev) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_pack, dim3((unsigned)n), dim3(256), 0, s, slot_source(d), d.src[d.cur], d.x[d.cur],
                     d.y[d.cur], d.h[d.cur], d.logw[d.cur], idx_dev, buf_dev);
}

// New generation of the shard: slot k <- local particle srcs[k] (>= 0) or received record
// -(srcs[k]) - 1.  Poses are gathered now; maps stay where they are (own buffer / record)
// until the next observe rewrites them.
__global__ void __launch_bounds__(256) k_adopt(const double* __restrict__ x, const double* __restrict__ y,
                                               const double* __restrict__ h, const double* __restrict__ lw,
                                               const int32_t* __restrict__ src, double* __restrict__ x2,
                                               double* __restrict__ y2, double* __restrict__ h2,
                                               double* __restrict__ lw2, int32_t* __restrict__ src2,
                                               const int64_t* __restrict__ srcs, const unsigned char* __restrict__ buf,
                                               size_t stride, int64_t P) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const int64_t a = srcs[k];
  if (a >= 0) {
    x2[k] = x[a];
    y2[k] = y[a];
    h2[k] = h[a];
    lw2[k] = lw[a];
    src2[k] = src[a];
  } else {
    const double* hd = reinterpret_cast<const double*>(buf + (size_t)(-(a + 1)) * stride);
    x2[k] = hd[0];
    y2[k] = hd[1];
    h2[k] = hd[2];
    lw2[k] = hd[3];
    src2[k] = (int32_t)a;
  }
}
void launch_adopt(hipStream_t s, DeviceState& d, const int64_t* src_dev, const unsigned char* buf_dev) {
  if (d.P == 0) return;
  const int c = d.cur, n = c ^ 1;
  const size_t stride = kPoseRecordBytes + d.lay.slot_bytes;
  hipLaunchKernelGGL(k_adopt, dim3((unsigned)((d.P + 255) / 256)), dim3(256), 0, s, d.x[c], d.y[c], d.h[c],
                     d.logw[c], d.src[c], d.x[n], d.y[n], d.h[n], d.logw[n], d.src[n], src_dev, buf_dev, stride,
                     d.P);
  d.cur = n;
  d.alt = buf_dev;
  d.alt_stride = stride;
  d.alt_off = kPoseRecordBytes;
}


// ---- device-resident exchange ------------------------------------------------------------
// ranges[2 d], ranges[2 d + 1] = [j0, j1): the local particles whose offspring overlap the
// output slots [d P, (d + 1) P) of rank d (hi from k_offspring, monotone).
__global__ void k_shard_ranges(const int64_t* __restrict__ hi, int64_t P, int world, int64_t* __restrict__ ranges) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= world) return;
  const int64_t start = (int64_t)d * P, end = start + P;
  int64_t lo = 0, up = P;  // first j with hi[j + 1] > start
  while (lo < up) {
    const int64_t mid = (lo + up) >> 1;
    if (hi[mid + 1] > start)
      up = mid;
    else
      lo = mid + 1;
  }
  const int64_t j0 = lo;
  lo = 0;
  up = P;  // first j with hi[j] >= end
  while (lo < up) {
    const int64_t mid = (lo + up) >> 1;
    if (hi[mid] >= end)
      up = mid;
    else
      lo = mid + 1;
  }
  ranges[2 * d] = j0;
  ranges[2 * d + 1] = lo < j0 ? j0 : lo;
}
void launch_shard_ranges(hipStream_t s, const int64_t* hi_dev, int64_t P_local, int world, int64_t* ranges_dev) {
  hipLaunchKernelGGL(k_shard_ranges, dim3((world + 63) / 64), dim3(64), 0, s, hi_dev, P_local, world, ranges_dev);
}

// Records of the contiguous local particles [j0, j0 + n) for one destination; the header
// carries the destination slots [lo, hi) each copy fills (empty for particles without
// offspring there: they ride along, the receiver skips them).
__global__ void __launch_bounds__(256) k_pack_range(SlotSource ss, const int32_t* __restrict__ src,
                                                    const double* __restrict__ x, const double* __restrict__ y,
                                                    const double* __restrict__ h, const double* __restrict__ lw,
                                                    const int64_t* __restrict__ hi, int64_t j0, int64_t slot_start,
                                                    int64_t slot_end, unsigned char* __restrict__ buf) {
  const int64_t i = blockIdx.x;
  const int64_t j = j0 + i;
  unsigned char* rec = buf + (size_t)i * (kPoseRecordBytes + ss.slot_bytes);
  if (threadIdx.x == 0) {
    double* hd = reinterpret_cast<double*>(rec);
    hd[0] = x[j];
    hd[1] = y[j];
    hd[2] = h[j];
    hd[3] = lw[j];
    int64_t lo = hi[j] > slot_start ? hi[j] : slot_start;
    int64_t up = hi[j + 1] < slot_end ? hi[j + 1] : slot_end;
    if (up < lo) up = lo;
    reinterpret_cast<int64_t*>(rec)[4] = lo;
    reinterpret_cast<int64_t*>(rec)[5] = up;
  }
  const uint4* s = reinterpret_cast<const uint4*>(ss.at(src[j]));
  uint4* d = reinterpret_cast<uint4*>(rec + kPoseRecordBytes);
  const size_t n = ss.slot_bytes / 16;
  for (size_t k = threadIdx.x; k < n; k += blockDim.x) d[k] = s[k];
}
void launch_pack_range(hipStream_t s, DeviceState& d, const int64_t* hi_dev, int64_t j0, int64_t n, int64_t slot_start,
                       int64_t slot_end, unsigned char* buf_dev) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_pack_range, dim3((unsigned)n), dim3(256), 0, s, slot_source(d), d.src[d.cur], d.x[d.cur],
                     d.y[d.cur], d.h[d.cur], d.logw[d.cur], hi_dev, j0, slot_start, slot_end, buf_dev);
}

__global__ void __launch_bounds__(256) k_extract_lohi(const unsigned char* __restrict__ buf, size_t stride, int64_t n,
                                                      int64_t* __restrict__ rlohi) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t* hd = reinterpret_cast<const int64_t*>(buf + (size_t)r * stride);
  rlohi[2 * r] = hd[4];
  rlohi[2 * r + 1] = hd[5];
}

// New generation from the device-side plan: local slot k (global slot K) takes the local
// particle j with hi[j] <= K < hi[j + 1] if K lies in this shard's offspring range, else the
// received record whose [lo, hi) holds K (records arrive in slot order).
__global__ void __launch_bounds__(256) k_adopt_dev(const double* __restrict__ x, const double* __restrict__ y,
                                                   const double* __restrict__ h, const double* __restrict__ lw,
                                                   const int32_t* __restrict__ src, double* __restrict__ x2,
                                                   double* __restrict__ y2, double* __restrict__ h2,
                                                   double* __restrict__ lw2, int32_t* __restrict__ src2,
                                                   const int64_t* __restrict__ hi, int64_t slot_start,
                                                   const unsigned char* __restrict__ buf, size_t stride,
                                                   const int64_t* __restrict__ rlohi, int64_t n_recv, int64_t P) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const int64_t K = slot_start + k;
  if (K >= hi[0] && K < hi[P]) {
    int64_t lo = 0, up = P - 1;  // first j with hi[j + 1] > K
    while (lo < up) {
      const int64_t mid = (lo + up) >> 1;
      if (hi[mid + 1] > K)
        up = mid;
      else
        lo = mid + 1;
    }
    x2[k] = x[lo];
    y2[k] = y[lo];
    h2[k] = h[lo];
    lw2[k] = lw[lo];
    src2[k] = src[lo];
  } else {
    int64_t lo = 0, up = n_recv - 1;  // first record with hi_r > K
    while (lo < up) {
      const int64_t mid = (lo + up) >> 1;
      if (rlohi[2 * mid + 1] > K)
        up = mid;
      else
        lo = mid + 1;
    }
    const double* hd = reinterpret_cast<const double*>(buf + (size_t)lo * stride);
    x2[k] = hd[0];
    y2[k] = hd[1];
    h2[k] = hd[2];
    lw2[k] = hd[3];
    src2[k] = (int32_t)(-(lo + 1));
  }
}
void launch_adopt_dev(hipStream_t s, DeviceState& d, const int64_t* hi_dev, int64_t slot_start,
                      const unsigned char* buf_dev, int64_t n_recv, int64_t* rlohi_dev) {
  if (d.P == 0) return;
  const int c = d.cur, n = c ^ 1;
  const size_t stride = kPoseRecordBytes + d.lay.slot_bytes;
  if (n_recv > 0)
    hipLaunchKernelGGL(k_extract_lohi, dim3((unsigned)((n_recv + 255) / 256)), dim3(256), 0, s, buf_dev, stride, n_recv,
                       rlohi_dev);
  hipLaunchKernelGGL(k_adopt_dev, dim3((unsigned)((d.P + 255) / 256)), dim3(256), 0, s, d.x[c], d.y[c], d.h[c],
                     d.logw[c], d.src[c], d.x[n], d.y[n], d.h[n], d.logw[n], d.src[n], hi_dev, slot_start, buf_dev,
                     stride, rlohi_dev, n_recv, d.P);
  d.cur = n;
  d.alt = n_recv > 0 ? buf_dev : nullptr;
  d.alt_stride = stride;
  d.alt_off = kPoseRecordBytes;
}

// ------------------------------------------------------------------ probe
// in: pose[3] mean[5] cov[25] blob[4] Qt[16] dir[2] (55 doubles); out: PK_PROBE_LEN doubles.
__global__ void k_probe(const double* __restrict__ in, double* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double sx = in[0], sy = in[1], sh = in[2];
  const double* mean = in + 3;
  const double* cov = in + 8;
  const double* blob = in + 33;
  const double* Qt = in + 37;
  Landmark<double> f{mean[0], mean[1], mean[2], mean[3], mean[4], cov[0], cov[1], cov[6],
                     cov[12], cov[13], cov[14], cov[18], cov[19], cov[24], 0};
  BlobT<double> z{blob[0], blob[1], blob[2], blob[3]};
  Noise<double> qt{Qt[0], Qt[5], Qt[6], Qt[7], Qt[10], Qt[11], Qt[15]};
  const double ux = in[53], uy = in[54];  // unit((cos b, sin b, 0)), host side like the kernels' input
  for (int i = 0; i < 79; ++i) out[i] = 0.0;
  out[0] = probability_of_match(f, sx, sy, sh, z, ux, uy);
  double pse = atan2(f.my - sy, f.mx - sx);
  out[1] = prob_position_match(f, sx, sy, pse, z.bearing, ux, uy, out + 2);
  out[4] = prob_color_match(f, z.r, z.g, z.b);
  EkfAux<double> aux;
  Landmark<double> g = f;
  double lw = ekf_update(g, sx, sy, z, qt, false, &aux);
  out[5] = aux.zhat0;
  out[6] = f.mr;
  out[7] = f.mg;
  out[8] = f.mb;
  out[9] = aux.h0;
  out[10] = aux.h1;
  double* Q = out + 11;
  Q[0] = aux.q00;
  Q[5] = aux.qc.a;
  Q[6] = aux.qc.b;
  Q[7] = aux.qc.c;
  Q[9] = aux.qc.b;
  Q[10] = aux.qc.d;
  Q[11] = aux.qc.e;
  Q[13] = aux.qc.c;
  Q[14] = aux.qc.e;
  Q[15] = aux.qc.f;
  double* K = out + 27;  // 5x4 row-major
  K[0] = aux.k0;
  K[4] = aux.k1;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K[(2 + i) * 4 + 1 + j] = aux.kc[i * 3 + j];
  out[47] = exp(lw);
  out[48] = g.mx;
  out[49] = g.my;
  out[50] = g.mr;
  out[51] = g.mg;
  out[52] = g.mb;
  double* S = out + 53;
  S[0] = g.pxx;
  S[1] = g.pxy;
  S[5] = g.pxy;
  S[6] = g.pyy;
  S[12] = g.crr;
  S[13] = g.crg;
  S[14] = g.crb;
  S[17] = g.crg;
  S[18] = g.cgg;
  S[19] = g.cgb;
  S[22] = g.crb;
  S[23] = g.cgb;
  S[24] = g.cbb;
  out[78] = lw;
}
void launch_probe(hipStream_t s, const double* in_dev, double* out_dev) {
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, s, in_dev, out_dev);
}

}  // namespace pk
