// K3 for maximum-likelihood association, settling the association hand-off inside the EKF kernel:
// k_observe_fast (L <= 512, whole map in registers), k_observe_sweep (any L, landmark chunks, two
// sweeps) and k_step_fused (K2 + K3 in one kernel: gates, settling, update).
//
// Hand-written gfx950 (CDNA4, wave64) kernels of the FastSLAM particle update; see DESIGN.md
// section 4.  No MFMA: the algebra is 2x2 / 3x3 and register resident (pk_math.hpp).
#include "pk_device.hpp"

// k_step_fused: where the lane's covariance rows are asked for -- 0: with the means, right behind the table words' requests
// (rounds 1-3); 1 (default, round 4): when this wave's table words have ARRIVED (the rows of the early waves no longer stand in the
// texture addresser's queue in front of the late waves' table words, which the whole workgroup waits for: 0.2423 -> 0.2406 ms at
// 10 000 x 500, three interleaved repetitions); 3: behind the gates (0.277 ms: the settling waits for them).
#ifndef PK_FUSED_LATE_COV
#define PK_FUSED_LATE_COV 1
#endif

namespace pk {

// ------------------------------------------------------------------ K3 (fast ML variant, L <= 512)
// One workgroup per particle, one landmark per lane, the particle's whole map in
// registers from the single coalesced load to the single coalesced store.  Input is the
// association kernel's hand-off: per landmark the (<= 4) blobs that pass its gates, per blob
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

typedef float Float2 __attribute__((ext_vector_type(2)));

// What a lane keeps about one gate-passing blob of its landmark between the phases of the settling.
struct FastSlot {
  int t;                    // blob (cell order) or -1
  unsigned qf;              // bits 0-3 flags: 1 contested, 2 apply the update, 4 unmatched (single, probability 0),
                            // 8 this lane bids for the contested blob; bits 4..: 1 + entry of the probability queue (0: none)
  unsigned long long bits;  // in-place evaluation only (queue full): probability bits of a contested candidate
};

// The few (landmark, blob) pairs whose probability VALUE is needed -- contested blobs, and pairs
// too close to the float64 underflow edge to call positive without evaluating -- are queued in
// LDS and evaluated densely by the first lanes of the workgroup (two log + two exp each), instead
// of dragging every wave through that code for a handful of its lanes.
constexpr int kFastQueue = 512;
struct FastQueue {
  double* det2;   // [kFastQueue]
  double* det3;
  double* maha2;  // numerator of the position Mahalanobis term (x 1 / det2); overwritten with the probability bits by the evaluation pass
  double* maha3;  // numerator of the colour term (x 1 / det3)
  int* meta;      // blob t | contested << 16
  int* n;         // entries pushed (may exceed kFastQueue: the excess is evaluated in place)
};
__host__ __device__ inline size_t fast_queue_bytes() { return (size_t)kFastQueue * (4 * 8 + 4) + 16; }
size_t observe_fast_lds_bytes(int B) { return fast_queue_bytes() + (size_t)B * 13 + 16; }

// exact: the scan's exact records [B][6], in global memory (a.exact) or staged in LDS by the caller.
// INPLACE: entries that do not fit the queue are evaluated on the spot (k_observe_fast); without it
// the caller checks *fq.n > kFastQueue afterwards and hands the particle to the general kernels.
// bc: landmarks passing each blob's gates (unsigned char from the hand-off, or the int LDS counters of k_step_fused)
template <bool INPLACE, typename CountT, int QCAP = kFastQueue>
__device__ __forceinline__ void fast_prepare(const FastArgs& a, const double* exact,
                                             const Landmark<double>& lm, double sx, double sy,
                                             double pse, uint2 packed, const CountT* bc,
                                             unsigned long long* best, const FastQueue& fq,
                                             FastSlot (&sl)[kFastSlots]) {
  const double det2 = lm.pxx * lm.pyy - lm.pxy * lm.pxy;
  double det3;
  const Sym3<double> adj3 = sym3_adjugate(Sym3<double>{lm.crr, lm.crg, lm.crb, lm.cgg, lm.cgb, lm.cbb}, det3);
  const bool dets_sane = det2 > 0.0 && det2 < 1e60 && det3 > 0.0 && det3 < 1e60;
  const unsigned w[2] = {packed.x, packed.y};
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    const int t = (int)((w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
    sl[k].t = t == 0xFFFF ? -1 : t;
    sl[k].qf = 0u;
    sl[k].bits = 0ull;
    if (sl[k].t < 0) continue;
    const double* rec = exact + 6 * (size_t)t;
    const double2 z01 = *reinterpret_cast<const double2*>(rec);
    const double2 z23 = *reinterpret_cast<const double2*>(rec + 2);
    const double2 dir = *reinterpret_cast<const double2*>(rec + 4);
    // prob_position_match :457-494, prob_color_match :524-544
    const bool angle_ok = !(fabs(pse - z01.x) > Consts<double>::half_pi);  // :473-475
    double nx, ny;
    closest_point(lm.mx, lm.my, sx, sy, dir.x, dir.y, nx, ny);
    const double ex = nx - lm.mx, ey = ny - lm.my;
    const double num2 = lm.pyy * ex * ex - 2.0 * lm.pxy * ex * ey + lm.pxx * ey * ey;  // maha2 = num2 / det2
    const double num3 = sym3_quad(adj3, z01.y - lm.mr, z23.x - lm.mg, z23.y - lm.mb);   // maha3 = num3 / det3
    const bool contested = bc[t] >= 2;
    // pr = (500 exp(a1)) (500 exp(a2)) / 250000 is certainly > 0 when a1 + a2 is far from the
    // float64 underflow edge: log det <= 138.2 for det <= 1e60, so a1 + a2 > -543 when
    // maha2 + maha3 < 800 -- tested without a division: num2 det3 + num3 det2 < 800 det2 det3
    // (both determinants positive; an overflow or NaN fails the test and the pair is evaluated)
    const bool surely_positive = angle_ok && dets_sane && num2 >= 0.0 && num3 >= 0.0 &&
                                 num2 * det3 + num3 * det2 < 800.0 * det2 * det3;
    if (contested) sl[k].qf = 1u;
    if (!angle_ok) {  // bp = 0 (:475): probability 0
      if (!contested) sl[k].qf = 4u;
      continue;
    }
    if (!contested && surely_positive) {
      sl[k].qf = 2u;
      continue;
    }
    const int qi = atomicAdd(fq.n, 1);
    if (qi < QCAP) {
      fq.det2[qi] = det2;
      fq.det3[qi] = det3;
      fq.maha2[qi] = num2;
      fq.maha3[qi] = num3;
      fq.meta[qi] = t | (contested ? 0x10000 : 0);
      sl[k].qf |= (unsigned)(qi + 1) << 4;
    } else if (INPLACE) {  // queue full (dense clusters of look-alike landmarks): evaluate in place
      // (opaque copies: keeps the compiler from hoisting the logs out of this rare branch into the
      // code every lane runs)
      double d2 = det2, d3 = det3;
      asm volatile("" : "+v"(d2), "+v"(d3));
      const double pr = pr_from_parts(d2, d3, num2, num3);
      if (contested) {
        if (pr > 0.0) {
          sl[k].bits = (unsigned long long)__double_as_longlong(pr);
          atomicMax(&best[t], sl[k].bits);
        }
      } else {
        sl[k].qf = pr > 0.0 ? 2u : 4u;
      }
    }
  }
}

// Dense evaluation of the queued probabilities, lanes over queue entries.
__device__ __forceinline__ void fast_evaluate_queue(const FastQueue& fq, unsigned long long* best, int tid,
                                                    int nthreads) {
  const int n = min(*fq.n, kFastQueue);
  // The pass is a serial section of the workgroup, so its latency counts, not its lane efficiency:
  // entries are dealt round-robin to the first four waves (one per SIMD), and each entry to a PAIR of
  // lanes -- the even lane evaluates the position pdf (:439), the odd lane the colour pdf (:446), one
  // log + one exp each instead of two in a row; the product (:455) is formed from the same two
  // values, in the same order, as pr_from_parts does.
  if (tid >= 256) return;
  const int lane = tid & 63, wave = tid >> 6, role = lane & 1;
  for (int base = 0; base < n; base += 128) {  // wave-uniform trip count (the shuffle needs both lanes of a pair)
    const int i = base + ((lane >> 1) << 2) + wave;
    const bool on = i < n;
    const double det = on ? (role ? fq.det3[i] : fq.det2[i]) : 1.0;
    const double maha = (on ? (role ? fq.maha3[i] : fq.maha2[i]) : 0.0) / det;
    const double k = role ? 3.0 : 2.0;
    const double mine = 500.0 * exp(-0.5 * (k * Consts<double>::log_two_pi + log_few_ulp(det) + maha));
    const double other = __shfl_xor(mine, 1, kWave);
    if (on && role == 0) {
      const double pr = mine * other / 250000.0;  // bp * cp / 250000
      const unsigned long long bits = pr > 0.0 ? (unsigned long long)__double_as_longlong(pr) : 0ull;
      reinterpret_cast<unsigned long long*>(fq.maha2)[i] = bits;
      const int m = fq.meta[i];
      if ((m & 0x10000) && bits != 0ull) atomicMax(&best[m & 0xFFFF], bits);
    }
  }
}

// Read the queued results back into the owner's slots; contested candidates that attain the
// blob's best probability bid for it with their landmark index (earliest wins, :377).
// results: probability bits of the queue entries (the evaluation pass wrote them over fq.maha2, or into an array of its own)
__device__ __forceinline__ void fast_collect(const unsigned long long* results, const unsigned long long* best, int* win, int l,
                                             FastSlot (&sl)[kFastSlots]) {
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    if (sl[k].t < 0) continue;
    unsigned long long bits = sl[k].bits;
    if (sl[k].qf >> 4) {
      bits = results[(sl[k].qf >> 4) - 1];
      if (!(sl[k].qf & 1u)) sl[k].qf = bits != 0ull ? 2u : 4u;
    }
    if ((sl[k].qf & 1u) && bits != 0ull && bits == best[sl[k].t]) {
      atomicMin(&win[sl[k].t], l);
      sl[k].qf |= 8u;
    }
  }
}

// imm: Feature.__immutable__ of this landmark (:909, :926), read by the caller together with the state
__device__ __forceinline__ double fast_apply(const FastArgs& a, const double* exact, const unsigned short* order,
                                             Landmark<double>& lm, int l, bool imm, double sx, double sy, double pse,
                                             FastSlot (&sl)[kFastSlots], const int* win) {
  double acc = 0.0;
  // per slot: scan index << 16 | blob when the update is applied, else all ones (sorts to the back)
  unsigned key[kFastSlots];
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    key[k] = 0xFFFFFFFFu;
    if (sl[k].t < 0) continue;
    bool apply = (sl[k].qf & 2u) != 0u;
    if ((sl[k].qf & 8u) && win[sl[k].t] == l) apply = true;          // the earliest of the best bidders
    if (sl[k].qf & 4u) acc += Consts<double>::log_no_match;          // single candidate, probability 0 (:94-95)
    if (apply) key[k] = ((unsigned)order[sl[k].t] << 16) | (unsigned)sl[k].t;
  }
  // the blobs to apply first, in scan order (:88) -- so that nearly every lane of the wave
  // applies its (usually only) update in the same iteration (5-comparator network on the keys)
  auto cswap = [&](unsigned& u, unsigned& v) {
    const unsigned lo = min(u, v), hi = max(u, v);
    u = lo;
    v = hi;
  };
  cswap(key[0], key[1]);
  cswap(key[2], key[3]);
  cswap(key[0], key[2]);
  cswap(key[1], key[3]);
  cswap(key[1], key[2]);
  bool fresh = true;
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    if (key[k] == 0xFFFFFFFFu) continue;
    const double* rec = exact + 6 * (size_t)(key[k] & 0xFFFFu);
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
  FastQueue fq;
  fq.det2 = reinterpret_cast<double*>(smem);
  fq.det3 = fq.det2 + kFastQueue;
  fq.maha2 = fq.det3 + kFastQueue;
  fq.maha3 = fq.maha2 + kFastQueue;
  fq.meta = reinterpret_cast<int*>(fq.maha3 + kFastQueue);
  fq.n = fq.meta + kFastQueue;
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem + fast_queue_bytes());
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
  unsigned char immA = 0;
  if (active) {
    lp = a.lmpass[(size_t)p * Lp + l];  // first: the blob records it points to are the next dependent loads
    A = load_landmark(sf, sc, Lp, l);
    immA = a.immutable[min(l, a.L - 1)];
  }
  for (int t = tid; t < B; t += kFastThreads) {
    best[t] = 0ull;
    win[t] = INT_MAX;
    bc[t] = a.bcount[(size_t)p * B + t];
  }
  if (tid == 0) *fq.n = 0;
  __syncthreads();
  int nun = 0;  // blobs no landmark passes
  for (int t = tid; t < B; t += kFastThreads) nun += bc[t] == 0;
  FastSlot sa[kFastSlots];
  // atan2(my - sy, mx - sx) of the untouched state, handed over by the association kernel
  const double pseA = __longlong_as_double((long long)(((unsigned long long)lp.w << 32) | lp.z));
  fast_prepare<true, unsigned char>(a, a.exact, A, sx, sy, pseA,
                     has ? make_uint2(lp.x, lp.y) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu), bc, best, fq, sa);
  __syncthreads();
  fast_evaluate_queue(fq, best, tid, kFastThreads);
  __syncthreads();
  fast_collect(reinterpret_cast<const unsigned long long*>(fq.maha2), best, win, l, sa);
  __syncthreads();
  for (int t = tid; t < B; t += kFastThreads) nun += (bc[t] >= 2 && best[t] == 0ull);  // contested, all 0
  double acc = (double)nun * Consts<double>::log_no_match;
  if (has) acc += fast_apply(a, a.exact, a.order, A, l, immA != 0, sx, sy, pseA, sa, win);
  asm volatile("" ::"v"(A.count));  // consumed on every path: no load left pending at the join below (it would cost a vmcnt(0) after the stores)
  if (active) {
    __builtin_nontemporal_store(A.mx, &df[(size_t)F_MX * Lp + l]);
    __builtin_nontemporal_store(A.my, &df[(size_t)F_MY * Lp + l]);
    __builtin_nontemporal_store(A.mr, &df[(size_t)F_MR * Lp + l]);
    __builtin_nontemporal_store(A.mg, &df[(size_t)F_MG * Lp + l]);
    __builtin_nontemporal_store(A.mb, &df[(size_t)F_MB * Lp + l]);
    __builtin_nontemporal_store(A.pxx, &df[(size_t)F_PXX * Lp + l]);
    __builtin_nontemporal_store(A.pxy, &df[(size_t)F_PXY * Lp + l]);
    __builtin_nontemporal_store(A.pyy, &df[(size_t)F_PYY * Lp + l]);
    __builtin_nontemporal_store(A.crr, &df[(size_t)F_CRR * Lp + l]);
    __builtin_nontemporal_store(A.crg, &df[(size_t)F_CRG * Lp + l]);
    __builtin_nontemporal_store(A.crb, &df[(size_t)F_CRB * Lp + l]);
    __builtin_nontemporal_store(A.cgg, &df[(size_t)F_CGG * Lp + l]);
    __builtin_nontemporal_store(A.cgb, &df[(size_t)F_CGB * Lp + l]);
    __builtin_nontemporal_store(A.cbb, &df[(size_t)F_CBB * Lp + l]);
    __builtin_nontemporal_store(A.count, &dc[l]);
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
  a.qt = make_noise(qt.q00, qt.rr, qt.rg, qt.rb, qt.gg, qt.gb, qt.bb);
  const size_t lds = observe_fast_lds_bytes(B);  // <= kMaxDynLds: checked by the caller
  static bool attr_set[kMaxDevices] = {false};
  if (first_time_on_this_device(attr_set)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_observe_fast), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kMaxDynLds) != hipSuccess)
      (void)hipGetLastError();
  }
  hipLaunchKernelGGL(k_observe_fast, dim3((unsigned)d.P), dim3(kFastThreads), lds, s, a);
}

// ------------------------------------------------------------------ K2 + K3 fused (ML, L <= 512, small scan tables)
// One workgroup per particle, one landmark per lane, from the association gates to the coalesced
// store of the updated map: what k_assoc_grid<hand-off> and k_observe_fast do in two launches,
// without the hand-off through HBM (no lmpass / bcount arrays, the means are read once).
//   1. the scan tables (exact records when they fit, cell starts, fp32 records, duplicated index list,
//      order table) are copied to LDS -- all table words of a lane requested in one batch; the lane
//      requests its landmark's 14 rows (means first), count and immutable flag right behind them;
//   2. gates: atan2, colour cell, 4-wide walk of the duplicated list with the conservative fp32
//      screen, exact float64 gates (:433, :441) on the survivors; the (<= 4) passing blobs stay in
//      registers, the per-blob counts are LDS atomics;
//   3. a particle in which some landmark passes more than kFastSlots blobs -- or, after step 4's
//      preparation, one that wants more probabilities than the LDS queue holds -- is flagged for the
//      general kernels and left untouched (nothing has been written by then);
//   4. everything is settled and applied exactly as in k_observe_fast (the covariance rows were
//      requested together with the means and arrived during the gates).
// A particle's pose component through the constant address space: the poses are not written while this kernel runs, and
// said so the (uniform) read becomes a scalar load -- its own counter, the scalar cache -- instead of a vector load that
// queues behind the rows in flight.
__device__ __forceinline__ double regs_pose(const double* a, int64_t p) {
  return ((const __attribute__((address_space(4))) double*)a)[p];
}
// src[p] the same way: the entry is written by the kernels that read it, but only by the workgroup that owns particle p and
// only after it has read it, so whichever copy a scalar cache holds is the current one.
__device__ __forceinline__ int32_t regs_source(const int32_t* src, int64_t p) {
  return ((const __attribute__((address_space(4))) int32_t*)src)[p];
}
struct FusedArgs {
  FastArgs f;                   // lmpass, bcount unused
  BlobGrid g;
  const unsigned char* tables;  // start u16[ncell+1] (16-byte padded) | rec32 float4[B] | idx9 u16[n9]
  const double* h;
  unsigned char* pflag_out;     // [P] 1 = general route
  unsigned* n_flagged;
  int n9;
  const unsigned* pub_skipped;  // NULL, or: == 0 -> k_step_pub takes this scan, this kernel stands back
};

// exact_lds: the exact records [B][6] and the order table are staged in LDS too (one contiguous
// image exact | start | rec32 | idx9 | order, as the host lays the scan block out)
size_t fused_lds_bytes(int ncell, int B, int n9, bool exact_lds) {
  size_t tab = grid_cs_bytes(ncell) + (size_t)B * 16 + (size_t)n9 * 2;
  if (exact_lds) tab += (size_t)B * 48 + (size_t)B * 2;
  tab = (tab + 15) & ~(size_t)15;
  return tab + (((size_t)B * 4 + 15) & ~(size_t)15) + fast_queue_bytes() + (size_t)B * 13 + 16;
}

// Diagnostic build only (-DPK_STAMPS): per-phase cycle sums of k_step_fused (slots 16.. of pk_debug_stamps).
#ifdef PK_STAMPS
__device__ unsigned long long pk_fstamp_acc[16];
#define PK_FSTAMP_ADD(slot, a, b) \
  if ((threadIdx.x & 63) == 0) atomicAdd(&pk_fstamp_acc[slot], (b) - (a));
void debug_read_fused_stamps(unsigned long long* out, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pk_fstamp_acc), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(pk_fstamp_acc), z, sizeof(z));
  }
}
#else
#define PK_FSTAMP_ADD(slot, a, b)
#endif

template <bool EXACT_LDS>
__global__ void __launch_bounds__(kFastThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) k_step_fused(FusedArgs fa) {
  PK_STAMP(f0)
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double red[kFastThreads / kWave];
  __shared__ int wg_flag;
  const FastArgs& a = fa.f;
  const BlobGrid& g = fa.g;
  if (fa.pub_skipped && *fa.pub_skipped == 0u) return;  // workgroup-uniform: the publish / subscribe instance works on this scan
  // Every argument the first loads depend on is wanted at once: one batch of kernarg loads and one
  // wait, not seven dependent round trips through the scalar cache before the first request leaves.
  // A pure asm (no side effects: a volatile one in front of them would turn the scalar loads of
  // src[p], x[p], ... into vector loads) whose result, an opaque zero, is added to the particle index.
  int zero;
  asm("s_mov_b32 %0, 0"
      : "=s"(zero)
      : "s"(fa.f.src), "s"(fa.f.x), "s"(fa.f.y), "s"(fa.h), "s"(fa.f.exact), "s"(fa.tables), "s"(fa.f.B), "s"(fa.n9),
        "s"(fa.g.ncell), "s"(fa.f.Lp), "s"(fa.f.L), "s"(fa.f.ss.map), "s"(fa.f.ss.slot_bytes), "s"(fa.f.map_dst),
        "s"(fa.f.count_off), "s"(fa.f.immutable), "s"(fa.f.ss.alt), "s"(fa.f.ss.alt_stride), "s"(fa.f.ss.alt_off));
  const int64_t p = (int64_t)blockIdx.x + zero;
  const int tid = threadIdx.x;
  const int B = a.B, Lp = a.Lp;
  const size_t cs_bytes = grid_cs_bytes(g.ncell);
  // LDS image: [exact records] | start | rec32 | idx9 | [order]  (a contiguous piece of the scan block)
  const size_t ex_bytes = EXACT_LDS ? (size_t)B * 48 : 0;
  const size_t tab_bytes = (ex_bytes + cs_bytes + (size_t)B * 16 + (size_t)fa.n9 * 2 + (EXACT_LDS ? (size_t)B * 2 : 0) + 15) & ~(size_t)15;
  const unsigned short* start = reinterpret_cast<const unsigned short*>(smem + ex_bytes);
  const float4* rec32 = reinterpret_cast<const float4*>(smem + ex_bytes + cs_bytes);
  const unsigned short* idx9 = reinterpret_cast<const unsigned short*>(smem + ex_bytes + cs_bytes + (size_t)B * 16);
  // the exact records and the order table the settling / update code reads: LDS copy or global
  const double* exact = EXACT_LDS ? reinterpret_cast<const double*>(smem) : a.exact;
  const unsigned short* order =
      EXACT_LDS ? reinterpret_cast<const unsigned short*>(smem + ex_bytes + cs_bytes + (size_t)B * 16 + (size_t)fa.n9 * 2)
                : a.order;
  int* ccount = reinterpret_cast<int*>(smem + tab_bytes);
  unsigned char* qbase = smem + tab_bytes + (((size_t)B * 4 + 15) & ~(size_t)15);
  FastQueue fq;
  fq.det2 = reinterpret_cast<double*>(qbase);
  fq.det3 = fq.det2 + kFastQueue;
  fq.maha2 = fq.det3 + kFastQueue;
  fq.maha3 = fq.maha2 + kFastQueue;
  fq.meta = reinterpret_cast<int*>(fq.maha3 + kFastQueue);
  fq.n = fq.meta + kFastQueue;
  unsigned long long* best = reinterpret_cast<unsigned long long*>(qbase + fast_queue_bytes());
  int* win = reinterpret_cast<int*>(best + B);

  // (the particle's own scalars -- src[p], its pose -- are fetched BEHIND the table requests: src[p] is a second, dependent
  // scalar round trip that the table words do not have to wait for)
  const double* sf;
  double* df;
  const int* sc;
  int* dc;
  double sx, sy, sh;
  const int l = tid;
  // ---- 1. tables -> LDS, own state ------------------------------------------------------------
  // All table words of the lane are requested in one batch (a copy loop of load / wait / LDS write
  // costs one L2 round trip per 8 KB: six in a row at B = 500, 44 % of the workgroup's lifetime
  // when measured with the -DPK_STAMPS build), the whole state right behind them: vmcnt retires
  // in order, so the LDS writes wait for the table words only, and every barrier of this kernel
  // orders LDS alone -- the covariance rows arrive while the gates are worked out.
  constexpr int kTabBatch = 8;  // x 512 lanes x 16 B = 64 KB in one batch; larger tables: a loop for the rest
  Landmark<double> A{};
  double lw0 = 0.0;    // the particle's log-weight so far (0 = log 1 after the fused reset, :73)
  unsigned char immA;  // Feature.__immutable__ of the lane's landmark: one more load of this batch, not a
                       // dependent L2 round trip in front of the update
  {
    const uint4* src = reinterpret_cast<const uint4*>(EXACT_LDS ? reinterpret_cast<const unsigned char*>(a.exact) : fa.tables);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const unsigned n16 = (unsigned)(tab_bytes / 16);
    uint4 tw[kTabBatch];
    PK_STAMP(h0)
    PK_FSTAMP_ADD(14, f0, h0)  // scalar prologue
#pragma unroll
    for (int j = 0; j < kTabBatch; ++j) {
      // unconditional (clamped index): a predicated load gets a branch and a wait of its own
      const unsigned i = (unsigned)tid + (unsigned)j * kFastThreads;
      tw[j] = src[min(i, n16 - 1u)];
    }
    asm volatile("" ::: "memory");  // the table requests first: they come back first
    {
      const unsigned char* sslot = a.ss.at(regs_source(a.src, p));
      unsigned char* dslot = a.map_dst + (size_t)p * a.ss.slot_bytes;
      sf = reinterpret_cast<const double*>(sslot);
      df = reinterpret_cast<double*>(dslot);
      sc = reinterpret_cast<const int*>(sslot + a.count_off);
      dc = reinterpret_cast<int*>(dslot + a.count_off);
      sx = regs_pose(a.x, p);
      sy = regs_pose(a.y, p);
      sh = regs_pose(fa.h, p);
    }
    // unconditional too (lanes beyond the map read its last landmark and never use or store it):
    // straight-line code, so that the wait below is vmcnt(15) -- table words only
#if PK_FUSED_LATE_COV
    // (round 4, as in k_step_pub: the covariance rows -- not needed before the settling -- are asked for only when
    // this wave's table words have arrived, so that they do not stand in the texture addresser's queue in front of the OTHER
    // waves' table words, which the whole workgroup waits for at the barrier below)
    {
      const int lq = min(l, Lp - 1);
      A.mx = sf[F_MX * Lp + lq];
      A.my = sf[F_MY * Lp + lq];
      A.mr = sf[F_MR * Lp + lq];
      A.mg = sf[F_MG * Lp + lq];
      A.mb = sf[F_MB * Lp + lq];
    }
#else
    A = load_landmark_means_first(sf, sc, Lp, min(l, Lp - 1));
#endif
    immA = a.immutable[min(l, a.L - 1)];
    if (!a.reset) lw0 = a.logw[p];  // requested with the state: read at the very end it would wait for every store of the wave
    // the table words are needed HERE (keeps the compiler from sinking each load into its
    // predicated LDS write, one round trip at a time); nothing moves across
#pragma unroll
    for (int j = 0; j < kTabBatch; ++j) asm volatile("" : "+v"(tw[j].x), "+v"(tw[j].y), "+v"(tw[j].z), "+v"(tw[j].w)::"memory");
    PK_STAMP(h1)
    PK_FSTAMP_ADD(15, h0, h1)  // table words arrived
#if PK_FUSED_LATE_COV == 1
    {
      const int lq = min(l, Lp - 1);
      A.pxx = sf[F_PXX * Lp + lq];
      A.pxy = sf[F_PXY * Lp + lq];
      A.pyy = sf[F_PYY * Lp + lq];
      A.crr = sf[F_CRR * Lp + lq];
      A.crg = sf[F_CRG * Lp + lq];
      A.crb = sf[F_CRB * Lp + lq];
      A.cgg = sf[F_CGG * Lp + lq];
      A.cgb = sf[F_CGB * Lp + lq];
      A.cbb = sf[F_CBB * Lp + lq];
      A.count = sc[lq];
      asm volatile("" ::: "memory");
    }
#endif
#pragma unroll
    for (int j = 0; j < kTabBatch; ++j) {
      const unsigned i = (unsigned)tid + (unsigned)j * kFastThreads;
      if (i < n16) dst[i] = tw[j];
    }
    for (unsigned i = (unsigned)tid + kTabBatch * kFastThreads; i < n16; i += kFastThreads) dst[i] = src[i];
  }
  for (int t = tid; t < B; t += kFastThreads) {
    ccount[t] = 0;
    best[t] = 0ull;
    win[t] = INT_MAX;
  }
  if (tid == 0) {
    *fq.n = 0;
    wg_flag = 0;
  }
  const bool active = l < Lp, has = l < a.L;
  lds_barrier();
  PK_STAMP(f1)
  PK_FSTAMP_ADD(0, f0, f1)
  // ---- 2. gates ----------------------------------------------------------------------------------
  unsigned pass01 = 0xFFFFFFFFu, pass23 = 0xFFFFFFFFu;
  double pseA = 0.0;
  if (has) {
    const double mx = A.mx, my = A.my, mr = A.mr, mg = A.mg, mb = A.mb;
    PK_STAMP(g0)
    PK_FSTAMP_ADD(1, f1, g0)  // wait for the means
    pseA = pk_atan2(my - sy, mx - sx);
    const double eb = pseA - sh;  // :408
    const float mr32 = (float)mr, mg32 = (float)mg, mb32 = (float)mb, eb32 = (float)eb;
    int c[3];
    const double m3[3] = {mr, mg, mb};
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // same cell function as the host, see k_assoc_grid
      double q = floor(__dmul_rn(__dsub_rn(m3[k], g.lo[k]), g.inv_h));
      q = fmin(fmax(q, -1.0), (double)g.G[k]);
      c[k] = (int)q;
    }
    const int k0 = max(c[2] - 1, 0), k1 = min(c[2] + 1, g.G[2] - 1);
    const int r = min(max(c[0], 0), g.G[0] - 1), gg = min(max(c[1], 0), g.G[1] - 1);
    const int base = (r * g.G[1] + gg) * g.G[2];
    int i0 = 0, i1 = 0;
    if (k0 <= k1) {
      i0 = start[base + k0];
      i1 = start[base + k1 + 1];
    }
    int npass = 0;
    // fp32 screen on packed pairs (v_pk_add_f32 / v_pk_mul_f32): (r, g) and (b, bearing)
    const Float2 m01 = {mr32, mg32}, m23 = {mb32, eb32};
    auto prefilter_q = [&](const float4& q) {
      const Float2 q01 = {q.x, q.y}, q23 = {q.z, q.w};
      const Float2 d01 = q01 - m01, d23 = q23 - m23;
      const Float2 s01 = d01 * d01;
      const float cd32 = fmaf(d23.x, d23.x, s01.x + s01.y);
      return !(cd32 > g.thr32) && !(fabsf(d23.y) > g.thrb32);
    };
    auto exact_gates = [&](int tt, const double2& z01, const double2& z23) {
      if (!(fabs(z01.x - eb) > 0.5) && !(fabs(color_distance2(mr, mg, mb, z01.y, z23.x, z23.y)) > 300.0)) {
        atomicAdd(&ccount[tt], 1);
        if (npass == 0) pass01 = (pass01 & 0xFFFF0000u) | (unsigned)tt;
        if (npass == 1) pass01 = (pass01 & 0x0000FFFFu) | ((unsigned)tt << 16);
        if (npass == 2) pass23 = (pass23 & 0xFFFF0000u) | (unsigned)tt;
        if (npass == 3) pass23 = (pass23 & 0x0000FFFFu) | ((unsigned)tt << 16);
        ++npass;
      }
    };
    // phase 1 (LDS only): survivors of the fp32 screen, the LAST four kept in registers -- a 64-bit
    // shift register of 16-bit blob indices (two instructions per survivor)
    unsigned slo = 0u, shi = 0u;
    int npc = 0;
    for (int i = i0; i < i1; i += 4) {
      int t4[4];
      float4 q4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) t4[j] = idx9[i + j];
#pragma unroll
      for (int j = 0; j < 4; ++j) q4[j] = rec32[t4[j]];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j < i1 && prefilter_q(q4[j])) {
          shi = __builtin_amdgcn_alignbit(shi, slo, 16);
          slo = (slo << 16) | (unsigned)t4[j];
          ++npc;
        }
    }
    PK_STAMP(g1)
    PK_FSTAMP_ADD(2, g0, g1)  // atan2, cell, walk
    // phase 2: the float64 gates of the (last four) survivors.  Records in global memory: all loads in
    // one batch (one L2 round trip instead of four); records in LDS: one at a time (16 instead of 32
    // VGPRs at the kernel's register peak)
    {
      const int pc[4] = {(int)(slo & 0xFFFFu), (int)(slo >> 16), (int)(shi & 0xFFFFu), (int)(shi >> 16)};
      if (EXACT_LDS) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (npc > k) {
            const double* rec = exact + 6 * (size_t)pc[k];
            const double2 z01 = *reinterpret_cast<const double2*>(rec);
            const double2 z23 = *reinterpret_cast<const double2*>(rec + 2);
            exact_gates(pc[k], z01, z23);
          }
      } else {
        double2 z01[4], z23[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (npc > k) {
            const double* rec = exact + 6 * (size_t)pc[k];
            z01[k] = *reinterpret_cast<const double2*>(rec);
            z23[k] = *reinterpret_cast<const double2*>(rec + 2);
          }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (npc > k) exact_gates(pc[k], z01[k], z23[k]);
      }
    }
    if (npc > 4) {  // dense colour clusters: walk again for the survivors before the last four
      int seen = 0;
      for (int i = i0; i < i1 && seen < npc - 4; ++i) {
        const int t = idx9[i];
        if (prefilter_q(rec32[t])) {
          const double* rec = exact + 6 * (size_t)t;
          exact_gates(t, *reinterpret_cast<const double2*>(rec), *reinterpret_cast<const double2*>(rec + 2));
          ++seen;
        }
      }
    }
    if (npass > kFastSlots) wg_flag = 1;
    PK_STAMP(g2)
    PK_FSTAMP_ADD(3, g1, g2)  // exact gates
  }
#if PK_FUSED_LATE_COV == 3  // (experiment: the covariance rows behind the gates)
  {
    const int lq = min(l, Lp - 1);
    A.pxx = sf[F_PXX * Lp + lq];
    A.pxy = sf[F_PXY * Lp + lq];
    A.pyy = sf[F_PYY * Lp + lq];
    A.crr = sf[F_CRR * Lp + lq];
    A.crg = sf[F_CRG * Lp + lq];
    A.crb = sf[F_CRB * Lp + lq];
    A.cgg = sf[F_CGG * Lp + lq];
    A.cgb = sf[F_CGB * Lp + lq];
    A.cbb = sf[F_CBB * Lp + lq];
    A.count = sc[lq];
    asm volatile("" ::: "memory");
  }
#endif
  PK_STAMP(f2)
  lds_barrier();
  PK_STAMP(f3)
  PK_FSTAMP_ADD(4, f2, f3)  // barrier wait after the gates
  // ---- 3. flagged particles go the general way ----------------------------------------------------
  if (wg_flag) {  // workgroup-uniform
    if (tid == 0) {
      fa.pflag_out[p] = 1;
      atomicAdd(fa.n_flagged, 1u);
    }
    return;
  }
  // ---- 4. k_observe_fast from here (same device functions, same bits); the per-blob counts are read
  // straight from the LDS counters (no conversion pass, one barrier less)
  int nun = 0;  // blobs no landmark passes
  for (int t = tid; t < B; t += kFastThreads) nun += ccount[t] == 0;
  FastSlot sa[kFastSlots];
  PK_STAMP(f4)
  PK_FSTAMP_ADD(5, f3, f4)  // counts
  fast_prepare<false, int>(a, exact, A, sx, sy, pseA, make_uint2(pass01, pass23), ccount, best, fq, sa);
  PK_STAMP(f5)
  PK_FSTAMP_ADD(6, f4, f5)  // prepare (first use of the covariance rows)
  lds_barrier();
  PK_STAMP(f6)
  PK_FSTAMP_ADD(7, f5, f6)  // barrier wait after prepare
  // more probabilities wanted than the queue holds (dense clusters of look-alike landmarks): nothing
  // has been written yet, the general kernels take the particle
  const bool overflow = *fq.n > kFastQueue;  // workgroup-uniform
  if (tid == 0) {
    fa.pflag_out[p] = overflow ? 1 : 0;
    if (overflow) atomicAdd(fa.n_flagged, 1u);
  }
  if (overflow) return;
  fast_evaluate_queue(fq, best, tid, kFastThreads);
  lds_barrier();
  PK_STAMP(f7)
  PK_FSTAMP_ADD(8, f6, f7)  // queue evaluation + barrier
  fast_collect(reinterpret_cast<const unsigned long long*>(fq.maha2), best, win, l, sa);
  lds_barrier();
  PK_STAMP(f8)
  PK_FSTAMP_ADD(9, f7, f8)  // collect + barrier
  for (int t = tid; t < B; t += kFastThreads) nun += (ccount[t] >= 2 && best[t] == 0ull);  // contested, all 0
  double acc = (double)nun * Consts<double>::log_no_match;
  if (has) acc += fast_apply(a, exact, order, A, l, immA != 0, sx, sy, pseA, sa, win);
  PK_STAMP(f9)
  PK_FSTAMP_ADD(10, f8, f9)  // apply
  // consumed on every path: no load is left pending at the join below -- it cost a vmcnt(0), i.e. every
  // wave sat out the acknowledgement of all its stores before the block sum
  asm volatile("" ::"v"(A.count));
  if (active) {
    __builtin_nontemporal_store(A.mx, &df[(size_t)F_MX * Lp + l]);
    __builtin_nontemporal_store(A.my, &df[(size_t)F_MY * Lp + l]);
    __builtin_nontemporal_store(A.mr, &df[(size_t)F_MR * Lp + l]);
    __builtin_nontemporal_store(A.mg, &df[(size_t)F_MG * Lp + l]);
    __builtin_nontemporal_store(A.mb, &df[(size_t)F_MB * Lp + l]);
    __builtin_nontemporal_store(A.pxx, &df[(size_t)F_PXX * Lp + l]);
    __builtin_nontemporal_store(A.pxy, &df[(size_t)F_PXY * Lp + l]);
    __builtin_nontemporal_store(A.pyy, &df[(size_t)F_PYY * Lp + l]);
    __builtin_nontemporal_store(A.crr, &df[(size_t)F_CRR * Lp + l]);
    __builtin_nontemporal_store(A.crg, &df[(size_t)F_CRG * Lp + l]);
    __builtin_nontemporal_store(A.crb, &df[(size_t)F_CRB * Lp + l]);
    __builtin_nontemporal_store(A.cgg, &df[(size_t)F_CGG * Lp + l]);
    __builtin_nontemporal_store(A.cgb, &df[(size_t)F_CGB * Lp + l]);
    __builtin_nontemporal_store(A.cbb, &df[(size_t)F_CBB * Lp + l]);
    __builtin_nontemporal_store(A.count, &dc[l]);
  }
  PK_STAMP(f10)
  PK_FSTAMP_ADD(11, f9, f10)  // stores issued
  const double tot = block_sum_lds_only<kFastThreads / kWave>(acc, red, tid);  // the stores stay in flight
  if (tid == 0) {
    const double v = lw0 + tot;
    a.logw[p] = v;
    if (a.gmax_key) atomicMax(a.gmax_key + (p & (kGmaxKeys - 1)), double_to_key(v));
    a.src[p] = (int32_t)p;
  }
  PK_STAMP(f11)
  PK_FSTAMP_ADD(12, f10, f11)  // block sum
  PK_FSTAMP_ADD(13, f0, f11)   // lifetime
}

void launch_step_fused(hipStream_t s, DeviceState& d, int B, const BlobGrid& grid, int n9,
                       const unsigned char* tables_dev, const double* exact_dev, const unsigned short* order_dev,
                       const FastHandoff& fh, const NoiseD& qt, const ObserveExtras& ex, const unsigned* pub_skipped_dev) {
  if (d.P == 0) return;
  static bool attr_set[kMaxDevices] = {false};
  if (first_time_on_this_device(attr_set)) {
    for (const void* fn : {reinterpret_cast<const void*>(k_step_fused<false>), reinterpret_cast<const void*>(k_step_fused<true>)})
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds) != hipSuccess)
        (void)hipGetLastError();
  }
  FusedArgs fa;
  FastArgs& a = fa.f;
  a.ss = slot_source(d);
  a.map_dst = d.map[d.mcur ^ 1];
  a.count_off = d.lay.count_off;
  a.src = d.src[d.cur];
  a.x = d.x[d.cur];
  a.y = d.y[d.cur];
  a.logw = d.logw[d.cur];
  a.exact = exact_dev;
  a.order = order_dev;
  a.lmpass = nullptr;
  a.bcount = nullptr;
  a.pflag = nullptr;
  a.immutable = d.immutable;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  a.reset = ex.reset ? 1 : 0;
  a.gmax_key = ex.gmax_key;
  a.qt = make_noise(qt.q00, qt.rr, qt.rg, qt.rb, qt.gg, qt.gb, qt.bb);
  fa.g = grid;
  fa.tables = tables_dev;
  fa.h = d.h[d.cur];
  fa.pflag_out = fh.pflag;
  fa.n_flagged = fh.n_flagged;
  fa.n9 = n9;
  fa.pub_skipped = pub_skipped_dev;
  // exact records + order table in LDS as well when two workgroups per CU still fit
  const bool exact_lds = fused_lds_bytes(grid.ncell, B, n9, true) <= kFusedMaxLds;
  const size_t lds = fused_lds_bytes(grid.ncell, B, n9, exact_lds);
  if (exact_lds)
    hipLaunchKernelGGL((k_step_fused<true>), dim3((unsigned)d.P), dim3(kFastThreads), lds, s, fa);
  else
    hipLaunchKernelGGL((k_step_fused<false>), dim3((unsigned)d.P), dim3(kFastThreads), lds, s, fa);
}

// ------------------------------------------------------------------ K3 (sweep ML variant, any L)
// The same hand-off as k_observe_fast, for maps that do not fit one landmark per lane: persistent
// workgroups, each particle's landmarks in chunks of kSweepThreads, two sweeps.
//   sweep 1 (read only): landmarks that pass a CONTESTED blob (one that several landmarks pass)
//     load their state and queue the pair's probability inputs in LDS; the queue is evaluated
//     densely (two log + two exp per pair), atomicMax of the probability bits per blob in LDS,
//     and every positive (blob, landmark, probability) goes to a result list (per-workgroup
//     scratch in global memory, L2 resident).  After the last chunk the pairs that attain their
//     blob's best probability bid with their landmark index: atomicMin -> the earliest wins (:377).
//   sweep 2: every landmark again (the second read comes from L2 / Infinity Cache: one
//     particle's map is <= a few hundred KB), uncontested blobs settled by the strict '>' from 0.0
//     as in k_observe_fast, contested ones applied by their winner, updates in scan order (:88),
//     coalesced store of all 14 rows into the other map buffer.
// A particle the association kernel flagged (a landmark passing more than kFastSlots blobs) is
// skipped here and taken by the general kernels.
constexpr int kSweepThreads = 256;  // 3 workgroups per CU at <= 168 VGPRs

struct SweepArgs {
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
  uint4* results;               // [gridDim.x][kSweepSlots * Lp]: probability bits (lo, hi), blob t, landmark l
  int64_t P;
  int L, Lp, B;
  int qcap;                     // entries of the LDS probability queue
  int reset;
  unsigned long long* gmax_key;
  Noise<double> qt;
  int only_value;               // 0: the particles whose flag is 0; else only the particles with this flag (second chance, 2)
  const unsigned* n_flagged;    // with only_value: nothing to do when 0
  const int32_t* row_of;        // with only_value: the particle's hand-off row (FastHandoff::row_of)
};

__host__ __device__ inline size_t sweep_lds_bytes(int B, int qcap) {
  return (size_t)qcap * 36 + 16 + (((size_t)B * 13 + 15) & ~(size_t)15);
}

SweepPlan observe_sweep_plan(const DeviceState& d, int B) {
  SweepPlan pl{};
  const size_t fixed = sweep_lds_bytes(B, 0);
  // workgroups per CU: three (the register budget of the kernel) when the blob tables leave room
  // for a queue of >= 256 entries each, else two, else one
  int per_cu = 3;
  size_t budget = (kMaxDynLds / 3) & ~(size_t)255;
  if (fixed + 256 * 36 > budget) {
    per_cu = 2;
    budget = (kMaxDynLds / 2) & ~(size_t)255;
  }
  if (fixed + 256 * 36 > budget) {
    per_cu = 1;
    budget = kMaxDynLds;
  }
  if (fixed + 64 * 36 > budget) return pl;  // scan too large for the LDS tables: grid = 0
  long q = (long)((budget - fixed) / 36);
  q = q > 1024 ? 1024 : q;  // kSweepThreads lanes x kFastSlots
  q &= ~63L;
  pl.qcap = (int)q;
  pl.lds = sweep_lds_bytes(B, pl.qcap);
  int64_t g = 256 * (int64_t)per_cu;
  pl.grid = (int)(g < d.P ? g : d.P);
  pl.results_per_wg = kSweepSlots * (size_t)d.lay.Lp;
  return pl;
}

// SLOTS = kFastSlots (hand-off entry 16 B: four blob fields + atan2) or kSweepSlots (32 B: eight + atan2)
template <int SLOTS>
__global__ void __launch_bounds__(kSweepThreads, 3) k_observe_sweep(SweepArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double red[kSweepThreads / kWave];
  const int tid = threadIdx.x;
  const int B = a.B, Lp = a.Lp, qcap = a.qcap;
  double* q_det2 = reinterpret_cast<double*>(smem);
  double* q_det3 = q_det2 + qcap;
  double* q_maha2 = q_det3 + qcap;
  double* q_maha3 = q_maha2 + qcap;
  unsigned* q_meta = reinterpret_cast<unsigned*>(q_maha3 + qcap);  // blob t | landmark l << 16
  int* q_n = reinterpret_cast<int*>(q_meta + qcap);                // [2] alternating per chunk, [2] = result count
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem + (size_t)qcap * 36 + 16);
  int* win = reinterpret_cast<int*>(best + B);
  unsigned char* bc = reinterpret_cast<unsigned char*>(win + B);
  uint4* results = a.results + (size_t)blockIdx.x * SLOTS * (size_t)Lp;

  if (a.only_value && *a.n_flagged == 0u) return;
  for (int64_t p = blockIdx.x; p < a.P; p += gridDim.x) {
    if (a.only_value ? a.pflag[p] != a.only_value : a.pflag[p] != 0) continue;  // workgroup-uniform: not this launch's particle
    const unsigned char* sslot = a.ss.at(a.src[p]);
    unsigned char* dslot = a.map_dst + (size_t)p * a.ss.slot_bytes;
    const double* sf = reinterpret_cast<const double*>(sslot);
    double* df = reinterpret_cast<double*>(dslot);
    const int* sc = reinterpret_cast<const int*>(sslot + a.count_off);
    int* dc = reinterpret_cast<int*>(dslot + a.count_off);
    const double sx = a.x[p], sy = a.y[p];
    const int64_t row = a.only_value ? (int64_t)a.row_of[p] : p;  // (second chance: the row k_assoc_grid dealt this particle)
    const uint4* lmp = a.lmpass + (SLOTS == 4 ? 1 : 2) * (size_t)row * Lp;  // entries as written by k_assoc_grid
    // blob fields of landmark l (unused words all ones) and its atan2(my - sy, mx - sx)
    auto entry_blobs = [&](int l) {
      if (SLOTS == 4) {
        const uint4 e = lmp[l];
        return make_uint4(e.x, e.y, 0xFFFFFFFFu, 0xFFFFFFFFu);
      }
      return lmp[2 * l];
    };
    auto entry_pse = [&](int l) {
      const uint4 e = SLOTS == 4 ? lmp[l] : lmp[2 * l + 1];
      const unsigned lo = SLOTS == 4 ? e.z : e.x, hi = SLOTS == 4 ? e.w : e.y;
      return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    for (int t = tid; t < B; t += kSweepThreads) {
      best[t] = 0ull;
      win[t] = INT_MAX;
      bc[t] = a.bcount[(size_t)row * B + t];
    }
    if (tid < 3) q_n[tid] = 0;
    __syncthreads();

    // ---- sweep 1: probabilities of the contested pairs ------------------------------------
    int par = 0;
    for (int base = 0; base < a.L; base += kSweepThreads, par ^= 1) {
      int l = base + tid;
      asm volatile("" : "+v"(l));
      uint4 lp = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
      if (l < a.L) lp = entry_blobs(l);
      // slot k of this landmark: blob (cell order) or 0xFFFF; the slots are filled from the front
      auto slot_of = [&](int k) {
        const unsigned w = k < 2 ? lp.x : (k < 4 ? lp.y : (k < 6 ? lp.z : lp.w));
        return (int)((w >> (16 * (k & 1))) & 0xFFFFu);
      };
      bool any = false;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        const int t = slot_of(k);
        any |= t != 0xFFFF && bc[t] >= 2;
      }
      if (any) {
        const Landmark<double> lm = load_landmark_nocount(sf, Lp, l);
        const double pse = entry_pse(l);
        const double det2 = lm.pxx * lm.pyy - lm.pxy * lm.pxy;
        double det3;
        const Sym3<double> adj3 = sym3_adjugate(Sym3<double>{lm.crr, lm.crg, lm.crb, lm.cgg, lm.cgb, lm.cbb}, det3);
#pragma unroll 1
        for (int k = 0; k < SLOTS; ++k) {  // rolled: one copy of the code, a lane leaves at its first empty slot
          const int t = slot_of(k);
          if (t == 0xFFFF) break;
          if (bc[t] < 2) continue;
          const double* rec = a.exact + 6 * (size_t)t;
          const double2 z01 = *reinterpret_cast<const double2*>(rec);
          const double2 z23 = *reinterpret_cast<const double2*>(rec + 2);
          const double2 dir = *reinterpret_cast<const double2*>(rec + 4);
          if (fabs(pse - z01.x) > Consts<double>::half_pi) continue;  // :473-475 -> probability 0
          double nx, ny;
          closest_point(lm.mx, lm.my, sx, sy, dir.x, dir.y, nx, ny);
          const double ex = nx - lm.mx, ey = ny - lm.my;
          // numerators of the two Mahalanobis terms (maha = num / det, divided in pr_from_parts)
          const double maha2 = lm.pyy * ex * ex - 2.0 * lm.pxy * ex * ey + lm.pxx * ey * ey;
          const double maha3 = sym3_quad(adj3, z01.y - lm.mr, z23.x - lm.mg, z23.y - lm.mb);
          const int qi = atomicAdd(&q_n[par], 1);
          if (qi < qcap) {
            q_det2[qi] = det2;
            q_det3[qi] = det3;
            q_maha2[qi] = maha2;
            q_maha3[qi] = maha3;
            q_meta[qi] = (unsigned)t | ((unsigned)l << 16);
          } else {  // queue full: evaluate in place (opaque copies keep the logs out of the common path)
            double d2 = det2, d3 = det3;
            asm volatile("" : "+v"(d2), "+v"(d3));
            const double pr = pr_from_parts(d2, d3, maha2, maha3);
            if (pr > 0.0) {
              const unsigned long long bits = (unsigned long long)__double_as_longlong(pr);
              atomicMax(&best[t], bits);
              results[atomicAdd(&q_n[2], 1)] = make_uint4((unsigned)bits, (unsigned)(bits >> 32), (unsigned)t, (unsigned)l);
            }
          }
        }
      }
      __syncthreads();
      const int n = min(q_n[par], qcap);
      for (int i = tid; i < n; i += kSweepThreads) {
        const double pr = pr_from_parts(q_det2[i], q_det3[i], q_maha2[i], q_maha3[i]);
        if (pr > 0.0) {
          const unsigned long long bits = (unsigned long long)__double_as_longlong(pr);
          const unsigned m = q_meta[i];
          atomicMax(&best[m & 0xFFFFu], bits);
          results[atomicAdd(&q_n[2], 1)] = make_uint4((unsigned)bits, (unsigned)(bits >> 32), m & 0xFFFFu, m >> 16);
        }
      }
      if (tid == 0) q_n[par ^ 1] = 0;  // the other counter: last read before the previous barrier
      __syncthreads();
    }
    // the pairs that attain their blob's best probability bid with their landmark index
    {
      const int nres = q_n[2];
      for (int i = tid; i < nres; i += kSweepThreads) {
        const uint4 e = results[i];
        const unsigned long long bits = ((unsigned long long)e.y << 32) | e.x;
        if (bits == best[e.z]) atomicMin(&win[e.z], (int)e.w);
      }
    }
    __syncthreads();
    int nun = 0;  // blobs nobody passes, and contested blobs whose probabilities are all 0
    for (int t = tid; t < B; t += kSweepThreads) nun += (bc[t] == 0) || (bc[t] >= 2 && best[t] == 0ull);
    double acc = (double)nun * Consts<double>::log_no_match;

    // ---- sweep 2: settle the uncontested blobs, apply, store ----------------------------------
    for (int base = 0; base < Lp; base += kSweepThreads) {
      int l = base + tid;
      asm volatile("" : "+v"(l));  // opaque: no strength-reduced row pointers kept live across the chunk loop
      if (l >= Lp) continue;
      Landmark<double> A = load_landmark(sf, sc, Lp, l);
      if (l < a.L) {
        const uint4 lp = entry_blobs(l);
        const double pse = entry_pse(l);
        // per slot: scan index << 16 | blob when the update is applied, else 0xFFFFFFFF
        unsigned key[SLOTS];
        // Division-free sufficient test for "probability certainly > 0" (all an uncontested blob
        // needs, :369-381).  With P, C positive definite (Sylvester's criterion),
        //   maha2 = e' P^-1 e <= |e|^2 / lmin(P) <= |e|^2 tr(P) / det(P)
        //   maha3 = d' C^-1 d <= |d|^2 / lmin(C) <= |d|^2 m2(C) / det(C),  m2 = sum of principal 2x2 minors
        // so maha2 + maha3 < 800 follows from |e|^2 tr det3 + |d|^2 m2 det2 < 800 det2 det3, and then
        // pr = (500 exp(a1)) (500 exp(a2)) / 250000 has a1 + a2 > -543 (log det <= 138.2 for det <= 1e60):
        // no underflow.  Pairs that fail it are evaluated exactly as the reference does.
        bool have_q = false, spd = false;
        double det2 = 0.0, det3 = 0.0, tr2 = 0.0, m2 = 0.0, lim = 0.0;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) key[k] = 0xFFFFFFFFu;
#pragma unroll 1
        for (int k = 0; k < SLOTS; ++k) {  // rolled: one copy of the settling code
          const unsigned wk = k < 2 ? lp.x : (k < 4 ? lp.y : (k < 6 ? lp.z : lp.w));
          const int t = (int)((wk >> (16 * (k & 1))) & 0xFFFFu);
          if (t == 0xFFFF) break;  // the slots are filled from the front
          bool apply;
          if (bc[t] >= 2) {
            apply = win[t] == l;
          } else {
            const double* rec = a.exact + 6 * (size_t)t;
            const double2 z01 = *reinterpret_cast<const double2*>(rec);
            const double2 z23 = *reinterpret_cast<const double2*>(rec + 2);
            const double2 dir = *reinterpret_cast<const double2*>(rec + 4);
            if (!have_q) {
              det2 = A.pxx * A.pyy - A.pxy * A.pxy;
              const double c00 = A.cgg * A.cbb - A.cgb * A.cgb;
              const double c11 = A.crr * A.cbb - A.crb * A.crb;
              const double c22 = A.crr * A.cgg - A.crg * A.crg;
              det3 = A.crr * c00 + A.crg * (A.crb * A.cgb - A.crg * A.cbb) + A.crb * (A.crg * A.cgb - A.crb * A.cgg);
              tr2 = A.pxx + A.pyy;
              m2 = c00 + c11 + c22;
              spd = A.pxx > 0.0 && det2 > 0.0 && det2 < 1e60 && A.crr > 0.0 && c22 > 0.0 && det3 > 0.0 && det3 < 1e60 &&
                    tr2 < 1e100 && m2 < 1e100;
              lim = 800.0 * det2 * det3;
              have_q = true;
            }
            apply = !(fabs(pse - z01.x) > Consts<double>::half_pi);  // :473-475
            if (apply) {
              double nx, ny;
              closest_point(A.mx, A.my, sx, sy, dir.x, dir.y, nx, ny);
              const double ex = nx - A.mx, ey = ny - A.my;
              const double d1 = z01.y - A.mr, d2c = z23.x - A.mg, d3c = z23.y - A.mb;
              const bool sure = spd && (ex * ex + ey * ey) * tr2 * det3 + (d1 * d1 + d2c * d2c + d3c * d3c) * m2 * det2 < lim;
              if (!sure) {  // rare: tiny or indefinite covariances, far-off closest points
                double e2 = ex, e3 = ey;
                asm volatile("" : "+v"(e2), "+v"(e3));  // opaque: keeps the divisions out of the code every lane runs
                double det3b;
                const Sym3<double> adj3 = sym3_adjugate(Sym3<double>{A.crr, A.crg, A.crb, A.cgg, A.cgb, A.cbb}, det3b);
                const double maha2 = A.pyy * e2 * e2 - 2.0 * A.pxy * e2 * e3 + A.pxx * e3 * e3;  // numerators
                const double maha3 = sym3_quad(adj3, d1, d2c, d3c);
                apply = pr_from_parts(det2, det3b, maha2, maha3) > 0.0;
              }
            }
            if (!apply) acc += Consts<double>::log_no_match;  // probability 0: unseen feature (:94-95)
          }
          if (apply) {
            const unsigned kv = ((unsigned)a.order[t] << 16) | (unsigned)t;
#pragma unroll
            for (int j = 0; j < SLOTS; ++j)
              if (k == j) key[j] = kv;
          }
        }
        // apply in scan order (:88): take the smallest remaining key each time (usually one or two)
        const bool imm = a.immutable[l] != 0;
        bool fresh = true;
        for (;;) {
          unsigned kk = key[0];
#pragma unroll
          for (int j = 1; j < SLOTS; ++j) kk = min(kk, key[j]);
          if (kk == 0xFFFFFFFFu) break;
#pragma unroll
          for (int j = 0; j < SLOTS; ++j)
            if (key[j] == kk) key[j] = 0xFFFFFFFFu;  // keys are distinct: one slot per blob
          const double* rec = a.exact + 6 * (size_t)(kk & 0xFFFFu);
          const double2 z01 = *reinterpret_cast<const double2*>(rec);
          const double2 z23 = *reinterpret_cast<const double2*>(rec + 2);
          BlobT<double> z{z01.x, z01.y, z23.x, z23.y};
          acc += ekf_update(A, sx, sy, z, a.qt, imm, (EkfAux<double>*)nullptr, fresh ? &pse : (const double*)nullptr);
          fresh = imm;
        }
      }
      __builtin_nontemporal_store(A.mx, &df[(size_t)F_MX * Lp + l]);
      __builtin_nontemporal_store(A.my, &df[(size_t)F_MY * Lp + l]);
      __builtin_nontemporal_store(A.mr, &df[(size_t)F_MR * Lp + l]);
      __builtin_nontemporal_store(A.mg, &df[(size_t)F_MG * Lp + l]);
      __builtin_nontemporal_store(A.mb, &df[(size_t)F_MB * Lp + l]);
      __builtin_nontemporal_store(A.pxx, &df[(size_t)F_PXX * Lp + l]);
      __builtin_nontemporal_store(A.pxy, &df[(size_t)F_PXY * Lp + l]);
      __builtin_nontemporal_store(A.pyy, &df[(size_t)F_PYY * Lp + l]);
      __builtin_nontemporal_store(A.crr, &df[(size_t)F_CRR * Lp + l]);
      __builtin_nontemporal_store(A.crg, &df[(size_t)F_CRG * Lp + l]);
      __builtin_nontemporal_store(A.crb, &df[(size_t)F_CRB * Lp + l]);
      __builtin_nontemporal_store(A.cgg, &df[(size_t)F_CGG * Lp + l]);
      __builtin_nontemporal_store(A.cgb, &df[(size_t)F_CGB * Lp + l]);
      __builtin_nontemporal_store(A.cbb, &df[(size_t)F_CBB * Lp + l]);
      __builtin_nontemporal_store(A.count, &dc[l]);
    }
    const double tot = block_sum<kSweepThreads / kWave>(acc, red);  // two barriers: LDS is free for the next particle
    if (tid == 0) {
      const double v = (a.reset ? 0.0 : a.logw[p]) + tot;
      a.logw[p] = v;
      if (a.gmax_key) atomicMax(a.gmax_key + (p & (kGmaxKeys - 1)), double_to_key(v));
      a.src[p] = (int32_t)p;
    }
  }
}

void launch_observe_sweep(hipStream_t s, DeviceState& d, int B, const double* exact_dev,
                          const unsigned short* order_dev, const FastHandoff& fh, const NoiseD& qt,
                          const ObserveExtras& ex, const SweepPlan& plan, uint4* results_dev) {
  if (d.P == 0 || plan.grid == 0) return;
  static bool attr_set[kMaxDevices] = {false};
  if (first_time_on_this_device(attr_set)) {
    for (const void* fn : {reinterpret_cast<const void*>(k_observe_sweep<kFastSlots>),
                           reinterpret_cast<const void*>(k_observe_sweep<kSweepSlots>)})
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds) != hipSuccess)
        (void)hipGetLastError();
  }
  SweepArgs a;
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
  a.row_of = fh.row_of;
  a.immutable = d.immutable;
  a.results = results_dev;
  a.P = d.P;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  a.qcap = plan.qcap;
  a.reset = ex.reset ? 1 : 0;
  a.gmax_key = ex.gmax_key;
  a.qt = make_noise(qt.q00, qt.rr, qt.rg, qt.rb, qt.gg, qt.gb, qt.bb);
  a.only_value = ex.sweep_only_value;
  a.n_flagged = ex.n_flagged;
  // never more workgroups than are resident at once (results_dev is sized for plan.grid)
  static size_t asked_lds = ~(size_t)0;
  static int asked_per_cu = 0;
  if (asked_lds != plan.lds) {
    asked_lds = plan.lds;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&asked_per_cu, reinterpret_cast<const void*>(k_observe_sweep<kSweepSlots>),
                                                     kSweepThreads, plan.lds) != hipSuccess) {
      (void)hipGetLastError();
      asked_per_cu = 0;
    }
  }
  int grid = plan.grid;
  if (asked_per_cu > 0 && 256 * asked_per_cu < grid) grid = 256 * asked_per_cu;
  if (fh.slots == kSweepSlots)
    hipLaunchKernelGGL(k_observe_sweep<kSweepSlots>, dim3((unsigned)grid), dim3(kSweepThreads), plan.lds, s, a);
  else
    hipLaunchKernelGGL(k_observe_sweep<kFastSlots>, dim3((unsigned)grid), dim3(kSweepThreads), plan.lds, s, a);
}

// ------------------------------------------------------------------ K2 + K3 in ONE pass, 512 < L <= 2048
// k_step_fused's shape for maps that do not fit one landmark per lane: a workgroup of 1024 lanes owns
// one particle at a time and keeps the particle's WHOLE map in registers, two landmarks per lane
// (l and l + 1024: every row access of a wave is one contiguous 512-byte run), from the single coalesced
// load to the single coalesced store -- the state is read once (k_assoc_grid + k_observe_sweep read the
// means twice and most covariance rows twice, and hand 32 B per landmark over through HBM).
// 16 waves x 128 VGPRs are the CU's whole register file, so exactly one particle is in flight per CU:
// the grid is persistent (one workgroup per CU), the scan tables are staged in LDS once per workgroup,
// and the NEXT particle's map slot is touched (one dword per 128-byte line, LDS-DMA into a dump word: no
// register is tied up) while this one is worked on, so that it waits in L2 when its turn comes -- the
// only way one resident workgroup overlaps its memory time with its arithmetic.
//   per particle: state requested (means first) -> gates of both landmarks of the lane (as in
//   k_step_fused) -> settling in two rounds, one per landmark of the lane, through ONE probability queue
//   in LDS (results of both rounds kept) -> bids -> EKF updates in scan order -> stores -> log-weight.
// Registers are the scarce resource (58 of the 128 hold the two landmarks): what a lane knows about its
// gate-passing blobs is packed into 16-bit fields between the phases, and while the first landmark is
// updated the six colour-covariance rows of the second wait in the (by then dead) queue area of LDS.
// A particle in which a landmark passes more than kFastSlots blobs, or whose queue overflows in a round,
// is flagged for the general kernels before anything of it has been written.
constexpr int kRegsThreads = 1024;
// (diagnostic builds only: -DPK_REGS_BOUND=512 lifts the 128-register cap to show what the allocator would like to have)
#ifndef PK_REGS_BOUND
#define PK_REGS_BOUND kRegsThreads
#endif
constexpr int kRegsQueue = 1024;  // probability queue entries per settling round

struct RegsArgs {
  FastArgs f;                   // lmpass, bcount, pflag unused
  BlobGrid g;
  const unsigned char* tables;  // start u16[ncell+1] (16-byte padded) | rec32 float4[B] | idx9 u16[n9]
  const double* h;
  unsigned char* pflag_out;     // [P] 1 = general route
  unsigned* n_flagged;
  int n9;
  int warm;                     // 0: no L2 warming of the next particle's slot, 1: mean rows, 2: whole slot
  int64_t P;                    // end of the particle range of this launch
  int64_t p_begin;              // its start (0 but for the split step of the sharded filter)
  const uint4* cand;            // candidate lists of the reference particle (CandTable), or NULL
  const unsigned* cand_over;    // != 0: the lists overflowed, this scan takes the grid walk
  const unsigned* cand_skip;    // != 0: the candidate-list instance stands back (lists overflowed, or k_step_pub takes the scan)
};

// LDS: tables | ccount int[B] (later: win) | best u64[B] | queue inputs 36 B x kRegsQueue | results u64[2 kRegsQueue] | counters
// (queue inputs + results = 52 KB: the parking area of the second landmark's colour block, 48 KB, during the updates)
size_t regs_lds_bytes(int ncell, int B, int n9) {
  size_t tab = (grid_cs_bytes(ncell) + (size_t)B * 16 + (size_t)n9 * 2 + 15) & ~(size_t)15;
  return tab + (((size_t)B * 4 + 15) & ~(size_t)15) + (size_t)B * 8 + (size_t)kRegsQueue * 36 + (size_t)kRegsQueue * 16 + 32;
}

// LDS of the candidate-list instance: where the grid tables stood, the lanes park the means of their two landmarks
// (10 rows of 1024 doubles, 80 KB) between the phases that use them; the candidate records are read from L2
constexpr size_t kRegsMeanPark = (size_t)10 * kRegsThreads * 8;
size_t regs_cand_lds_bytes(int Lp, int B) {
  (void)Lp;
  return kRegsMeanPark + (((size_t)B * 4 + 15) & ~(size_t)15) + (size_t)B * 8 + (size_t)kRegsQueue * 36 + (size_t)kRegsQueue * 16 + 32;
}

// Gates of one landmark (prkt_core_v2.py:433, :441) against the scan tables in LDS: the same walk as in
// k_step_fused (4-wide over the duplicated list, fp32 screen on packed pairs, exact float64 gates on the
// survivors, their records from L2 two at a time).  pass01 / pass23: the (first four) passing blobs, 16 bits each.
struct RegsGated {
  unsigned pass01, pass23;
  double pse;
};
__device__ __forceinline__ RegsGated regs_gates(const BlobGrid& g, const unsigned short* start, const float4* rec32,
                                                const unsigned short* idx9, const double* exact, int* ccount, int* wg_flag,
                                                double mx, double my, double mr, double mg, double mb, double sx, double sy,
                                                double sh) {
  unsigned pass01 = 0xFFFFFFFFu, pass23 = 0xFFFFFFFFu;
  const double pse = pk_atan2(my - sy, mx - sx);
  const double eb = pse - sh;  // :408
  const float mr32 = (float)mr, mg32 = (float)mg, mb32 = (float)mb, eb32 = (float)eb;
  int c[3];
  const double m3[3] = {mr, mg, mb};
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // same cell function as the host, see k_assoc_grid
    double q = floor(__dmul_rn(__dsub_rn(m3[k], g.lo[k]), g.inv_h));
    q = fmin(fmax(q, -1.0), (double)g.G[k]);
    c[k] = (int)q;
  }
  const int k0 = max(c[2] - 1, 0), k1 = min(c[2] + 1, g.G[2] - 1);
  const int r = min(max(c[0], 0), g.G[0] - 1), gg = min(max(c[1], 0), g.G[1] - 1);
  const int base = (r * g.G[1] + gg) * g.G[2];
  int i0 = 0, i1 = 0;
  if (k0 <= k1) {
    i0 = start[base + k0];
    i1 = start[base + k1 + 1];
  }
  int npass = 0;
  const Float2 m01 = {mr32, mg32}, m23 = {mb32, eb32};
  auto prefilter_q = [&](const float4& q) {
    const Float2 q01 = {q.x, q.y}, q23 = {q.z, q.w};
    const Float2 d01 = q01 - m01, d23 = q23 - m23;
    const Float2 s01 = d01 * d01;
    const float cd32 = fmaf(d23.x, d23.x, s01.x + s01.y);
    return !(cd32 > g.thr32) && !(fabsf(d23.y) > g.thrb32);
  };
  auto exact_gates = [&](int tt, const double2& z01, const double2& z23) {
    if (!(fabs(z01.x - eb) > 0.5) && !(fabs(color_distance2(mr, mg, mb, z01.y, z23.x, z23.y)) > 300.0)) {
      atomicAdd(&ccount[tt], 1);
      if (npass == 0) pass01 = (pass01 & 0xFFFF0000u) | (unsigned)tt;
      if (npass == 1) pass01 = (pass01 & 0x0000FFFFu) | ((unsigned)tt << 16);
      if (npass == 2) pass23 = (pass23 & 0xFFFF0000u) | (unsigned)tt;
      if (npass == 3) pass23 = (pass23 & 0x0000FFFFu) | ((unsigned)tt << 16);
      ++npass;
    }
  };
  unsigned slo = 0u, shi = 0u;  // the LAST four survivors of the fp32 screen, 16 bits each
  int npc = 0;
  for (int i = i0; i < i1; i += 4) {
    int t4[4];
    float4 q4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t4[j] = idx9[i + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) q4[j] = rec32[t4[j]];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i + j < i1 && prefilter_q(q4[j])) {
        shi = __builtin_amdgcn_alignbit(shi, slo, 16);
        slo = (slo << 16) | (unsigned)t4[j];
        ++npc;
      }
  }
  {
    const int pc[4] = {(int)(slo & 0xFFFFu), (int)(slo >> 16), (int)(shi & 0xFFFFu), (int)(shi >> 16)};
#pragma unroll
    for (int k2 = 0; k2 < 4; k2 += 2) {  // two records per L2 round trip (16 VGPRs; all four at once: 32)
      double2 z01[2], z23[2];
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (npc > k2 + k) {
          const double* rec = exact + 6 * (size_t)pc[k2 + k];
          z01[k] = *reinterpret_cast<const double2*>(rec);
          z23[k] = *reinterpret_cast<const double2*>(rec + 2);
        }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (npc > k2 + k) exact_gates(pc[k2 + k], z01[k], z23[k]);
    }
  }
  if (npc > 4) {  // dense colour clusters: walk again for the survivors before the last four
    int seen = 0;
    for (int i = i0; i < i1 && seen < npc - 4; ++i) {
      const int t = idx9[i];
      if (prefilter_q(rec32[t])) {
        const double* rec = exact + 6 * (size_t)t;
        exact_gates(t, *reinterpret_cast<const double2*>(rec), *reinterpret_cast<const double2*>(rec + 2));
        ++seen;
      }
    }
  }
  if (npass > kFastSlots) *wg_flag = 1;
  return RegsGated{pass01, pass23, pse};
}

// Gates of one landmark against the reference particle's candidate list (pk_kernels.hpp, CandTable): the landmark's
// expected bearing and colour must lie within the margins of the reference's -- else the particle is flagged for the
// general kernels -- and then only the listed blobs can pass: exact float64 gates on those, two records per L2 round trip.
__device__ __forceinline__ RegsGated regs_gates_cand(uint4 ref, uint4 cands, const double* exact, int* ccount, int* wg_flag,
                                                     double mx, double my, double mr, double mg, double mb, double sx,
                                                     double sy, double sh) {
  unsigned pass01 = 0xFFFFFFFFu, pass23 = 0xFFFFFFFFu;
  const double pse = pk_atan2(my - sy, mx - sx);
  const double eb = pse - sh;  // :408
  // (written so that a NaN anywhere breaks the margin)
  const double deb = eb - (double)__uint_as_float(ref.x);  // 2 pi off: the other side of a branch cut, listed too (k_candidates)
  const bool inside = (fabs(deb) <= kCandBearing || fabs(deb - Consts<double>::two_pi) <= kCandBearing ||
                       fabs(deb + Consts<double>::two_pi) <= kCandBearing) &&
                      fabs(mr - (double)__uint_as_float(ref.y)) <= kCandColour &&
                      fabs(mg - (double)__uint_as_float(ref.z)) <= kCandColour && fabs(mb - (double)__uint_as_float(ref.w)) <= kCandColour;
  int npass = 0;
  auto exact_gates = [&](int tt, const double2& z01, const double2& z23) {
    if (!(fabs(z01.x - eb) > 0.5) && !(fabs(color_distance2(mr, mg, mb, z01.y, z23.x, z23.y)) > 300.0)) {
      atomicAdd(&ccount[tt], 1);
      if (npass == 0) pass01 = (pass01 & 0xFFFF0000u) | (unsigned)tt;
      if (npass == 1) pass01 = (pass01 & 0x0000FFFFu) | ((unsigned)tt << 16);
      if (npass == 2) pass23 = (pass23 & 0xFFFF0000u) | (unsigned)tt;
      if (npass == 3) pass23 = (pass23 & 0x0000FFFFu) | ((unsigned)tt << 16);
      ++npass;
    }
  };
  unsigned c0 = cands.x, c1 = cands.y, c2 = cands.z, c3 = cands.w;  // the list is filled from the front
#pragma unroll 1
  for (int k = 0; k < kCandSlots; k += 2) {
    const int ta = (int)(c0 & 0xFFFFu), tb = (int)(c0 >> 16);
    if (ta == 0xFFFF) break;
    c0 = c1;
    c1 = c2;
    c2 = c3;
    c3 = 0xFFFFFFFFu;
    const double* ra = exact + 6 * (size_t)ta;
    const double* rb = exact + 6 * (size_t)(tb == 0xFFFF ? ta : tb);
    const double2 a01 = *reinterpret_cast<const double2*>(ra);
    const double2 a23 = *reinterpret_cast<const double2*>(ra + 2);
    const double2 b01 = *reinterpret_cast<const double2*>(rb);
    const double2 b23 = *reinterpret_cast<const double2*>(rb + 2);
    exact_gates(ta, a01, a23);
    if (tb != 0xFFFF) exact_gates(tb, b01, b23);
  }
  if (!inside || npass > kFastSlots) *wg_flag = 1;
  return RegsGated{pass01, pass23, pse};
}

// What a lane keeps about one landmark's (<= 4) gate-passing blobs between the phases: the blobs in `pass`
// (2 registers) and the four FastSlot::qf words (4 flag bits + 12 bits of queue entry) in 2 more.
__device__ __forceinline__ void regs_pack(const FastSlot (&sl)[kFastSlots], unsigned& q01, unsigned& q23) {
  q01 = (sl[0].qf & 0xFFFFu) | (sl[1].qf << 16);
  q23 = (sl[2].qf & 0xFFFFu) | (sl[3].qf << 16);
  asm volatile("" : "+v"(q01), "+v"(q23));  // these two are what stays live, not the slots they came from
}
__device__ __forceinline__ void regs_unpack(unsigned p01, unsigned p23, unsigned q01, unsigned q23, FastSlot (&sl)[kFastSlots]) {
  const unsigned pw[2] = {p01, p23}, qw[2] = {q01, q23};
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    const int t = (int)((pw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
    sl[k].t = t == 0xFFFF ? -1 : t;
    sl[k].qf = (qw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
    sl[k].bits = 0ull;
  }
}

// fast_apply with ONE copy of the update code (the slots to apply are sorted to the front; a lane leaves at its
// first empty one): same arithmetic, same order, a quarter of the instructions in the cache.
__device__ __forceinline__ double regs_apply(const double* exact, const unsigned short* order, const Noise<double>& qt,
                                             Landmark<double>& lm, int l, bool imm, double sx, double sy, double pse,
                                             FastSlot (&sl)[kFastSlots], const int* win) {
  double acc = 0.0;
  unsigned key[kFastSlots];
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    key[k] = 0xFFFFFFFFu;
    if (sl[k].t < 0) continue;
    bool apply = (sl[k].qf & 2u) != 0u;
    if ((sl[k].qf & 8u) && win[sl[k].t] == l) apply = true;          // the earliest of the best bidders
    if (sl[k].qf & 4u) acc += Consts<double>::log_no_match;          // single candidate, probability 0 (:94-95)
    if (apply) key[k] = ((unsigned)order[sl[k].t] << 16) | (unsigned)sl[k].t;
  }
  auto cswap = [&](unsigned& u, unsigned& v) {
    const unsigned lo = min(u, v), hi = max(u, v);
    u = lo;
    v = hi;
  };
  cswap(key[0], key[1]);
  cswap(key[2], key[3]);
  cswap(key[0], key[2]);
  cswap(key[1], key[3]);
  cswap(key[1], key[2]);
  bool fresh = true;
#pragma unroll 1
  for (int it = 0; it < kFastSlots; ++it) {
    const unsigned kk = key[0];
    if (kk == 0xFFFFFFFFu) break;
    key[0] = key[1];
    key[1] = key[2];
    key[2] = key[3];
    key[3] = 0xFFFFFFFFu;
    const double* rec = exact + 6 * (size_t)(kk & 0xFFFFu);
    const double2 z01 = *reinterpret_cast<const double2*>(rec);
    const double2 z23 = *reinterpret_cast<const double2*>(rec + 2);
    BlobT<double> z{z01.x, z01.y, z23.x, z23.y};
    acc += ekf_update(lm, sx, sy, z, qt, imm, (EkfAux<double>*)nullptr, fresh ? &pse : (const double*)nullptr);
    fresh = imm;
  }
  return acc;
}

// Dense evaluation of one round's queue by ALL the waves of the workgroup (an entry per pair of lanes, as
// in fast_evaluate_queue); the probability bits go to results[] (kept for both rounds), contested ones bid
// for their blob's best probability.
template <int NWAVES>
__device__ __forceinline__ void regs_evaluate_queue(const FastQueue& fq, int n, unsigned long long* results,
                                                    unsigned long long* best, int tid) {
  const int lane = tid & 63, wave = tid >> 6, role = lane & 1;
  for (int base = 0; base < n; base += 32 * NWAVES) {  // workgroup-uniform trip count
    const int i = base + (lane >> 1) * NWAVES + wave;
    const bool on = i < n;
    const double det = on ? (role ? fq.det3[i] : fq.det2[i]) : 1.0;
    const double maha = (on ? (role ? fq.maha3[i] : fq.maha2[i]) : 0.0) / det;
    const double k = role ? 3.0 : 2.0;
    const double mine = 500.0 * exp(-0.5 * (k * Consts<double>::log_two_pi + log_few_ulp(det) + maha));
    const double other = __shfl_xor(mine, 1, kWave);
    if (on && role == 0) {
      const double pr = mine * other / 250000.0;  // bp * cp / 250000
      const unsigned long long bits = pr > 0.0 ? (unsigned long long)__double_as_longlong(pr) : 0ull;
      results[i] = bits;
      const int m = fq.meta[i];
      if ((m & 0x10000) && bits != 0ull) atomicMax(&best[m & 0xFFFF], bits);
    }
  }
}

// off: byte offset of the landmark inside a row (uniform row base + 32-bit lane offset addressing)
__device__ __forceinline__ void regs_store_landmark(double* df, int* dc, int Lp, unsigned off, const Landmark<double>& A) {
  auto at = [&](int f) { return reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(df + (size_t)f * Lp) + off); };
  __builtin_nontemporal_store(A.mx, at(F_MX));
  __builtin_nontemporal_store(A.my, at(F_MY));
  __builtin_nontemporal_store(A.mr, at(F_MR));
  __builtin_nontemporal_store(A.mg, at(F_MG));
  __builtin_nontemporal_store(A.mb, at(F_MB));
  __builtin_nontemporal_store(A.pxx, at(F_PXX));
  __builtin_nontemporal_store(A.pxy, at(F_PXY));
  __builtin_nontemporal_store(A.pyy, at(F_PYY));
  __builtin_nontemporal_store(A.crr, at(F_CRR));
  __builtin_nontemporal_store(A.crg, at(F_CRG));
  __builtin_nontemporal_store(A.crb, at(F_CRB));
  __builtin_nontemporal_store(A.cgg, at(F_CGG));
  __builtin_nontemporal_store(A.cgb, at(F_CGB));
  __builtin_nontemporal_store(A.cbb, at(F_CBB));
  __builtin_nontemporal_store(A.count, reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(dc) + (off >> 1)));
}

// The kernel's arguments are read from the kernarg segment where they are needed, phase by phase, instead
// of sitting in ~100 SGPRs for the whole particle loop (the register allocator spilled 87 of them into VGPR
// lanes, 340 v_readlane / v_writelane per particle): an empty asm makes the pointer opaque, so that nothing
// loaded through it before is kept alive across it.
typedef const __attribute__((address_space(4))) RegsArgs* RegsArgsPtr;
__device__ __forceinline__ RegsArgsPtr regs_args_now(RegsArgsPtr rp) {
  asm volatile("" : "+s"(rp));
  return rp;
}
// member-wise copies out of the kernarg segment (a struct copy cannot bind across address spaces)
__device__ __forceinline__ SlotSource regs_slot_source(RegsArgsPtr R) {
  return SlotSource{R->f.ss.map, R->f.ss.slot_bytes, R->f.ss.alt, R->f.ss.alt_stride, R->f.ss.alt_off};
}
__device__ __forceinline__ BlobGrid regs_blob_grid(RegsArgsPtr R) {
  BlobGrid g;
  g.lo[0] = R->g.lo[0];
  g.lo[1] = R->g.lo[1];
  g.lo[2] = R->g.lo[2];
  g.inv_h = R->g.inv_h;
  g.G[0] = R->g.G[0];
  g.G[1] = R->g.G[1];
  g.G[2] = R->g.G[2];
  g.ncell = R->g.ncell;
  g.thr32 = R->g.thr32;
  g.thrb32 = R->g.thrb32;
  return g;
}
__device__ __forceinline__ Noise<double> regs_noise(RegsArgsPtr R) {
  return Noise<double>{R->f.qt.q00, R->f.qt.rr, R->f.qt.rg, R->f.qt.rb, R->f.qt.gg, R->f.qt.gb, R->f.qt.bb, R->f.qt.diag};
}

// LDS offsets of the workgroup's arrays (see regs_lds_bytes), all derived from three numbers
struct RegsLds {
  unsigned cs_bytes, tab_bytes, best_off;  // start at 0, rec32 at cs_bytes, idx9 behind it; ccount at tab_bytes
  int B;
  __device__ __forceinline__ const unsigned short* start(unsigned char* m) const { return reinterpret_cast<const unsigned short*>(m); }
  __device__ __forceinline__ const float4* rec32(unsigned char* m) const { return reinterpret_cast<const float4*>(m + cs_bytes); }
  __device__ __forceinline__ const unsigned short* idx9(unsigned char* m) const { return reinterpret_cast<const unsigned short*>(m + cs_bytes + (size_t)B * 16); }
  __device__ __forceinline__ int* ccount(unsigned char* m) const { return reinterpret_cast<int*>(m + tab_bytes); }
  __device__ __forceinline__ unsigned long long* best(unsigned char* m) const { return reinterpret_cast<unsigned long long*>(m + best_off); }
  __device__ __forceinline__ FastQueue queue(unsigned char* m, int* n) const {
    FastQueue fq;
    fq.det2 = reinterpret_cast<double*>(m + best_off + (size_t)B * 8);
    fq.det3 = fq.det2 + kRegsQueue;
    fq.maha2 = fq.det3 + kRegsQueue;
    fq.maha3 = fq.maha2 + kRegsQueue;
    fq.meta = reinterpret_cast<int*>(fq.maha3 + kRegsQueue);
    fq.n = n;
    return fq;
  }
  __device__ __forceinline__ unsigned long long* results(unsigned char* m) const {
    return reinterpret_cast<unsigned long long*>(m + best_off + (size_t)B * 8 + (size_t)kRegsQueue * 36);
  }
  __device__ __forceinline__ int* qn(unsigned char* m) const { return reinterpret_cast<int*>(results(m) + 2 * kRegsQueue); }
  __device__ __forceinline__ double* park(unsigned char* m) const { return reinterpret_cast<double*>(m + best_off + (size_t)B * 8); }
};

// Diagnostic build only (-DPK_STAMPS): per-phase cycle sums of k_step_regs (slots 32.. of pk_debug_stamps).
#ifdef PK_STAMPS
__device__ unsigned long long pk_rstamp_acc[16];
#define PK_RSTAMP(slot, a, b) \
  if ((threadIdx.x & 63) == 0) atomicAdd(&pk_rstamp_acc[slot], (b) - (a));
void debug_read_regs_stamps(unsigned long long* out, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pk_rstamp_acc), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(pk_rstamp_acc), z, sizeof(z));
  }
}
#else
#define PK_RSTAMP(slot, a, b)
#endif

template <bool CAND>
__global__ void __launch_bounds__(PK_REGS_BOUND) k_step_regs(RegsArgs ra_unused) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double red[kRegsThreads / kWave];
  __shared__ unsigned warm_dump[kWave];  // where the L2-warming loads of every wave land (never read)
  __shared__ int wg_flag;
  RegsArgsPtr rp = (RegsArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const int tid0 = threadIdx.x;
  RegsLds lds;
  int Lp, L;
  bool immA, immB;
  {
    const int tid = tid0;
    RegsArgsPtr R = regs_args_now(rp);
    // which instance works on this scan: candidate lists, unless some landmark's list overflowed (workgroup-uniform)
    {
      const unsigned* over = R->cand_over;
      if (CAND ? *R->cand_skip != 0u : (over != nullptr && *over == 0u)) return;
    }
    lds.B = R->f.B;
    Lp = R->f.Lp;
    L = R->f.L;
    lds.cs_bytes = CAND ? 0u : (unsigned)grid_cs_bytes(R->g.ncell);
    lds.tab_bytes = CAND ? (unsigned)kRegsMeanPark : (unsigned)((lds.cs_bytes + (size_t)lds.B * 16 + (size_t)R->n9 * 2 + 15) & ~(size_t)15);
    lds.best_off = lds.tab_bytes + (unsigned)(((size_t)lds.B * 4 + 15) & ~(size_t)15);
    // the scan tables of the grid walk: once per workgroup
    if (!CAND) {
      const uint4* src = reinterpret_cast<const uint4*>(R->tables);
      uint4* dst = reinterpret_cast<uint4*>(smem);
      for (unsigned i = (unsigned)tid; i < lds.tab_bytes / 16; i += kRegsThreads) dst[i] = src[i];
    }
    const unsigned char* imm = R->f.immutable;
    immA = imm[min(tid, L - 1)] != 0;
    immB = imm[min(tid + kRegsThreads, L - 1)] != 0;
  }
  const int B = lds.B;
  // the lane's two landmarks: lA = tid and lB = tid + 1024 (lanes beyond the map re-read its last landmark and
  // never use or store it)

  // The log-weight of a particle is finished one barrier late: the waves leave their partial sums in red[] and go on to
  // request the next particle's rows; the barrier that separates the two particles' use of the LDS counters is passed
  // with those requests in flight (the slowest wave's updates hide the others' first round trip).
  int64_t prev = -1;  // the particle whose partial sums wait in red[] (-1: none, or it went to the general kernels)
  for (int64_t p = regs_args_now(rp)->p_begin + blockIdx.x;; p += gridDim.x) {
    Landmark<double> A, Bq;
    // everything derived from the lane index is derived afresh for every particle: hoisted out of this loop
    // (LDS addresses, row offsets, queue-entry indices ...) those values filled the register file and spilled
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    PK_STAMP(r0)
    const int lA = tid, lB = tid + kRegsThreads;
    const bool hasA = lA < L, hasB = lB < L;
    // byte offsets of the lane's two landmarks inside a row: the only per-lane part of the 30 addresses (row
    // bases are uniform: SGPR pair + 32-bit VGPR offset addressing)
    const unsigned oA = (unsigned)min(lA, Lp - 1) * 8u, oB = (unsigned)min(lB, Lp - 1) * 8u;
    const double* sf;
    const int* sc;
    auto row = [&](int f, unsigned off) {
      return *reinterpret_cast<const double*>(reinterpret_cast<const unsigned char*>(sf + (size_t)f * Lp) + off);
    };
    // Candidate lists: between the phases the means of the lane's two landmarks wait in LDS (each lane reads back what it
    // wrote itself: no barrier involved), not in registers -- from the gates to the update they were the longest-lived
    // values of the kernel, the ones the register allocator sent to scratch, and a scratch reload waits behind every row
    // that is on its way from HBM (one in-order counter)
    double* mpark = reinterpret_cast<double*>(smem);
    auto park_means = [&](int h, const Landmark<double>& m) {
      double* q = mpark + (size_t)(5 * h) * kRegsThreads + tid;
      q[0 * kRegsThreads] = m.mx;
      q[1 * kRegsThreads] = m.my;
      q[2 * kRegsThreads] = m.mr;
      q[3 * kRegsThreads] = m.mg;
      q[4 * kRegsThreads] = m.mb;
    };
    auto unpark_means = [&](int h, Landmark<double>& m) {
      const double* q = mpark + (size_t)(5 * h) * kRegsThreads + tid;
      m.mx = q[0 * kRegsThreads];
      m.my = q[1 * kRegsThreads];
      m.mr = q[2 * kRegsThreads];
      m.mg = q[3 * kRegsThreads];
      m.mb = q[4 * kRegsThreads];
    };
    uint4 cA0 = {}, cA1 = {}, cB0 = {}, cB1 = {};  // the two landmarks' candidate records
    bool done;
    {
      RegsArgsPtr R = regs_args_now(rp);
      done = p >= R->P;
      if (!done) {
      const SlotSource ss = regs_slot_source(R);
      const unsigned char* sslot = ss.at(R->f.src[p]);
      sf = reinterpret_cast<const double*>(sslot);
      sc = reinterpret_cast<const int*>(sslot + R->f.count_off);
      // ---- 1. the state is requested in three instalments, each one phase before it is needed (a row that is
      // requested now but used three phases later holds its registers all the way): the means of both landmarks
      // now, the first landmark's covariance rows behind its gates, the second one's behind the second gates
      A.mx = row(F_MX, oA);
      A.my = row(F_MY, oA);
      A.mr = row(F_MR, oA);
      A.mg = row(F_MG, oA);
      A.mb = row(F_MB, oA);
      Bq.mx = row(F_MX, oB);
      Bq.my = row(F_MY, oB);
      Bq.mr = row(F_MR, oB);
      Bq.mg = row(F_MG, oB);
      Bq.mb = row(F_MB, oB);
      if (CAND) {
        // straight to their places in LDS (the lane's own: whatever it kept there of the previous particle it has used): nothing
        // of the state is held in a register across the barriers and the zeroing
        park_means(0, A);
        park_means(1, Bq);
      }
      }
    }
    lds_barrier();  // every wave is through with the previous particle: its bids have been read, its partial sums are in
    if (prev >= 0 && tid == 0) {
      RegsArgsPtr R = regs_args_now(rp);
      double tot = red[0];
#pragma unroll
      for (int i = 1; i < kRegsThreads / kWave; ++i) tot += red[i];
      double* logw = R->f.logw;
      const double w = (R->f.reset ? 0.0 : logw[prev]) + tot;
      logw[prev] = w;
      unsigned long long* gk = R->f.gmax_key;
      if (gk) atomicMax(gk + (prev & (kGmaxKeys - 1)), double_to_key(w));
      R->f.src[prev] = (int32_t)prev;
    }
    if (done) break;
    prev = -1;
    {
      int* ccount = lds.ccount(smem);
      unsigned long long* best = lds.best(smem);
      for (int t = tid; t < B; t += kRegsThreads) {
        ccount[t] = 0;
        best[t] = 0ull;
      }
      if (tid == 0) {
        int* qn = lds.qn(smem);
        qn[0] = 0;
        qn[1] = 0;
        wg_flag = 0;
      }
    }
    PK_STAMP(r1)
    PK_RSTAMP(0, r0, r1)  // scalars, requests, zeroing
    lds_barrier();
    // ---- 2. gates -------------------------------------------------------------------------------------
    PK_STAMP(r2)
    PK_RSTAMP(1, r1, r2)  // barrier
    unsigned pA01 = 0xFFFFFFFFu, pA23 = 0xFFFFFFFFu, pB01 = 0xFFFFFFFFu, pB23 = 0xFFFFFFFFu;
    double pseA = 0.0, pseB = 0.0;
    {
      RegsArgsPtr R = regs_args_now(rp);
      const BlobGrid g = regs_blob_grid(R);
      const double* exact = R->f.exact;
      // the particle's pose is read again in every phase that uses it (scalar loads, their own counter): kept in
      // registers from the first phase to the last it was spilled to scratch, and scratch reloads queue behind the
      // rows that are on their way from HBM
      const double sx = regs_pose(R->f.x, p), sy = regs_pose(R->f.y, p), sh = regs_pose(R->h, p);
      if (CAND) {  // the candidate records of both landmarks (L2), the first landmark's means
        const uint4* crec = R->cand;
        const int iA = 2 * min(lA, Lp - 1), iB = 2 * min(lB, Lp - 1);
        cA0 = crec[iA];
        cA1 = crec[iA + 1];
        cB0 = crec[iB];
        cB1 = crec[iB + 1];
        unpark_means(0, A);
      }
      PK_STAMP(r3)
      PK_RSTAMP(2, r2, r3)  // gate arguments (scalar loads)
      if (hasA) {
        const RegsGated r = CAND ? regs_gates_cand(cA0, cA1, exact, lds.ccount(smem), &wg_flag, A.mx, A.my, A.mr, A.mg, A.mb, sx, sy, sh)
                                 : regs_gates(g, lds.start(smem), lds.rec32(smem), lds.idx9(smem), exact, lds.ccount(smem), &wg_flag, A.mx, A.my, A.mr, A.mg, A.mb, sx, sy, sh);
        pA01 = r.pass01;
        pA23 = r.pass23;
        pseA = r.pse;
      }
      // the first landmark's covariance rows.  Grid walk: requested here, behind its gates (the offset is made to depend on
      // their result: the requests cannot be moved up), they arrive while the second landmark's gates are worked out.
      // Candidate lists: requested behind BOTH gates -- the second landmark's gates read blob records from L2, and the
      // vector memory counter retires in order: behind rows that come from HBM those reads would wait for the rows.
      auto request_cov_a = [&](unsigned dep) {
        unsigned oA2 = oA;
        asm volatile("" : "+v"(oA2) : "v"(dep));
        A.pxx = row(F_PXX, oA2);
        A.pxy = row(F_PXY, oA2);
        A.pyy = row(F_PYY, oA2);
        A.crr = row(F_CRR, oA2);
        A.crg = row(F_CRG, oA2);
        A.crb = row(F_CRB, oA2);
        A.cgg = row(F_CGG, oA2);
        A.cgb = row(F_CGB, oA2);
        A.cbb = row(F_CBB, oA2);
        A.count = *reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(sc) + (oA2 >> 1));
      };
      if (!CAND) request_cov_a(pA01);
      PK_STAMP(r4)
      PK_RSTAMP(3, r3, r4)  // gates of the first landmark (waits for its means)
      if (CAND) unpark_means(1, Bq);
      if (hasB) {
        const RegsGated r = CAND ? regs_gates_cand(cB0, cB1, exact, lds.ccount(smem), &wg_flag, Bq.mx, Bq.my, Bq.mr, Bq.mg, Bq.mb, sx, sy, sh)
                                 : regs_gates(g, lds.start(smem), lds.rec32(smem), lds.idx9(smem), exact, lds.ccount(smem), &wg_flag, Bq.mx, Bq.my, Bq.mr, Bq.mg, Bq.mb, sx, sy, sh);
        pB01 = r.pass01;
        pB23 = r.pass23;
        pseB = r.pse;
      }
      if (CAND) request_cov_a(pB01);
      PK_STAMP(r5)
      PK_RSTAMP(4, r4, r5)  // gates of the second landmark
    }
    PK_STAMP(r6)
    lds_barrier();
    PK_STAMP(r7)
    PK_RSTAMP(5, r6, r7)  // barrier behind the gates
    // ---- 3. flagged particles go the general way (nothing has been written) ------------------------------
    if (wg_flag) {  // workgroup-uniform
      if (tid == 0) {
        RegsArgsPtr R = regs_args_now(rp);
        R->pflag_out[p] = 1;
        atomicAdd(R->n_flagged, 1u);
      }
      continue;  // (the barrier at the top of the next turn: everybody has read the flag before it is cleared)
    }
    // ---- 4. settling: one round per landmark of the lane -----------------------------------------------------
    int nun = 0;  // blobs no landmark passes
    {
      const int* ccount = lds.ccount(smem);
      for (int t = tid; t < B; t += kRegsThreads) nun += ccount[t] == 0;
    }
    unsigned qA01, qA23, qB01, qB23;
    PK_STAMP(r8)
    PK_RSTAMP(6, r7, r8)  // warming requests, count of unseen blobs
    {
      RegsArgsPtr R = regs_args_now(rp);
      const double sx = regs_pose(R->f.x, p), sy = regs_pose(R->f.y, p);
      if (CAND) unpark_means(0, A);
      FastSlot sl[kFastSlots];
      const FastQueue fq = lds.queue(smem, lds.qn(smem));
      fast_prepare<false, int, kRegsQueue>(ra_unused.f, R->f.exact, A, sx, sy, pseA, hasA ? make_uint2(pA01, pA23) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu), lds.ccount(smem), lds.best(smem), fq, sl);
      regs_pack(sl, qA01, qA23);
      // the second landmark's covariance rows: requested here (the offset depends on the first round's result:
      // the requests cannot be moved up), they arrive while the first round's queue is evaluated
      unsigned oB2 = oB;
      asm volatile("" : "+v"(oB2) : "v"(qA01));
      Bq.pxx = row(F_PXX, oB2);
      Bq.pxy = row(F_PXY, oB2);
      Bq.pyy = row(F_PYY, oB2);
      Bq.crr = row(F_CRR, oB2);
      Bq.crg = row(F_CRG, oB2);
      Bq.crb = row(F_CRB, oB2);
      Bq.cgg = row(F_CGG, oB2);
      Bq.cgb = row(F_CGB, oB2);
      Bq.cbb = row(F_CBB, oB2);
      Bq.count = *reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(sc) + (oB2 >> 1));
    }
    PK_STAMP(r9)
    PK_RSTAMP(7, r8, r9)  // first settling round: prepare (waits for the covariance rows)
    lds_barrier();
    const int nA = lds.qn(smem)[0];
    PK_STAMP(r10)
    regs_evaluate_queue<kRegsThreads / kWave>(lds.queue(smem, nullptr), min(nA, kRegsQueue), lds.results(smem), lds.best(smem), tid);
    PK_STAMP(r11)
    PK_RSTAMP(8, r10, r11)  // first round: queue evaluation
    lds_barrier();
    PK_STAMP(r12)
    {
      RegsArgsPtr R = regs_args_now(rp);
      const double sx = regs_pose(R->f.x, p), sy = regs_pose(R->f.y, p);
      if (CAND) unpark_means(1, Bq);
      FastSlot sl[kFastSlots];
      const FastQueue fq = lds.queue(smem, lds.qn(smem) + 1);
      fast_prepare<false, int, kRegsQueue>(ra_unused.f, R->f.exact, Bq, sx, sy, pseB, hasB ? make_uint2(pB01, pB23) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu), lds.ccount(smem), lds.best(smem), fq, sl);
#pragma unroll
      for (int k = 0; k < kFastSlots; ++k)
        if (sl[k].qf >> 4) sl[k].qf += (unsigned)kRegsQueue << 4;  // the second round's results live behind the first's
      regs_pack(sl, qB01, qB23);
    }
    PK_STAMP(r13)
    PK_RSTAMP(9, r12, r13)  // second round: prepare
    lds_barrier();
    const int nB = lds.qn(smem)[1];
    // more probabilities wanted in a round than the queue holds (dense clusters of look-alike landmarks):
    // nothing has been written yet, the general kernels take the particle
    if (nA > kRegsQueue || nB > kRegsQueue) {  // workgroup-uniform
      if (tid == 0) {
        RegsArgsPtr R = regs_args_now(rp);
        R->pflag_out[p] = 1;
        atomicAdd(R->n_flagged, 1u);
      }
      continue;
    }
    // which of the lane's blobs are contested (the counts then make room for the bids: win aliases ccount)
    unsigned cont = 0u;
    {
      const int* ccount = lds.ccount(smem);
      int j = 0;
      for (int t = tid; t < B; t += kRegsThreads, ++j) cont |= (ccount[t] >= 2 ? 1u : 0u) << j;
    }
    PK_STAMP(r14)
    regs_evaluate_queue<kRegsThreads / kWave>(lds.queue(smem, nullptr), nB, lds.results(smem) + kRegsQueue, lds.best(smem), tid);
    PK_STAMP(r15)
    PK_RSTAMP(10, r14, r15)  // second round: queue evaluation
    lds_barrier();  // every count has been read (fast_prepare, cont), every probability is in
    int* win = lds.ccount(smem);
    for (int t = tid; t < B; t += kRegsThreads) win[t] = INT_MAX;
    if (tid == 0) regs_args_now(rp)->pflag_out[p] = 0;
    lds_barrier();
    {
      FastSlot sl[kFastSlots];
      regs_unpack(pA01, pA23, qA01, qA23, sl);
      fast_collect(lds.results(smem), lds.best(smem), win, lA, sl);
      regs_pack(sl, qA01, qA23);
      regs_unpack(pB01, pB23, qB01, qB23, sl);
      fast_collect(lds.results(smem), lds.best(smem), win, lB, sl);
      regs_pack(sl, qB01, qB23);
    }
    lds_barrier();
    {
      const unsigned long long* best = lds.best(smem);
      int j = 0;
      for (int t = tid; t < B; t += kRegsThreads, ++j) nun += (((cont >> j) & 1u) && best[t] == 0ull);  // contested, all 0
    }
    double acc = (double)nun * Consts<double>::log_no_match;
    PK_STAMP(r16)
    PK_RSTAMP(11, r15, r16)  // bids: barriers, win, collect
    {  // the next particle's map slot: one dword per 128-byte line, so that the lines wait in L2
      RegsArgsPtr R = regs_args_now(rp);
      const int warm = R->warm;
      if (warm != 0 && p + gridDim.x < R->P) {
        const SlotSource ss = regs_slot_source(R);
        const size_t warm_lines = warm >= 2 ? ss.slot_bytes >> 7 : ((size_t)5 * Lp * 8 + 127) >> 7;  // slot_bytes: a multiple of 256
        const unsigned char* nslot = ss.at(R->f.src[p + gridDim.x]);
        for (size_t i = (size_t)tid; i < warm_lines; i += kRegsThreads)
          // (address spaces spelled out: through generic pointers the builtin is accepted but M0, the LDS base of
          // the transfer, is never set)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned*)(nslot + (i << 7)),
                                           (__attribute__((address_space(3))) unsigned*)warm_dump, 4, 0, 0);
      }
    }
    // ---- 5. updates in scan order, stores: the first landmark, then the second ---------------------------------
    // (the second one's colour block waits in LDS meanwhile: the queue and its results are dead)
    double* park = lds.park(smem);
    park[0 * kRegsThreads + tid] = Bq.crr;
    park[1 * kRegsThreads + tid] = Bq.crg;
    park[2 * kRegsThreads + tid] = Bq.crb;
    park[3 * kRegsThreads + tid] = Bq.cgg;
    park[4 * kRegsThreads + tid] = Bq.cgb;
    park[5 * kRegsThreads + tid] = Bq.cbb;
    {
      RegsArgsPtr R = regs_args_now(rp);
      const double* exact = R->f.exact;
      const unsigned short* order = R->f.order;
      const Noise<double> qt = regs_noise(R);
      const double sx = regs_pose(R->f.x, p), sy = regs_pose(R->f.y, p);
      unsigned char* dslot = R->f.map_dst + (size_t)p * R->f.ss.slot_bytes;
      double* df = reinterpret_cast<double*>(dslot);
      int* dc = reinterpret_cast<int*>(dslot + R->f.count_off);
      Landmark<double> cur = A;
      if (CAND) {
        // the first landmark's means come back; where they stood, the rest of the second landmark waits out the first
        // one's update (position block, expected bearing, blob and flag words): nothing of it is in a register then
        unpark_means(0, cur);
        asm volatile("" ::: "memory");  // the words below overwrite doubles that have just been read (type-based aliasing would let them pass)
        double* q = mpark + tid;
        q[0 * kRegsThreads] = Bq.pxx;
        q[1 * kRegsThreads] = Bq.pxy;
        q[2 * kRegsThreads] = Bq.pyy;
        q[3 * kRegsThreads] = pseB;
        reinterpret_cast<uint2*>(mpark + 4 * kRegsThreads)[tid] = make_uint2(pB01, pB23);  // the lane's own 8 bytes of the row
        // (after fast_collect only the four flag bits of a slot's word are read)
        reinterpret_cast<unsigned*>(park + 6 * kRegsThreads)[tid] = (qB01 & 0x000F000Fu) | ((qB23 & 0x000F000Fu) << 4);
      }
      double pseC = pseA;
      unsigned pc01 = pA01, pc23 = pA23, qc01 = qA01, qc23 = qA23;
      bool immC = immA, hasC = hasA;
      int lC = lA;
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        if (hasC) {
          FastSlot sl[kFastSlots];
          regs_unpack(pc01, pc23, qc01, qc23, sl);
          acc += regs_apply(exact, order, qt, cur, lC, immC, sx, sy, pseC, sl, win);
        }
        if (lC < Lp) {
          unsigned off = (unsigned)lC * 8u;
          asm volatile("" : "+v"(off));
          regs_store_landmark(df, dc, Lp, off, cur);
        }
        if (CAND) {
          unpark_means(1, cur);
        } else {
          cur.mx = Bq.mx;
          cur.my = Bq.my;
          cur.mr = Bq.mr;
          cur.mg = Bq.mg;
          cur.mb = Bq.mb;
        }
        if (CAND) {
          const double* q = mpark + tid;
          cur.pxx = q[0 * kRegsThreads];
          cur.pxy = q[1 * kRegsThreads];
          cur.pyy = q[2 * kRegsThreads];
        } else {
          cur.pxx = Bq.pxx;
          cur.pxy = Bq.pxy;
          cur.pyy = Bq.pyy;
        }
        cur.count = Bq.count;
        cur.crr = park[0 * kRegsThreads + tid];
        cur.crg = park[1 * kRegsThreads + tid];
        cur.crb = park[2 * kRegsThreads + tid];
        cur.cgg = park[3 * kRegsThreads + tid];
        cur.cgb = park[4 * kRegsThreads + tid];
        cur.cbb = park[5 * kRegsThreads + tid];
        if (CAND) {
          pseC = mpark[3 * kRegsThreads + tid];
          const uint2 w = reinterpret_cast<const uint2*>(mpark + 4 * kRegsThreads)[tid];
          pc01 = w.x;
          pc23 = w.y;
          const unsigned qp = reinterpret_cast<const unsigned*>(park + 6 * kRegsThreads)[tid];
          qc01 = qp & 0x000F000Fu;
          qc23 = (qp >> 4) & 0x000F000Fu;
        } else {
          pseC = pseB;
          pc01 = pB01;
          pc23 = pB23;
          qc01 = qB01;
          qc23 = qB23;
        }
        immC = immB;
        hasC = hasB;
        lC = lB;
      }
    }
    PK_STAMP(r17)
    PK_RSTAMP(12, r16, r17)  // updates + stores issued
    {
      const double ws = wave_sum(acc);  // the sum over the workgroup is finished behind the next barrier (top of the loop)
      if ((tid & (kWave - 1)) == 0) red[tid / kWave] = ws;
      prev = p;
    }
    PK_STAMP(r18)
    PK_RSTAMP(13, r17, r18)  // block sum + tail
    PK_RSTAMP(14, r0, r18)   // particle
#ifdef PK_STAMPS
    if (tid == 0) atomicAdd(&pk_rstamp_acc[15], (unsigned long long)(nA + nB));  // queued probabilities
#endif
  }
}

void launch_step_regs(hipStream_t s, DeviceState& d, int B, const BlobGrid& grid, int n9, const unsigned char* tables_dev,
                      const double* exact_dev, const unsigned short* order_dev, const FastHandoff& fh, const NoiseD& qt,
                      const ObserveExtras& ex, int warm, const CandTable& cand, int64_t p0, int64_t p1, int reserve_cus) {
  if (p1 < 0) p1 = d.P;
  if (d.P == 0 || p1 <= p0) return;
  static bool attr_set[kMaxDevices] = {false};
  if (first_time_on_this_device(attr_set)) {
    for (const void* fn : {reinterpret_cast<const void*>(k_step_regs<false>), reinterpret_cast<const void*>(k_step_regs<true>)})
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds) != hipSuccess) (void)hipGetLastError();
  }
  RegsArgs ra;
  FastArgs& a = ra.f;
  a.ss = slot_source(d);
  a.map_dst = d.map[d.mcur ^ 1];
  a.count_off = d.lay.count_off;
  a.src = d.src[d.cur];
  a.x = d.x[d.cur];
  a.y = d.y[d.cur];
  a.logw = d.logw[d.cur];
  a.exact = exact_dev;
  a.order = order_dev;
  a.lmpass = nullptr;
  a.bcount = nullptr;
  a.pflag = nullptr;
  a.immutable = d.immutable;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  a.reset = ex.reset ? 1 : 0;
  a.gmax_key = ex.gmax_key;
  a.qt = make_noise(qt.q00, qt.rr, qt.rg, qt.rb, qt.gg, qt.gb, qt.bb);
  ra.g = grid;
  ra.tables = tables_dev;
  ra.h = d.h[d.cur];
  ra.pflag_out = fh.pflag;
  ra.n_flagged = fh.n_flagged;
  ra.n9 = n9;
  ra.warm = warm;
  ra.P = p1;
  ra.p_begin = p0;
  ra.cand = cand.rec;
  ra.cand_over = cand.rec ? cand.over : nullptr;
  ra.cand_skip = cand.rec ? (cand.skip_cand ? cand.skip_cand : cand.over) : nullptr;
  // persistent grid: the workgroups that are resident at once (one per CU: 1024 lanes x 128 VGPRs)
  const int n_cu = device_cu_count();
  // reserve_cus: leave that many CUs without a workgroup -- this kernel's workgroups hold a CU's whole register file for the
  // whole launch, and a collective that is to run meanwhile (the sharded filter's all-to-all) needs somewhere to run
  int64_t grid_n = n_cu - (reserve_cus > 0 && reserve_cus < n_cu ? reserve_cus : 0);
  if (grid_n > p1 - p0) grid_n = p1 - p0;
  if (cand.rec) {
    hipLaunchKernelGGL(k_step_regs<true>, dim3((unsigned)grid_n), dim3(kRegsThreads), regs_cand_lds_bytes(d.lay.Lp, B), s, ra);
    // a scan in which some candidate list overflowed: every particle goes to the fall-back kernels (second chance on the
    // eight-slot hand-off, then the general kernels)
    launch_flag_range_if(s, cand.over, fh.pflag, fh.n_flagged, p0, p1);
  } else {
    hipLaunchKernelGGL(k_step_regs<false>, dim3((unsigned)grid_n), dim3(kRegsThreads), regs_lds_bytes(grid.ncell, B, n9), s, ra);
  }
}

}  // namespace pk
