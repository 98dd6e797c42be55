// K2 maximum-likelihood data association: brute-force reference kernel and the colour-grid kernel.
//
// Hand-written gfx950 (CDNA4, wave64) kernels of the FastSLAM particle update; see DESIGN.md
// section 4.  No MFMA: the algebra is 2x2 / 3x3 and register resident (pk_math.hpp).
#include "pk_device.hpp"
#include "pk_pub_math.hpp"

namespace pk {

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
    double pse = pk_atan2(lm.my - sy, lm.mx - sx);
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
  const size_t lds = assoc_brute_lds_bytes(B);  // <= kMaxDynLds: checked by the caller
  static bool attr_set[kMaxDevices] = {false};
  if (first_time_on_this_device(attr_set)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_assoc_brute), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kMaxDynLds) != hipSuccess)
      (void)hipGetLastError();
  }
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
  // SLOTS = 4: [P][Lp]     x,y: four 16-bit fields = blob (cell order) or 0xFFFF; z,w: atan2(my-sy, mx-sx)
  // SLOTS = 8: [P][Lp][2]  first uint4: eight 16-bit blob fields; second: x,y = atan2(my-sy, mx-sx)
  uint4* lmpass;
  unsigned char* bcount;  // [P][B]   saturating count
  unsigned char* pflag;   // [P]      1 = general path
  const unsigned char* only_flagged;  // GENERAL instance: skip particles whose flag is 0
  unsigned* n_flagged;                // count of flagged particles (zeroed by the scan upload)
  // hand-off instance: the candidate lists of a reference particle (CandTable, pk_kernels.hpp) stand in for the walk through
  // the colour grid as long as no list overflowed (*cand_over == 0); NULL: always the grid
  const uint4* cand_rec;
  const unsigned* cand_over;
  int cand_slots;  // kCandSlots (records of two uint4) or 2 kCandSlots (three)
  int retry;       // hand-off instance: only the particles whose flag is 1; leaves 2 (handed off) or 1 (general kernels)
  int32_t* row_of;     // retry: hand-off row of each particle (FastHandoff)
  unsigned* row_next;  //        rows dealt out so far
  int64_t row_cap;     //        rows there are
};

// tables: start u16[ncell+1] (16-byte padded) | rec32 float4[B] | idx9 u16[n9] (DUP only) | order u16[B]
// `start` is cell_start (offsets into rec32) or, with DUP, col_start (offsets into idx9).
size_t blob_grid_table_bytes(int ncell, int B, int n9) {
  return grid_cs_bytes(ncell) + (size_t)B * 16 + (size_t)n9 * 2 + (size_t)B * 2;
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
#define PK_STAMP_ADD(slot, a, b) \
  if ((threadIdx.x & 63) == 0) atomicAdd(&pk_stamp_acc[slot], (b) - (a));
#else
#define PK_STAMP_ADD(slot, a, b)
#endif

// GENERAL = false: S1 + hand-off only (light on registers); particles it flags are redone by
// the GENERAL = true instance launched with only_flagged.
template <int THREADS, bool DUP, bool GENERAL, int SLOTS>
__global__ void __launch_bounds__(THREADS) k_assoc_grid(AssocGridArgs ga) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int n_few, n_many, wg_flag;
  __shared__ long long s_row;
  __shared__ unsigned long long s3_best[THREADS / 4];
  __shared__ int s3_win[THREADS / 4];
  const AssocArgs& a = ga.a;
  const BlobGrid& g = ga.g;
  const int B = a.B;
  if (GENERAL && ga.only_flagged && *ga.n_flagged == 0u) return;  // nothing was flagged
  if (!GENERAL && ga.retry && *ga.n_flagged == 0u) return;
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
  // hand-off instance: gates from the reference particle's candidate lists (workgroup-uniform choice, once per launch)
  const bool use_cand = !GENERAL && ga.cand_rec != nullptr && *ga.cand_over == 0u;
  for (int64_t p = blockIdx.x; p < ga.P; p += gridDim.x) {
    if (GENERAL && ga.only_flagged && ga.only_flagged[p] != 1) continue;  // workgroup-uniform
    if (!GENERAL && ga.retry && ga.pflag[p] != 1) continue;
    int64_t row = p;  // where this particle's hand-off entries go
    if (!GENERAL && ga.retry) {  // second chance: the next free row, if there is one (workgroup-uniform)
      __syncthreads();
      if (threadIdx.x == 0) s_row = (long long)atomicAdd(ga.row_next, 1u);
      __syncthreads();
      row = s_row;
      if (row >= ga.row_cap) continue;  // none left: the flag stays 1, the general kernels take the particle
      if (threadIdx.x == 0) ga.row_of[p] = (int32_t)row;
    }
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
      const double pse = pk_atan2(my - sy, mx - sx);
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
      // survivors of the fp32 screen: the LAST kCand = 4 kept in a 64-bit shift register of 16-bit
      // blob indices (two instructions per survivor instead of a compare-and-select chain)
      unsigned slo = 0u, shi = 0u;
      int npc = 0;
      auto push = [&](int t) {
        shi = __builtin_amdgcn_alignbit(shi, slo, 16);
        slo = (slo << 16) | (unsigned)t;
        ++npc;
      };
      PK_STAMP(ta1)
      PK_STAMP_ADD(1, ta0, ta1)
      // fp32 screen on packed pairs (v_pk_add_f32 / v_pk_mul_f32): (r, g) and (b, bearing)
      typedef float Float2 __attribute__((ext_vector_type(2)));
      const Float2 m01 = {mr32, mg32}, m23 = {mb32, eb32};
      auto prefilter_q = [&](const float4& q) {
        const Float2 q01 = {q.x, q.y}, q23 = {q.z, q.w};
        const Float2 d01 = q01 - m01, d23 = q23 - m23;
        const Float2 s01 = d01 * d01;
        const float cd32 = fmaf(d23.x, d23.x, s01.x + s01.y);
        // conservative fp32 gates; NaN/inf fall through to the exact float64 tests
        return !(cd32 > g.thr32) && !(fabsf(d23.y) > g.thrb32);
      };
      auto prefilter = [&](int t) { return prefilter_q(rec32[t]); };
      unsigned pass[SLOTS / 2];  // the blobs that pass this landmark's gates (first SLOTS), two per word
#pragma unroll
      for (int j = 0; j < SLOTS / 2; ++j) pass[j] = 0xFFFFFFFFu;
      int npass = 0;
      auto exact_gates = [&](int tt, const double2& z01, const double2& z23) {
        if (!(fabs(z01.x - eb) > 0.5) && !(fabs(color_distance2(mr, mg, mb, z01.y, z23.x, z23.y)) > 300.0)) {
          const int n = atomicAdd(&ccount[tt], 1);
          if (GENERAL && n < 4) cand[4 * tt + n] = (unsigned short)l;  // the hand-off instance keeps no candidate lists
#pragma unroll
          for (int j = 0; j < SLOTS; ++j)
            if (npass == j)
              pass[j >> 1] = (j & 1) ? ((pass[j >> 1] & 0x0000FFFFu) | ((unsigned)tt << 16))
                                     : ((pass[j >> 1] & 0xFFFF0000u) | (unsigned)tt);
          ++npass;
        }
      };
      if (!GENERAL && use_cand) {
        // ---- candidate lists (k_step_regs' gates, pk_k_observe_ml.hip: regs_gates_cand): the landmark's expected bearing
        // and colour must lie within the margins of the reference particle's -- else the particle goes the general way --
        // and then only the listed blobs can pass; exact float64 gates on those, two records per L2 round trip
        const bool wide = ga.cand_slots > kCandSlots;
        const uint4* crec = ga.cand_rec + (wide ? 3 : 2) * (size_t)l;
        const uint4 ref = crec[0], cl = crec[1];
        uint4 cl2 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (wide) cl2 = crec[2];
        const double deb = eb - (double)__uint_as_float(ref.x);  // 2 pi off: the other side of a branch cut, listed too
        const bool inside = (fabs(deb) <= kCandBearing || fabs(deb - Consts<double>::two_pi) <= kCandBearing ||
                             fabs(deb + Consts<double>::two_pi) <= kCandBearing) &&
                            fabs(mr - (double)__uint_as_float(ref.y)) <= kCandColour &&
                            fabs(mg - (double)__uint_as_float(ref.z)) <= kCandColour &&
                            fabs(mb - (double)__uint_as_float(ref.w)) <= kCandColour;
        if (!inside) wg_flag = 1;
        unsigned c0 = cl.x, c1 = cl.y, c2 = cl.z, c3 = cl.w, c4 = cl2.x, c5 = cl2.y, c6 = cl2.z, c7 = cl2.w;  // filled from the front
#pragma unroll 1
        for (int k = 0; k < 2 * kCandSlots; k += 2) {
          const int ta = (int)(c0 & 0xFFFFu), tb = (int)(c0 >> 16);
          if (ta == 0xFFFF) break;
          c0 = c1;
          c1 = c2;
          c2 = c3;
          c3 = c4;
          c4 = c5;
          c5 = c6;
          c6 = c7;
          c7 = 0xFFFFFFFFu;
          const double* ra = ga.exact + 6 * (size_t)ta;
          const double* rb = ga.exact + 6 * (size_t)(tb == 0xFFFF ? ta : tb);
          const double2 a01 = *reinterpret_cast<const double2*>(ra);
          const double2 a23 = *reinterpret_cast<const double2*>(ra + 2);
          const double2 b01 = *reinterpret_cast<const double2*>(rb);
          const double2 b23 = *reinterpret_cast<const double2*>(rb + 2);
          exact_gates(ta, a01, a23);
          if (tb != 0xFFFF) exact_gates(tb, b01, b23);
        }
      } else {
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
        // four list entries per round trip: the index reads and then the record reads are
        // independent, so a walk of n blobs costs ceil(n / 4) dependent LDS latencies, not n
        // (idx9 is padded: reading up to three entries past i1 stays inside the list)
        for (; i < i1; i += 4) {
          int t4[4];
          float4 q4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) t4[j] = idx9[i + j];
#pragma unroll
          for (int j = 0; j < 4; ++j) q4[j] = rec32[t4[j]];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (i + j < i1 && prefilter_q(q4[j])) push(t4[j]);
          }
        }
      } else {
        // the 9 (r, g) columns x [k0, k1]: nine contiguous record ranges, four records per round trip
        // (reads past a range end stay inside rec32 or return the next table's bytes: masked out)
        if (k0 <= k1) {
          for (int j = 0; j < 9; ++j) {
            const int jr = (j * 11) >> 5;  // j / 3 for j < 9
            const int r = c[0] - 1 + jr, gg = c[1] - 1 + (j - 3 * jr);
            if ((unsigned)r >= (unsigned)g.G[0] || (unsigned)gg >= (unsigned)g.G[1]) continue;
            const int base = (r * g.G[1] + gg) * g.G[2];
            const int t1 = start[base + k1 + 1];
            for (int t = start[base + k0]; t < t1; t += 4) {
              float4 q4[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) q4[i] = rec32[min(t + i, B - 1)];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (t + i < t1 && prefilter_q(q4[i])) push(t + i);
              }
            }
          }
        }
      }
      PK_STAMP(ta2)
      PK_STAMP_ADD(2, ta1, ta2)
      // ---- phase 2 (global, convergent): all survivors' exact records in one batch -----
      if (__any(npc > 0)) {
        static_assert(kCand == 4, "the shift register holds four 16-bit indices");
        const int pc[kCand] = {(int)(slo & 0xFFFFu), (int)(slo >> 16), (int)(shi & 0xFFFFu), (int)(shi >> 16)};
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
        // the ones before the last kCand as they come (nine-range walk works for both layouts
        // only without DUP; with DUP repeat the single range)
        int seen = 0;
        auto late = [&](int t) {
          if (prefilter(t)) {
            if (seen < npc - kCand) {
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
      }  // grid walk
      if (!GENERAL) {
        const unsigned long long pb = (unsigned long long)__double_as_longlong(pse);
        if (SLOTS == 4) {
          ga.lmpass[(size_t)row * a.Lp + l] = make_uint4(pass[0], pass[1], (unsigned)pb, (unsigned)(pb >> 32));
        } else {
          uint4* e = ga.lmpass + 2 * ((size_t)row * a.Lp + l);
          e[0] = make_uint4(pass[0], pass[1], pass[SLOTS / 2 - 2], pass[SLOTS / 2 - 1]);
          e[1] = make_uint4((unsigned)pb, (unsigned)(pb >> 32), 0u, 0u);
        }
        if (npass > SLOTS) wg_flag = 1;
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
        ga.bcount[(size_t)row * B + t] = (unsigned char)(n > 255 ? 255 : n);
      }
      if (threadIdx.x == 0) {
        if (ga.retry) {  // (already counted by the kernel that flagged it)
          ga.pflag[p] = (unsigned char)(wg_flag != 0 ? 1 : 2);
        } else {
          ga.pflag[p] = (unsigned char)(wg_flag != 0);
          if (wg_flag) atomicAdd(ga.n_flagged, 1u);
        }
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
        const double eb = pk_atan2(my - sy, mx - sx) - sh;
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

template <int THREADS, bool DUP, bool GENERAL, int SLOTS>
static void launch_assoc_grid_t(hipStream_t s, const AssocGridArgs& ga, size_t lds, int64_t P) {
  static bool attr_set[kMaxDevices] = {false};
  if (first_time_on_this_device(attr_set)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_assoc_grid<THREADS, DUP, GENERAL, SLOTS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds) != hipSuccess)
      (void)hipGetLastError();  // leave no sticky error behind for other users of the runtime
  }
  // persistent grid = the workgroups that are resident at once (registers, LDS and the wave limit all
  // count: a workgroup that has to wait for a slot would run its particles after the others finished)
  static size_t asked_lds = ~(size_t)0;
  static int asked_per_cu = 0;
  if (asked_lds != lds) {
    asked_lds = lds;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&asked_per_cu,
                                                     reinterpret_cast<const void*>(k_assoc_grid<THREADS, DUP, GENERAL, SLOTS>),
                                                     THREADS, lds) != hipSuccess) {
      (void)hipGetLastError();
      asked_per_cu = 0;
    }
  }
  int per_cu = asked_per_cu;
  if (per_cu < 1) {  // fall back to the LDS / thread-count estimate
    per_cu = (int)((160 * 1024) / (lds + 64));
    per_cu = per_cu < 1 ? 1 : per_cu;
    const int by_threads = 2048 / THREADS;
    per_cu = per_cu > by_threads ? by_threads : per_cu;
  }
  int64_t blocks = 256 * (int64_t)per_cu;
  if (blocks > P) blocks = P;
  hipLaunchKernelGGL((k_assoc_grid<THREADS, DUP, GENERAL, SLOTS>), dim3((unsigned)blocks), dim3(THREADS), lds, s, ga);
}

void launch_assoc_grid(hipStream_t s, DeviceState& d, int B, const BlobGrid& grid, int n9,
                       const unsigned char* tables_dev, const double* exact_dev, int32_t* ids_dev, bool finalize,
                       const FastHandoff& fh, const CandTable& cand) {
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
  ga.cand_rec = cand.rec;
  ga.cand_over = cand.rec ? cand.over : nullptr;
  ga.cand_slots = cand.slots;
  ga.retry = fh.retry ? 1 : 0;
  ga.row_of = fh.row_of;
  ga.row_next = fh.row_next;
  ga.row_cap = fh.row_cap;
  auto go = [&](auto general, auto slots) {
    constexpr bool G = decltype(general)::value;
    constexpr int S = decltype(slots)::value;
    // the hand-off instance needs the tables and the per-blob counts only (20 B per blob, not 30)
    const size_t lds = G ? assoc_grid_lds_bytes(grid.ncell, B, n9) : assoc_grid_lds_bytes(grid.ncell, B, n9) - (size_t)B * 10;
    // bigger workgroups when the LDS tables are large (fewer copies of them per CU)
    const bool big = lds > 40 * 1024;
    // one workgroup per CU when the tables are that large: the hand-off instance fits three waves per
    // SIMD in its registers (768 threads), the general one two (512)
    constexpr int kBigThreads = G ? 512 : 768;
    if (n9 > 0) {
      if (big)
        launch_assoc_grid_t<kBigThreads, true, G, S>(s, ga, lds, d.P);
      else
        launch_assoc_grid_t<256, true, G, S>(s, ga, lds, d.P);
    } else {
      if (big)
        launch_assoc_grid_t<kBigThreads, false, G, S>(s, ga, lds, d.P);
      else
        launch_assoc_grid_t<256, false, G, S>(s, ga, lds, d.P);
    }
  };
  using Slots4 = std::integral_constant<int, kFastSlots>;
  using Slots8 = std::integral_constant<int, kSweepSlots>;
  if (fh.flags_only) {
    ga.only_flagged = fh.pflag;
    go(std::true_type{}, Slots4{});  // k_step_fused did the others
  } else if (fh.lmpass && fh.retry) {
    go(std::false_type{}, Slots8{});  // the flagged particles only; the caller runs the sweep and the general kernels
  } else if (fh.lmpass) {
    // gate tests + hand-off for every particle
    if (fh.slots == kSweepSlots)
      go(std::false_type{}, Slots8{});
    else
      go(std::false_type{}, Slots4{});
    ga.only_flagged = fh.pflag;
    go(std::true_type{}, Slots4{});  // the flagged ones (a landmark with more gate-passing blobs than slots) the general way
  } else {
    go(std::true_type{}, Slots4{});
  }
}

void launch_assoc_grid(hipStream_t s, DeviceState& d, int B, const BlobGrid& grid, int n9,
                       const unsigned char* tables_dev, const double* exact_dev, int32_t* ids_dev, bool finalize,
                       const FastHandoff& fh) {
  launch_assoc_grid(s, d, B, grid, n9, tables_dev, exact_dev, ids_dev, finalize, fh, CandTable{});
}

// ------------------------------------------------------------------ K2' candidate lists from a reference particle
// See pk_kernels.hpp (CandTable).  A workgroup takes 64 landmarks of the reference particle; its sixteen waves share the
// scan's blobs (uniform reads of the exact records), candidates are appended per landmark through LDS atomics.
struct CandArgs {
  SlotSource ss;
  const int32_t* src;
  const double *x, *y, *h;
  const double* exact;  // [B][6] in cell order: bearing, r, g, b, ux, uy
  uint4* rec;           // [Lp][2] (SLOTS = kCandSlots) or [Lp][3] (twice as many)
  unsigned* over;
  unsigned* bcnt;            // [B] entries of each blob's inverse list (cleared by the launcher), or NULL
  unsigned short* blist;     // [B][inv_slots] landmarks listing each blob (0xFFFF-filled by the launcher)
  int inv_slots;             // entries per inverse list: kCandSlots, or as many as SLOTS
  int64_t ref;
  int L, Lp, B;
  const double* pose4;  // sums of x, y, sin h, cos h over the P particles (k_summary_*): the reference POSE is their mean, or NULL
  const double* part;   // ... or the per-block sums the motion launch left ([n_part][4]), reduced here in a fixed order
  int64_t n_part;
  int64_t P;
  unsigned char* npass;  // [Lp + kCandSpare] out: blobs inside the reference's OWN gates (:433, :441) -- what a particle's verdict rounds
                         // will be about; k_cand_entries orders the lanes of k_step_pub by it.  Or NULL
  uint4* far;            // [Lp + kCandSpare][1 + SLOTS / 8] out, or NULL: (Kb, Ib as float, entries of the far list, 0) | the FAR list --
                         // look-alikes whose key is certainly beyond the underflow edge for every particle whose own bound is at
                         // least (Kb, Ib) leave the landmark's list (pk_pub_math.hpp); NULL: nothing is taken off
};

constexpr int kCandThreads = 1024;  // 64 landmarks x 16 waves that share the scan's blobs
template <int SLOTS>
__global__ void __launch_bounds__(kCandThreads) k_candidates(CandArgs a) {
  static_assert(SLOTS == kCandSlots || SLOTS == 2 * kCandSlots, "records of two or three uint4");
  __shared__ unsigned short s_c[64][SLOTS], s_f[64][SLOTS];
  __shared__ int s_n[64], s_pass[64], s_nf[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int l = blockIdx.x * 64 + lane;
  const unsigned char* slot = a.ss.at(a.src[a.ref]);
  const double* f = reinterpret_cast<const double*>(slot);
  // the reference: particle ref's MAP seen from the MEAN pose of the particles (circular mean of the heading, as summary
  // :254-276 takes it) -- the particles' expected bearings scatter around it with their heading spread, and a particle
  // drawn at random (particle 0 itself) sits one sigma off the middle: twice the margin for the same cloud
  double sx = a.x[a.ref], sy = a.y[a.ref], sh = a.h[a.ref];
  if (a.part) {  // (kernel-uniform)
    // every WAVE adds the motion launch's per-block sums up for itself, in a fixed order (lane i: blocks i, i + 64, ...; then the
    // butterfly): no barrier, and the loads fly beside the landmark's rows and the scan's records
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = lane; i < a.n_part; i += 64) {
      const double2 p01 = *reinterpret_cast<const double2*>(a.part + 4 * i), p23 = *reinterpret_cast<const double2*>(a.part + 4 * i + 2);
      v[0] += p01.x;
      v[1] += p01.y;
      v[2] += p23.x;
      v[3] += p23.y;
    }
    for (int c = 0; c < 4; ++c) v[c] = wave_sum(v[c]);
    const double n = (double)a.P;
    sx = v[0] / n;
    sy = v[1] / n;
    sh = atan2(v[2], v[3]);
  } else if (a.pose4) {
    const double n = (double)a.P;
    sx = a.pose4[0] / n;
    sy = a.pose4[1] / n;
    sh = atan2(a.pose4[2], a.pose4[3]);
  }
  float ebf = 0.f, rf = 0.f, gf = 0.f, bf = 0.f;
  const bool has = l < a.L;
  if (has) {
    const double mx = f[(size_t)F_MX * a.Lp + l], my = f[(size_t)F_MY * a.Lp + l];
    ebf = (float)(pk_atan2(my - sy, mx - sx) - sh);  // :408 for the reference particle
    rf = (float)f[(size_t)F_MR * a.Lp + l];
    gf = (float)f[(size_t)F_MG * a.Lp + l];
    bf = (float)f[(size_t)F_MB * a.Lp + l];
  }
  // the reference's far bound with its margins, ROUNDED as the particles will read it (Ib = 0: nothing is far)
  float kbf = -3.0e38f, ibf = 0.f;
  if (has && a.far) {
    Landmark<double> lm = load_landmark_nocount(f, a.Lp, l);
    double fk, fi;
    pub_far_bound(lm, fk, fi);
    if (fi > 0.0 && fk == fk && fabs(fk) < 1e30) {  // (fi > 0: sane determinants, a positive definite colour block)
      kbf = pub_round_down_to_float(fk - kFarKeySlack);
      ibf = pub_round_down_to_float(fi / kFarVarFactor);
      if (!(ibf > 0.f)) ibf = 0.f;
    }
  }
  const double Kb = (double)kbf, Ib = (double)ibf;
  // the centres are the ROUNDED values the particles will compare themselves with
  const double cb = (double)ebf, cr = (double)rf, cg = (double)gf, cc = (double)bf;
  const double tb = 0.5 + kCandBearing + 1e-9;
  const double rad = 17.320508075688775 + 1.7320508075688773 * kCandColour + 1e-6;  // sqrt(300) + sqrt(3) margin
  const double tc = rad * rad;
  if (w == 0) {
    s_n[lane] = 0;
    s_pass[lane] = 0;
    s_nf[lane] = 0;
  }
  __syncthreads();
  // The scan's records go through LDS, 1 024 at a time (one coalesced round trip; read back at wave-uniform addresses: broadcasts).
  // Round 4 read them through the scalar cache where they were used -- a chain of cache misses per wave, 22 us of the
  // kernel's 25 at 500 x 500, where the work itself is a microsecond.
  __shared__ double s_rec[1024][4];
  for (int c0 = 0; c0 < a.B; c0 += 1024) {
    const int nb = min(1024, a.B - c0);
    __syncthreads();
    if ((int)threadIdx.x < nb) {
      const double2* r = reinterpret_cast<const double2*>(a.exact + 6 * (size_t)(c0 + threadIdx.x));
      const double2 r01 = r[0], r23 = r[1];
      s_rec[threadIdx.x][0] = r01.x;
      s_rec[threadIdx.x][1] = r01.y;
      s_rec[threadIdx.x][2] = r23.x;
      s_rec[threadIdx.x][3] = r23.y;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = w; i < nb; i += kCandThreads / 64) {  // wave-uniform
      const int t = c0 + i;
      const double zb = s_rec[i][0], zr = s_rec[i][1], zg = s_rec[i][2], zc = s_rec[i][3];
      const double dr = zr - cr, dg = zg - cg, dc = zc - cc;
      // NaN / inf in the reference's state fail both tests: no candidates, and every particle near such a state
      // breaks the margin test (comparisons with NaN are false) and goes the general way
      // the expected bearing is not wrapped (:416-423 are commented out in the reference): particles on the other side
      // of atan2's branch cut for this landmark, or of the heading wrap, sit 2 pi away from the reference -- their
      // blobs are listed as well
      const double db = zb - cb;
      const bool near = fabs(db) <= tb || fabs(db - Consts<double>::two_pi) <= tb || fabs(db + Consts<double>::two_pi) <= tb;
      if (has && near && dr * dr + dg * dg + dc * dc <= tc) {
        // certainly beyond the underflow edge for every particle inside the margins whose own bound is at least (Kb, Ib)?
        const double er = fmax(fabs(dr) - kCandColour, 0.0), eg = fmax(fabs(dg) - kCandColour, 0.0), ec = fmax(fabs(dc) - kCandColour, 0.0);
        const bool far = Ib > 0.0 && Kb + (er * er + eg * eg + ec * ec) * Ib > kPubFarKey + 0.5;
        if (far) {
          const int n = atomicAdd(&s_nf[lane], 1);
          if (n < SLOTS) s_f[lane][n] = (unsigned short)t;
        } else {
          const int n = atomicAdd(&s_n[lane], 1);
          if (n < SLOTS) s_c[lane][n] = (unsigned short)t;
          if (fabs(db) <= 0.5 && dr * dr + dg * dg + dc * dc <= 300.0) atomicAdd(&s_pass[lane], 1);
        }
      }
    }
  }
  __syncthreads();
  if (w == 0 && l < a.Lp + kCandSpare) {  // (the spare records: empty lists)
    const int n = has ? s_n[lane] : 0;
    unsigned short c[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) c[k] = k < n ? s_c[lane][k] : (unsigned short)0xFFFF;
    uint4* out = a.rec + (1 + SLOTS / 8) * (size_t)l;
    out[0] = make_uint4(__float_as_uint(ebf), __float_as_uint(rf), __float_as_uint(gf), __float_as_uint(bf));
#pragma unroll
    for (int j = 0; j < SLOTS / 8; ++j)
      out[1 + j] = make_uint4((unsigned)c[8 * j + 0] | ((unsigned)c[8 * j + 1] << 16), (unsigned)c[8 * j + 2] | ((unsigned)c[8 * j + 3] << 16),
                              (unsigned)c[8 * j + 4] | ((unsigned)c[8 * j + 5] << 16), (unsigned)c[8 * j + 6] | ((unsigned)c[8 * j + 7] << 16));
    if (n > SLOTS) atomicAdd(a.over, 1u);
    if (a.far) {
      const int nf = has ? s_nf[lane] : 0;
      unsigned short cf[SLOTS];
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) cf[k] = k < nf ? s_f[lane][k] : (unsigned short)0xFFFF;
      uint4* fo = a.far + (1 + SLOTS / 8) * (size_t)l;
      fo[0] = make_uint4(__float_as_uint(kbf), __float_as_uint(ibf), (unsigned)min(nf, SLOTS), 0u);
#pragma unroll
      for (int j = 0; j < SLOTS / 8; ++j)
        fo[1 + j] = make_uint4((unsigned)cf[8 * j + 0] | ((unsigned)cf[8 * j + 1] << 16), (unsigned)cf[8 * j + 2] | ((unsigned)cf[8 * j + 3] << 16),
                               (unsigned)cf[8 * j + 4] | ((unsigned)cf[8 * j + 5] << 16), (unsigned)cf[8 * j + 6] | ((unsigned)cf[8 * j + 7] << 16));
      if (nf > SLOTS) atomicAdd(a.over, 1u);  // (a far list that does not hold its blobs: the scan goes the general way)
    }
    if (a.npass) a.npass[l] = (unsigned char)min(has ? s_pass[lane] : 0, 255);
  }
  // the inverse lists: the landmark joins the list of each of its blobs -- wave w appends candidate w (sixteen waves, at most sixteen
  // candidates: every atomic of the workgroup is in flight at once; one lane per landmark did them one round trip after the other)
  if (a.bcnt && has && w < SLOTS && w < min(s_n[lane], SLOTS)) {
    const unsigned t = s_c[lane][w];
    const unsigned m = atomicAdd(&a.bcnt[t], 1u);
    if (m < (unsigned)a.inv_slots)
      a.blist[(size_t)t * a.inv_slots + m] = (unsigned short)l;
    else
      atomicAdd(a.over, 1u);
  }
}

void launch_candidates(hipStream_t s, DeviceState& d, int B, const double* exact_dev, int64_t ref_particle, uint4* rec_dev,
                       unsigned* over_dev, unsigned* bcnt_dev, uint4* brec_dev, unsigned* stray_dev, int slots,
                       const double* pose_sums4_dev, unsigned char* npass_dev, uint4* far_dev, const double* pose_part_dev) {
  if (d.P == 0 || d.lay.Lp == 0) return;
  // (inverse lists as wide as the lists themselves: brec_dev holds B x slots u16.  bcnt_dev is all 0 and brec_dev all 0xFF when
  // this is called: cleared at their allocation, and again by k_cand_entries behind its last read of them -- two memset
  // launches per scan less.  stray_dev: unused since round 3, kept in the signature)
  (void)stray_dev;
  CandArgs a;
  a.inv_slots = slots;
  a.bcnt = brec_dev ? bcnt_dev : nullptr;
  a.blist = reinterpret_cast<unsigned short*>(brec_dev);
  a.ss = slot_source(d);
  a.src = d.src[d.cur];
  a.x = d.x[d.cur];
  a.y = d.y[d.cur];
  a.h = d.h[d.cur];
  a.exact = exact_dev;
  a.rec = rec_dev;
  a.over = over_dev;
  a.ref = ref_particle;
  a.pose4 = pose_sums4_dev;
  a.part = pose_part_dev;
  a.n_part = motion_pose_blocks(d.P);
  a.npass = npass_dev;
  a.far = far_dev;
  a.P = d.P;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  if (slots > kCandSlots)
    hipLaunchKernelGGL(k_candidates<2 * kCandSlots>, dim3((unsigned)((d.lay.Lp + kCandSpare + 63) / 64)), dim3(kCandThreads), 0, s, a);
  else
    hipLaunchKernelGGL(k_candidates<kCandSlots>, dim3((unsigned)((d.lay.Lp + kCandSpare + 63) / 64)), dim3(kCandThreads), 0, s, a);
}

// A candidate list overflowed (a landmark with more than kCandSlots blobs inside the widened gates, or a blob listed by
// more landmarks): the candidate-list kernels stand back, and every particle is handed to the general kernels.
// (the particles [p0, p1) of a ranged launch; the count is added to)
__global__ void __launch_bounds__(256) k_flag_range_if(const unsigned* over, unsigned char* pflag, unsigned* n_flagged, int64_t p0, int64_t p1) {
  if (*over == 0u) return;
  const int64_t p = p0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < p1) pflag[p] = 1;
  if (p == p0) atomicAdd(n_flagged, (unsigned)(p1 - p0));
}
void launch_flag_range_if(hipStream_t s, const unsigned* over_dev, unsigned char* pflag_dev, unsigned* n_flagged_dev, int64_t p0, int64_t p1) {
  if (p1 <= p0) return;
  hipLaunchKernelGGL(k_flag_range_if, dim3((unsigned)((p1 - p0 + 255) / 256)), dim3(256), 0, s, over_dev, pflag_dev, n_flagged_dev, p0, p1);
}


}  // namespace pk
