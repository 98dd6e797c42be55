// K2 + K3 with NO synchronisation between the landmarks of a particle: k_step_owner.
//
// Hand-written gfx950 (CDNA4, wave64) kernel of the FastSLAM particle update; see DESIGN.md section 4.
// No MFMA: the algebra is 2x2 / 3x3 and register resident (pk_math.hpp).
//
// What couples the landmarks of a particle in maximum-likelihood association (prkt_core_v2.py:353-381) is a blob
// that passes the gates of several landmarks: it goes to the most probable one.  The other ML kernels settle that
// through LDS -- per-blob counters, a probability queue, bids -- behind workgroup barriers, which ties a whole particle
// to one workgroup, in lockstep.  Here the coupling is made LOCAL with the reference particle's candidate lists
// (k_candidates, pk_kernels.hpp) in both directions:
//     landmark -> the blobs that can pass its gates        (lcand, <= 8)
//     blob     -> the landmarks whose gates it can pass    (bcand, <= 8)
// A lane owns one landmark l.  For every blob t of lcand[l] that passes l's exact gates it looks the rivals up in
// bcand[t], reads THEIR states (pre-update: the source buffer is not written by this kernel), evaluates their gates and
// probabilities with the same device functions, and takes t iff it is the most probable -- the earliest landmark on a
// tie, nobody at probability 0 (:369-381).  Every rival reaches the same verdict from the same bits.  The blob's
// weight factor 0.1 when nobody takes it (:94-95) is added by the blob's "accountant", the lowest landmark of bcand[t].
// No LDS tables, no barriers but the final log-weight sum: 256-lane workgroups, one particle each, landmark chunks,
// as many workgroups per CU as the registers allow -- occupancy hides the memory latency, as in k_observe.
// A particle that leaves the lists' margins anywhere (or wins more than kOwnWins blobs with one landmark) is flagged;
// the general kernels then redo it from the untouched source buffer.
#include "pk_device.hpp"

namespace pk {

constexpr int kOwnThreads = 256;
constexpr int kOwnWins = 4;  // blobs one landmark can take in one scan (updates applied in scan order, :88)

struct OwnerArgs {
  SlotSource ss;
  unsigned char* map_dst;
  size_t count_off;
  int32_t* src;
  const double *x, *y, *h;
  double* logw;
  const double* exact;          // [B][6] cell order: bearing, r, g, b, ux, uy
  const unsigned short* order;  // [B] cell order -> scan order
  const unsigned char* immutable;
  const uint4* lcand;           // [Lp][2]: reference (eb, r, g, b as float), 8 x u16 blobs
  const uint4* bcand;           // [B]: 8 x u16 landmarks (0xFFFF = empty)
  const unsigned* cand_over;    // != 0: a list overflowed, this scan is not ours
  const unsigned* n_stray;      // blobs on nobody's list
  unsigned char* pflag_out;     // [P] 1 = general route
  unsigned* n_flagged;
  int L, Lp, B;
  int reset;
  unsigned long long* gmax_key;
  Noise<double> qt;
};

// What a landmark state says about one blob record (bearing, r, g, b, ux, uy) for the particle at (sx, sy, sh):
// gates (:433, :441) and probability_of_match (:439-455) exactly as the other ML kernels evaluate them.
//   pass: both gates passed;  bits: the probability's bit pattern, 0 when it is not > 0
struct OwnerVerdict {
  bool pass;
  unsigned long long bits;
};
__device__ __forceinline__ OwnerVerdict owner_verdict(const Landmark<double>& lm, double sx, double sy, double sh, double pse,
                                                      const double2& z01, const double2& z23, const double2& dir) {
  OwnerVerdict v{false, 0ull};
  const double eb = pse - sh;  // :408
  if ((fabs(z01.x - eb) > 0.5) || (fabs(color_distance2(lm.mr, lm.mg, lm.mb, z01.y, z23.x, z23.y)) > 300.0)) return v;
  v.pass = true;
  if (fabs(pse - z01.x) > Consts<double>::half_pi) return v;  // :473-475 -> probability 0
  const double det2 = lm.pxx * lm.pyy - lm.pxy * lm.pxy;
  double det3;
  const Sym3<double> adj3 = sym3_adjugate(Sym3<double>{lm.crr, lm.crg, lm.crb, lm.cgg, lm.cgb, lm.cbb}, det3);
  double nx, ny;
  closest_point(lm.mx, lm.my, sx, sy, dir.x, dir.y, nx, ny);
  const double ex = nx - lm.mx, ey = ny - lm.my;
  const double num2 = lm.pyy * ex * ex - 2.0 * lm.pxy * ex * ey + lm.pxx * ey * ey;
  const double num3 = sym3_quad(adj3, z01.y - lm.mr, z23.x - lm.mg, z23.y - lm.mb);
  const double pr = pr_from_parts(det2, det3, num2, num3);
  v.bits = pr > 0.0 ? (unsigned long long)__double_as_longlong(pr) : 0ull;
  return v;
}

// Is the probability of a pair that passed both gates > 0?  (the single-contender case: no value needed; same
// shortcut as fast_prepare, same answer as the evaluation)
__device__ __forceinline__ bool owner_positive(const Landmark<double>& lm, double sx, double sy, double pse, const double2& z01,
                                               const double2& z23, const double2& dir) {
  if (fabs(pse - z01.x) > Consts<double>::half_pi) return false;
  const double det2 = lm.pxx * lm.pyy - lm.pxy * lm.pxy;
  double det3;
  const Sym3<double> adj3 = sym3_adjugate(Sym3<double>{lm.crr, lm.crg, lm.crb, lm.cgg, lm.cgb, lm.cbb}, det3);
  double nx, ny;
  closest_point(lm.mx, lm.my, sx, sy, dir.x, dir.y, nx, ny);
  const double ex = nx - lm.mx, ey = ny - lm.my;
  const double num2 = lm.pyy * ex * ex - 2.0 * lm.pxy * ex * ey + lm.pxx * ey * ey;
  const double num3 = sym3_quad(adj3, z01.y - lm.mr, z23.x - lm.mg, z23.y - lm.mb);
  const bool dets_sane = det2 > 0.0 && det2 < 1e60 && det3 > 0.0 && det3 < 1e60;
  if (dets_sane && num2 >= 0.0 && num3 >= 0.0 && num2 * det3 + num3 * det2 < 800.0 * det2 * det3) return true;
  double d2 = det2, d3 = det3;
  asm volatile("" : "+v"(d2), "+v"(d3));  // opaque: keeps the logs out of the common path
  return pr_from_parts(d2, d3, num2, num3) > 0.0;
}

__global__ void __launch_bounds__(kOwnThreads) k_step_owner(OwnerArgs a) {
  __shared__ double red[kOwnThreads / kWave];
  __shared__ int s_viol;
  if (*a.cand_over != 0u) return;  // a candidate list overflowed: the grid-walk kernels take this scan
  const int64_t p = blockIdx.x;
  const int tid = threadIdx.x;
  const int Lp = a.Lp;
  const unsigned char* sslot = a.ss.at(a.src[p]);
  unsigned char* dslot = a.map_dst + (size_t)p * a.ss.slot_bytes;
  const double* sf = reinterpret_cast<const double*>(sslot);
  double* df = reinterpret_cast<double*>(dslot);
  const int* sc = reinterpret_cast<const int*>(sslot + a.count_off);
  int* dc = reinterpret_cast<int*>(dslot + a.count_off);
  const double sx = a.x[p], sy = a.y[p], sh = a.h[p];
  if (tid == 0) s_viol = 0;
  double acc = 0.0;
  bool viol = false;
  for (int l = tid; l < Lp; l += kOwnThreads) {
    Landmark<double> A = load_landmark(sf, sc, Lp, l);
    if (l < a.L) {
      const uint4 ref = a.lcand[2 * (size_t)l], cands = a.lcand[2 * (size_t)l + 1];
      const double pse = atan2(A.my - sy, A.mx - sx);
      {  // the particle must sit inside the margins the lists were made with (k_candidates), modulo one turn
        const double deb = (pse - sh) - (double)__uint_as_float(ref.x);
        const bool inside = (fabs(deb) <= kCandBearing || fabs(deb - Consts<double>::two_pi) <= kCandBearing ||
                             fabs(deb + Consts<double>::two_pi) <= kCandBearing) &&
                            fabs(A.mr - (double)__uint_as_float(ref.y)) <= kCandColour &&
                            fabs(A.mg - (double)__uint_as_float(ref.z)) <= kCandColour &&
                            fabs(A.mb - (double)__uint_as_float(ref.w)) <= kCandColour;
        viol |= !inside;
      }
      unsigned key[kOwnWins];  // scan index << 16 | blob of the blobs this landmark takes
#pragma unroll
      for (int k = 0; k < kOwnWins; ++k) key[k] = 0xFFFFFFFFu;
      int nwin = 0;
      unsigned c0 = cands.x, c1 = cands.y, c2 = cands.z, c3 = cands.w;  // filled from the front
#pragma unroll 1
      for (int k = 0; k < kCandSlots; ++k) {
        const unsigned t = c0 & 0xFFFFu;
        if (t == 0xFFFFu) break;
        c0 = (c0 >> 16) | (c1 << 16);
        c1 = (c1 >> 16) | (c2 << 16);
        c2 = (c2 >> 16) | (c3 << 16);
        c3 = (c3 >> 16) | 0xFFFF0000u;
        const double* rec = a.exact + 6 * (size_t)t;
        const double2 z01 = *reinterpret_cast<const double2*>(rec);
        const double2 z23 = *reinterpret_cast<const double2*>(rec + 2);
        const double2 dir = *reinterpret_cast<const double2*>(rec + 4);
        const bool mine = !(fabs(z01.x - (pse - sh)) > 0.5) && !(fabs(color_distance2(A.mr, A.mg, A.mb, z01.y, z23.x, z23.y)) > 300.0);
        // the blob's other candidate landmarks, and its accountant (the lowest of them all)
        const uint4 bl = a.bcand[t];
        unsigned r0 = bl.x, r1 = bl.y, r2 = bl.z, r3 = bl.w;
        unsigned lowest = 0xFFFFu, others = 0u;
        {
          const unsigned w[4] = {r0, r1, r2, r3};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned lj = (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            lowest = min(lowest, lj);
            others += (lj != 0xFFFFu && lj != (unsigned)l) ? 1u : 0u;
          }
        }
        const bool accountant = lowest == (unsigned)l;
        if (!mine && !accountant) continue;  // not my blob, not my books
        bool take = false, anybody = false;
        if (others == 0u) {  // the common case: nobody else can pass this blob
          take = mine && owner_positive(A, sx, sy, pse, z01, z23, dir);
          anybody = take;
        } else {
          unsigned long long my_bits = 0ull;
          if (mine) my_bits = owner_verdict(A, sx, sy, sh, pse, z01, z23, dir).bits;
          take = my_bits != 0ull;
          anybody = take;
          if (take || accountant) {
#pragma unroll 1
            for (int j = 0; j < 8; ++j) {
              const unsigned lj = r0 & 0xFFFFu;
              r0 = (r0 >> 16) | (r1 << 16);
              r1 = (r1 >> 16) | (r2 << 16);
              r2 = (r2 >> 16) | (r3 << 16);
              r3 = (r3 >> 16) | 0xFFFF0000u;
              if (lj == 0xFFFFu) break;  // filled from the front
              if (lj == (unsigned)l) continue;
              const Landmark<double> R = load_landmark_nocount(sf, Lp, (int)lj);
              const double pr = atan2(R.my - sy, R.mx - sx);
              const OwnerVerdict v = owner_verdict(R, sx, sy, sh, pr, z01, z23, dir);
              if (v.bits != 0ull) {
                anybody = true;
                // the larger probability wins, the earlier landmark on a tie (:377)
                if (v.bits > my_bits || (v.bits == my_bits && lj < (unsigned)l)) take = false;
              }
              if (!take && (anybody || !accountant)) break;  // lost (and, for the accountant, somebody matches): nothing left to learn
            }
          }
        }
        if (take) {
          const unsigned kv = ((unsigned)a.order[t] << 16) | t;
#pragma unroll
          for (int j = 0; j < kOwnWins; ++j)
            if (nwin == j) key[j] = kv;
          ++nwin;
        }
        if (accountant && !anybody) acc += Consts<double>::log_no_match;  // unseen feature: weight *= 0.1 (:94-95)
      }
      viol |= nwin > kOwnWins;
      // the blobs taken, in scan order (:88)
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
      const bool imm = a.immutable[l] != 0;
      bool fresh = true;
#pragma unroll 1
      for (int it = 0; it < kOwnWins; ++it) {
        const unsigned kk = key[0];
        if (kk == 0xFFFFFFFFu) break;
        key[0] = key[1];
        key[1] = key[2];
        key[2] = key[3];
        key[3] = 0xFFFFFFFFu;
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
  if (viol) s_viol = 1;  // (benign race: everybody writes the same value)
  const double tot = block_sum<kOwnThreads / kWave>(acc, red);  // its two barriers also publish s_viol
  if (tid == 0) {
    if (s_viol) {  // the general kernels redo this particle from the source buffer; its weight and slot stay as they were
      a.pflag_out[p] = 1;
      atomicAdd(a.n_flagged, 1u);
    } else {
      a.pflag_out[p] = 0;
      const double v = (a.reset ? 0.0 : a.logw[p]) + tot + (double)(*a.n_stray) * Consts<double>::log_no_match;
      a.logw[p] = v;
      if (a.gmax_key) atomicMax(a.gmax_key + (p & (kGmaxKeys - 1)), double_to_key(v));
      a.src[p] = (int32_t)p;
    }
  }
}

// A candidate list overflowed (a landmark with more than kCandSlots blobs inside the widened gates, or a blob listed by
// more landmarks): k_step_owner stood back, and every particle is handed to the general kernels.
__global__ void __launch_bounds__(256) k_flag_all_if(const unsigned* over, unsigned char* pflag, unsigned* n_flagged, int64_t P) {
  if (*over == 0u) return;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < P) pflag[p] = 1;
  if (p == 0) *n_flagged = (unsigned)P;
}
// the same for the particles [p0, p1) of a ranged launch (the count is added to)
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

void launch_step_owner(hipStream_t s, DeviceState& d, int B, const double* exact_dev, const unsigned short* order_dev,
                       const FastHandoff& fh, const NoiseD& qt, const ObserveExtras& ex, const CandTable& cand) {
  if (d.P == 0) return;
  OwnerArgs a;
  a.ss = slot_source(d);
  a.map_dst = d.map[d.mcur ^ 1];
  a.count_off = d.lay.count_off;
  a.src = d.src[d.cur];
  a.x = d.x[d.cur];
  a.y = d.y[d.cur];
  a.h = d.h[d.cur];
  a.logw = d.logw[d.cur];
  a.exact = exact_dev;
  a.order = order_dev;
  a.immutable = d.immutable;
  a.lcand = cand.rec;
  a.bcand = cand.brec;
  a.cand_over = cand.over;
  a.n_stray = cand.n_stray;
  a.pflag_out = fh.pflag;
  a.n_flagged = fh.n_flagged;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  a.reset = ex.reset ? 1 : 0;
  a.gmax_key = ex.gmax_key;
  a.qt = make_noise(qt.q00, qt.rr, qt.rg, qt.rb, qt.gg, qt.gb, qt.bb);
  hipLaunchKernelGGL(k_step_owner, dim3((unsigned)d.P), dim3(kOwnThreads), 0, s, a);
  hipLaunchKernelGGL(k_flag_all_if, dim3((unsigned)((d.P + 255) / 256)), dim3(256), 0, s, cand.over, fh.pflag, fh.n_flagged, d.P);
}

}  // namespace pk
