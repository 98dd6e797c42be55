// The per-scan preparation of the publish / subscribe kernels (pk_k_step_pub.hip): k_cand_entries, ONE workgroup behind k_candidates
// (pk_k_assoc.hip) -- the blobs' inverse candidate lists sorted, the publish table laid out (which (blob, landmark) pair owns which
// 8-byte entry of the LDS table, prkt_core_v2.py:353-381's contest settled by static publish / subscribe), the contested blobs
// listed, the lanes' order of sixteen-landmark groups, the float gate table of k_step_pub_big.  Hand-written gfx950 (CDNA4, wave64).
// (Round 5: its own file; it used to stand at the head of pk_k_step_pub.hip.)
#include "pk_device.hpp"
#include "pk_pub_math.hpp"
#include "pk_pub_layout.hpp"

namespace pk {

// ------------------------------------------------------------------ the publish table's layout, once per scan
// One workgroup.  (a) every blob's inverse list sorted ascending (rank = landmark order: the tie rule of :377),
// (b) exclusive scan of the list lengths of the blobs at least two landmarks list -> offs, (c) every landmark's
// candidates get their entry index.  Control words: skip_pub (a list overflowed, or more entries than the LDS table
// holds) and skip_cand (k_step_regs' candidate-list instance stands back when this route runs).
struct CandEntriesArgs {
  const uint4* cand;      // [Lp][1 + SLOTS / 8]
  unsigned short* erec;   // [Lp][SLOTS]
  unsigned* bcnt;         // [B]
  unsigned short* brec;   // [B][SLOTS], sorted in place
  unsigned* binfo;        // [B] per blob: first entry | contenders << 16 (scratch of this kernel)
  unsigned* glist;        // [B] the blobs at least two landmarks list, compacted: first entry | contenders << 16; [B] = their number
  const unsigned* over;   // candidate-list overflow
  const double* exact;    // [B][6] the scan's records (for gate4), or null
  float4* gate4;          // [B] out: the gate quantities of every blob as float (k_step_pub_big's first look), or null
  const unsigned char* npass;  // [Lp] blobs inside the reference's own gates (k_candidates), or null
  unsigned* skip_pub;
  unsigned* skip_cand;
  unsigned char* flag_all;  // [flag_P] or null: a scan the publish / subscribe kernel stands back from (skip_pub) flags every particle for the
  unsigned* n_flagged;      // fall-back kernels HERE (whole observes; a split step flags its ranges with k_flag_range_if): one launch less
  int64_t flag_P;
  unsigned* skip_duo;  // k_step_pub_duo (two workgroups per CU: half the LDS each) stands back when != 0, or null
  unsigned* skip_big;  // k_step_pub_big stands back when != 0 (nobody's scan, or k_step_pub_duo's), or null
  unsigned* stats;     // [4] out, or null: entries of the publish table, contested blobs, landmarks of the reference particle with two
                       // or more blobs inside their own gates, the longest candidate list
  DuoLimits duo;
  uint4* prim;  // out, or null: the PRIMARY blob of every landmark -- of its candidates the closest in colour -- moved to the front of its
                // list, and that blob's records in LANDMARK order (pk_pub_layout.hpp: prim_*), so that the two-pass kernels read them
                // side by side instead of gathering them blob by blob
  int L, Lp, B, ecap;
  int pruned;  // the lists have had their far look-alikes taken off (k_candidates): only the publish / subscribe kernels, which check
               // every landmark's own bound against the scan's, may use them -- k_step_regs' candidate-list instance always stands back
};

// Exclusive scan of one value per thread over a 1 024-lane workgroup in thread order (Kogge-Stone across the wave, the sixteen wave
// totals through LDS); returns the prefix, `total` = the sum.  sw: 16 words nobody else uses between the call's two barriers.
__device__ __forceinline__ unsigned block_excl_scan_1024(unsigned v, unsigned* sw, unsigned& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  __syncthreads();
  if (lane == 63) sw[wave] = inc;
  __syncthreads();
  unsigned woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned x = sw[i];
    woff += i < wave ? x : 0u;
    tot += x;
  }
  total = tot;
  return woff + inc - v;
}

// SLOTS: entries per candidate list and per inverse list (kCandSlots, or twice that for the scans of several thousand blobs)
// (Round 5: the kernel's chains of dependent round trips taken apart -- the scans over the 1 024 threads' partial sums by one thread,
// a landmark's entries looked up candidate by candidate (two round trips to L2 each), the octets' lists read landmark by landmark:
// 27 us at 500 x 500, where it stood between the motion update and the one-pass kernel of a 270-us step.)
template <int SLOTS>
__global__ void __launch_bounds__(1024) k_cand_entries(CandEntriesArgs a) {
  constexpr bool kRankMajor = SLOTS > kCandSlots;  // (sixteen-entry lists: k_step_pub_big; k_step_pub keeps blob-major, +0.6 % otherwise)
  __shared__ unsigned s_tot[SLOTS + 1], s_cbase[SLOTS + 1], s_rbase[SLOTS];
  __shared__ unsigned s_total, s_sw[16];
  __shared__ unsigned s_multi, s_longest, s_gtotal;
  if (threadIdx.x == 0) {
    s_multi = 0u;
    s_longest = 0u;
  }
  __shared__ unsigned short s_cls[kRankMajor ? SLOTS + 1 : 1][1024];  // per class (number of contenders) and thread: its blobs of the class, then their scan
  __shared__ unsigned char s_len[kPubBigMaxL + kCandSpare + 14], s_np[kPubBigMaxL + kCandSpare + 14];  // per landmark: list length, npass
  const int tid = threadIdx.x;
  const int chunk = (a.B + 1023) / 1024;
  const int t0 = tid * chunk, t1 = min(a.B, t0 + chunk);
  unsigned mine = 0, gmine = 0;
  int ncls[SLOTS + 1];
#pragma unroll
  for (int c = 0; c <= SLOTS; ++c) ncls[c] = 0;
  for (int t = t0; t < t1; ++t) {
    const unsigned n = min(a.bcnt[t], (unsigned)SLOTS);
    unsigned short* row = a.brec + (size_t)t * SLOTS;
    unsigned short v[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) v[i] = row[i];
    // insertion sort (empty = 0xFFFF sorts to the back)
#pragma unroll
    for (int i = 1; i < SLOTS; ++i)
#pragma unroll
      for (int j = i; j > 0; --j)
        if (v[j] < v[j - 1]) {
          const unsigned short x = v[j];
          v[j] = v[j - 1];
          v[j - 1] = x;
        }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) row[i] = v[i];
    mine += n >= 2u ? n : 0u;
    gmine += n >= 2u ? 1u : 0u;
    if constexpr (kRankMajor) {
#pragma unroll
      for (int c = 2; c <= SLOTS; ++c) ncls[c] += n == (unsigned)c ? 1 : 0;
    }
  }
  if constexpr (kRankMajor) {
    // RANK-MAJOR table: the contested blobs ordered by their number of contenders, most first (g = a blob's place in that order),
    // entry of (blob g, rank r) = rbase[r] + g with rbase[r] = how many entries of lower rank there are.  The settling's lane g
    // then reads rank r of ITS blob next to lane g + 1's: consecutive 8-byte words, no bank conflict -- blob-major (a blob's
    // entries side by side) the lanes read at a stride of three to six entries, an 8-way conflict on every read: the settling
    // loop was 8 % of k_step_pub_big's time and LDS-bound -- and a wave reads no further than its longest list.
#pragma unroll
    for (int c = 2; c <= SLOTS; ++c) s_cls[c][tid] = (unsigned short)ncls[c];  // (through LDS: held in registers across the scans they spilled)
#pragma unroll 1
    for (int c = 2; c <= SLOTS; ++c) {  // per class: blobs of that class in the threads before this one
      unsigned tot;
      const unsigned pre = block_excl_scan_1024((unsigned)s_cls[c][tid], s_sw, tot);
      s_cls[c][tid] = (unsigned short)pre;
      if (tid == 0) s_tot[c] = tot;
    }
    __syncthreads();
    if (tid == 0) {
      unsigned base = 0, entries = 0;
      for (int c = SLOTS; c >= 2; --c) {  // most contenders first
        s_cbase[c] = base;
        base += s_tot[c];
        entries += (unsigned)c * s_tot[c];
      }
      s_total = entries;
      s_gtotal = base;
      a.glist[a.B] = base;  // G
      unsigned rb = 0;
      unsigned* rbg = a.glist + a.B + 1 + kPubTailWords;
      for (int r = 0; r < SLOTS; ++r) {  // rank r exists for the blobs with more than r contenders
        s_rbase[r] = rb;
        rbg[r] = rb;
        unsigned more = 0;
        for (int c = max(r + 1, 2); c <= SLOTS; ++c) more += s_tot[c];
        rb += more;
      }
    }
    __syncthreads();
    {
      int k[SLOTS + 1];
#pragma unroll
      for (int c = 0; c <= SLOTS; ++c) k[c] = 0;
      for (int t = t0; t < t1; ++t) {
        const unsigned n = min(a.bcnt[t], (unsigned)SLOTS);
        unsigned g = 0;
#pragma unroll
        for (int c = 2; c <= SLOTS; ++c)
          if (n == (unsigned)c) g = s_cbase[c] + s_cls[c][tid] + (unsigned)(k[c]++);
        a.binfo[t] = (g & 0xFFFFu) | (n << 16);
        if (n >= 2u) a.glist[g] = (g & 0xFFFFu) | (n << 16);
      }
    }
  } else {
  unsigned tot_e, tot_g;
  const unsigned pre_e = block_excl_scan_1024(mine, s_sw, tot_e);
  const unsigned pre_g = block_excl_scan_1024(gmine, s_sw, tot_g);
  if (tid == 0) {
    s_total = tot_e;
    s_gtotal = tot_g;
    a.glist[a.B] = tot_g;
  }
  __syncthreads();
  {
    unsigned run = pre_e, grun = pre_g;
    for (int t = t0; t < t1; ++t) {
      const unsigned n = min(a.bcnt[t], (unsigned)SLOTS);
      const unsigned c = n >= 2u ? n : 0u;
      a.binfo[t] = (run & 0xFFFFu) | (n << 16);
      if (n >= 2u) a.glist[grun++] = (run & 0xFFFFu) | (n << 16);
      run += c;
    }
  }
  }
  const bool fits = *a.over == 0u && s_total <= (unsigned)a.ecap && s_total < 0xFFFFu;
  if (tid == 0) {
    *a.skip_pub = fits ? 0u : 1u;
    *a.skip_cand = (*a.over != 0u || fits || a.pruned != 0) ? 1u : 0u;
  }
  if (!fits && a.flag_all != nullptr) {  // (workgroup-uniform) nobody's scan: every particle to the fall-back kernels
    for (int64_t p = tid; p < a.flag_P; p += 1024) a.flag_all[p] = 1;
    if (tid == 0) atomicAdd(a.n_flagged, (unsigned)a.flag_P);
  }
  unsigned my_multi = 0u, my_longest = 0u;
  __syncthreads();  // brec / binfo written above are read below by other threads of this (the only) workgroup
  __threadfence_block();
  constexpr int RW = 1 + SLOTS / 8;  // uint4 per landmark record
  for (int l = tid; l < a.Lp + kCandSpare; l += 1024) {  // (the spare records get empty entry words)
    // the landmark's list in one go, then eight candidates at a time: their blob words in one batch, their inverse lists in one
    // batch (clamped indices instead of branches: a branch per candidate made every lookup a round trip of its own)
    uint4 lw[SLOTS / 8];
#pragma unroll
    for (int j = 0; j < SLOTS / 8; ++j) lw[j] = a.cand[RW * (size_t)l + 1 + j];
    int len = 0;
    constexpr int BATCH = SLOTS > 8 ? 4 : 8;  // candidates looked up side by side (sixteen-entry lists: four -- their rows are 32 bytes)
    unsigned cw[SLOTS / 2];
#pragma unroll
    for (int j = 0; j < SLOTS / 8; ++j) {
      cw[4 * j + 0] = lw[j].x;
      cw[4 * j + 1] = lw[j].y;
      cw[4 * j + 2] = lw[j].z;
      cw[4 * j + 3] = lw[j].w;
    }
    if (a.prim) {  // (kernel-uniform)
      // The landmark's PRIMARY blob to the front of its list: of its candidates the one closest in colour to the reference particle's
      // landmark -- its own blob wherever one is in sight.  (The order of a list decides nothing: k_candidates fills it in the order
      // its atomics arrive in.)  The gates' first candidate, the verdicts' first slot and the blob the update applies are then nearly
      // always THIS blob, whose records the kernels read in landmark order -- sixteen lanes a cache line -- instead of blob by blob, a
      // line per lane: those gathers were 70 % of the two-pass kernels' accesses to the vector cache (profiles/r06/pmc_ta_*.json).
      const uint4 ref = a.cand[RW * (size_t)l];
      const double rr = (double)__uint_as_float(ref.y), rg = (double)__uint_as_float(ref.z), rb = (double)__uint_as_float(ref.w);
      int best = 0;
      double bd = 1e300;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        const unsigned t = (cw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        if (t != 0xFFFFu && l < a.L) {
          const double* e = a.exact + 6 * (size_t)t;
          const double d1 = e[1] - rr, d2 = e[2] - rg, d3 = e[3] - rb;
          const double d = d1 * d1 + d2 * d2 + d3 * d3;
          if (d < bd) {  // (NaN: never the best)
            bd = d;
            best = k;
          }
        }
      }
      if (best != 0) {  // swap the entries 0 and best
        unsigned tb = 0xFFFFu;
        const unsigned t0 = cw[0] & 0xFFFFu;
#pragma unroll
        for (int k = 1; k < SLOTS; ++k)
          if (k == best) tb = (cw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
#pragma unroll
        for (int k = 1; k < SLOTS; ++k)
          if (k == best) cw[k >> 1] = (cw[k >> 1] & ~(0xFFFFu << (16 * (k & 1)))) | (t0 << (16 * (k & 1)));
        cw[0] = (cw[0] & 0xFFFF0000u) | tb;
        uint4* cand_rw = const_cast<uint4*>(a.cand);
#pragma unroll
        for (int j = 0; j < SLOTS / 8; ++j) cand_rw[RW * (size_t)l + 1 + j] = make_uint4(cw[4 * j + 0], cw[4 * j + 1], cw[4 * j + 2], cw[4 * j + 3]);
      }
      const unsigned t0 = l < a.L ? (cw[0] & 0xFFFFu) : 0xFFFFu;
      const size_t Lpp = (size_t)a.Lp + kCandSpare;
      double z[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (t0 != 0xFFFFu) {
#pragma unroll
        for (int c = 0; c < 6; ++c) z[c] = a.exact[6 * (size_t)t0 + c];
      }
      // (the float record exactly as gate4[t0] below has it)
      const bool ok = fabs(z[0]) <= 8.0 && fabs(z[1]) <= 1000.0 && fabs(z[2]) <= 1000.0 && fabs(z[3]) <= 1000.0;  // NaN: false
      const unsigned nanw = 0x7FC00000u;
      a.prim[l] = ok ? make_uint4(__float_as_uint((float)z[0]), __float_as_uint((float)z[1]), __float_as_uint((float)z[2]), __float_as_uint((float)z[3]))
                     : make_uint4(nanw, nanw, nanw, nanw);
      double2* pz = reinterpret_cast<double2*>(a.prim);
      pz[Lpp + l] = make_double2(z[0], z[1]);
      pz[2 * Lpp + l] = make_double2(z[2], z[3]);
      pz[3 * Lpp + l] = make_double2(z[4], z[5]);
      reinterpret_cast<unsigned*>(a.prim + 4 * Lpp)[l] = t0;
    }
#pragma unroll
    for (int h = 0; h < SLOTS / BATCH; ++h) {
      unsigned t[BATCH], bi[BATCH];
      bool on[BATCH];
      uint4 rows[BATCH][SLOTS / 8];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int kk = BATCH * h + k;
        t[k] = (cw[kk >> 1] >> (16 * (kk & 1))) & 0xFFFFu;
        len += t[k] != 0xFFFFu ? 1 : 0;
        on[k] = l < a.L && t[k] != 0xFFFFu && fits;
        bi[k] = __hip_atomic_load(&a.binfo[on[k] ? t[k] : 0u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k)
#pragma unroll
        for (int j = 0; j < SLOTS / 8; ++j)
          rows[k][j] = reinterpret_cast<const uint4*>(a.brec)[(size_t)(on[k] ? t[k] : 0u) * (SLOTS / 8) + j];
      unsigned ev[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        unsigned rank = 0;
#pragma unroll
        for (int j = 0; j < SLOTS / 8; ++j) {
          const unsigned w4[4] = {rows[k][j].x, rows[k][j].y, rows[k][j].z, rows[k][j].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) rank += ((w4[q] & 0xFFFFu) < (unsigned)l ? 1u : 0u) + ((w4[q] >> 16) < (unsigned)l ? 1u : 0u);
        }
        const unsigned n = bi[k] >> 16;
        ev[k] = (on[k] && n >= 2u) ? (kRankMajor ? s_rbase[rank < (unsigned)SLOTS ? rank : 0u] + (bi[k] & 0xFFFFu) : (bi[k] & 0xFFFFu) + rank) : 0xFFFFu;
      }
#pragma unroll
      for (int k = 0; k < BATCH; k += 2)
        reinterpret_cast<unsigned*>(a.erec)[(size_t)l * (SLOTS / 2) + (BATCH * h + k) / 2] = (ev[k] & 0xFFFFu) | (ev[k + 1] << 16);
    }
    s_len[l] = (unsigned char)len;
    s_np[l] = (a.npass && l < a.L) ? a.npass[l] : (unsigned char)0;
    my_multi += (l < a.L && s_np[l] >= 2) ? 1u : 0u;
    my_longest = max(my_longest, l < a.L ? (unsigned)len : 0u);
  }
  {  // (one LDS atomic per WAVE: a thousand lanes on one address took 2.7 us of this kernel's 13)
    unsigned wm = my_multi, wl = my_longest;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      wm += __shfl_xor(wm, o);
      wl = max(wl, (unsigned)__shfl_xor(wl, o));
    }
    if ((tid & 63) == 0) {
      if (wm) atomicAdd(&s_multi, wm);
      if (wl) atomicMax(&s_longest, wl);
    }
  }
  __syncthreads();
  if (tid == 0) {
    // The two-workgroups-per-CU instance (k_step_pub_duo) has half a CU's LDS: it takes the scan when the list of contested blobs
    // fits its place and the publish table leaves room for the landmarks that will park their slots -- estimated from the reference
    // particle: the landmarks with two or more blobs inside their own gates, a quarter and 64 to spare (a particle that needs more
    // goes to the fall-back kernels, as exact as ever); k_step_pub_big takes every other scan that a publish / subscribe kernel can.
    const unsigned G_ = s_gtotal;  // (= glist[B]: read back from global memory it was a round trip at the kernel's end)
    const unsigned tab = ((G_ + 3u) & ~3u) * 4u + ((s_total + 1u) & ~1u) * 8u + 16u, park = 16u * (s_multi + s_multi / 4u + 64u);
    const bool duo_ok = fits && a.duo.tbytes > 0 && s_total <= (unsigned)a.duo.ecap && G_ <= (unsigned)a.duo.gcap && tab + park <= (unsigned)a.duo.tbytes;
    if (a.skip_duo) *a.skip_duo = duo_ok ? 0u : 1u;
    if (a.skip_big) *a.skip_big = (fits && !duo_ok) ? 0u : 1u;
    if (a.stats) {
      a.stats[0] = s_total;
      a.stats[1] = G_;
      a.stats[2] = s_multi;
      a.stats[3] = s_longest;
    }
  }
  // k_step_pub_big's first look at a candidate: bearing and colour as FLOAT, 16 bytes in one gather instead of 32 in two (the
  // kernel is bound by the texture addresser's gathers, one cache line a cycle: DESIGN.md section 4).  The margins of that
  // look (pub_gatesN<GT>) hold for |bearing| <= 8 and |colour| <= 1000; any other blob -- NaN and infinities included -- gets
  // NaN here and is always looked at exactly.
  if (a.gate4) {
    for (int t = tid; t < a.B; t += 1024) {
      const double z0 = a.exact[6 * (size_t)t], z1 = a.exact[6 * (size_t)t + 1], z2 = a.exact[6 * (size_t)t + 2], z3 = a.exact[6 * (size_t)t + 3];
      const bool ok = fabs(z0) <= 8.0 && fabs(z1) <= 1000.0 && fabs(z2) <= 1000.0 && fabs(z3) <= 1000.0;  // NaN: false
      const float nanf_ = __uint_as_float(0x7FC00000u);
      a.gate4[t] = ok ? make_float4((float)z0, (float)z1, (float)z2, (float)z3) : make_float4(nanf_, nanf_, nanf_, nanf_);
    }
  }
  // The lane order of k_step_pub<2, 512>: 128 places of eight lanes -- sixteen landmarks, an "octet" -- each, place 8 w + k of
  // pair q being lanes 8 k ... 8 k + 7 of wave w.  The octets are ranked by their longest candidate list (then by the sum of
  // their lists) and dealt out eight at a time: the sixteen (wave, pair) groups get octets of like cost -- a wave's gate and
  // verdict loops run as long as its longest list --, the eight costliest groups go to waves 0-3, the others to waves 4-7.
  {
    // (k_step_pub_big, SLOTS = 16: up to six chunks of 64 places -- the same ranking, chunk c takes ranks 512 c ... 512 c + 511 and
    // wave w of it the ranks 64 w ... : all eight waves work through lists of like length at the same time)
    constexpr bool kBig = SLOTS != kCandSlots;
    // (groups of FOUR lanes -- eight landmarks, 64 bytes of a row -- measured +18 %: profiles/r04/ab_perm_groups_of_four_lanes.log)
    constexpr int kLm = 16;  // landmarks per group: eight lanes, 128 bytes of a row
    constexpr int kPlaces = kBig ? kPubBigPlaces : 2 * kPubOctets;
    __shared__ int s_cost[kPlaces];
    unsigned short* perm = reinterpret_cast<unsigned short*>(a.glist + a.B + 1) + (kBig ? 2 * kPubOctets : 0);
    const int n_oct = a.Lp / kLm;
    if (n_oct <= kPlaces) {  // (uniform)
      if (tid < kPlaces) {
        int c = -1;
        if (tid < n_oct) {
          int longest = 0, sum = 0, passes = 0;
#pragma unroll
          for (int i = 0; i < kLm; ++i) {  // (list lengths and gate passes: left in LDS by the loop above)
            const int l = kLm * tid + i;
            const int n = l < a.L ? (int)s_len[l] : 0;
            longest = max(longest, n);
            sum += n;
            passes = max(passes, l < a.L ? (int)s_np[l] : 0);
          }
          // (first by the blobs inside the reference's own gates -- a verdict round each, and a round costs the whole wave its
          // arithmetic --, then by the longest list -- two candidates a gate round)
          c = min(passes, 15) * 4096 + longest * 256 + sum;
        }
        s_cost[tid] = c;
        perm[tid] = 0xFFFFu;  // a place without an octet: its lanes are beyond the map
      }
      __syncthreads();
      if (tid < n_oct) {
        const int c = s_cost[tid];
        int r = 0;
#pragma unroll 8
        for (int o = 0; o < kPlaces; ++o) r += (s_cost[o] > c || (s_cost[o] == c && o < tid)) ? 1 : 0;
        if constexpr (kBig) {
          // place = chunk 64 + wave 8 + k = the rank itself (a last, partial chunk thus goes to the first waves -- the ones
          // with time to spare at barrier A; dealt out evenly or to the last waves: +3-4 %, profiles/r04/ab_big_last_chunk_*.log)
          perm[r] = (unsigned short)tid;
        } else {
          // every wave one costly and one cheap group (the costly ones all to waves 0-3: -0.3 %, to waves 4-7: +2.4 %, a snake over
          // the SIMDs: -0.5 % -- profiles/r04/ab_perm_*.log)
          const int g = r >> 3, k = r & 7;
          const int wave = g & 7, pair = g >> 3;
          perm[kPubOctets * pair + 8 * wave + k] = (unsigned short)tid;
        }
      }
    }
  }
  // the inverse lists have done their work: back to "empty" for the next scan's k_candidates (which appends with atomics)
  __syncthreads();
  for (int t = tid; t < a.B; t += 1024) a.bcnt[t] = 0u;
  {
    unsigned* bw = reinterpret_cast<unsigned*>(a.brec);
    const int nw = a.B * (SLOTS / 2);
    for (int i = tid; i < nw; i += 1024) bw[i] = 0xFFFFFFFFu;
  }
}

void launch_cand_entries(hipStream_t s, const DeviceState& d, int B, const uint4* cand_dev, uint4* erec_dev, unsigned* bcnt_dev,
                         uint4* brec_dev, unsigned* binfo_dev, unsigned* glist_dev, const unsigned* over_dev, unsigned* skip_pub_dev,
                         unsigned* skip_cand_dev, int ecap, int slots, const double* exact_dev, float4* gate4_dev,
                         const unsigned char* npass_dev, bool pruned, unsigned* stats_dev, unsigned* skip_duo_dev, unsigned* skip_big_dev,
                         const DuoLimits& duo, uint4* prim_dev, unsigned char* flag_all_dev, unsigned* n_flagged_dev) {
  CandEntriesArgs a;
  a.flag_all = n_flagged_dev ? flag_all_dev : nullptr;
  a.n_flagged = n_flagged_dev;
  a.flag_P = d.P;
  a.prim = (exact_dev && slots > kCandSlots) ? prim_dev : nullptr;
  a.stats = stats_dev;
  a.skip_duo = skip_duo_dev;
  a.skip_big = skip_big_dev;
  a.duo = duo;
  a.pruned = pruned ? 1 : 0;
  a.npass = npass_dev;
  a.exact = exact_dev;
  a.gate4 = exact_dev ? gate4_dev : nullptr;
  a.cand = cand_dev;
  a.erec = reinterpret_cast<unsigned short*>(erec_dev);
  a.bcnt = bcnt_dev;
  a.brec = reinterpret_cast<unsigned short*>(brec_dev);
  a.binfo = binfo_dev;
  a.glist = glist_dev;
  a.over = over_dev;
  a.skip_pub = skip_pub_dev;
  a.skip_cand = skip_cand_dev;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  a.ecap = ecap;
  if (slots > kCandSlots)
    hipLaunchKernelGGL(k_cand_entries<2 * kCandSlots>, dim3(1), dim3(1024), 0, s, a);
  else
    hipLaunchKernelGGL(k_cand_entries<kCandSlots>, dim3(1), dim3(1024), 0, s, a);
}

}  // namespace pk
