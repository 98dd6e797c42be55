// k_step_pub_duo: the two-pass publish / subscribe kernel (k_step_pub_big, maps of 2 049 .. 6 144 landmarks) at <= 128 VGPRs, so that
// TWO 512-lane workgroups share a CU -- four waves per SIMD -- and one workgroup's row latency is the other's float64 issue.
// Included at the end of pk_k_step_pub.hip (inside namespace pk): it is made of that file's device functions -- pub_gatesN, pub_keysN,
// pub_far_recheck, pub_keep_positive, pub_settle_blobs, pub_apply_loop -- with the same inputs, so its maps are k_step_pub_big's bit
// for bit (the log-weight's partial sums add up in another order: 1e-13).  Hand-written gfx950 (CDNA4, wave64), DESIGN.md section 4.
//
// Why: k_step_pub_big sits at 235-250 VGPRs -- two waves per SIMD -- and in the window the driver times it is bound by neither pipe
// (issue share 0.39, 0.43 of the HBM roofline): between a pair's row request and its gates, and in every gather round of the
// verdicts, a wave waits for memory with one other wave on its SIMD to cover for it (the traffic fell by a
// third there and the time did not move).  Forcing that kernel to 128 VGPRs spills 450 of them (25.9 ms where it takes 6.3).
// What makes 128 possible BY CONSTRUCTION:
//   one landmark per lane and turn (29 VGPRs of state instead of 58, one chain of gate / verdict arithmetic instead of two);
//   ONE carried word per landmark from pass 1 to pass 2 instead of seven (k_step_pub_big: four slot words, their state, the expected
//     bearing): nearly every landmark ends pass 1 with at most one blob of probability > 0 -- its own -- and that blob | entry word
//     is all pass 2 needs; a landmark with two or more parks its four slot words in an LDS overflow area (places dealt out by an LDS
//     counter; a particle that needs more than there are goes to the fall-back kernels) and carries the place.  Behind the settling
//     such a landmark nearly always TAKES one blob and becomes an ordinary one-word landmark again; the few that take two or more
//     (a landmark sighted twice, :88) keep their slots in a small second area for pass 2;
//   the expected bearing (prkt_core_v2.py:871) is worked out again in pass 2 from the rows it reads anyway -- the same expression on
//     the same bits.
// Each of the two workgroups has half the CU's LDS: the publish table, the list of contested blobs and the overflow area must fit
// 78 KB.  Whether they do is the device's knowledge (k_cand_entries: *skip_duo, DuoLimits); the scans that do not fit -- the first
// steps of a fresh map, whose look-alikes are all real contenders -- go to k_step_pub_big as before (both kernels are launched, one
// returns at once).
//
// Lanes and landmarks: turn q, lanes 16 k .. 16 k + 15 of the workgroup take the sixteen landmarks of octet s_bperm[32 q + k] --
// k_cand_entries' ranking of the octets by the length of their lists (the table k_step_pub_big reads with eight lanes per octet), so
// a wave's 64 landmarks have lists of like length and a row access of sixteen lanes is one 128-byte line.
constexpr int kDuoHeld = 64;    // landmarks of ONE particle that TAKE two or more blobs (their slots are kept for pass 2)
constexpr size_t kDuoMaxDynLds = 78 * 1024;   // two 512-lane workgroups per CU (160 KB less the kernels' static __shared__)
constexpr size_t kTrioMaxDynLds = 50 * 1024;  // three 256-lane workgroups per CU (the pair instance, NL = 2)
constexpr int kDuoTurn = 512;   // landmarks a workgroup works on per turn: 512 lanes x 1, or 256 lanes x 2
constexpr int kDuoMaxTurns = 10;  // turns of 512 landmarks: maps up to 5 120 (twelve carried words and the update no longer fit 128 VGPRs: two spilled)

// dynamic LDS: any 2 x (Bp + 16) | held [2][kDuoHeld] uint4 | glist 4 G | publish table 8 (E + 2) | overflow area, 16 B a place
// The last three share what is left (DuoLimits::tbytes): the list of contested blobs and the table take what THIS scan needs -- the
// kernel reads the numbers from k_cand_entries' figures --, the overflow area the rest; k_cand_entries gives the scan to this kernel
// when that rest holds the reference particle's landmarks with two or more blobs inside their gates, with a quarter to spare.
__host__ __device__ inline size_t pub_duo_fixed_lds_bytes(int B) {
  const size_t Bp = ((size_t)B + 15) & ~(size_t)15;
  return 2 * (Bp + 16) + 2 * (size_t)kDuoHeld * 16;
}
// nl: landmarks per lane and turn -- 1: 512-lane workgroups, two per CU, <= 128 VGPRs; 2: 256-lane workgroups, three per CU, <= 168
void step_pub_duo_limits(int B, int Lp, int nl, DuoLimits* out) {
  *out = DuoLimits();
  if (Lp <= kRegsMaxL || Lp > kDuoMaxTurns * kDuoTurn || B <= 0 || (nl != 1 && nl != 2)) return;
  const size_t fixed = pub_duo_fixed_lds_bytes(B), total = nl == 1 ? kDuoMaxDynLds : kTrioMaxDynLds;
  if (fixed + 4096 > total) return;
  out->tbytes = (int)((total - fixed) & ~(size_t)15);
  out->ecap = 65534;
  out->gcap = 65535;
  out->nl = nl;
}
size_t step_pub_duo_lds_bytes(int B, const DuoLimits& lim) { return pub_duo_fixed_lds_bytes(B) + (size_t)lim.tbytes; }

constexpr unsigned kDuoNone = 0xFFFFFFFFu;   // carried word: the landmark takes no blob
constexpr unsigned kDuoMulti = 0xFFFEu;      // ... entry field of a landmark whose slots are parked: pass 1 -> take: place | kDuoMulti << 16
                                             // (the overflow area); take -> pass 2: place | take bits << 8 | kDuoMulti << 16 (the held area)

// NL: landmarks per lane and turn.  1: 512 lanes, <= 128 VGPRs, two workgroups per CU, 8-byte row accesses; 2: an adjacent PAIR per lane
// (16-byte row accesses, as k_step_pub_big's) worked on one landmark after the other, 256 lanes, <= 168 VGPRs, THREE workgroups per CU.
template <int NT, int NL>
__global__ void __launch_bounds__(NL == 1 ? 512 : 256, NL == 1 ? 4 : 3) k_step_pub_duo(PubArgs a_unused) {
  constexpr int TH = NL == 1 ? 512 : 256;  // lanes of the workgroup
  constexpr int NW = NT * NL;              // carried words of a lane
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double red[2][TH / kWave];
  __shared__ int wg_flag[2];
  __shared__ unsigned s_novf;      // places of the overflow area dealt out to the particle in pass 1 (read again before barrier C only)
  __shared__ unsigned s_nheld[2];  // places of the held area dealt out to the particle of either parity (read in its pass 2)
  __shared__ unsigned short s_bperm[kPubBigPlaces];
  __shared__ unsigned s_rb[kPubBigSlots];
  __shared__ unsigned s_ubits[kPubUnmWords];  // growing maps: the particle's unmatched blobs, scan order (pub_note_unmatched)
  // the lane's (first) landmark of turn q_: sixteen lanes an octet of sixteen landmarks (NL = 1), or eight lanes with a pair each
#define PK_DUO_L(q_, t_) (NL == 1 ? (int)(16u * (unsigned)s_bperm[32 * (q_) + ((t_) >> 4)]) + ((t_)&15) \
                                  : (int)(16u * (unsigned)s_bperm[32 * (q_) + ((t_) >> 3)]) + 2 * ((t_)&7))
  constexpr int kPubWaves = TH / kWave;
  PubArgsPtr rp = (PubArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const int tid0 = threadIdx.x;
  int B, Lp, L;
  unsigned ecap, tbytes;
  int park_limit;
  bool long_lists;  // some landmark lists more than eight candidates: the second list word is read at all
  {
    PubArgsPtr R = pub_args_now(rp);
    if (*R->skip != 0u) return;  // workgroup-uniform: k_step_pub_big (or the fall-back kernels) take this scan
    B = R->B;
    Lp = R->Lp;
    L = R->L;
    ecap = (R->stats[0] + 1u) & ~1u;  // THIS scan's entries (k_cand_entries), rounded up to even: the overflow area starts on 16 bytes
    tbytes = (unsigned)R->tbytes;
    park_limit = R->ecap;  // (tests: the overflow area treated as this small; < 0: what LDS holds)
    long_lists = R->stats[3] > (unsigned)kCandSlots;
  }
  const unsigned Bp = ((unsigned)B + 15u) & ~15u;
  unsigned G;
  {
    PubArgsPtr R = pub_args_now(rp);
    G = R->glist[B];
  }
  // LDS offsets (bytes): any[2][Bp + 16] | held[2][kDuoHeld] | glist (G words) | pub (ecap + 2 entries) | overflow area (what is left)
  const unsigned o_any = 0u, o_held = 2u * (Bp + 16u), o_glist = o_held + 2u * (unsigned)kDuoHeld * 16u, o_pub = o_glist + ((G + 3u) & ~3u) * 4u;
  const unsigned o_ovf = o_pub + (ecap + 2u) * 8u;
  unsigned n_places = o_glist + tbytes > o_ovf ? (o_glist + tbytes - o_ovf) / 16u : 0u;  // (>= what k_cand_entries asked for when it gave the scan to this kernel)
  if (n_places > 0xFFFFu) n_places = 0xFFFFu;
  if (park_limit >= 0 && n_places > (unsigned)park_limit) n_places = (unsigned)park_limit;
  const unsigned dump = ecap, anydump = Bp;
  {
    const int tid = tid0;
    PubArgsPtr R = pub_args_now(rp);
    unsigned* glist = reinterpret_cast<unsigned*>(smem + o_glist);
    const unsigned* gb = R->glist;
    for (unsigned i = (unsigned)tid; i < G; i += TH) glist[i] = gb[i];
    for (unsigned i = (unsigned)tid; i < 2u * (Bp + 16u) / 4u; i += TH) reinterpret_cast<unsigned*>(smem + o_any)[i] = 0u;
    for (int i = tid; i < kPubBigPlaces; i += TH) s_bperm[i] = reinterpret_cast<const unsigned short*>(gb + B + 1)[2 * kPubOctets + i];
    if (tid < kPubBigSlots) s_rb[tid] = gb[B + 1 + kPubTailWords + tid];
    for (int i = tid; i < kPubUnmWords; i += TH) s_ubits[i] = 0u;
    if (tid == 0) {
      wg_flag[0] = 0;
      wg_flag[1] = 0;
      s_novf = 0u;
      s_nheld[0] = 0u;
      s_nheld[1] = 0u;
    }
  }
  __syncthreads();

  // the rows of landmark lb_ of the slot at source src_: the five mean rows first (the gates need nothing else)
#define PK_DUO_LOAD(field, F)                                                                         \
  if constexpr (NL == 1) {                                                                            \
    S[0].field = sf_[(size_t)F * Lp + lb_];                                                           \
  } else {                                                                                            \
    const Double2 v_ = *reinterpret_cast<const Double2*>(sf_ + (size_t)F * Lp + lb_);                 \
    S[0].field = v_.x;                                                                                \
    S[NL - 1].field = v_.y;                                                                           \
  }
#define PK_DUO_ROWS(lbv_, src_)                                                                       \
  {                                                                                                   \
    PubArgsPtr R2 = pub_args_now(rp);                                                                 \
    const SlotSource ss_ = pub_slot_source(R2);                                                       \
    const unsigned char* sslot_ = ss_.at(src_);                                                       \
    const double* sf_ = reinterpret_cast<const double*>(sslot_);                                      \
    const int* sc_ = reinterpret_cast<const int*>(sslot_ + R2->count_off);                            \
    const int lb_ = (lbv_);                                                                           \
    PK_DUO_LOAD(mx, F_MX)                                                                             \
    PK_DUO_LOAD(my, F_MY)                                                                             \
    PK_DUO_LOAD(mr, F_MR)                                                                             \
    PK_DUO_LOAD(mg, F_MG)                                                                             \
    PK_DUO_LOAD(mb, F_MB)                                                                             \
    asm volatile("" ::: "memory");                                                                    \
    PK_DUO_LOAD(pxx, F_PXX)                                                                           \
    PK_DUO_LOAD(pxy, F_PXY)                                                                           \
    PK_DUO_LOAD(pyy, F_PYY)                                                                           \
    PK_DUO_LOAD(crr, F_CRR)                                                                           \
    PK_DUO_LOAD(crg, F_CRG)                                                                           \
    PK_DUO_LOAD(crb, F_CRB)                                                                           \
    PK_DUO_LOAD(cgg, F_CGG)                                                                           \
    PK_DUO_LOAD(cgb, F_CGB)                                                                           \
    PK_DUO_LOAD(cbb, F_CBB)                                                                           \
    if constexpr (NL == 1) {                                                                          \
      S[0].count = sc_[lb_];                                                                          \
    } else {                                                                                          \
      const Int2 c_ = *reinterpret_cast<const Int2*>(sc_ + lb_);                                      \
      S[0].count = c_.x;                                                                              \
      S[NL - 1].count = c_.y;                                                                         \
    }                                                                                                 \
    asm volatile("" ::: "memory");                                                                    \
  }
  int64_t prev = -1;
  int cur = 0;
#ifdef PK_STAMPS
  // (diagnostic build, k_step_pub_big's slots: 0 pass 1 waits for records and rows | 1 gates | 2 verdicts | 3 next rows asked for |
  //  4 barrier A | 5 settling, B | 6 take, C | 7 pass 2 waits for rows | 8 updates | 9 particles | 10 stores, next rows | 11 particle)
  unsigned long long pst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PK_DUO_WAIT_ALL asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
#define PK_DUO_WAIT_ALL
#endif
  int32_t nsrc;  // the next particle's source slot, asked for a whole particle ahead (as in k_step_pub)
  {
    PubArgsPtr R = pub_args_now(rp);
    const int64_t p0 = PK_BIG_FIRST(R), pl = PK_BIG_LIMIT(R);
    nsrc = regs_source_pub(R->src, p0 < pl ? p0 : pl - 1);
  }
  // The rows of a turn are asked for as soon as the turn before it is through, into the registers it has just let go; pass 2 walks
  // back to front from the turn pass 1 ended on (its rows are still there), and asks for the NEXT particle's first turn at its end.
  Landmark<double> S[NL];
  {
    const int lb0 = min(PK_DUO_L(0, tid0), Lp - NL);
    PK_DUO_ROWS(lb0, nsrc)
  }
  for (int64_t p = PK_BIG_FIRST(pub_args_now(rp));; p += PK_BIG_STRIDE(), cur ^= 1) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int32_t csrc = nsrc;
    double* pub = reinterpret_cast<double*>(smem + o_pub);
    uint4* ovf = reinterpret_cast<uint4*>(smem + o_ovf);
    uint4* held = reinterpret_cast<uint4*>(smem + o_held) + (unsigned)cur * (unsigned)kDuoHeld;
    const unsigned* glist = reinterpret_cast<const unsigned*>(smem + o_glist);
    unsigned char* anyc = smem + o_any + (unsigned)cur * (Bp + 16u);
    unsigned W[NW];  // the carried words: W[0 .. NL) the turn worked on last
#pragma unroll
    for (int i = 0; i < NW; ++i) W[i] = kDuoNone;
    bool done;
    {
      PubArgsPtr R = pub_args_now(rp);
      done = p >= PK_BIG_LIMIT(R);
    }
    PK_STAMP(b0)
    // ---- pass 1: gates and verdicts, one landmark per lane and turn (one copy of the code, the carried words rotating)
    if (!done) {
#pragma unroll 1
      for (int q = 0; q < NT; ++q) {
        unsigned wq[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) wq[j] = kDuoNone;
        if (kDuoTurn * q < Lp) {  // workgroup-uniform
          const int l0 = PK_DUO_L(q, tid);
#pragma unroll
          for (int j = 0; j < NL; ++j) {  // (NL = 2: the pair's landmarks one after the other -- one chain of gate / verdict arithmetic at a time)
          PubArgsPtr R = pub_args_now(rp);
          PK_STAMP(c0)
          const double sx = pose_scalar(R->x, p), sy = pose_scalar(R->y, p), sh = pose_scalar(R->h, p);
          const int lc = min(l0, Lp) + j;  // (lanes beyond the map: the spare records, empty lists -- see k_step_pub)
          const uint4* cr = R->cand + 3 * (size_t)lc;
          const uint4* er = R->erec + 2 * (size_t)lc;
          PubGateIn gi[1];
          gi[0].ref = cr[0];
          gi[0].cw[0] = cr[1];
          gi[0].ew[0] = er[0];
          gi[0].cw[1] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
          gi[0].ew[1] = gi[0].cw[1];
          if (long_lists) {  // kernel-uniform: once the lists are pruned, hardly ever
            gi[0].cw[1] = cr[2];
            gi[0].ew[1] = er[1];
          }
          const uint4* frow = R->far;
          const bool far_hdr_on = frow != nullptr;
          uint4 fh0 = make_uint4(0u, 0u, 0u, 0u);
          if (far_hdr_on) fh0 = frow[3 * (size_t)lc];
          asm volatile("" ::: "memory");
          if (q == 0 && j == 0) {  // the next particle's source slot (as in k_step_pub)
            PubArgsPtr R4 = pub_args_now(rp);
            const int64_t pn = p + PK_BIG_STRIDE(), pl = PK_BIG_LIMIT(R4);
            nsrc = regs_source_pub(R4->src, pn < pl ? pn : pl - 1);
            asm volatile("" : "+s"(nsrc));
          }
          PK_DUO_WAIT_ALL
          PK_STAMP(c1)
          PK_PSTAMP(0, c0, c1)
          gi[0].mx = S[j].mx;
          gi[0].my = S[j].my;
          gi[0].mr = S[j].mr;
          gi[0].mg = S[j].mg;
          gi[0].mb = S[j].mb;
          gi[0].has = l0 + j < L;
          pub_far_bound(S[j], gi[0].fk, gi[0].fi);
          bool viol[1] = {false};
          if (far_hdr_on)  // (uniform) do the scan's pruned lists hold for this landmark?
            viol[0] = gi[0].has && !(gi[0].fi >= (double)__uint_as_float(fh0.y) && gi[0].fk >= (double)__uint_as_float(fh0.x));
          double pp[1] = {0.0};
          const Landmark<double>* const l1[1] = {&S[j]};
          PubSlotsT<kPubBigGateSlots> qq[1];
          // the landmark's primary blob -- the front of its list -- comes from the table in landmark order: its float record in the gates'
          // first round, its exact records where the verdicts want them
          pub_gatesN<1, 2, kPubBigGateSlots, false, true, true>(qq, pp, gi, R->exact, pub, dump, &wg_flag[cur], sx, sy, sh, R->gate4,
                                                                reinterpret_cast<const float4*>(R->prim), lc);
          PK_STAMP(c2)
          PK_PSTAMP(1, c1, c2)
          {
            const double kb_[1] = {gi[0].fk}, it_[1] = {gi[0].fi};
            {
              PubArgsPtr R8 = pub_args_now(rp);
              const uint4* fr8 = R8->far;
              pub_far_recheck<1, 2>(viol, l1, pp, kb_, it_, [&]() { return fr8 ? fr8 + 3 * (size_t)lc : (const uint4*)nullptr; }, R8->exact, sh, &wg_flag[cur]);
            }
            {
              PubArgsPtr R9 = pub_args_now(rp);
              PubPrim prim;
              prim.tab = R9->prim;
              prim.Lpp = (size_t)R9->Lp + kCandSpare;
              prim.lc = lc;
              pub_keysN<1, kPubBigGateSlots, true, false, 1, PubNoChk, true>(qq, l1, pp, R9->exact, pub, dump, anyc, anydump, &wg_flag[cur], sx, sy, kb_, it_,
                                                                             PubNoChk(), &prim);
            }
          }
          const PubSlots qa = pub_keep_positive(qq[0], &wg_flag[cur]);
          // the carried word: the one blob of probability > 0 (pub_keep_positive puts it in front), or the place of the parked slots
          const int npos = __popc(qa.st & 0x1111u);
          wq[j] = npos == 0 ? kDuoNone : qa.s[0];
          if (__ballot(npos >= 2) != 0ull) {  // wave-uniform
            if (npos >= 2) {
              const unsigned place = atomicAdd(&s_novf, 1u);
              if (place < n_places) {
                ovf[place] = make_uint4(qa.s[0], qa.s[1], qa.s[2], qa.s[3]);
                wq[j] = place | (kDuoMulti << 16);
              } else {  // more such landmarks than the area holds: the fall-back kernels take the particle
                wg_flag[cur] = 1;
                wq[j] = kDuoNone;
              }
            }
          }
          PK_STAMP(c3)
          PK_PSTAMP(2, c2, c3)
          }
          // the next turn of this pass; the LAST turn's rows stay where they are: pass 2 starts with them
          if (q + 1 < NT && kDuoTurn * (q + 1) < Lp) {  // workgroup-uniform
#if defined(PK_STAMPS)
            PK_STAMP(c3b)
#endif
            const int ln = min(PK_DUO_L(q + 1, tid), Lp - NL);
            PK_DUO_ROWS(ln, csrc)
#if defined(PK_STAMPS)
            PK_STAMP(c4)
            PK_PSTAMP(3, c3b, c4)
#endif
          }
        }
#pragma unroll
        for (int i = NW - 1; i >= NL; --i) W[i] = W[i - NL];  // (a turn's words enter at the FRONT: pass 2 walks back to front)
#pragma unroll
        for (int j = 0; j < NL; ++j) W[j] = wq[j];
      }
    }
    PK_STAMP(b1)
    lds_barrier();  // A: every verdict of this particle is in the table
    PK_STAMP(b2)
    PK_PSTAMP(4, b1, b2)
    if (prev >= 0 && tid == 0) {  // the previous particle's log-weight (its partial sums were written before A)
      PubArgsPtr R = pub_args_now(rp);
      double tot = red[cur ^ 1][0];
#pragma unroll
      for (int i = 1; i < kPubWaves; ++i) tot += red[cur ^ 1][i];
      double* logw = R->logw;
      const double w = (R->reset ? 0.0 : logw[prev]) + tot;
      logw[prev] = w;
      unsigned long long* gk = R->gmax_key;
      if (gk) atomicMax(gk + (prev & (kGmaxKeys - 1)), double_to_key(w));
      R->src[prev] = (int32_t)prev;
    }
    if (done) break;
    prev = -1;
    double acc;
    {
      int nun = 0;
      for (unsigned w = (unsigned)tid; w < Bp / 4u; w += TH) {
        const unsigned v = reinterpret_cast<const unsigned*>(anyc)[w];
#pragma unroll
        for (int b = 0; b < 4; ++b) nun += ((int)(4 * w + b) < B && ((v >> (8 * b)) & 0xFFu) == 0u) ? 1 : 0;
      }
      acc = (double)nun * Consts<double>::log_no_match;  // unseen features: weight *= 0.1 each (:94-95)
      unsigned* anyn = reinterpret_cast<unsigned*>(smem + o_any + (unsigned)(cur ^ 1) * (Bp + 16u));
      for (unsigned w = (unsigned)tid; w < Bp / 4u; w += TH) anyn[w] = 0u;
      if (tid == 0) {  // (the other parity's flag and held places: its particle's pass 2 ended before barrier A, the next one's
                       // pass 1 starts behind barrier C)
        wg_flag[cur ^ 1] = 0;
        s_nheld[cur ^ 1] = 0u;
      }
    }
    {
      PubArgsPtr Ru = pub_args_now(rp);
      if (Ru->unm != nullptr) pub_note_unmatched<TH>(tid, anyc, Ru->order, B, Bp, s_ubits);  // kernel-uniform: growing maps only
    }
    pub_settle_blobs<TH, kPubBigSlots>(tid, glist, G, pub, dump, &wg_flag[cur], s_rb);
    lds_barrier();  // B: every winner is marked, every flag is set
    PK_STAMP(b3)
    PK_PSTAMP(5, b2, b3)
    {
      PubArgsPtr Ru = pub_args_now(rp);
      if (Ru->unm != nullptr) pub_store_unmatched<TH>(tid, s_ubits, Ru->unm + (size_t)p * Ru->unm_words, Ru->unm_words);
    }
    // which of its blobs every landmark takes (pub_take): a blob nobody else lists, or the entry that carries the winner's marker
    {
      double m[NW];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const unsigned e = W[i] >> 16;
        m[i] = pub[e < kDuoMulti ? e : dump];
      }
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const unsigned e = W[i] >> 16;
        W[i] = (e < kDuoMulti && m[i] != pub_marker()) ? kDuoNone : W[i];  // somebody else's
      }
      bool anymulti = false;
#pragma unroll
      for (int i = 0; i < NW; ++i) anymulti |= (W[i] >> 16) == kDuoMulti;
      if (__ballot(anymulti) != 0ull) {  // wave-uniform: some landmark of the wave has its slots parked
#pragma unroll 1
        for (int r = 0; r < NW; ++r) {  // (one copy of the code; the words rotate once round)
          const unsigned w0 = W[0];
          const bool multi = (w0 >> 16) == kDuoMulti;
          unsigned wn = w0;
          if (__ballot(multi) != 0ull) {
            const uint4 s4 = ovf[multi ? (w0 & 0xFFFFu) : 0u];
            const unsigned sw[4] = {s4.x, s4.y, s4.z, s4.w};
            double mm[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const unsigned e = sw[s] >> 16;
              mm[s] = pub[(multi && e < kDuoMulti) ? e : dump];
            }
            unsigned tk = 0u, one = kDuoNone;
#pragma unroll
            for (int s = 3; s >= 0; --s) {
              const unsigned e = sw[s] >> 16;
              const bool valid = (sw[s] & 0xFFFFu) != 0xFFFFu;  // (every parked slot that holds a blob has probability > 0)
              const bool take = valid && (e == 0xFFFFu || mm[s] == pub_marker());
              tk |= take ? (1u << s) : 0u;
              one = take ? sw[s] : one;
            }
            // nothing taken: an ordinary landmark without a blob; one: an ordinary one-word landmark; more: the slots are held for pass 2
            const bool many = multi && (tk & (tk - 1u)) != 0u;
            wn = !multi ? w0 : (one | 0xFFFF0000u);  // (kDuoNone stays kDuoNone; the entry has done its work)
            if (__ballot(many) != 0ull) {  // wave-uniform, rare
              if (many) {
                const unsigned h = atomicAdd(&s_nheld[cur], 1u);
                if (h < (unsigned)kDuoHeld) {
                  held[h] = s4;
                  wn = h | (tk << 8) | (kDuoMulti << 16);
                } else {  // more such landmarks than the area holds: the fall-back kernels take the particle
                  wg_flag[cur] = 1;
                  wn = kDuoNone;
                }
              }
            }
          }
#pragma unroll
          for (int i = 0; i + 1 < NW; ++i) W[i] = W[i + 1];
          W[NW - 1] = wn;
        }
      }
    }
    if (tid == 0) s_novf = 0u;  // (the next particle's pass 1 starts behind barrier C; this one's places were all read above)
    lds_barrier();  // C: every marker has been read -- the table is the next particle's
    PK_STAMP(b4)
    PK_PSTAMP(6, b3, b4)
    if (wg_flag[cur]) {  // workgroup-uniform: nothing has been written; the fall-back kernels take the particle
      if (tid == 0) {
        PubArgsPtr R = pub_args_now(rp);
        R->pflag_out[p] = 1;
        atomicAdd(R->n_flagged, 1u);
      }
      {  // (the registers hold this particle's last turn: the next particle's first instead)
        const int lb0 = min(PK_DUO_L(0, tid), Lp - NL);
        PK_DUO_ROWS(lb0, nsrc)
      }
      continue;
    }
    if (tid == 0) pub_args_now(rp)->pflag_out[p] = 0;
    // ---- pass 2, back to front: rows in again (the last turn's are still here), updates in scan order, rows out
    double nprod = 1.0;  // (the product of the norms of this lane's updates: pub_fold_norms)
#pragma unroll 1
    for (int qr = 0; qr < NT; ++qr) {
      const int q = NT - 1 - qr;
      if (kDuoTurn * q < Lp) {  // workgroup-uniform
        const int l0 = PK_DUO_L(q, tid);
        PK_STAMP(d0)
        PK_DUO_WAIT_ALL
        PK_STAMP(d1)
        PK_PSTAMP(7, d0, d1)
#pragma unroll
        for (int j = 0; j < NL; ++j) {
        PubArgsPtr R = pub_args_now(rp);
        const Noise<double> qt = pub_noise(R);
        const double sx = pose_scalar(R->x, p), sy = pose_scalar(R->y, p);
        const unsigned char* immutable = R->immutable;
        // the landmark's primary blob again (the blob it applies, nearly always): which one it is, and where its exact record stands
        const int lc2 = min(l0, Lp) + j;
        const uint4* ptab = R->prim;
        const size_t Lpp = (size_t)Lp + kCandSpare;
        const unsigned t0 = reinterpret_cast<const unsigned*>(ptab + 4 * Lpp)[lc2];
        // the landmark's slots again: its one blob, or what it parked
        const unsigned w0 = W[j];
        const bool multi = (w0 >> 16) == kDuoMulti;
        PubSlots qs = kPubNoSlots;
        qs.s[0] = multi ? 0xFFFFFFFFu : w0;
        qs.st = (!multi && (w0 & 0xFFFFu) != 0xFFFFu) ? 5u : 0u;  // probability > 0, taken
        if (__ballot(multi) != 0ull) {  // wave-uniform
          const uint4 s4 = held[multi ? (w0 & 0xFFu) : 0u];
          if (multi) {
            qs.s[0] = s4.x;
            qs.s[1] = s4.y;
            qs.s[2] = s4.z;
            qs.s[3] = s4.w;
            const unsigned tk = (w0 >> 8) & 0xFu;
            qs.st = ((tk & 1u) ? 0x0005u : 0u) | ((tk & 2u) ? 0x0050u : 0u) | ((tk & 4u) ? 0x0500u : 0u) | ((tk & 8u) ? 0x5000u : 0u);
          }
        }
        // the expected bearing (:871), as pass 1 worked it out: the same expression on the same bits
        const double pse = pk_atan2(S[j].my - sy, S[j].mx - sx);
        acc += pub_apply_loop<true>(qs, R->exact, R->order, qt, S[j], immutable[min(l0 + j, L - 1)] != 0, sx, sy, pse,
                                    reinterpret_cast<const char*>(ptab + Lpp), Lpp * 16, lc2, t0, PK_PROD_PTR(nprod));
        }
        pub_fold_norms(acc, nprod, false);
        PK_STAMP(d2)
        PK_PSTAMP(8, d1, d2)
        if (l0 < Lp) {
          PubArgsPtr R3 = pub_args_now(rp);
          unsigned char* dslot = R3->map_dst + (size_t)p * R3->ss.slot_bytes;
          double* df = reinterpret_cast<double*>(dslot);
          int* dc = reinterpret_cast<int*>(dslot + R3->count_off);
#define PK_DUO_STORE(field, F)                                                                  \
  if constexpr (NL == 1) {                                                                      \
    __builtin_nontemporal_store(S[0].field, df + (size_t)F * Lp + l0);                          \
  } else {                                                                                      \
    const Double2 v_ = {S[0].field, S[NL - 1].field};                                           \
    __builtin_nontemporal_store(v_, reinterpret_cast<Double2*>(df + (size_t)F * Lp + l0));     \
  }
          PK_DUO_STORE(mx, F_MX)
          PK_DUO_STORE(my, F_MY)
          PK_DUO_STORE(mr, F_MR)
          PK_DUO_STORE(mg, F_MG)
          PK_DUO_STORE(mb, F_MB)
          PK_DUO_STORE(pxx, F_PXX)
          PK_DUO_STORE(pxy, F_PXY)
          PK_DUO_STORE(pyy, F_PYY)
          PK_DUO_STORE(crr, F_CRR)
          PK_DUO_STORE(crg, F_CRG)
          PK_DUO_STORE(crb, F_CRB)
          PK_DUO_STORE(cgg, F_CGG)
          PK_DUO_STORE(cgb, F_CGB)
          PK_DUO_STORE(cbb, F_CBB)
#undef PK_DUO_STORE
          if constexpr (NL == 1) {
            __builtin_nontemporal_store(S[0].count, dc + l0);
          } else {
            const Int2 c2_ = {S[0].count, S[NL - 1].count};
            __builtin_nontemporal_store(c2_, reinterpret_cast<Int2*>(dc + l0));
          }
        }
        {  // the turn before it, or the next particle's first one
          const bool more = q > 0;  // workgroup-uniform
          const int ln = min(PK_DUO_L(more ? q - 1 : 0, tid), Lp - NL);
          const int32_t sn = more ? csrc : nsrc;
          PK_DUO_ROWS(ln, sn)
        }
        PK_STAMP(d3)
        PK_PSTAMP(10, d2, d3)
      }
#pragma unroll
      for (int i = 0; i + NL < NW; ++i) W[i] = W[i + NL];  // the words of the turn before it to the front
    }
    {
      pub_fold_norms(acc, nprod, true);
      const double ws = wave_sum(acc);  // the sum over the workgroup is finished behind the next barrier A
      if ((tid & (kWave - 1)) == 0) red[cur][tid / kWave] = ws;
      prev = p;
    }
#ifdef PK_STAMPS
    {
      PK_STAMP(b5)
      PK_PSTAMP(11, b0, b5)
      pst[9] += 1ull;
    }
#endif
  }
#ifdef PK_STAMPS
  if ((tid0 & 63) == 0)
    for (int k = 0; k < 12; ++k) atomicAdd(&pk_pstamp_wave[(tid0 >> 6) + (NL == 2 ? 4 * (int)(blockIdx.x & 1u) : 0)][k], pst[k]);
#endif
#undef PK_DUO_WAIT_ALL
#undef PK_DUO_ROWS
#undef PK_DUO_LOAD
#undef PK_DUO_L
}

void launch_step_pub_duo(hipStream_t s, DeviceState& d, int B, const double* exact_dev, const unsigned short* order_dev,
                         const FastHandoff& fh, const NoiseD& qt, const ObserveExtras& ex, const CandTable& cand, const uint4* erec_dev,
                         const unsigned* glist_dev, const unsigned* skip_duo_dev, const unsigned* stats_dev, const DuoLimits& lim,
                         const float4* gate4_dev, const uint4* prim_dev, int64_t p0, int64_t p1, int reserve_cus) {
  if (p1 < 0) p1 = d.P;
  if (d.P == 0 || p1 <= p0 || lim.tbytes <= 0 || !prim_dev || !stats_dev) return;
  static bool attr_set[kMaxDevices] = {false};
  if (first_time_on_this_device(attr_set)) {
    for (const void* fn : {reinterpret_cast<const void*>(k_step_pub_duo<6, 1>), reinterpret_cast<const void*>(k_step_pub_duo<10, 1>)})
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDuoMaxDynLds) != hipSuccess) (void)hipGetLastError();
    for (const void* fn : {reinterpret_cast<const void*>(k_step_pub_duo<6, 2>), reinterpret_cast<const void*>(k_step_pub_duo<10, 2>)})
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrioMaxDynLds) != hipSuccess) (void)hipGetLastError();
  }
  PubArgs a;
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
  a.cand = cand.rec;
  a.erec = erec_dev;
  a.glist = glist_dev;
  a.skip = skip_duo_dev;
  a.gate4 = gate4_dev;
  a.far = cand.far;
  a.prim = prim_dev;
  a.unm = ex.unm;
  a.unm_words = ex.unm_words;
  a.pflag_out = fh.pflag;
  a.n_flagged = fh.n_flagged;
  a.P = p1;
  a.p_begin = p0;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  a.ecap = lim.park_limit;  // (the kernel reads THIS scan's entry count from a.stats; this field: the overflow area's debug limit)
  a.stats = stats_dev;
  a.tbytes = lim.tbytes;
  a.reset = ex.reset ? 1 : 0;
  a.gmax_key = ex.gmax_key;
  a.qt = make_noise(qt.q00, qt.rr, qt.rg, qt.rb, qt.gg, qt.gb, qt.bb);
  const int n_cu = device_cu_count();
  // persistent grid, TWO 512-lane (or THREE 256-lane) workgroups per CU; reserve_cus as in launch_step_regs
  const int per_cu = lim.nl == 2 ? 3 : 2, th = lim.nl == 2 ? 256 : 512;
  int64_t grid_n = per_cu * (int64_t)(n_cu - (reserve_cus > 0 && reserve_cus < n_cu ? reserve_cus : 0));
  if (grid_n > p1 - p0) grid_n = p1 - p0;
  const size_t lds = step_pub_duo_lds_bytes(B, lim);
  const int nt = (d.lay.Lp + kDuoTurn - 1) / kDuoTurn;
  if (lim.nl == 2) {
    if (nt <= 6)
      hipLaunchKernelGGL((k_step_pub_duo<6, 2>), dim3((unsigned)grid_n), dim3(th), lds, s, a);
    else
      hipLaunchKernelGGL((k_step_pub_duo<10, 2>), dim3((unsigned)grid_n), dim3(th), lds, s, a);
  } else {
    if (nt <= 6)
      hipLaunchKernelGGL((k_step_pub_duo<6, 1>), dim3((unsigned)grid_n), dim3(th), lds, s, a);
    else
      hipLaunchKernelGGL((k_step_pub_duo<10, 1>), dim3((unsigned)grid_n), dim3(th), lds, s, a);
  }
}
