// Host-visible launchers of the gfx950 kernels (defined in pk_k_motion / pk_k_assoc / pk_k_observe / pk_k_resample .hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pk_layout.hpp"

namespace pk {

struct NoiseD {
  double q00, rr, rg, rb, gg, gb, bb;
};

// Device-side view of one filter shard.
struct DeviceState {
  int64_t P;
  MapLayout lay;
  // pose SoA, double-buffered for the resample gather: [cur] is live
  double* x[2];
  double* y[2];
  double* h[2];
  double* logw[2];
  int32_t* src[2];  // map slot holding the particle's landmarks (in map[mcur])
  int cur;          // live pose buffer
  unsigned char* map[2];
  int mcur;  // live map buffer
  unsigned char* immutable;  // [L]
  // Particles adopted from another shard (multi-GPU resample) are read in place from the
  // receive buffer until the next observe rewrites them: src[p] < 0 means record -src[p]-1 of
  // `alt`, whose map slot starts alt_off bytes into a record of alt_stride bytes.
  const unsigned char* alt = nullptr;
  size_t alt_stride = 0, alt_off = 0;
  int64_t global_offset = 0;  // index of local particle 0 in the whole filter (Philox counters)
  // balanced placement of the sharded filter: the LOGICAL index of the particle in every physical slot
  // (its index in one filter holding all particles -- what keys the Philox counters and orders the weight scan), double-buffered
  // with the poses; NULL until the balanced exchange is first used (the logical index is global_offset + slot then)
  int64_t* logical[2] = {nullptr, nullptr};
};

// section 8(f4) on the device (pk_k_grow.hip): the per-particle bookkeeping of the new-landmark machinery, double-buffered like the
// poses (the resample gathers it by ancestors)
struct GrowState {
  double* hyp[2] = {nullptr, nullptr};       // [P][R][8] stored readings: id, x, y, heading, bearing, r, g, b
  int32_t* cnt[2] = {nullptr, nullptr};      // [P][4]: readings stored, spare slots in use, next_id, readings dropped (ring full)
  int32_t* slot_id[2] = {nullptr, nullptr};  // [P][S] feature id of every spare slot in use
  int cur = 0;
  int L0 = 0, S = 0, R = 0;                  // preset landmarks, spare slots, ring capacity
  double pair_threshold = 0.0;
};
// the bookkeeping of one particle behind its record in the sharded exchange: counters (16 B) | slot ids | readings
__host__ __device__ inline size_t grow_tail_readings_off(int S) { return 16 + (((size_t)S * 4 + 15) & ~(size_t)15); }
inline size_t grow_tail_bytes(const GrowState& g) { return g.R > 0 ? grow_tail_readings_off(g.S) + (size_t)g.R * 64 : 0; }

// Where the landmark slot of a particle lives: its own map buffer or the adoption buffer.
struct SlotSource {
  const unsigned char* map;
  size_t slot_bytes;
  const unsigned char* alt;
  size_t alt_stride, alt_off;
  __host__ __device__ const unsigned char* at(int32_t src) const {
    return src >= 0 ? map + (size_t)src * slot_bytes : alt + (size_t)(-(src + 1)) * alt_stride + alt_off;
  }
};
inline SlotSource slot_source(const DeviceState& d) {
  return SlotSource{d.map[d.mcur], d.lay.slot_bytes, d.alt, d.alt_stride, d.alt_off};
}

constexpr int kScanBlock = 1024;  // particles per weight-scan block (256 threads x 4)

// K1
// up_dst_dev != NULL: the same launch also pulls the per-scan block (up_bytes from the pinned,
// device-mapped staging slot) into HBM with a few extra workgroups -- one launch less per step
void launch_motion(hipStream_t s, DeviceState& d, double v, double w, double dt, const double* z_dev,
                   uint64_t seed, uint64_t draw, int64_t global_offset, void* up_dst_dev = nullptr,
                   const void* up_src_host_mapped = nullptr, size_t up_bytes = 0, double* pose_part_dev = nullptr);
// pose_part_dev != NULL: the launch also leaves, per block of 256 particles, the sums of x, y, sin h, cos h of the moved particles
// ([4] doubles per block, motion_pose_blocks(P) blocks): k_candidates reduces them to the particles' mean pose
inline int64_t motion_pose_blocks(int64_t P) { return (P + 255) / 256; }
void launch_motion_range(hipStream_t s, DeviceState& d, double v, double w, double dt, uint64_t seed, uint64_t draw,
                         int64_t p0, int64_t p1);
void launch_reset_weights(hipStream_t s, DeviceState& d);
// K2: maximum-likelihood association -> ids[P*B]
// (a) reference kernel: every (landmark, blob) pair is gate-tested
void launch_assoc_brute(hipStream_t s, DeviceState& d, const double* blobs_dev, const double* blobdir_dev,
                        int B, int32_t* ids_dev);
// (b) production kernel: blobs bucketed on the host into a 3-D colour grid whose cell edge
// exceeds the colour gate radius sqrt(300), so a landmark only meets the blobs of its 27
// neighbouring cells.  Tables (device): start u16[ncell+1] (16-byte padded) | rec32
// float4[B] = (r, g, b, bearing) in cell order | idx9 u16[n9] | order u16[B]; exact records
// double[B][6] = (bearing, r, g, b, ux, uy) in cell order.  With n9 > 0, idx9 lists for every
// (r, g) column the blobs within one cell in r and g, ordered by b cell, and `start` indexes
// it: a landmark's whole 27-cell neighbourhood is then one contiguous range.
struct BlobGrid {
  double lo[3];
  double inv_h;
  int G[3];
  int ncell;
  float thr32;   // conservative fp32 pre-filter threshold for the colour gate (300)
  float thrb32;  // conservative fp32 pre-filter threshold for the bearing gate (0.5)
};
constexpr double kGridCell = 17.5;  // > sqrt(300) = 17.3205 (prkt_core_v2.py:441)
constexpr int kGridMax = 16;
__host__ __device__ inline size_t grid_cs_bytes(int ncell) { return ((size_t)(ncell + 1) * 2 + 15) & ~(size_t)15; }
size_t blob_grid_table_bytes(int ncell, int B, int n9);
size_t assoc_grid_lds_bytes(int ncell, int B, int n9);
constexpr size_t kMaxDynLds = 156 * 1024;
// Per-device "already done" flag for one-time function attributes (hipFuncSetAttribute is per
// device; one process may hold filters on several GPUs).  Returns true the first time it is asked
// for the CURRENT device.
constexpr int kMaxDevices = 64;
inline bool first_time_on_this_device(bool (&done)[kMaxDevices]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return true;
  if (done[dev]) return false;
  done[dev] = true;
  return true;
}  // 160 KiB per workgroup minus the kernels' static __shared__
// Compute units of the CURRENT device (what sizes the persistent grids), cached per device: a process may drive filters on
// devices with different CU counts.
inline int device_cu_count() {
  static int n_cu[kMaxDevices] = {0};
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices;
  if (known && n_cu[dev] > 0) return n_cu[dev];
  int n = 256;
  hipDeviceProp_t prop;
  if (known && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
  (void)hipGetLastError();
  if (known) n_cu[dev] = n;
  return n;
}
// Hand-off from the association kernel to k_observe_fast (all three NULL = not used).
struct FastHandoff {
  uint4* lmpass = nullptr;          // [rows][Lp] (slots = 4) or [rows][Lp][2] (slots = 8): see k_assoc_grid; rows = P, or with
                                    // `retry` the capped number of second-chance rows (row_of / row_next / row_cap below)
  int slots = 4;                    // gate-passing blobs a landmark can hand over: kFastSlots or kSweepSlots
  bool flags_only = false;          // pflag / n_flagged come from k_step_fused: run the general instance on the flagged only
  unsigned char* bcount = nullptr;  // [P][B]
  unsigned char* pflag = nullptr;   // [P] 0: done by the fast kernel, 1: for the general kernels, 2: handed off on the second chance
  unsigned* n_flagged = nullptr;    // number of flagged particles of this scan
  // second chance only: the hand-off rows are dealt out in order of arrival, row_cap of them (a fraction of P: what a scan
  // flags is a few percent at worst) -- row_of[p] = the row of particle p, *row_next = rows dealt so far (zeroed by the scan
  // upload); a particle that finds none left keeps flag 1 and goes to the general kernels
  int32_t* row_of = nullptr;
  unsigned* row_next = nullptr;
  int64_t row_cap = 0;
  // Second chance for the particles a one-pass kernel flagged (a landmark passing more than its four register slots): the
  // hand-off instance with eight slots works on the particles whose flag is 1 only and leaves 2 where it succeeded (for
  // k_observe_sweep, ObserveExtras::sweep_only_value) and 1 where not even eight slots do (general kernels).
  bool retry = false;
};
constexpr int kGmaxKeys = 256;  // the running max of the log-weights is kept in this many keys (one scan-block thread each)
// Optional behaviour of one observe launch.
struct ObserveExtras {
  const unsigned char* only_flagged = nullptr;  // general kernel: only the particles flagged by the fast path (flag == 1)
  const unsigned* n_flagged = nullptr;
  int sweep_only_value = 0;                     // k_observe_sweep: 0 = the particles whose flag is 0, else only those with this flag
  bool flip = true;                             // swap the map buffers after this launch
  bool reset = false;                           // weights restart from 1 (fused pk_reset_weights)
  unsigned long long* gmax_key = nullptr;       // keep the running max of the new log-weights here
  bool single_sightings = false;                // known ids: no landmark is matched by more than one blob
  unsigned* unm = nullptr;                      // growing maps on the publish / subscribe routes: out [P][unm_words] -- every particle's unmatched
  int unm_words = 0;                            //   blobs as a bit row in scan order (pk_k_step_pub.hip: pub_note_unmatched)
};
constexpr int kFastSlots = 4;   // gate-passing blobs a landmark can hand over to k_observe_fast; more -> general path
constexpr int kSweepSlots = 8;  // ... to k_observe_sweep (large maps: a landmark's colour neighbourhood is busier)
constexpr int kFastMaxL = 512;  // k_observe_fast / k_step_fused keep a particle's whole map in registers (one landmark per lane, 512 lanes)
void launch_assoc_grid(hipStream_t s, DeviceState& d, int B, const BlobGrid& grid, int n9,
                       const unsigned char* tables_dev, const double* exact_dev, int32_t* ids_dev,
                       bool finalize, const FastHandoff& fh);
// ML observe for L <= kFastMaxL straight from the hand-off: contested blobs are settled here,
// with the landmark state in registers.  Particles flagged in fh.pflag are skipped (the general
// k_observe, launched with only_flagged, takes them).
size_t observe_fast_lds_bytes(int B);  // dynamic LDS of k_observe_fast: must fit kMaxDynLds
void launch_observe_fast(hipStream_t s, DeviceState& d, int B, const double* exact_dev,
                         const unsigned short* order_dev, const FastHandoff& fh, const NoiseD& qt,
                         const ObserveExtras& ex = ObserveExtras());
// K3: EKF update + log-weight.  known: first/next chains shared by all particles (device
// arrays, built on the host); otherwise built per particle in LDS from ids_dev.
// ML ids are TENTATIVE: k_observe keeps a match only if its probability is > 0 (needs blobdir).
void launch_observe(hipStream_t s, DeviceState& d, const double* blobs_dev, const double* blobdir_dev, int B,
                    const int32_t* first_dev, const int32_t* next_dev, int n_unmatched,
                    int32_t* ids_dev, const NoiseD& qt, const ObserveExtras& ex = ObserveExtras());
// ML observe for any L from the same hand-off, two sweeps over the particle's landmarks in
// chunks (persistent workgroups; see k_observe_sweep).  plan.grid == 0: scan too large for its LDS tables.
struct SweepPlan {
  int grid = 0;               // persistent workgroups
  size_t lds = 0;             // dynamic LDS per workgroup
  int qcap = 0;               // entries of the LDS probability queue
  size_t results_per_wg = 0;  // uint4 entries of global scratch per workgroup
};
SweepPlan observe_sweep_plan(const DeviceState& d, int B);
void launch_observe_sweep(hipStream_t s, DeviceState& d, int B, const double* exact_dev,
                          const unsigned short* order_dev, const FastHandoff& fh, const NoiseD& qt,
                          const ObserveExtras& ex, const SweepPlan& plan, uint4* results_dev);
// K2 + K3 fused for L <= kFastMaxL and scan tables small enough for two workgroups per CU: gates,
// settling and EKF update of a particle in one workgroup, no hand-off through HBM.  Writes fh.pflag /
// fh.n_flagged (particles left to the general kernels).
size_t fused_lds_bytes(int ncell, int B, int n9, bool exact_lds = false);
constexpr size_t kFusedMaxLds = 78 * 1024;
void launch_step_fused(hipStream_t s, DeviceState& d, int B, const BlobGrid& grid, int n9,
                       const unsigned char* tables_dev, const double* exact_dev, const unsigned short* order_dev,
                       const FastHandoff& fh, const NoiseD& qt, const ObserveExtras& ex, const unsigned* pub_skipped_dev = nullptr);
// Candidate lists from a reference particle (k_candidates).  All particles of a filter see the same scan and hold
// nearly the same map (same initial map, same blobs matched), so the blobs that can pass a landmark's two gates
// (prkt_core_v2.py:433, :441) are nearly the same for every particle.  Once per scan, for every landmark of ONE
// reference particle, the blobs within the gates WIDENED by (kCandBearing, kCandColour) are listed; a particle whose
// own expected bearing and colour of that landmark lie within those margins of the reference's can only match blobs of
// that list (triangle inequality), and tests them with the exact float64 gates -- no walk through the colour grid.
// The expected bearing is unwrapped in the reference (:408-423), so "within the margin" is taken modulo one turn of
// 2 pi either way, and the list holds the blobs of all three centres.
// A particle that breaks a margin anywhere is flagged and goes the general way; a landmark with more than kCandSlots
// candidates switches the whole scan back to the grid walk (cand_over).  Record per landmark, 32 B:
//   float ebref, r, g, b  |  kCandSlots x u16 blob (cell order) or 0xFFFF
constexpr int kCandSlots = 8;
// Every candidate / entry table carries kCandSpare records with EMPTY lists behind its Lp landmarks: the lanes of k_step_pub /
// k_step_pub_big that stand beyond the map read those (index Lp), so they never see a real landmark's candidates.
constexpr int kCandSpare = 2;
constexpr double kCandBearing = 0.2;   // rad: |expected bearing - reference's| of every particle, else flagged (the particles' HEADING spread goes here: sigma 0.02 rad after 25 steps of the bench)
constexpr double kCandColour = 1.5;    // per channel: |colour mean - reference's|
// The inverse lists, blob -> the (<= kCandSlots) landmarks that list it, 16 B per blob (8 x u16, 0xFFFF = empty, filled from
// the front in arbitrary order), are what k_cand_entries lays the publish table of k_step_pub out from.
struct CandTable {
  const uint4* rec = nullptr;         // [Lp][2] landmark records
  const uint4* brec = nullptr;        // [B] blob records (the inverse lists), or NULL
  const unsigned* over = nullptr;     // != 0: some list has more entries than slots -> grid walk for this scan
  const unsigned* skip_cand = nullptr; // k_step_regs' candidate-list instance stands back when != 0 (NULL: when *over != 0)
  const unsigned* n_stray = nullptr;  // blobs on no landmark's list
  const uint4* far = nullptr;         // [Lp][1 + slots / 8] (Kb, Ib, n, 0) | far list: look-alikes k_candidates took off the lists (pk_pub_math.hpp), or NULL
  int slots = kCandSlots;             // entries per list: kCandSlots ([Lp][2] records) or twice that ([Lp][3]; no inverse lists)
};
// launch_assoc_grid whose hand-off instance takes its gates from the candidate lists (while no list overflowed)
void launch_assoc_grid(hipStream_t s, DeviceState& d, int B, const BlobGrid& grid, int n9,
                       const unsigned char* tables_dev, const double* exact_dev, int32_t* ids_dev,
                       bool finalize, const FastHandoff& fh, const CandTable& cand);
// bcnt_dev / brec_dev: NULL, or u32[B] / u16[B][slots] (cleared / filled with 0xFF by this call); stray_dev: their count
void launch_candidates(hipStream_t s, DeviceState& d, int B, const double* exact_dev, int64_t ref_particle, uint4* rec_dev,
                       unsigned* over_dev, unsigned* bcnt_dev = nullptr, uint4* brec_dev = nullptr, unsigned* stray_dev = nullptr,
                       int slots = kCandSlots, const double* pose_sums4_dev = nullptr, unsigned char* npass_dev = nullptr,
                       uint4* far_dev = nullptr, const double* pose_part_dev = nullptr);  // pose_part_dev: the motion launch's per-block sums instead of pose_sums4_dev
// K2 + K3 in one pass (pk_k_observe_ml.hip, pk_k_step_pub.hip).  (k_step_owner, a barrier-free variant in which every
// landmark settled its blobs against the rivals named by the two-way lists, was measured at 56 ms against 13 and removed
// in round 3: DESIGN.md section 4.)
// 512 < L <= kRegsMaxL: persistent 1024-lane workgroups, a particle's whole map in registers (two
// landmarks per lane), state read once; warm: how much of the next particle's slot is pulled into L2 ahead of time.
constexpr int kRegsMaxL = 2048;
size_t regs_lds_bytes(int ncell, int B, int n9);
size_t regs_cand_lds_bytes(int Lp, int B);
// cand.rec != NULL: the candidate-list instance (returns at once when *cand.skip_cand != 0), and a scan in which a list
// overflowed hands all its particles to the fall-back kernels (the grid-walk instance -- cand.rec == NULL, option
// "cand_lists" = 0 -- still spills eight registers and is no default route's kernel any more)
void launch_step_regs(hipStream_t s, DeviceState& d, int B, const BlobGrid& grid, int n9, const unsigned char* tables_dev,
                      const double* exact_dev, const unsigned short* order_dev, const FastHandoff& fh, const NoiseD& qt,
                      const ObserveExtras& ex, int warm, const CandTable& cand = CandTable(), int64_t p0 = 0, int64_t p1 = -1,
                      int reserve_cus = 0);
// K2 + K3 in one pass with STATIC publish / subscribe settling (pk_k_step_pub.hip): 512 < L <= kRegsMaxL, candidate lists
// both ways (cand.rec, and the inverse lists that launch_cand_entries turns into the publish table's layout: erec, binfo).
// 512-lane persistent workgroups, four landmarks per lane, two barriers per particle.  Returns at once when *skip != 0.
int step_pub_entry_capacity(int B);  // publish-table entries that fit LDS beside the scan's tables (0: the scan does not fit)
int step_pub_entry_capacity_small(int B);  // ... with three 256-lane workgroups per CU (the L <= 512 instance)
size_t step_pub_lds_bytes(int B, int ecap, bool small = false);
// What k_step_pub_duo (the two-workgroups-per-CU instance of the two-pass kernel) has room for: entries of the publish table,
// contested blobs, landmarks with several gate-passing blobs (ecap = 0: the instance is not in use)
struct DuoLimits {
  int tbytes = 0;  // bytes of LDS the publish table (8 per entry) and the overflow area (16 per landmark with several blobs) share; 0: off
  int ecap = 0;    // entries at most (the option "pub_entry_limit")
  int gcap = 0;    // contested blobs at most
  int nl = 1;      // landmarks per lane and turn: 1 (512 lanes, two workgroups per CU) or 2 (256 lanes with a pair each, three per CU)
  int park_limit = -1;  // >= 0 (tests, option "pub_duo_park_limit"): the kernel treats its overflow area as this many places
};
void launch_cand_entries(hipStream_t s, const DeviceState& d, int B, const uint4* cand_dev, uint4* erec_dev, unsigned* bcnt_dev,
                         uint4* brec_dev, unsigned* binfo_dev, unsigned* glist_dev, const unsigned* over_dev, unsigned* skip_pub_dev,
                         unsigned* skip_cand_dev, int ecap, int slots = kCandSlots, const double* exact_dev = nullptr,
                         float4* gate4_dev = nullptr, const unsigned char* npass_dev = nullptr, bool pruned = false,
                         unsigned* stats_dev = nullptr, unsigned* skip_duo_dev = nullptr, unsigned* skip_big_dev = nullptr,
                         const DuoLimits& duo = DuoLimits(), uint4* prim_dev = nullptr, unsigned char* flag_all_dev = nullptr,
                         unsigned* n_flagged_dev = nullptr);  // flag_all_dev: [P] a scan nobody takes flags every particle in this very launch
// The primary-blob table of the two-pass kernels (k_cand_entries writes it, once per scan): for every landmark l (and the kCandSpare
// spare records) the records of the FIRST blob of its candidate list, in landmark order -- four planes of Lp + kCandSpare uint4
// (bearing, r, g, b as float | exact bearing, r | exact g, b | ray direction ux, uy), then the blob's index per landmark (u32;
// 0xFFFF: the list is empty)
inline size_t prim_table_uint4(int Lp) { const size_t Lpp = (size_t)Lp + kCandSpare; return 4 * Lpp + (Lpp + 3) / 4; }
void launch_step_pub(hipStream_t s, DeviceState& d, int B, const double* exact_dev, const unsigned short* order_dev,
                     const FastHandoff& fh, const NoiseD& qt, const ObserveExtras& ex, const CandTable& cand, const uint4* erec_dev,
                     const unsigned* glist_dev, const unsigned* skip_dev, int ecap, int64_t p0 = 0, int64_t p1 = -1,
                     int reserve_cus = 0);
// every particle of [p0, p1) flagged for the fall-back kernels when *over != 0 (a candidate list overflowed)
void launch_flag_range_if(hipStream_t s, const unsigned* over_dev, unsigned char* pflag_dev, unsigned* n_flagged_dev, int64_t p0, int64_t p1);
// The same for maps beyond kRegsMaxL landmarks (up to kPubBigMaxL), in two passes over the map (the second from L2 / Infinity
// Cache); candidate and inverse lists of 2 kCandSlots entries: cand.rec [Lp][3], erec [Lp][2].
constexpr int kPubBigMaxL = 6144;  // six pairs per lane: beyond that the slot words of a lane no longer fit its registers
int step_pub_big_entry_capacity(int B);
size_t step_pub_big_lds_bytes(int B, int ecap);
void launch_step_pub_big(hipStream_t s, DeviceState& d, int B, const double* exact_dev, const unsigned short* order_dev,
                         const FastHandoff& fh, const NoiseD& qt, const ObserveExtras& ex, const CandTable& cand, const uint4* erec_dev,
                         const unsigned* glist_dev, const unsigned* skip_dev, int ecap, const float4* gate4_dev = nullptr, int64_t p0 = 0,
                         int64_t p1 = -1, int reserve_cus = 0, const uint4* prim_dev = nullptr, const unsigned* stats_dev = nullptr);
// The two-workgroups-per-CU instance of the two-pass kernel (pk_k_step_duo.hip): ONE landmark per lane and turn, one carried word
// per landmark, the expected bearing worked out again in pass 2 -- at most 128 VGPRs, so that two 512-lane workgroups share a CU
// (four waves per SIMD) and one's row latency is the other's float64 issue.  Each has half the CU's LDS: k_cand_entries decides per
// scan whether the publish table fits (step_pub_duo_limits -> DuoLimits; *skip_duo), k_step_pub_big takes the scans that do not.
void step_pub_duo_limits(int B, int Lp, int nl, DuoLimits* out);
size_t step_pub_duo_lds_bytes(int B, const DuoLimits& lim);
void launch_step_pub_duo(hipStream_t s, DeviceState& d, int B, const double* exact_dev, const unsigned short* order_dev,
                         const FastHandoff& fh, const NoiseD& qt, const ObserveExtras& ex, const CandTable& cand, const uint4* erec_dev,
                         const unsigned* glist_dev, const unsigned* skip_duo_dev, const unsigned* stats_dev, const DuoLimits& lim,
                         const float4* gate4_dev, const uint4* prim_dev, int64_t p0 = 0, int64_t p1 = -1, int reserve_cus = 0);
extern int g_observe_nv;
// dynamic LDS of the general ML instance of k_observe (per-particle chains first[Lp], next[B], ids[B]) and of
// k_assoc_brute (best[B] u64 + bid[B]): callers check them against kMaxDynLds BEFORE anything is enqueued
inline size_t observe_general_lds_bytes(int Lp, int B) { return sizeof(int32_t) * ((size_t)Lp + 2 * (size_t)B); }
inline size_t assoc_brute_lds_bytes(int B) { return (size_t)B * 12; }
// K4: weights -> block totals / local scans
void launch_block_max(hipStream_t s, DeviceState& d, double* partial_dev, double* gmax_dev);
void launch_keys_max(hipStream_t s, const unsigned long long* keys_dev, double* gmax_dev);
void launch_scan_local(hipStream_t s, DeviceState& d, const double* gmax_dev, int domain,
                       double* clocal_dev, double* totals_dev, const unsigned long long* gmax_key_dev = nullptr);
// exclusive scan of the (global) block totals, sequential in block order: offsets[nb], sum[1]
void launch_scan_blocks(hipStream_t s, const double* totals_dev, int64_t nb, double* offsets_dev,
                        double* sum_dev);
// ancestors of the local output slots [slot0, slot0 + n); offsets_dev == sum_dev == NULL with
// nb <= kAncestorsScanMaxBlocks: the kernel scans the block totals itself (same order, same bits)
constexpr int64_t kAncestorsScanMaxBlocks = 256;
void launch_ancestors(hipStream_t s, const double* clocal_dev, const double* totals_dev,
                      const double* offsets_dev, const double* sum_dev, int64_t nb, int64_t P_global,
                      int64_t P_scan, double u, int64_t slot0, int64_t n, int32_t* anc_dev,
                      DeviceState* gather = nullptr);  // gather: also gather the poses (fused K5)
// K6
void launch_summary_partials(hipStream_t s, DeviceState& d, double* partial_dev, double* out4_dev);
// map maintenance
void launch_materialise(hipStream_t s, DeviceState& d);
void launch_broadcast_slot(hipStream_t s, DeviceState& d, const unsigned char* slot_dev);
void launch_probe(hipStream_t s, const double* in_dev, double* out_dev);
// the general dense path (pk_k_dense.hip): 30-row slots, full 5x5 covariances, any 4x4 Qt; one kernel does the (brute-force)
// maximum-likelihood association -- or takes supplied ids --, the EKF updates in scan order and the log-weight.
// update = false: association only (ids_out_dev), no state is touched.
size_t dense_lds_bytes(int Lp, int B);
void launch_observe_dense(hipStream_t s, DeviceState& d, const double* blobs_dev, const double* blobdir_dev, int B,
                          const int32_t* ids_in_dev, int32_t* ids_out_dev, const double Qt[16], bool update,
                          const ObserveExtras& ex);
void launch_probe_dense(hipStream_t s, const double* in_dev, double* out_dev);
// sharded resample
// record header in front of the map slot: x, y, h, logw, slot_lo, slot_hi (the last two int64:
// the global output slots this copy fills at its destination; 0, 0 when the caller plans on the host)
// balanced placement: + klo (int64, the logical index of the child in slot_lo) + one spare word -- 64 bytes
constexpr size_t kPoseRecordBytes = 8 * sizeof(double);
// device tables of the balanced plan (all sized for the WHOLE filter: every rank derives the whole plan by itself)
struct BalancedBuffers {
  double* glogw = nullptr;    // [Pg] log-weights in logical order
  double* clocal = nullptr;   // [Pg] block-local weight scans, [nbg] totals, [nbg + 1] offsets: the 1-GPU scan's tables
  double* totals = nullptr;
  double* offsets = nullptr;
  double* sum = nullptr;
  int64_t* H = nullptr;       // [Pg + 1] offspring table in logical order
  long long* cloc = nullptr;  // [Pg] packed (children << 32 | has children) block-local scans in PHYSICAL order
  long long* ctot = nullptr;  // [nbg], [nbg + 1]
  long long* coff = nullptr;
  int64_t* rel = nullptr;     // [P + 1] this rank's child positions
  int64_t* Hl = nullptr;      // [P]
  int32_t* alive = nullptr;   // [P]
  int* bad = nullptr;         // consistency failures seen by the kernels (must stay 0)
  int64_t cap = 0;            // Pg the tables were sized for
};
void launch_iota64(hipStream_t s, int64_t* p, int64_t n, int64_t off);
void launch_bal_state(hipStream_t s, const DeviceState& d, double* out_dev);
void launch_bal_plan(hipStream_t s, const DeviceState& d, const double* gstate_dev, int64_t Pg, int world, int rank,
                     const double* gmax_dev, int domain, double u, BalancedBuffers& b, int64_t* table_dev);
void launch_bal_pack(hipStream_t s, DeviceState& d, const BalancedBuffers& b, int64_t a0, int64_t n, int64_t ebase_s,
                     int64_t dbase_d, int64_t dd_d, int64_t m_d, unsigned char* buf_dev, size_t stride, const GrowState* g,
                     int64_t keep = -1);
void launch_bal_adopt(hipStream_t s, DeviceState& d, const BalancedBuffers& b, int64_t m, const unsigned char* buf_dev,
                      int64_t n_recv, int64_t* rh_dev, int mode, size_t stride, int32_t* anc_dev);
void launch_offspring(hipStream_t s, const double* clocal_dev, const double* offsets_dev, const double* sum_dev,
                      int64_t first_block, int64_t P_local, int64_t P_global, double u, int last_shard,
                      int64_t* hi_dev);
void launch_offspring_plan(hipStream_t s, const double* clocal_dev, const double* global_totals_dev,
                           int64_t n_global_blocks, int64_t first_block, int64_t P_local, int64_t P_global, double u,
                           int last_shard, int64_t* hi_dev, int world, int64_t* ranges_dev, unsigned* ticket_dev);
void launch_offspring_global(hipStream_t s, const double* clocal_global_dev, const double* offsets_dev, const double* sum_dev,
                             int64_t goff, int64_t P_local, int64_t P_global, double u, int last_shard, int64_t* hi_dev);
void launch_scan_local_of(hipStream_t s, const double* logw_dev, int64_t n, const double* gmax_dev, int domain,
                          double* clocal_dev, double* totals_dev);
void launch_pack(hipStream_t s, DeviceState& d, const int64_t* idx_dev, int64_t n, unsigned char* buf_dev);
void launch_adopt(hipStream_t s, DeviceState& d, const int64_t* src_dev, const unsigned char* buf_dev);
// device-resident variant of the exchange (no per-particle host metadata)
void launch_shard_ranges(hipStream_t s, const int64_t* hi_dev, int64_t P_local, int world, int64_t* ranges_dev);
void launch_pack_range(hipStream_t s, DeviceState& d, const int64_t* hi_dev, int64_t j0, int64_t n, int64_t slot_start,
                       int64_t slot_end, unsigned char* buf_dev);
// mode 0: the whole new generation; 1: only the slots this shard's own particles fill (the generation becomes current);
// 2: only the slots filled by received records (into the generation mode 1 made current)
void launch_adopt_dev(hipStream_t s, DeviceState& d, const int64_t* hi_dev, int64_t slot_start,
                      const unsigned char* buf_dev, int64_t n_recv, int64_t* rlohi_dev, int mode = 0, int64_t span_lo = INT64_MIN, int64_t span_hi = INT64_MAX);
void launch_iota(hipStream_t s, int32_t* p, int64_t n);
// unm_dev != NULL: the unmatched blobs of particle p come from its bit row there (the one-pass kernels leave it) unless pflag_dev[p] != 0
// (the fall-back kernels took the particle and left its ids in ids_dev's row)
void launch_new_landmarks(hipStream_t s, DeviceState& d, GrowState& g, const int32_t* ids_dev, const double* blobs_dev, int B,
                          const unsigned* unm_dev = nullptr, int unm_words = 0, const unsigned char* pflag_dev = nullptr);
// anc >= 0: a particle of this filter; anc < 0: record -anc - 1 of buf (stride bytes apart, its bookkeeping tail_off bytes in)
void launch_grow_gather(hipStream_t s, GrowState& g, const int32_t* anc_dev, int64_t P, const unsigned char* buf_dev = nullptr,
                        size_t stride = 0, size_t tail_off = 0);
// scan block: pinned (device-mapped) host memory -> HBM by a kernel, in stream order
void launch_upload(hipStream_t s, void* dst_dev, const void* src_host_mapped, size_t bytes);

}  // namespace pk
