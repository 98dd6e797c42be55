// C ABI of the FastSLAM particle update (include/parakeet_slam.h) on top of the gfx950
// kernels.  Host orchestration only: argument checks, host<->device staging, launch
// order, hipEvent instrumentation.  No arithmetic of the path runs on the host except
// the three reductions to scalars that the reference also does in Python floats
// (summary's division and atan2, prkt_core_v2.py:273-275) and the per-blob unit ray
// direction cos/sin (prkt_core_v2.py:510), which are O(B), not O(P*L).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/parakeet_slam.h"
#include "pk_kernels.hpp"

using namespace pk;

extern "C" {
int pk_download_landmarks(pk_filter* f, int64_t p0, int64_t p1, double* means, double* covs, int32_t* counts);
int pk_upload_landmarks(pk_filter* f, int64_t p0, int64_t p1, const double* means, const double* covs, const int32_t* counts);
}

#ifdef PK_STAMPS
namespace pk { void debug_read_stamps(unsigned long long* out, bool reset); void debug_read_fused_stamps(unsigned long long* out, bool reset); void debug_read_regs_stamps(unsigned long long* out, bool reset); void debug_read_pub_stamps(unsigned long long* out, bool reset); void debug_read_pub_wave_stamps(unsigned long long* out, bool reset); }
#endif

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define PK_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (void)hipGetLastError();                                                               \
      return fail(e_ == hipErrorOutOfMemory ? PK_ERR_NOMEM : PK_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                  hipGetErrorString(e_), __FILE__, __LINE__);                                \
    }                                                                                        \
  } while (0)

// Kernel launches report configuration errors (too much dynamic LDS, bad grid) through the
// runtime's last-error slot, not through a return value: every entry point that enqueued kernels
// asks for it before it reports success.
#define PK_LAUNCH_CHECK(what)                                                                \
  do {                                                                                       \
    hipError_t e_ = hipGetLastError();                                                       \
    if (e_ != hipSuccess)                                                                    \
      return fail(PK_ERR_HIP, "%s: a kernel launch failed: %s", what, hipGetErrorString(e_)); \
  } while (0)

struct TimedSpan {
  int slot;
  hipEvent_t a, b;
};

}  // namespace

struct AssocLaunch {
  bool fused = false;  // nothing launched yet: k_step_fused does gates + EKF in one kernel
  bool regs = false;   // nothing launched yet: k_step_regs does the same for 512 < L <= 2048
  bool retry = false;  // with regs: the hand-off lists for the flagged particles' second chance are allocated
  bool big = false;    // nothing launched yet: k_step_pub_big (2 048 < L <= 6 144: publish / subscribe in two passes over the map)
  BlobGrid grid{};
  int n9 = 0;
  const unsigned char* tables = nullptr;
  bool fast = false;  // hand-off written: k_observe_fast can run
  const double* blobs = nullptr;
  const double* dir = nullptr;
  const double* exact = nullptr;
  const unsigned short* order = nullptr;
};

struct pk_filter {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  DeviceState d{};
  NoiseD qt{0.1, 0.1, 0.0, 0.0, 0.1, 0.0, 0.1};
  double qt16[16] = {0.1, 0, 0, 0, 0, 0.1, 0, 0, 0, 0, 0.1, 0, 0, 0, 0, 0.1};  // the same, dense (prkt_core_v2.py:50-53)
  bool qt_dense = false;   // Qt couples bearing and colour or is not symmetric: only the dense kernels take it
  bool dense = false;      // maps in the dense 30-row layout (pk_layout.hpp): the general dense kernels run the observes
  std::vector<double> dense_staged;  // pk_stage_scan in dense mode: the blobs, kept on the host
  bool map_loaded = false;
  bool src_identity = true;
  int64_t nblocks = 0;  // weight-scan blocks
  int64_t device_bytes = 0;
  // workspaces
  double* z_dev = nullptr;        // P x 3
  unsigned char* scan_dev = nullptr;  // per-scan block: ctl | blobs | chains or association tables
  size_t scan_cap = 0;
  bool gmax_fused = false;  // ctl holds the max of the current log-weights (set by observe)
  int32_t* ids_dev = nullptr;     // P x B
  int64_t ids_cap = 0;
  double* g_totals = nullptr;   // sharded resample: every shard's block totals
  double* g_offsets = nullptr;
  int64_t gblocks_cap = 0;
  double* gl_clocal = nullptr;  // global-scan mode of the sharded resample: block-local scans of ALL particles' weights
  double* gl_totals = nullptr;
  double* gl_offsets = nullptr;
  int64_t gl_cap = 0;
  int64_t* hi_dev = nullptr;    // P + 1
  unsigned* plan_ticket = nullptr;  // workgroup counter of the one-launch shard plan
  int64_t* idx_dev = nullptr;   // P
  int64_t* srcs_dev = nullptr;  // P
  int64_t* rlohi_dev = nullptr; // (lo, hi) of the received records
  int64_t rlohi_cap = 0;
  BalancedBuffers bal{};        // balanced placement of the sharded filter: the plan's tables (every rank holds the whole plan)
  int64_t bal_m = -1;           // slots this rank's own children fill in the plan that is being carried out (-1: none)
  int64_t bal_loop_keep = -1;   // "balanced_loopback_keep" (debug, one-rank tests of the exchange): the next balanced adoption fills only the
                                // slots [0, keep) with this rank's own children; the slots [keep, P) come from records -- which the caller
                                // packs with pk_shard_pack_balanced_loop_dev and sends through the all-to-all to itself
  int assoc_kernel = 0;  // 0 = colour-grid kernel, 1 = brute-force reference kernel
  int assoc_dup = 1;     // grid kernel: use the 9x column-duplicated index list when it fits in LDS
  // host half of an ML scan upload done ahead of time (pk_stage_scan): tables built in a staging slot
  struct Staged {
    bool valid = false;
    int B = 0;
    unsigned char* st = nullptr;
    int slot = 0;
    BlobGrid g{};
    int n9 = 0;
    bool use_grid = false;
    size_t tab_bytes = 0;
    bool uploaded = false;  // pk_step sent the block to the device together with the motion kernel
  } staged;
  int route = PK_ROUTE_NONE;  // kernels used by the last observe
  int upload_kernel = 1; // per-scan block: read from pinned host memory by a kernel (1) or hipMemcpyAsync (0)
  int fused_step = 1;    // L <= 512 and small scan tables: k_step_fused instead of hand-off + k_observe_fast
  unsigned* bcnt_dev = nullptr;  // [bcand_cap] entries of the blobs' inverse candidate lists
  uint4* brec_dev = nullptr;     // [bcand_cap] the lists
  int64_t bcand_cap = 0;
  int cand_lists = 1;    // k_step_regs: gates against the reference particle's candidate lists (k_candidates) instead of the grid walk
  int64_t loop_lo = INT64_MIN, loop_hi = INT64_MAX;  // "split_loopback_lo/hi" (debug): local slots outside come from records
  uint4* cand_dev = nullptr;  // [Lp + kCandSpare][3] candidate records (two or three uint4 per landmark in use)
  int regs_step = 1;     // 512 < L <= 2048 and scan tables that fit LDS: k_step_regs (one pass, map in registers)
  int pub_step = 1;      // ... with the contested blobs settled by static publish / subscribe (k_step_pub) while the publish table fits LDS
  bool flag_folded = false;  // this scan's k_cand_entries flags every particle itself when nobody takes the scan (whole observes)
  int pub_small = -1;    // L <= 512: k_step_pub<256 lanes> instead of k_step_fused -- 1 / 0, or -1 (default): where it is measured faster
                         // (pub_small_now below)
  int duo_park_limit = -1;  // >= 0: k_step_pub_duo's overflow area is treated as this small (tests: particles that need more go to the fall-back kernels)
  int duo_on = 0;        // "pub_duo" (measured, off: DESIGN.md section 4): 2 048 < L <= 5 120, scans whose publish table fits its share of a CU's LDS go to
                         // k_step_pub_duo -- 1: two 512-lane workgroups per CU (<= 128 VGPRs), 2: three 256-lane workgroups (<= 168) -- the others to k_step_pub_big
  int pub_entry_limit = 0;  // > 0: the publish table is treated as this small (tests: scans whose table "does not fit" fall back to k_step_regs)
  uint4* erec_dev = nullptr;     // [Lp] publish entries of every landmark's candidates (k_cand_entries)
  uint4* erec_dev2 = nullptr;    // [Lp][2] the same for sixteen-entry lists (k_step_pub_big)
  unsigned* binfo_dev = nullptr; // [bcand_cap] per blob: first entry | contenders << 16
  unsigned char* npass_dev = nullptr; // [Lp + kCandSpare] per landmark: blobs inside the reference particle's own gates (k_candidates)
  unsigned* unm_dev = nullptr;   // growing maps on the publish / subscribe routes: [P][unm_words] every particle's unmatched blobs, bits in scan order
  int unm_words = 0;
  int64_t unm_cap = 0;
  bool grow_bits = false;        // the last observe's one-pass kernel left those rows (k_new_landmarks reads them where the particle was not handed on)
  uint4* prim_dev = nullptr;     // the two-pass kernels' primary-blob table: every landmark's first candidate in landmark order (prim_table_uint4; k_cand_entries)
  float4* gate4_dev = nullptr;   // [bcand_cap] every blob's bearing and colour as float: k_step_pub_big's first look (k_cand_entries)
  uint4* far_dev = nullptr;      // [Lp + kCandSpare][3] per landmark: the bound its list was pruned with | its far list (k_candidates, pk_pub_math.hpp)
  int far_prune = 1;             // look-alikes certainly beyond the underflow edge leave the candidate lists once per scan (0: as round 4)
  unsigned* glist_dev = nullptr; // [bcand_cap + 1 + 256] the same for the blobs several landmarks list, compacted; then their number; then the octet orders of k_step_pub (128 u16) and k_step_pub_big (384 u16)
  // a split observe in progress (pk_observe_staged_range): what the first call set up for the later ones
  struct Split {
    bool active = false;
    AssocLaunch al;
    CandTable cand;
    int B = 0;
    bool reset = false;
  } split;
  bool adopt_local_done = false;  // pk_shard_adopt_local_dev made the new generation current; pk_shard_adopt_remote_dev may fill it
  int pub_ecap = 0;       // k_step_pub was prepared for the current scan with this many publish entries (0: not prepared)
  int split_reserve_cus = 16;  // CUs the first part of a split step leaves free for the all-to-all's kernels
  int regs_retry = 1;    // k_step_regs: 1 = the particles it flags get a second chance (eight-slot hand-off + k_observe_sweep) before the general kernels
  int regs_warm = 1;     // k_step_regs: L2 warming of the next particle's slot: 0 none, 1 its mean rows (default), 2 the whole slot (measured slower)
  int fast_observe = 1;  // association hand-off + k_observe_fast (L <= 512) / k_observe_sweep; 2 = always the sweep kernel
  uint4* sweep_results = nullptr;  // k_observe_sweep: per-workgroup result lists
  size_t sweep_cap = 0;
  unsigned* retry_seen = nullptr;  // pinned host word: second-chance rows the last scan WANTED (copied behind every second chance)
  int64_t retry_rows_min = 0;      // what retry_rows() grows to when a scan wanted more rows than there were
  FastHandoff fh{};      // device buffers of the hand-off
  int64_t fh_cap_l = 0, fh_cap_b = 0;
  // pinned host staging ring for the per-scan uploads (blobs, ray directions, chains):
  // lets pk_observe/pk_step return without synchronising the stream
  static constexpr int kRing = 8;
  unsigned char* stage[kRing] = {nullptr};
  hipEvent_t stage_done[kRing] = {nullptr};
  // which enqueued upload last read each slot, and up to which upload each slot's event covers
  // (an event is recorded behind every 4th upload only; a slot whose covering record never came --
  // its scan was staged and then discarded -- gets one when the slot is next handed out)
  uint64_t upload_seq = 0;
  uint64_t slot_seq[kRing] = {0};
  uint64_t event_seq[kRing] = {0};
  size_t stage_cap = 0;
  int stage_next = 0;
  double* partial = nullptr;  // 4 * 1024
  double* gmax = nullptr;
  double* clocal = nullptr;   // P
  double* totals = nullptr;   // nblocks
  double* offsets = nullptr;  // nblocks
  double* sum = nullptr;
  double* out4 = nullptr;
  double* pose_part = nullptr;   // [motion_pose_blocks(P)][4]: per-block sums of x, y, sin h, cos h the last whole-filter motion launch left
  bool pose_part_ok = false;     // ... and nothing has touched the poses since
  GrowState grow{};                 // section 8(f4) on the device (pk_grow_enable): per-particle new-landmark bookkeeping
  bool grow_on = false;
  int32_t* anc = nullptr;           // P
  unsigned char* slot_tmp = nullptr;  // one slot
  // timing
  uint32_t timing_mask = 0;  // bit i: PK_T_* slot i is bracketed by hipEvents
  int timing_stride = 1;     // ... every timing_stride-th time the slot comes up (sampling keeps the probe cheap)
  int64_t timing_seen[PK_T_COUNT] = {0};
  std::vector<TimedSpan> pending;
  std::vector<hipEvent_t> pool;
  double ms[PK_T_COUNT] = {0};
  int64_t launches[PK_T_COUNT] = {0};
};

namespace {

template <typename T>
int dev_alloc(pk_filter* f, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(PK_ERR_NOMEM, "hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
  }
  f->device_bytes += (int64_t)(n * sizeof(T));
  return PK_OK;
}

struct Span {
  pk_filter* f;
  int slot;
  hipEvent_t a = nullptr, b = nullptr;
  Span(pk_filter* f_, int slot_) : f(f_), slot(slot_) {
    if (!((f->timing_mask >> slot_) & 1u)) return;
    if (f->timing_seen[slot_]++ % f->timing_stride != 0) return;
    a = take();
    b = take();
    if (a) (void)hipEventRecord(a, f->stream);
  }
  hipEvent_t take() {
    if (!f->pool.empty()) {
      hipEvent_t e = f->pool.back();
      f->pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
  ~Span() {
    if (!a || !b) return;
    (void)hipEventRecord(b, f->stream);
    f->pending.push_back(TimedSpan{slot, a, b});
  }
};

int drain_timings(pk_filter* f) {
  if (f->pending.empty()) return PK_OK;
  PK_HIP(hipStreamSynchronize(f->stream));
  for (auto& t : f->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
      f->ms[t.slot] += ms;
      f->launches[t.slot] += 1;
    }
    f->pool.push_back(t.a);
    f->pool.push_back(t.b);
  }
  f->pending.clear();
  return PK_OK;
}

int use_device(pk_filter* f) {
  PK_HIP(hipSetDevice(f->device));
  return PK_OK;
}

// One device block per scan, filled by ONE host->device copy:
//   [ctl: kGmaxKeys running-max keys (u64), flagged-particle count (u32)] [blobs 4B f64] then either
//   known ids:  [first Lp i32] [next B i32]
//   ML:         [dir 2B f64] [exact 6B f64] [association tables]
// The copy also zeroes ctl, which is how every observe starts with a fresh max / count.
constexpr size_t kCtlBytes = 8 * kGmaxKeys + 64;  // running-max keys, then the flagged-particle count, the route control words and the publish table's figures
int ensure_scan_capacity(pk_filter* f, size_t bytes) {
  if (bytes <= f->scan_cap) return PK_OK;
  PK_HIP(hipStreamSynchronize(f->stream));
  if (f->scan_dev) (void)hipFree(f->scan_dev);
  f->scan_dev = nullptr;
  f->scan_cap = 0;
  const size_t cap = bytes + bytes / 4 + 4096;
  int rc;
  if ((rc = dev_alloc(f, &f->scan_dev, cap))) return rc;
  f->scan_cap = cap;
  f->gmax_fused = false;  // the running-max keys lived in the block that was just freed
  return PK_OK;
}

int ensure_ids_capacity(pk_filter* f, int B) {
  int64_t need = f->d.P * (int64_t)B;
  if (need <= f->ids_cap) return PK_OK;
  PK_HIP(hipStreamSynchronize(f->stream));
  if (f->ids_dev) (void)hipFree(f->ids_dev);
  f->ids_dev = nullptr;
  f->ids_cap = 0;
  int rc;
  if ((rc = dev_alloc(f, &f->ids_dev, (size_t)need))) return rc;
  f->ids_cap = need;
  return PK_OK;
}

// One scan block host -> device in stream order.  The staging ring is pinned and device-mapped, so
// a small kernel reads it directly (k_upload); sizes are padded to 16 bytes on both sides.  Falls
// back to the copy engine when the mapping is not available.
int upload_scan(pk_filter* f, const unsigned char* st, size_t bytes) {
  void* dev_view = nullptr;
  if (f->upload_kernel && hipHostGetDevicePointer(&dev_view, const_cast<unsigned char*>(st), 0) == hipSuccess && dev_view) {
    launch_upload(f->stream, f->scan_dev, dev_view, bytes);
    return PK_OK;
  }
  (void)hipGetLastError();
  PK_HIP(hipMemcpyAsync(f->scan_dev, st, bytes, hipMemcpyHostToDevice, f->stream));
  return PK_OK;
}

// An upload that reads staging slot `slot` has just been enqueued on the stream.
int note_upload(pk_filter* f, int slot) {
  f->slot_seq[slot] = ++f->upload_seq;
  if ((slot & 3) == 3) {  // an event record costs the stream a few microseconds: one per four uploads
    PK_HIP(hipEventRecord(f->stage_done[slot], f->stream));
    f->event_seq[slot] = f->upload_seq;
  }
  return PK_OK;
}

// Next pinned staging block of at least `bytes`; waits for the upload that last read it.
int take_stage(pk_filter* f, size_t bytes, unsigned char** out, int* slot) {
  if (bytes > f->stage_cap) {
    PK_HIP(hipStreamSynchronize(f->stream));
    size_t cap = bytes + bytes / 4 + 4096;
    for (int i = 0; i < pk_filter::kRing; ++i) {
      if (f->stage[i]) (void)hipHostFree(f->stage[i]);
      f->stage[i] = nullptr;
      PK_HIP(hipHostMalloc((void**)&f->stage[i], cap, hipHostMallocMapped));
      if (!f->stage_done[i]) PK_HIP(hipEventCreateWithFlags(&f->stage_done[i], hipEventDisableTiming));
      f->slot_seq[i] = 0;  // the stream is idle: nothing reads the old blocks any more
    }
    f->stage_cap = cap;
  }
  int i = f->stage_next;
  f->stage_next = (i + 1) % pk_filter::kRing;
  // Slot i may still be read by upload number slot_seq[i]; any event recorded at or after that upload
  // covers it.  Normally that is the record behind slot (i | 3) of the previous trip round the ring;
  // when that slot's scan was discarded before its upload (pk_stage_scan followed by a supplied-ids
  // observe, an error between take_stage and the upload) no such record exists and one is made now.
  const uint64_t need = f->slot_seq[i];
  if (need != 0) {
    int ev = -1;
    for (int j = 0; j < pk_filter::kRing; ++j)
      if (f->event_seq[j] >= need && (ev < 0 || f->event_seq[j] < f->event_seq[ev])) ev = j;
    if (ev < 0) {
      PK_HIP(hipEventRecord(f->stage_done[i], f->stream));
      f->event_seq[i] = f->upload_seq;
      ev = i;
    }
    PK_HIP(hipEventSynchronize(f->stage_done[ev]));
    f->slot_seq[i] = 0;
  }
  *out = f->stage[i];
  *slot = i;
  return PK_OK;
}

int materialise(pk_filter* f) {
  if (f->src_identity) return PK_OK;
  {
    Span t(f, PK_T_MATERIALISE);
    launch_materialise(f->stream, f->d);
  }
  f->src_identity = true;
  f->d.alt = nullptr;  // every slot now lives in the shard's own buffer
  return PK_OK;
}

// Pack one landmark (dense 5x5 host form) into the compact fields of a host slot image.
void pack_landmark(const MapLayout& lay, unsigned char* slot, int l, const double* mean, const double* cov) {
  double* fl = reinterpret_cast<double*>(slot);
  const int Lp = lay.Lp;
  for (int i = 0; i < 5; ++i) fl[(size_t)i * Lp + l] = mean[i];
  if (lay.fields == kDenseFields) {  // the reference's full state, entry by entry
    for (int i = 0; i < 25; ++i) fl[(size_t)(5 + i) * Lp + l] = cov[i];
    return;
  }
  fl[(size_t)F_PXX * Lp + l] = cov[0];
  fl[(size_t)F_PXY * Lp + l] = 0.5 * (cov[1] + cov[5]);
  fl[(size_t)F_PYY * Lp + l] = cov[6];
  fl[(size_t)F_CRR * Lp + l] = cov[12];
  fl[(size_t)F_CRG * Lp + l] = 0.5 * (cov[13] + cov[17]);
  fl[(size_t)F_CRB * Lp + l] = 0.5 * (cov[14] + cov[22]);
  fl[(size_t)F_CGG * Lp + l] = cov[18];
  fl[(size_t)F_CGB * Lp + l] = 0.5 * (cov[19] + cov[23]);
  fl[(size_t)F_CBB * Lp + l] = cov[24];
}

void unpack_landmark(const MapLayout& lay, const unsigned char* slot, int l, double* mean, double* cov) {
  const double* fl = reinterpret_cast<const double*>(slot);
  const int Lp = lay.Lp;
  if (mean)
    for (int i = 0; i < 5; ++i) mean[i] = fl[(size_t)i * Lp + l];
  if (cov && lay.fields == kDenseFields) {
    for (int i = 0; i < 25; ++i) cov[i] = fl[(size_t)(5 + i) * Lp + l];
  } else if (cov) {
    for (int i = 0; i < 25; ++i) cov[i] = 0.0;
    cov[0] = fl[(size_t)F_PXX * Lp + l];
    cov[1] = cov[5] = fl[(size_t)F_PXY * Lp + l];
    cov[6] = fl[(size_t)F_PYY * Lp + l];
    cov[12] = fl[(size_t)F_CRR * Lp + l];
    cov[13] = cov[17] = fl[(size_t)F_CRG * Lp + l];
    cov[14] = cov[22] = fl[(size_t)F_CRB * Lp + l];
    cov[18] = fl[(size_t)F_CGG * Lp + l];
    cov[19] = cov[23] = fl[(size_t)F_CGB * Lp + l];
    cov[24] = fl[(size_t)F_CBB * Lp + l];
  }
}

// unit((cos b, sin b, 0.0)) of closest_point (prkt_core_v2.py:510, utils.py:68-76): the ray
// direction of each blob, scaled by 1/length exactly like utils.scale(vector, 1.0/length).
void blob_directions(const double* blobs, int B, double* dir) {
  for (int b = 0; b < B; ++b) {
    double c = std::cos(blobs[4 * b]), s = std::sin(blobs[4 * b]);
    double len = std::sqrt(c * c + s * s + 0.0 * 0.0);
    dir[2 * b] = c * (1.0 / len);
    dir[2 * b + 1] = s * (1.0 / len);
  }
}

// Bucket the blobs of one scan into a 3-D colour grid (cell edge kGridCell > sqrt(300), the
// colour gate radius of prkt_core_v2.py:441) and lay out what k_assoc_grid reads:
//   tables: start u16[ncell+1] (padded to 16 bytes) | rec32 float4[B] | idx9 u16[n9] | order u16[B]
//   exact:  double[B][6] = bearing, r, g, b, ux, uy      (rec32, exact, order in cell order)
// With dup, idx9 lists for every (r, g) column and b cell the blobs (cell-order index) within
// one cell in r and g, and `start` = offsets into idx9; otherwise n9 = 0 and `start` = cell
// offsets into rec32.  Returns n9 through *n9_out (0 when the duplicated list is not built).
void build_blob_grid(const double* blobs, const double* dir, int B, bool want_dup, BlobGrid& g,
                     unsigned char* tables, double* exact, int* n9_out) {
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, M = 0.0, Mb = 0.0;
  for (int k = 0; k < 3; ++k) {
    lo[k] = hi[k] = B ? blobs[1 + k] : 0.0;
    for (int b = 0; b < B; ++b) {
      double v = blobs[4 * b + 1 + k];
      lo[k] = std::fmin(lo[k], v);
      hi[k] = std::fmax(hi[k], v);
      M = std::fmax(M, std::fabs(v));
    }
  }
  for (int b = 0; b < B; ++b) Mb = std::fmax(Mb, std::fabs(blobs[4 * b]));
  g.inv_h = 1.0 / kGridCell;
  g.ncell = 1;
  for (int k = 0; k < 3; ++k) {
    g.lo[k] = lo[k];
    double span = std::floor((hi[k] - lo[k]) * g.inv_h) + 1.0;
    g.G[k] = span > (double)kGridMax ? kGridMax : (int)span;
    g.ncell *= g.G[k];
  }
  // fp32 pre-filter bounds.  Colour: with d the exact channel difference of a pair inside the
  // gate and d' its fp32 evaluation from fp32-rounded colours, |d' - d| <= eta = 2^-24 (2M + 35);
  // then sum d'^2 <= 300 + 60 eta + 3 eta^2 (+ fp32 summation error).  Bearing: a pair inside
  // the gate has |expected| <= Mb + 0.5, so |d' - d| <= 2^-24 (2 Mb + 1.5).  Both doubled.
  const double eta = 2.0 * std::ldexp(2.0 * M + 35.0, -24);
  const double thr = (300.0 + 60.0 * eta + 3.0 * eta * eta + 1e-3) * (1.0 + 1e-6);
  g.thr32 = thr < 3.0e38 ? (float)thr : INFINITY;
  const double thrb = (0.5 + 2.0 * std::ldexp(2.0 * Mb + 1.5, -24) + 1e-6) * (1.0 + 1e-6);
  g.thrb32 = thrb < 3.0e38 ? (float)thrb : INFINITY;
  auto cell_of = [&](int b) {
    int c[3];
    for (int k = 0; k < 3; ++k) {
      double q = std::floor((blobs[4 * b + 1 + k] - g.lo[k]) * g.inv_h);
      c[k] = q < 0.0 ? 0 : (q > (double)(g.G[k] - 1) ? g.G[k] - 1 : (int)q);
    }
    return (c[0] * g.G[1] + c[1]) * g.G[2] + c[2];
  };
  std::vector<int> cell((size_t)std::max(B, 1)), cs((size_t)g.ncell + 1, 0), fill;
  for (int b = 0; b < B; ++b) {
    cell[b] = cell_of(b);
    ++cs[cell[b] + 1];
  }
  for (int c = 0; c < g.ncell; ++c) cs[c + 1] += cs[c];
  fill = cs;
  const size_t cs_bytes = ((size_t)(g.ncell + 1) * 2 + 15) & ~(size_t)15;
  memset(tables, 0, cs_bytes);
  uint16_t* start = reinterpret_cast<uint16_t*>(tables);
  float* rec32 = reinterpret_cast<float*>(tables + cs_bytes);
  uint16_t* idx9 = reinterpret_cast<uint16_t*>(tables + cs_bytes + (size_t)B * 16);
  // duplicated column lists: size known before laying out `order`
  int n9 = 0;
  std::vector<int> col_start;
  if (want_dup) {
    col_start.assign((size_t)g.ncell + 1, 0);
    long total = 0;
    for (int r = 0; r < g.G[0]; ++r)
      for (int gg = 0; gg < g.G[1]; ++gg)
        for (int k = 0; k < g.G[2]; ++k) {
          col_start[(r * g.G[1] + gg) * g.G[2] + k] = (int)total;
          for (int r2 = std::max(r - 1, 0); r2 <= std::min(r + 1, g.G[0] - 1); ++r2)
            for (int g2 = std::max(gg - 1, 0); g2 <= std::min(gg + 1, g.G[1] - 1); ++g2) {
              const int c = (r2 * g.G[1] + g2) * g.G[2] + k;
              total += cs[c + 1] - cs[c];
            }
        }
    col_start[g.ncell] = (int)total;
    if (total + 3 <= 65535) n9 = (int)((total + 3 + 7) & ~7L);  // the walk reads up to three entries past a range
  }
  uint16_t* order = reinterpret_cast<uint16_t*>(tables + cs_bytes + (size_t)B * 16 + (size_t)n9 * 2);
  for (int b = 0; b < B; ++b) {  // ascending b inside a cell
    const int pos = fill[cell[b]]++;
    order[pos] = (uint16_t)b;
    rec32[4 * pos] = (float)blobs[4 * b + 1];
    rec32[4 * pos + 1] = (float)blobs[4 * b + 2];
    rec32[4 * pos + 2] = (float)blobs[4 * b + 3];
    rec32[4 * pos + 3] = (float)blobs[4 * b];
    double* e = exact + 6 * (size_t)pos;
    e[0] = blobs[4 * b];
    e[1] = blobs[4 * b + 1];
    e[2] = blobs[4 * b + 2];
    e[3] = blobs[4 * b + 3];
    e[4] = dir[2 * b];
    e[5] = dir[2 * b + 1];
  }
  if (n9 > 0) {
    int w = 0;
    for (int r = 0; r < g.G[0]; ++r)
      for (int gg = 0; gg < g.G[1]; ++gg)
        for (int k = 0; k < g.G[2]; ++k) {
          start[(r * g.G[1] + gg) * g.G[2] + k] = (uint16_t)w;
          for (int r2 = std::max(r - 1, 0); r2 <= std::min(r + 1, g.G[0] - 1); ++r2)
            for (int g2 = std::max(gg - 1, 0); g2 <= std::min(gg + 1, g.G[1] - 1); ++g2) {
              const int c = (r2 * g.G[1] + g2) * g.G[2] + k;
              for (int t = cs[c]; t < cs[c + 1]; ++t) idx9[w++] = (uint16_t)t;
            }
        }
    start[g.ncell] = (uint16_t)w;
    for (; w < n9; ++w) idx9[w] = 0;
  } else {
    for (int c = 0; c <= g.ncell; ++c) start[c] = (uint16_t)cs[c];
  }
  *n9_out = n9;
}

// rows: particles the lists have room for -- all of them (the hand-off routes), or the capped number of second-chance rows
// (ADVICE round 2 / VERDICT round 3: the second chance used to allocate lists for ALL P particles with the route, 6.4 GB at
// 100 000 x 2 000, for the few percent a scan flags at worst)
// (a scan the one-pass kernel stands back from as a WHOLE -- a list overflowed, the publish table does not fit -- flags
// all P particles; the rows grow to what the last scan wanted, so only the first such scan sends particles beyond P / 16 through
// the general kernels)
int64_t retry_rows(const pk_filter* f) {
  return std::min<int64_t>(f->d.P, std::max<int64_t>(std::max<int64_t>(1024, f->d.P / 16), f->retry_rows_min));
}
int ensure_handoff(pk_filter* f, int B, int slots, bool lists = true, bool retry_only = false) {
  if (retry_only && f->retry_seen) {
    const int64_t wanted = *reinterpret_cast<volatile unsigned*>(f->retry_seen);  // (the last finished scan's, or the one before)
    if (wanted > retry_rows(f)) f->retry_rows_min = std::min<int64_t>(f->d.P, wanted + wanted / 4);
  }
  const int64_t rows = retry_only ? retry_rows(f) : f->d.P;
  const int64_t need_l = lists ? rows * (int64_t)f->d.lay.Lp * (slots == kSweepSlots ? 2 : 1) : 0;
  const int64_t need_b = lists ? rows * (int64_t)std::max(B, 1) : 0;
  int rc;
  if (retry_only && !f->fh.row_of && (rc = dev_alloc(f, &f->fh.row_of, (size_t)f->d.P))) return rc;
  if (need_l > f->fh_cap_l) {
    PK_HIP(hipStreamSynchronize(f->stream));
    if (f->fh.lmpass) (void)hipFree(f->fh.lmpass);
    f->fh.lmpass = nullptr;
    f->fh_cap_l = 0;
    if ((rc = dev_alloc(f, &f->fh.lmpass, (size_t)need_l))) return rc;
    f->fh_cap_l = need_l;
  }
  if (need_b > f->fh_cap_b) {
    PK_HIP(hipStreamSynchronize(f->stream));
    if (f->fh.bcount) (void)hipFree(f->fh.bcount);
    f->fh.bcount = nullptr;
    f->fh_cap_b = 0;
    if ((rc = dev_alloc(f, &f->fh.bcount, (size_t)need_b))) return rc;
    f->fh_cap_b = need_b;
  }
  if (!f->fh.pflag && (rc = dev_alloc(f, &f->fh.pflag, (size_t)f->d.P))) return rc;
  return PK_OK;
}

inline unsigned long long* ctl_gmax_key(pk_filter* f) { return reinterpret_cast<unsigned long long*>(f->scan_dev); }
inline unsigned* ctl_n_flagged(pk_filter* f) { return reinterpret_cast<unsigned*>(f->scan_dev + 8 * kGmaxKeys); }
inline unsigned* ctl_cand_over(pk_filter* f) { return reinterpret_cast<unsigned*>(f->scan_dev + 8 * kGmaxKeys + 4); }
inline unsigned* ctl_n_stray(pk_filter* f) { return reinterpret_cast<unsigned*>(f->scan_dev + 8 * kGmaxKeys + 8); }
// written by k_cand_entries: != 0 -> k_step_pub stands back (a candidate list overflowed, or the publish table does not fit LDS)
inline unsigned* ctl_skip_pub(pk_filter* f) { return reinterpret_cast<unsigned*>(f->scan_dev + 8 * kGmaxKeys + 12); }
// != 0 -> the candidate-list instance of k_step_regs stands back (k_step_pub runs, or the grid walk does)
inline unsigned* ctl_skip_cand(pk_filter* f) { return reinterpret_cast<unsigned*>(f->scan_dev + 8 * kGmaxKeys + 16); }
// rows of the second-chance hand-off lists dealt out so far (FastHandoff::row_next)
inline unsigned* ctl_retry_rows(pk_filter* f) { return reinterpret_cast<unsigned*>(f->scan_dev + 8 * kGmaxKeys + 20); }
// written by k_cand_entries: != 0 -> the two-workgroups-per-CU instance of the two-pass kernel (k_step_pub_duo) stands back and
// k_step_pub_big takes the scan (the publish table, the contested blobs or the landmarks with several blobs exceed its share of LDS)
inline unsigned* ctl_skip_duo(pk_filter* f) { return reinterpret_cast<unsigned*>(f->scan_dev + 8 * kGmaxKeys + 24); }
// ... != 0 -> k_step_pub_big stands back (no publish / subscribe kernel takes the scan, or k_step_pub_duo does)
inline unsigned* ctl_skip_big(pk_filter* f) { return reinterpret_cast<unsigned*>(f->scan_dev + 8 * kGmaxKeys + 28); }
// what the scan's publish table came to (k_cand_entries; pk_observe_pub_stats): entries, contested blobs, landmarks of the reference
// particle with two or more blobs inside their gates, the longest candidate list
inline unsigned* ctl_pub_stats(pk_filter* f) { return reinterpret_cast<unsigned*>(f->scan_dev + 8 * kGmaxKeys + 32); }

// Host half of the ML scan upload: blobs, ray directions, exact records and the association tables
// are laid out in a pinned staging slot (no device work; may synchronise only to grow buffers).
// block: ctl | blobs (4B) | dir (2B) | exact (6B) doubles | tables
int stage_ml_scan(pk_filter* f, const double* blobs, int B) {
  int rc;
  if (B > 65535) return fail(PK_ERR_UNSUPPORTED, "maximum-likelihood association handles at most 65535 blobs per scan (got %d)", B);
  // the general EKF kernel (every ML route's last resort) builds per-particle chains in LDS
  if (observe_general_lds_bytes(f->d.lay.Lp, B) > kMaxDynLds)
    return fail(PK_ERR_UNSUPPORTED,
                "maximum-likelihood association: %d landmarks + 2 x %d blobs need %zu bytes of LDS chains per particle, the "
                "workgroup has %zu", f->d.lay.Lp, B, observe_general_lds_bytes(f->d.lay.Lp, B), (size_t)kMaxDynLds);
  if ((rc = ensure_ids_capacity(f, B))) return rc;
  pk_filter::Staged& sg = f->staged;
  sg.valid = false;
  sg.uploaded = false;
  const int ncell_max = kGridMax * kGridMax * kGridMax;
  const size_t tab_max = (blob_grid_table_bytes(ncell_max, B, 9 * B + 16) + 15) & ~(size_t)15;
  const size_t o_blobs = kCtlBytes;
  const size_t o_dir = o_blobs + (size_t)B * 4 * sizeof(double);
  const size_t o_exact = o_dir + (size_t)B * 2 * sizeof(double);
  const size_t o_tab = o_exact + (size_t)B * 6 * sizeof(double);
  unsigned char* st = nullptr;
  int slot = 0;
  if ((rc = take_stage(f, o_tab + tab_max + 16, &st, &slot))) return rc;
  if ((rc = ensure_scan_capacity(f, o_tab + tab_max + 16))) return rc;
  memset(st, 0, kCtlBytes);
  memmove(st + o_blobs, blobs, (size_t)B * 4 * sizeof(double));
  double* dir = reinterpret_cast<double*>(st + o_dir);
  blob_directions(blobs, B, dir);
  // the grid kernel keeps landmark indices as u16 and its tables in LDS
  sg.use_grid = f->assoc_kernel == 0 && f->d.lay.L <= 65535;
  sg.tab_bytes = 0;
  sg.n9 = 0;
  sg.g = BlobGrid{};
  if (sg.use_grid) {
    // duplicated column lists when they fit in LDS (about 48 B per blob), else the 9-range walk
    bool dup = f->assoc_dup && assoc_grid_lds_bytes(ncell_max, B, 9 * B + 16) <= kMaxDynLds;
    build_blob_grid(blobs, dir, B, dup, sg.g, st + o_tab, reinterpret_cast<double*>(st + o_exact), &sg.n9);
    sg.tab_bytes = (blob_grid_table_bytes(sg.g.ncell, B, sg.n9) + 15) & ~(size_t)15;
    if (assoc_grid_lds_bytes(sg.g.ncell, B, sg.n9) > kMaxDynLds) sg.use_grid = false;  // scan too large for LDS tables
  }
  sg.B = B;
  sg.st = st;
  sg.slot = slot;
  sg.valid = true;
  return PK_OK;
}

// Will a production observe of a scan with these tables take the register route (k_step_regs)?
// ... or, for maps beyond it, the two-pass publish / subscribe route (k_step_pub_big)?  Both run on particle ranges, which is
// what the split step of the sharded filter needs (the conditions are enqueue_association's).
static bool regs_route_taken(pk_filter* f, const BlobGrid& g, int B, int n9) {
  if (f->fast_observe != 1 || B <= 0 || f->d.lay.L >= 65535) return false;
  if (f->regs_step && f->d.lay.L > kFastMaxL && f->d.lay.L <= kRegsMaxL && n9 > 0 && regs_lds_bytes(g.ncell, B, n9) <= kMaxDynLds) return true;
  return f->pub_step && f->cand_lists && f->d.lay.L > kRegsMaxL && f->d.lay.L <= kPubBigMaxL && step_pub_big_entry_capacity(B) > 0 &&
         observe_sweep_plan(f->d, B).grid > 0;
}

// Upload one scan for maximum-likelihood association (one block) and enqueue the association.
// `blobs` may be the staged copy itself (pk_observe_staged).
// onepass_only (growing maps: the bookkeeping kernel needs every particle's unmatched blobs -- the publish / subscribe kernels leave them
// as bit rows, the general association as ids): a one-pass route whose kernel is of the publish / subscribe family, or else the
// general association; never the hand-off routes, k_step_fused or k_step_regs
int enqueue_association(pk_filter* f, const double* blobs, int B, bool finalize, bool want_fast, AssocLaunch* out, bool onepass_only = false) {
  int rc;
  const bool pub_ok = !onepass_only || (f->pub_step && f->cand_lists && B <= 32 * 176);
  pk_filter::Staged& sg = f->staged;
  const size_t o_blobs = kCtlBytes;
  const size_t o_dir = o_blobs + (size_t)B * 4 * sizeof(double);
  const size_t o_exact = o_dir + (size_t)B * 2 * sizeof(double);
  const size_t o_tab = o_exact + (size_t)B * 6 * sizeof(double);
  if (!(sg.valid && sg.B == B && blobs == reinterpret_cast<const double*>(sg.st + o_blobs)))
    if ((rc = stage_ml_scan(f, blobs, B))) return rc;
  sg.valid = false;  // consumed
  unsigned char* st = sg.st;
  const int slot = sg.slot;
  const BlobGrid g = sg.g;
  const int n9 = sg.n9;
  const bool use_grid = sg.use_grid;
  const size_t tab_bytes = sg.tab_bytes;
  if (!sg.uploaded) {
    if ((rc = upload_scan(f, st, use_grid ? o_tab + tab_bytes : o_exact))) return rc;
    if ((rc = note_upload(f, slot))) return rc;
  }
  sg.uploaded = false;
  f->gmax_fused = false;
  const double* blobs_dev = reinterpret_cast<const double*>(f->scan_dev + o_blobs);
  const double* dir_dev = reinterpret_cast<const double*>(f->scan_dev + o_dir);
  if (out) {
    out->blobs = blobs_dev;
    out->dir = dir_dev;
  }
  if (use_grid) {
    FastHandoff fh{};
    const bool sweep = f->d.lay.L > kFastMaxL || f->fast_observe >= 2;
    if (out && want_fast && !finalize && f->fast_observe == 1 && f->fused_step && !sweep && B > 0 && n9 > 0 &&
        fused_lds_bytes(g.ncell, B, n9) <= kFusedMaxLds && pub_ok && (!onepass_only || (step_pub_entry_capacity_small(B) > 0 && B <= 32 * 32))) {
      if ((rc = ensure_handoff(f, B, kFastSlots, false))) return rc;
      out->fused = true;
      out->grid = g;
      out->n9 = n9;
      out->tables = f->scan_dev + o_tab;
      out->exact = reinterpret_cast<const double*>(f->scan_dev + o_exact);
      const size_t cs_b = ((size_t)(g.ncell + 1) * 2 + 15) & ~(size_t)15;
      out->order = reinterpret_cast<const unsigned short*>(f->scan_dev + o_tab + cs_b + (size_t)B * 16 + (size_t)n9 * 2);
      return PK_OK;
    }
    if (out && want_fast && !finalize && f->fast_observe == 1 && f->regs_step && f->d.lay.L > kFastMaxL &&
        f->d.lay.L <= kRegsMaxL && B > 0 && n9 > 0 && regs_lds_bytes(g.ncell, B, n9) <= kMaxDynLds && pub_ok &&
        (!onepass_only || (step_pub_entry_capacity(B) > 0 && regs_cand_lds_bytes(f->d.lay.Lp, B) <= kMaxDynLds && B <= 32 * 96))) {
      // the flags, and -- when the two-sweep kernel can take this scan -- eight-slot hand-off lists for the second chance of
      // the particles k_step_regs flags (allocated here, not at the first flagged particle in the middle of a run)
      out->retry = f->regs_retry && observe_sweep_plan(f->d, B).grid > 0;
      if ((rc = ensure_handoff(f, B, out->retry ? kSweepSlots : kFastSlots, out->retry, true))) return rc;
      out->regs = true;
      out->grid = g;
      out->n9 = n9;
      out->tables = f->scan_dev + o_tab;
      out->exact = reinterpret_cast<const double*>(f->scan_dev + o_exact);
      const size_t cs_b = ((size_t)(g.ncell + 1) * 2 + 15) & ~(size_t)15;
      out->order = reinterpret_cast<const unsigned short*>(f->scan_dev + o_tab + cs_b + (size_t)B * 16 + (size_t)n9 * 2);
      return PK_OK;
    }
    if (out && want_fast && !finalize && f->fast_observe == 1 && f->pub_step && f->cand_lists && f->d.lay.L > kRegsMaxL &&
        f->d.lay.L <= kPubBigMaxL && B > 0 && step_pub_big_entry_capacity(B) > 0 && observe_sweep_plan(f->d, B).grid > 0 && pub_ok) {
      // maps beyond the register route: publish / subscribe in two passes (k_step_pub_big); what it flags -- or the whole scan,
      // when a sixteen-entry list overflows or the publish table does not fit LDS -- goes through the eight-slot hand-off
      // and k_observe_sweep, then the general kernels
      if ((rc = ensure_handoff(f, B, kSweepSlots, true, true))) return rc;
      out->big = true;
      out->retry = true;
      out->grid = g;
      out->n9 = n9;
      out->tables = f->scan_dev + o_tab;
      out->exact = reinterpret_cast<const double*>(f->scan_dev + o_exact);
      const size_t cs_b = ((size_t)(g.ncell + 1) * 2 + 15) & ~(size_t)15;
      out->order = reinterpret_cast<const unsigned short*>(f->scan_dev + o_tab + cs_b + (size_t)B * 16 + (size_t)n9 * 2);
      return PK_OK;
    }
    if (want_fast && !onepass_only && !finalize && f->fast_observe && B > 0 &&
        (sweep ? observe_sweep_plan(f->d, B).grid > 0 : observe_fast_lds_bytes(B) <= kMaxDynLds)) {
      // eight hand-off slots per landmark for the large scans (a landmark's colour neighbourhood gets
      // busier with B: at B = 5 000 random colours some landmark of every particle passes 5-7 blobs),
      // four (16-byte entries) otherwise; "fast_observe" = 3 forces eight
      const int slots = (sweep && (B >= 3000 || f->fast_observe == 3)) ? kSweepSlots : kFastSlots;
      if ((rc = ensure_handoff(f, B, slots))) return rc;
      f->fh.slots = slots;
      fh = f->fh;
      fh.n_flagged = ctl_n_flagged(f);
    }
    Span t(f, PK_T_ASSOC);
    CandTable cand;
    if (fh.lmpass && f->cand_lists && f->d.lay.L < 65535) {
      // the hand-off instance tests each landmark against the reference particle's candidate list (k_candidates, once per
      // scan) instead of walking the colour grid; a list that overflows leaves the scan to the walk
      if (!f->cand_dev && (rc = dev_alloc(f, &f->cand_dev, ((size_t)f->d.lay.Lp + kCandSpare) * 3))) return rc;
      // (sixteen entries per list: with several thousand blobs around the robot eight overflow somewhere in every scan)
      launch_summary_partials(f->stream, f->d, f->partial, f->out4);  // the reference pose: the particles' mean
      launch_candidates(f->stream, f->d, B, reinterpret_cast<const double*>(f->scan_dev + o_exact), 0, f->cand_dev, ctl_cand_over(f),
                        nullptr, nullptr, nullptr, 2 * kCandSlots, f->out4);
      cand.rec = f->cand_dev;
      cand.over = ctl_cand_over(f);
      cand.slots = 2 * kCandSlots;
    }
    launch_assoc_grid(f->stream, f->d, B, g, n9, f->scan_dev + o_tab, reinterpret_cast<const double*>(f->scan_dev + o_exact),
                      f->ids_dev, finalize, fh, cand);
    if (out) {
      out->fast = fh.lmpass != nullptr;
      out->exact = reinterpret_cast<const double*>(f->scan_dev + o_exact);
      const size_t cs_b = ((size_t)(g.ncell + 1) * 2 + 15) & ~(size_t)15;
      out->order = reinterpret_cast<const unsigned short*>(f->scan_dev + o_tab + cs_b + (size_t)B * 16 + (size_t)n9 * 2);
    }
    return PK_OK;
  }
  if (assoc_brute_lds_bytes(B) > kMaxDynLds)
    return fail(PK_ERR_UNSUPPORTED,
                "maximum-likelihood association of %d blobs: neither the colour-grid tables nor the brute-force kernel's "
                "%zu bytes of per-blob state fit the workgroup's %zu bytes of LDS", B, assoc_brute_lds_bytes(B), (size_t)kMaxDynLds);
  Span t(f, PK_T_ASSOC);
  launch_assoc_brute(f->stream, f->d, blobs_dev, dir_dev, B, f->ids_dev);
  return PK_OK;
}

// 0: fits the compact layout (block diagonal xy (+) rgb, symmetric); 1: finite but needs the dense layout
// (xy-rgb coupling, or not symmetric); PK_ERR_INVALID: not finite.
int classify_covariance(const double* cov, int l) {
  double scale = 0.0;
  for (int i = 0; i < 25; ++i) {
    if (!std::isfinite(cov[i])) return fail(PK_ERR_INVALID, "landmark %d: covariance not finite", l + 1);
    if (i % 6 == 0) scale = fmax(scale, fabs(cov[i]));
  }
  const double tol = 1e-9 * (scale > 0 ? scale : 1.0);
  for (int i = 0; i < 2; ++i)
    for (int j = 2; j < 5; ++j)
      if (fabs(cov[i * 5 + j]) > 1e-14 * scale || fabs(cov[j * 5 + i]) > 1e-14 * scale) return 1;
  for (int i = 0; i < 5; ++i)
    for (int j = i + 1; j < 5; ++j)
      if (fabs(cov[i * 5 + j] - cov[j * 5 + i]) > tol) return 1;
  return 0;
}

// The compact layout stores Sigma = Pxy (+) C: reject anything else loudly.
int check_block_diagonal(const double* cov, int l) {
  double scale = 0.0;
  for (int i = 0; i < 5; ++i) scale = fmax(scale, fabs(cov[i * 6]));
  if (!(scale >= 0.0) || !std::isfinite(scale))
    return fail(PK_ERR_INVALID, "landmark %d: covariance diagonal is not finite", l + 1);
  const double tol = 1e-9 * (scale > 0 ? scale : 1.0);
  for (int i = 0; i < 2; ++i)
    for (int j = 2; j < 5; ++j)
      if (fabs(cov[i * 5 + j]) > 1e-14 * scale || fabs(cov[j * 5 + i]) > 1e-14 * scale)
        return fail(PK_ERR_UNSUPPORTED,
                    "landmark %d: covariance couples position and colour (entry [%d][%d]); the compact "
                    "device layout holds xy 2x2 (+) rgb 3x3 only",
                    l + 1, i, j);
  for (int i = 0; i < 5; ++i)
    for (int j = i + 1; j < 5; ++j) {
      if (!std::isfinite(cov[i * 5 + j])) return fail(PK_ERR_INVALID, "landmark %d: covariance not finite", l + 1);
      if (fabs(cov[i * 5 + j] - cov[j * 5 + i]) > tol)
        return fail(PK_ERR_UNSUPPORTED, "landmark %d: covariance is not symmetric", l + 1);
    }
  return PK_OK;
}

// Switch the filter's maps between the compact (14-row) and the dense (30-row) layout.  keep: the particles' landmark
// states survive (through the host, in chunks: a rare, slow path -- a coupled Qt set after the map, coupled covariances
// uploaded into a compact filter); otherwise the maps are simply reallocated (pk_upload_map refills them).
int relayout(pk_filter* f, bool dense, bool keep) {
  if (f->dense == dense) return PK_OK;
  int rc;
  DeviceState& d = f->d;
  const MapLayout old = d.lay;
  const MapLayout neu = MapLayout::make(old.L, sizeof(double), dense ? kDenseFields : (int)F_COUNT_FIELDS);
  const int64_t P = d.P;
  const int L = old.L;
  std::vector<double> means, covs;
  std::vector<int32_t> counts;
  if (keep && L > 0) {
    if ((rc = materialise(f))) return rc;
    if (!dense)  // dense -> compact drops what the compact layout cannot hold: only when nothing is coupled
      return fail(PK_ERR_STATE, "relayout: a dense map cannot be folded back into the compact layout");
    try {  // the conversion goes through the host (a rare path): P L 30 doubles -- 48 GB at 100 000 x 2 000
      means.resize((size_t)P * L * 5);
      covs.resize((size_t)P * L * 25);
      counts.resize((size_t)P * L);
    } catch (const std::bad_alloc&) {
      return fail(PK_ERR_NOMEM, "relayout: no host memory to stage %lld x %d landmarks for the layout conversion", (long long)P, L);
    }
    if ((rc = pk_download_landmarks(f, 0, P, means.data(), covs.data(), counts.data()))) return rc;
  }
  PK_HIP(hipStreamSynchronize(f->stream));
  for (int i = 0; i < 2; ++i) {
    if (d.map[i]) (void)hipFree(d.map[i]);
    d.map[i] = nullptr;
    f->device_bytes -= (int64_t)((size_t)P * old.slot_bytes);
  }
  if (f->slot_tmp) (void)hipFree(f->slot_tmp);
  f->slot_tmp = nullptr;
  f->device_bytes -= (int64_t)old.slot_bytes;
  d.lay = neu;
  f->dense = dense;
  d.alt = nullptr;
  for (int i = 0; i < 2; ++i) {
    if ((rc = dev_alloc(f, &d.map[i], (size_t)P * neu.slot_bytes))) return rc;
    PK_HIP(hipMemsetAsync(d.map[i], 0, (size_t)P * neu.slot_bytes, f->stream));
  }
  if ((rc = dev_alloc(f, &f->slot_tmp, neu.slot_bytes))) return rc;
  launch_iota(f->stream, d.src[0], P);
  launch_iota(f->stream, d.src[1], P);
  f->src_identity = true;
  d.mcur = 0;
  if (keep && L > 0) {
    const bool was_loaded = f->map_loaded;
    f->map_loaded = true;
    rc = pk_upload_landmarks(f, 0, P, means.data(), covs.data(), counts.data());
    f->map_loaded = was_loaded;
    if (rc) return rc;
  }
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

}  // namespace

extern "C" {

int pk_abi_version(void) { return PK_ABI_VERSION; }

const char* pk_status_string(int status) {
  switch (status) {
    case PK_OK: return "ok";
    case PK_ERR_INVALID: return "invalid argument";
    case PK_ERR_HIP: return "HIP runtime error";
    case PK_ERR_STATE: return "invalid call order";
    case PK_ERR_UNSUPPORTED: return "unsupported by the device path";
    case PK_ERR_NOMEM: return "out of memory";
    default: return "unknown status";
  }
}

const char* pk_last_error(void) { return g_last_error.c_str(); }

int pk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int pk_create(int64_t P, int32_t L, int32_t device, pk_filter** out) {
  if (!out) return fail(PK_ERR_INVALID, "pk_create: out is NULL");
  *out = nullptr;
  if (P < 1 || P > 2147483647LL) return fail(PK_ERR_INVALID, "pk_create: num_particles %lld out of range", (long long)P);
  if (L < 0 || L > (1 << 24)) return fail(PK_ERR_INVALID, "pk_create: num_landmarks %d out of range", L);
  int ndev = pk_device_count();
  if (ndev <= 0) return fail(PK_ERR_HIP, "pk_create: no HIP device visible (the HIP path has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(PK_ERR_INVALID, "pk_create: device %d of %d", device, ndev);
  pk_filter* f = new (std::nothrow) pk_filter;
  if (!f) return fail(PK_ERR_NOMEM, "pk_create: host allocation failed");
  f->device = device;
  int rc = PK_OK;
  auto bail = [&](int code) {
    pk_destroy(f);
    return code;
  };
  if ((rc = use_device(f))) return bail(rc);
  if (hipStreamCreateWithFlags(&f->own_stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(PK_ERR_HIP, "hipStreamCreate failed"));
  f->stream = f->own_stream;
  DeviceState& d = f->d;
  d.P = P;
  d.lay = MapLayout::make(L, sizeof(double));
  d.cur = 0;
  d.mcur = 0;
  f->nblocks = (P + kScanBlock - 1) / kScanBlock;
  for (int i = 0; i < 2 && !rc; ++i) {
    if (!rc) rc = dev_alloc(f, &d.x[i], (size_t)P);
    if (!rc) rc = dev_alloc(f, &d.y[i], (size_t)P);
    if (!rc) rc = dev_alloc(f, &d.h[i], (size_t)P);
    if (!rc) rc = dev_alloc(f, &d.logw[i], (size_t)P);
    if (!rc) rc = dev_alloc(f, &d.src[i], (size_t)P);
    if (!rc) rc = dev_alloc(f, &d.map[i], (size_t)P * d.lay.slot_bytes);
  }
  if (!rc) rc = dev_alloc(f, &d.immutable, (size_t)d.lay.Lp);
  if (!rc) rc = dev_alloc(f, &f->z_dev, (size_t)P * 3);
  if (!rc) rc = dev_alloc(f, &f->partial, (size_t)4 * 1024);
  if (!rc) rc = dev_alloc(f, &f->gmax, 1);
  if (!rc) rc = dev_alloc(f, &f->clocal, (size_t)P);
  if (!rc) rc = dev_alloc(f, &f->totals, (size_t)f->nblocks);
  if (!rc) rc = dev_alloc(f, &f->offsets, (size_t)f->nblocks);
  if (!rc) rc = dev_alloc(f, &f->sum, 1);
  if (!rc) rc = dev_alloc(f, &f->out4, 4);
  if (!rc) rc = dev_alloc(f, &f->pose_part, (size_t)(4 * motion_pose_blocks(P)));
  if (!rc) rc = dev_alloc(f, &f->anc, (size_t)P);
  if (!rc) rc = dev_alloc(f, &f->slot_tmp, d.lay.slot_bytes);
  if (rc) return bail(rc);
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; ++i) {
    e = hipMemsetAsync(d.x[i], 0, P * sizeof(double), f->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d.y[i], 0, P * sizeof(double), f->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d.h[i], 0, P * sizeof(double), f->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d.logw[i], 0, P * sizeof(double), f->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d.map[i], 0, (size_t)P * d.lay.slot_bytes, f->stream);
  }
  if (e == hipSuccess) e = hipMemsetAsync(d.immutable, 0, d.lay.Lp, f->stream);
  if (e != hipSuccess) return bail(fail(PK_ERR_HIP, "pk_create: memset failed: %s", hipGetErrorString(e)));
  launch_iota(f->stream, d.src[0], P);
  launch_iota(f->stream, d.src[1], P);
  if (hipStreamSynchronize(f->stream) != hipSuccess) return bail(fail(PK_ERR_HIP, "pk_create: sync failed"));
  f->map_loaded = (L == 0);
  *out = f;
  return PK_OK;
}

int pk_destroy(pk_filter* f) {
  if (!f) return PK_OK;
  (void)hipSetDevice(f->device);
  if (f->stream) (void)hipStreamSynchronize(f->stream);
  for (auto& t : f->pending) {
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  for (auto e : f->pool) (void)hipEventDestroy(e);
  DeviceState& d = f->d;
  for (int i = 0; i < 2; ++i) {
    (void)hipFree(d.x[i]);
    (void)hipFree(d.y[i]);
    (void)hipFree(d.h[i]);
    (void)hipFree(d.logw[i]);
    (void)hipFree(d.src[i]);
    (void)hipFree(d.map[i]);
  }
  if (f->scan_dev) (void)hipFree(f->scan_dev);
  for (void* q : {(void*)f->fh.lmpass, (void*)f->fh.bcount, (void*)f->fh.pflag, (void*)f->fh.row_of, (void*)f->sweep_results, (void*)f->cand_dev, (void*)f->bcnt_dev, (void*)f->brec_dev, (void*)f->erec_dev, (void*)f->erec_dev2, (void*)f->binfo_dev, (void*)f->glist_dev, (void*)f->gate4_dev, (void*)f->npass_dev, (void*)f->far_dev, (void*)f->prim_dev, (void*)f->unm_dev})
    if (q) (void)hipFree(q);
  for (int i = 0; i < 2; ++i)
    for (void* q : {(void*)f->grow.hyp[i], (void*)f->grow.cnt[i], (void*)f->grow.slot_id[i]})
      if (q) (void)hipFree(q);
  for (void* q : {(void*)f->g_totals, (void*)f->g_offsets, (void*)f->hi_dev, (void*)f->gl_clocal, (void*)f->gl_totals, (void*)f->gl_offsets, (void*)f->plan_ticket, (void*)f->idx_dev, (void*)f->srcs_dev, (void*)f->rlohi_dev})
    if (q) (void)hipFree(q);
  for (void* q : {(void*)d.logical[0], (void*)d.logical[1], (void*)f->bal.glogw, (void*)f->bal.clocal, (void*)f->bal.totals, (void*)f->bal.offsets,
                  (void*)f->bal.sum, (void*)f->bal.H, (void*)f->bal.cloc, (void*)f->bal.ctot, (void*)f->bal.coff, (void*)f->bal.rel, (void*)f->bal.Hl,
                  (void*)f->bal.alive, (void*)f->bal.bad})
    if (q) (void)hipFree(q);
  void* rest[] = {d.immutable, f->z_dev,  f->ids_dev,
                  f->partial,  f->gmax,   f->clocal,    f->totals,      f->offsets,   f->sum,      f->out4, f->pose_part,
                  f->anc,      f->slot_tmp};
  for (void* p : rest)
    if (p) (void)hipFree(p);
  for (int i = 0; i < pk_filter::kRing; ++i) {
    if (f->stage[i]) (void)hipHostFree(f->stage[i]);
    if (f->stage_done[i]) (void)hipEventDestroy(f->stage_done[i]);
  }
  if (f->retry_seen) (void)hipHostFree(f->retry_seen);
  if (f->own_stream) (void)hipStreamDestroy(f->own_stream);
  delete f;
  return PK_OK;
}

int pk_set_stream(pk_filter* f, void* hip_stream) {
  if (!f) return fail(PK_ERR_INVALID, "pk_set_stream: NULL handle");
  int rc;
  if ((rc = use_device(f))) return rc;
  if ((rc = drain_timings(f))) return rc;
  PK_HIP(hipStreamSynchronize(f->stream));
  f->stream = hip_stream ? (hipStream_t)hip_stream : f->own_stream;
  return PK_OK;
}

int pk_synchronize(pk_filter* f) {
  if (!f) return fail(PK_ERR_INVALID, "pk_synchronize: NULL handle");
  int rc;
  if ((rc = use_device(f))) return rc;
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

int64_t pk_num_particles(const pk_filter* f) { return f ? f->d.P : -1; }
int32_t pk_num_landmarks(const pk_filter* f) { return f ? f->d.lay.L : -1; }
int64_t pk_device_bytes(const pk_filter* f) { return f ? f->device_bytes : -1; }

int pk_set_measurement_noise(pk_filter* f, const double Qt[16]) {
  if (!f || !Qt) return fail(PK_ERR_INVALID, "pk_set_measurement_noise: NULL argument");
  for (int i = 0; i < 16; ++i)
    if (!std::isfinite(Qt[i])) return fail(PK_ERR_INVALID, "Qt is not finite");
  double scale = 0;
  for (int i = 0; i < 4; ++i) scale = fmax(scale, fabs(Qt[i * 5]));
  // Qt = [q00] (+) symmetric 3x3 is what the compact layout and the fast kernels take; anything else (bearing-colour
  // coupling, asymmetry -- the reference accepts any 4x4, prkt_core_v2.py:50-53, :817-818) needs the dense kernels
  bool coupled = false;
  for (int j = 1; j < 4; ++j)
    if (fabs(Qt[j]) > 1e-14 * scale || fabs(Qt[j * 4]) > 1e-14 * scale) coupled = true;
  for (int i = 1; i < 4; ++i)
    for (int j = i + 1; j < 4; ++j)
      if (fabs(Qt[i * 4 + j] - Qt[j * 4 + i]) > 1e-9 * (scale > 0 ? scale : 1.0)) coupled = true;
  int rc;
  if ((rc = use_device(f))) return rc;
  if (coupled && !f->dense && (rc = relayout(f, true, f->map_loaded))) return rc;
  for (int i = 0; i < 16; ++i) f->qt16[i] = Qt[i];
  f->qt_dense = coupled;
  f->qt = NoiseD{Qt[0], Qt[5], 0.5 * (Qt[6] + Qt[9]), 0.5 * (Qt[7] + Qt[13]), Qt[10], 0.5 * (Qt[11] + Qt[14]), Qt[15]};
  return PK_OK;
}

int pk_upload_map(pk_filter* f, const double* means, const double* covs, const uint8_t* immutable) {
  if (!f) return fail(PK_ERR_INVALID, "pk_upload_map: NULL handle");
  const MapLayout& lay = f->d.lay;
  const int L = lay.L;
  if (L > 0 && (!means || !covs)) return fail(PK_ERR_INVALID, "pk_upload_map: NULL means/covs");
  int rc;
  if ((rc = use_device(f))) return rc;
  bool need_dense = f->qt_dense;
  for (int l = 0; l < L; ++l) {
    for (int i = 0; i < 5; ++i)
      if (!std::isfinite(means[l * 5 + i])) return fail(PK_ERR_INVALID, "landmark %d: mean is not finite", l + 1);
    const int c = classify_covariance(covs + (size_t)l * 25, l);
    if (c < 0) return c;
    need_dense |= c == 1;  // xy-rgb coupling or asymmetry: the whole filter takes the dense layout and kernels
  }
  if ((rc = relayout(f, need_dense, false))) return rc;  // (lay refers to f->d.lay: the new layout from here on)
  std::vector<unsigned char> slot(lay.slot_bytes, 0);
  std::vector<unsigned char> imm((size_t)lay.Lp, 0);
  for (int l = 0; l < L; ++l) {
    pack_landmark(lay, slot.data(), l, means + (size_t)l * 5, covs + (size_t)l * 25);
    imm[l] = immutable ? (immutable[l] != 0) : 0;
  }
  PK_HIP(hipMemcpyAsync(f->slot_tmp, slot.data(), lay.slot_bytes, hipMemcpyHostToDevice, f->stream));
  PK_HIP(hipMemcpyAsync(f->d.immutable, imm.data(), imm.size(), hipMemcpyHostToDevice, f->stream));
  launch_broadcast_slot(f->stream, f->d, f->slot_tmp);
  PK_HIP(hipStreamSynchronize(f->stream));  // host staging buffers die here
  f->src_identity = true;
  f->map_loaded = true;
  return PK_OK;
}

int pk_upload_poses(pk_filter* f, const double* xyhw) {
  if (f) f->pose_part_ok = false;  // (the poses change: the motion launch's pose sums are no longer theirs)
  if (!f || !xyhw) return fail(PK_ERR_INVALID, "pk_upload_poses: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  const int64_t P = f->d.P;
  std::vector<double> soa((size_t)P * 4);
  for (int64_t i = 0; i < P; ++i) {
    soa[i] = xyhw[4 * i];
    soa[P + i] = xyhw[4 * i + 1];
    soa[2 * P + i] = xyhw[4 * i + 2];
    double w = xyhw[4 * i + 3];
    if (!(w >= 0.0) || !std::isfinite(w)) return fail(PK_ERR_INVALID, "pk_upload_poses: weight of particle %lld is negative, infinite or NaN", (long long)i);
    soa[3 * P + i] = std::log(w);
  }
  const int c = f->d.cur;
  PK_HIP(hipMemcpyAsync(f->d.x[c], soa.data(), P * 8, hipMemcpyHostToDevice, f->stream));
  PK_HIP(hipMemcpyAsync(f->d.y[c], soa.data() + P, P * 8, hipMemcpyHostToDevice, f->stream));
  PK_HIP(hipMemcpyAsync(f->d.h[c], soa.data() + 2 * P, P * 8, hipMemcpyHostToDevice, f->stream));
  PK_HIP(hipMemcpyAsync(f->d.logw[c], soa.data() + 3 * P, P * 8, hipMemcpyHostToDevice, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  f->gmax_fused = false;
  return PK_OK;
}

/* One particle's pose and weight (FastSLAM.particles[i] = p, prkt_core_v2.py:162): the other particles' log-weights are not
 * touched -- a round trip of ALL poses through their linear weights (pk_download_poses / pk_upload_poses) loses every weight that
 * underflows as exp(log w). */
int pk_upload_pose(pk_filter* f, int64_t p, const double xyhw[4]) {
  if (!f || !xyhw) return fail(PK_ERR_INVALID, "pk_upload_pose: NULL argument");
  if (p < 0 || p >= f->d.P) return fail(PK_ERR_INVALID, "pk_upload_pose: particle %lld of %lld", (long long)p, (long long)f->d.P);
  for (int i = 0; i < 3; ++i)
    if (!std::isfinite(xyhw[i])) return fail(PK_ERR_INVALID, "pk_upload_pose: pose component %d is not finite", i);
  // (+inf passed "w >= 0", and log(inf) turns the log-domain scan into NaN / all-zero weights)
  if (!(xyhw[3] >= 0.0) || !std::isfinite(xyhw[3])) return fail(PK_ERR_INVALID, "pk_upload_pose: the weight is negative, infinite or NaN");
  int rc;
  if ((rc = use_device(f))) return rc;
  f->pose_part_ok = false;
  const double v4[4] = {xyhw[0], xyhw[1], xyhw[2], std::log(xyhw[3])};
  const int c = f->d.cur;
  double* const dst[4] = {f->d.x[c] + p, f->d.y[c] + p, f->d.h[c] + p, f->d.logw[c] + p};
  for (int i = 0; i < 4; ++i) PK_HIP(hipMemcpyAsync(dst[i], &v4[i], 8, hipMemcpyHostToDevice, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  f->gmax_fused = false;
  return PK_OK;
}

int pk_download_poses(pk_filter* f, double* xyhw) {
  if (!f || !xyhw) return fail(PK_ERR_INVALID, "pk_download_poses: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  const int64_t P = f->d.P;
  std::vector<double> soa((size_t)P * 4);
  const int c = f->d.cur;
  PK_HIP(hipMemcpyAsync(soa.data(), f->d.x[c], P * 8, hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipMemcpyAsync(soa.data() + P, f->d.y[c], P * 8, hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipMemcpyAsync(soa.data() + 2 * P, f->d.h[c], P * 8, hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipMemcpyAsync(soa.data() + 3 * P, f->d.logw[c], P * 8, hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  for (int64_t i = 0; i < P; ++i) {
    xyhw[4 * i] = soa[i];
    xyhw[4 * i + 1] = soa[P + i];
    xyhw[4 * i + 2] = soa[2 * P + i];
    xyhw[4 * i + 3] = std::exp(soa[3 * P + i]);
  }
  return PK_OK;
}

int pk_download_log_weights(pk_filter* f, double* logw) {
  if (!f || !logw) return fail(PK_ERR_INVALID, "pk_download_log_weights: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  PK_HIP(hipMemcpyAsync(logw, f->d.logw[f->d.cur], (size_t)f->d.P * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

int pk_download_landmarks(pk_filter* f, int64_t p0, int64_t p1, double* means, double* covs, int32_t* counts) {
  if (!f) return fail(PK_ERR_INVALID, "pk_download_landmarks: NULL handle");
  if (p0 < 0 || p1 < p0 || p1 > f->d.P) return fail(PK_ERR_INVALID, "pk_download_landmarks: bad particle range");
  if (!f->map_loaded) return fail(PK_ERR_STATE, "pk_download_landmarks: no map uploaded");
  int rc;
  if ((rc = use_device(f))) return rc;
  if ((rc = materialise(f))) return rc;
  const MapLayout& lay = f->d.lay;
  const int L = lay.L;
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(64u << 20) / (int64_t)lay.slot_bytes);
  std::vector<unsigned char> host((size_t)std::min<int64_t>(chunk, p1 - p0) * lay.slot_bytes);
  for (int64_t q0 = p0; q0 < p1; q0 += chunk) {
    int64_t q1 = std::min(p1, q0 + chunk);
    PK_HIP(hipMemcpyAsync(host.data(), f->d.map[f->d.mcur] + (size_t)q0 * lay.slot_bytes,
                          (size_t)(q1 - q0) * lay.slot_bytes, hipMemcpyDeviceToHost, f->stream));
    PK_HIP(hipStreamSynchronize(f->stream));
    for (int64_t q = q0; q < q1; ++q) {
      const unsigned char* slot = host.data() + (size_t)(q - q0) * lay.slot_bytes;
      const int32_t* cnt = reinterpret_cast<const int32_t*>(slot + lay.count_off);
      for (int l = 0; l < L; ++l) {
        size_t o = (size_t)(q - p0) * L + l;
        unpack_landmark(lay, slot, l, means ? means + o * 5 : nullptr, covs ? covs + o * 25 : nullptr);
        if (counts) counts[o] = cnt[l];
      }
    }
  }
  return PK_OK;
}

int pk_upload_landmarks(pk_filter* f, int64_t p0, int64_t p1, const double* means, const double* covs,
                        const int32_t* counts) {
  if (!f) return fail(PK_ERR_INVALID, "pk_upload_landmarks: NULL handle");
  if (p0 < 0 || p1 < p0 || p1 > f->d.P) return fail(PK_ERR_INVALID, "pk_upload_landmarks: bad particle range");
  if (!f->map_loaded) return fail(PK_ERR_STATE, "pk_upload_landmarks: no map uploaded");
  int rc;
  if ((rc = use_device(f))) return rc;
  if ((rc = materialise(f))) return rc;
  const int L = f->d.lay.L;
  // finite means and covariances whatever the layout (as pk_upload_map checks them); the layout decision only while compact
  if (means)
    for (int64_t o = 0; o < (p1 - p0) * L * 5; ++o)
      if (!std::isfinite(means[o])) return fail(PK_ERR_INVALID, "pk_upload_landmarks: mean of landmark %d not finite", (int)((o / 5) % L) + 1);
  if (covs) {
    bool need_dense = false;
    for (int64_t o = 0; o < (p1 - p0) * L; ++o) {
      const int c = classify_covariance(covs + (size_t)o * 25, (int)(o % L));
      if (c < 0) return c;
      need_dense |= c == 1;
    }
    if (need_dense && !f->dense && (rc = relayout(f, true, true))) return rc;
  }
  const MapLayout& lay = f->d.lay;  // (after the conversion)
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(64u << 20) / (int64_t)lay.slot_bytes);
  std::vector<unsigned char> host;
  try {
    host.resize((size_t)std::min<int64_t>(chunk, p1 - p0) * lay.slot_bytes);
  } catch (const std::bad_alloc&) {
    return fail(PK_ERR_NOMEM, "pk_upload_landmarks: no host memory for the staging buffer");
  }
  for (int64_t q0 = p0; q0 < p1; q0 += chunk) {
    int64_t q1 = std::min(p1, q0 + chunk);
    unsigned char* dev = f->d.map[f->d.mcur] + (size_t)q0 * lay.slot_bytes;
    PK_HIP(hipMemcpyAsync(host.data(), dev, (size_t)(q1 - q0) * lay.slot_bytes, hipMemcpyDeviceToHost, f->stream));
    PK_HIP(hipStreamSynchronize(f->stream));
    for (int64_t q = q0; q < q1; ++q) {
      unsigned char* slot = host.data() + (size_t)(q - q0) * lay.slot_bytes;
      int32_t* cnt = reinterpret_cast<int32_t*>(slot + lay.count_off);
      for (int l = 0; l < L; ++l) {
        size_t o = (size_t)(q - p0) * L + l;
        double m[5], c[25];
        unpack_landmark(lay, slot, l, m, c);
        pack_landmark(lay, slot, l, means ? means + o * 5 : m, covs ? covs + o * 25 : c);
        if (counts) cnt[l] = counts[o];
      }
    }
    PK_HIP(hipMemcpyAsync(dev, host.data(), (size_t)(q1 - q0) * lay.slot_bytes, hipMemcpyHostToDevice, f->stream));
    PK_HIP(hipStreamSynchronize(f->stream));
  }
  return PK_OK;
}

int pk_reset_weights(pk_filter* f) {
  if (!f) return fail(PK_ERR_INVALID, "pk_reset_weights: NULL handle");
  int rc;
  if ((rc = use_device(f))) return rc;
  launch_reset_weights(f->stream, f->d);
  f->gmax_fused = false;
  PK_LAUNCH_CHECK("pk_reset_weights");
  return PK_OK;
}

// bytes of a staged ML scan block that the device needs: ctl | blobs + directions | exact records | [tables]
static size_t staged_upload_bytes(const pk_filter::Staged& sg) {
  const size_t o_exact = kCtlBytes + (size_t)sg.B * 6 * sizeof(double);
  const size_t o_tab = o_exact + (size_t)sg.B * 6 * sizeof(double);
  return sg.use_grid ? o_tab + sg.tab_bytes : o_exact;
}

int pk_motion(pk_filter* f, double v, double w, double dt, const double* z, uint64_t seed, uint64_t draw) {
  if (!f) return fail(PK_ERR_INVALID, "pk_motion: NULL handle");
  if (!std::isfinite(v) || !std::isfinite(w) || !std::isfinite(dt)) return fail(PK_ERR_INVALID, "pk_motion: non-finite control");
  int rc;
  if ((rc = use_device(f))) return rc;
  const double* zd = nullptr;
  if (z) {
    // Parity mode: the caller's (pageable) buffer must be consumed before we return.
    PK_HIP(hipMemcpyAsync(f->z_dev, z, (size_t)f->d.P * 3 * sizeof(double), hipMemcpyHostToDevice, f->stream));
    PK_HIP(hipStreamSynchronize(f->stream));
    zd = f->z_dev;
  }
  Span t(f, PK_T_MOTION);
  // a scan staged by pk_stage_scan and not uploaded yet rides in extra workgroups of this launch
  // (what pk_step does; the sharded step stages the next scan before it calls pk_motion)
  pk_filter::Staged& sg = f->staged;
  if (!z && sg.valid && !sg.uploaded && f->upload_kernel) {
    void* dev_view = nullptr;
    if (hipHostGetDevicePointer(&dev_view, sg.st, 0) == hipSuccess && dev_view) {
      launch_motion(f->stream, f->d, v, w, dt, nullptr, seed, draw, 0, f->scan_dev, dev_view, staged_upload_bytes(sg), f->pose_part);
      f->pose_part_ok = true;
      sg.uploaded = true;
      f->gmax_fused = false;  // the block's control words (running weight maximum) were just overwritten
      if ((rc = note_upload(f, sg.slot))) return rc;
      PK_LAUNCH_CHECK("pk_motion");
      return PK_OK;
    }
    (void)hipGetLastError();
  }
  launch_motion(f->stream, f->d, v, w, dt, zd, seed, draw, 0, nullptr, nullptr, 0, f->pose_part);
  f->pose_part_ok = true;
  PK_LAUNCH_CHECK("pk_motion");
  return PK_OK;
}

// The dense path (maps with xy-rgb coupling or a coupled Qt): one general kernel does association (or takes the ids),
// EKF updates and weights.  update = false: association only.
static int dense_observe(pk_filter* f, const double* blobs, int32_t B, const int32_t* ids, int32_t* ids_out, bool reset,
                         bool update) {
  int rc;
  const MapLayout& lay = f->d.lay;
  f->staged.valid = false;
  if (dense_lds_bytes(lay.Lp, B) > kMaxDynLds)
    return fail(PK_ERR_UNSUPPORTED, "dense observe: %d landmarks and %d blobs need %zu bytes of LDS per particle, the workgroup has %zu",
                lay.Lp, B, dense_lds_bytes(lay.Lp, B), (size_t)kMaxDynLds);
  // block: ctl | blobs (4B doubles) | dir (2B doubles) | ids (B int32)
  const size_t o_blobs = kCtlBytes;
  const size_t o_dir = o_blobs + (size_t)B * 4 * sizeof(double);
  const size_t o_ids = o_dir + (size_t)B * 2 * sizeof(double);
  const size_t total = (o_ids + (size_t)B * sizeof(int32_t) + 15) & ~(size_t)15;
  unsigned char* st = nullptr;
  int slot = 0;
  if ((rc = take_stage(f, total, &st, &slot))) return rc;
  if ((rc = ensure_scan_capacity(f, total))) return rc;
  if (ids_out && !ids && (rc = ensure_ids_capacity(f, B))) return rc;
  memset(st, 0, kCtlBytes);
  if (B > 0) {
    memmove(st + o_blobs, blobs, (size_t)B * 4 * sizeof(double));
    blob_directions(reinterpret_cast<const double*>(st + o_blobs), B, reinterpret_cast<double*>(st + o_dir));
    if (ids) memcpy(st + o_ids, ids, (size_t)B * sizeof(int32_t));
  }
  if ((rc = upload_scan(f, st, total))) return rc;
  if ((rc = note_upload(f, slot))) return rc;
  ObserveExtras ex;
  ex.reset = reset;
  ex.gmax_key = ctl_gmax_key(f);
  {
    Span t(f, update ? PK_T_OBSERVE : PK_T_ASSOC);
    launch_observe_dense(f->stream, f->d, reinterpret_cast<const double*>(f->scan_dev + o_blobs),
                         reinterpret_cast<const double*>(f->scan_dev + o_dir), B,
                         ids ? reinterpret_cast<const int32_t*>(f->scan_dev + o_ids) : nullptr,
                         (ids_out && !ids) ? f->ids_dev : nullptr, f->qt16, update, ex);
  }
  PK_LAUNCH_CHECK("pk_observe (dense)");
  if (update) {
    f->src_identity = true;
    f->d.alt = nullptr;  // every map slot was just rewritten from its source: the receive buffer of the last exchange is free
    f->gmax_fused = true;
    f->route = PK_ROUTE_DENSE;
  } else {
    f->gmax_fused = false;  // the control words were overwritten by this upload
  }
  if (ids_out && B > 0) {
    if (ids) {
      for (int64_t p = 0; p < f->d.P; ++p) memcpy(ids_out + (size_t)p * B, ids, (size_t)B * 4);
    } else {
      PK_HIP(hipMemcpyAsync(ids_out, f->ids_dev, (size_t)f->d.P * B * 4, hipMemcpyDeviceToHost, f->stream));
      PK_HIP(hipStreamSynchronize(f->stream));
    }
  }
  return PK_OK;
}

// ---- the one-pass routes (k_step_fused / k_step_regs) in three pieces, so that the sharded filter can run the middle one
// on a part of the particles while the rest are still on the wire (pk_observe_staged_range) -------------------------------
// 1. the reference particle's candidate lists (register route), timed with the association
static int ensure_inverse_lists(pk_filter* f, int B, int slots = kCandSlots) {
  int rc;
  B = B * (slots / kCandSlots);  // (capacity in units of eight-entry lists: sixteen-entry lists take two)
  if (B > f->bcand_cap) {
    PK_HIP(hipStreamSynchronize(f->stream));
    for (void* q : {(void*)f->bcnt_dev, (void*)f->brec_dev, (void*)f->binfo_dev, (void*)f->glist_dev, (void*)f->gate4_dev})
      if (q) (void)hipFree(q);
    f->gate4_dev = nullptr;
    f->bcnt_dev = nullptr;
    f->brec_dev = nullptr;
    f->binfo_dev = nullptr;
    f->glist_dev = nullptr;
    f->bcand_cap = 0;
    const int64_t cap = (int64_t)B + B / 4 + 64;
    if ((rc = dev_alloc(f, &f->bcnt_dev, (size_t)cap))) return rc;
    if ((rc = dev_alloc(f, &f->brec_dev, (size_t)cap))) return rc;
    if ((rc = dev_alloc(f, &f->binfo_dev, (size_t)cap))) return rc;
    if ((rc = dev_alloc(f, &f->glist_dev, 2 * (size_t)cap + 1 + 256 + 16))) return rc;
    if ((rc = dev_alloc(f, &f->gate4_dev, (size_t)cap))) return rc;
    // (empty inverse lists: k_candidates appends to them, k_cand_entries empties them again behind its last read)
    PK_HIP(hipMemsetAsync(f->bcnt_dev, 0, (size_t)cap * sizeof(unsigned), f->stream));
    PK_HIP(hipMemsetAsync(f->brec_dev, 0xFF, (size_t)cap * sizeof(uint4), f->stream));
    f->bcand_cap = cap;
  }
  return PK_OK;
}
// what k_step_pub_duo has room for at this scan size (all zero: the instance is off, k_step_pub_big takes every scan)
static DuoLimits duo_limits(const pk_filter* f, int B) {
  DuoLimits d;
  if (!f->duo_on) return d;
  step_pub_duo_limits(B, f->d.lay.Lp, f->duo_on, &d);
  if (f->pub_entry_limit > 0 && d.ecap > f->pub_entry_limit) d.ecap = f->pub_entry_limit;
  d.park_limit = f->duo_park_limit;
  return d;
}
// ref: the particle whose MAP the candidate lists are made from -- particle 0, or in a split step the first particle of the range
// that has been filled already (the slots at either end still hold the old generation then)
// Maps of at most 512 landmarks: the publish / subscribe instance (k_step_pub<1, 256> on candidate lists) or k_step_fused?  The former's
// kernel is the faster one (11 % at 10 000 x 500, 18 % at 100 000 x 256) and costs two per-scan kernels and a flag launch (27 us) whatever
// the number of particles.  Measured over P x L (profiles/r06/pub_small_sweep*.log): the whole step loses with it at 5 000 x 500 (+14 %)
// and 8 000 x 500 (+3 %), ties at 10 000 x 500 (+0.6 %, -0.4 % at 12 000) and wins from there on (-5 % at 16 000 x 500, -8 % at
// 100 000 x 500, -15 % at 100 000 x 256); at 10 000 x 256 it is +1.5 %, at 10 000 x 128 +7 %, at 100 000 x 128 -7.5 %.  The gain grows
// with P x L, the cost does not: the publish / subscribe instance from 5e6 particle.landmarks on -- BASELINE configs[1], 10 000 x 500,
// is the tie.  Below 128 landmarks nothing was measured: k_step_fused.
constexpr int64_t kPubSmallAutoWork = 5000000;
constexpr int kPubSmallAutoLandmarks = 128;
static bool pub_small_now(const pk_filter* f) {
  if (f->grow_on) return true;  // (growing maps: only the publish / subscribe kernels leave the unmatched blobs' bit rows)
  if (f->pub_small >= 0) return f->pub_small != 0;
  return f->d.P * (int64_t)f->d.lay.L >= kPubSmallAutoWork && f->d.lay.L >= kPubSmallAutoLandmarks;
}
// whole: the launch that follows covers every particle -- a scan the publish / subscribe kernel stands back from and nobody else takes
// then flags the particles in k_cand_entries itself (f->flag_folded) instead of a k_flag_range_if launch of its own (3 us a step)
static int onepass_prepare(pk_filter* f, const AssocLaunch& al, int B, CandTable* cand, int64_t ref = 0, bool whole = false) {
  int rc;
  f->pub_ecap = 0;
  f->flag_folded = false;
  // the register route; with "pub_small" (on from 5e6 particle.landmarks: pub_small_now) also the L <= 512 route through the publish /
  // subscribe instance of three 256-lane workgroups per CU
  const bool small_pub = al.fused && pub_small_now(f) && f->pub_step && f->cand_lists && step_pub_entry_capacity_small(B) > 0;
  if (al.big) {  // sixteen-entry lists both ways and the publish table's layout
    if (!f->cand_dev && (rc = dev_alloc(f, &f->cand_dev, ((size_t)f->d.lay.Lp + kCandSpare) * 3))) return rc;
    if (!f->npass_dev && (rc = dev_alloc(f, &f->npass_dev, (size_t)f->d.lay.Lp + kCandSpare))) return rc;
    if (!f->erec_dev2 && (rc = dev_alloc(f, &f->erec_dev2, ((size_t)f->d.lay.Lp + kCandSpare) * 2))) return rc;
    if ((rc = ensure_inverse_lists(f, B, 2 * kCandSlots))) return rc;
    if (!f->far_dev && (rc = dev_alloc(f, &f->far_dev, ((size_t)f->d.lay.Lp + kCandSpare) * 3))) return rc;
    if (!f->prim_dev && (rc = dev_alloc(f, &f->prim_dev, prim_table_uint4(f->d.lay.Lp)))) return rc;
    int ecap = step_pub_big_entry_capacity(B);
    if (f->pub_entry_limit > 0 && ecap > f->pub_entry_limit) ecap = f->pub_entry_limit;
    Span t(f, PK_T_ASSOC);
    uint4* far = f->far_prune ? f->far_dev : nullptr;
    // the reference pose: the particles' mean -- from the sums the motion launch left, or (poses touched since) two launches
    const double* part = f->pose_part_ok ? f->pose_part : nullptr;
    if (!part) launch_summary_partials(f->stream, f->d, f->partial, f->out4);
    launch_candidates(f->stream, f->d, B, al.exact, ref, f->cand_dev, ctl_cand_over(f), f->bcnt_dev, f->brec_dev, ctl_n_stray(f),
                      2 * kCandSlots, f->out4, f->npass_dev, far, part);
    launch_cand_entries(f->stream, f->d, B, f->cand_dev, f->erec_dev2, f->bcnt_dev, f->brec_dev, f->binfo_dev, f->glist_dev, ctl_cand_over(f),
                        ctl_skip_pub(f), ctl_skip_cand(f), ecap, 2 * kCandSlots, al.exact, f->gate4_dev, f->npass_dev, far != nullptr,
                        ctl_pub_stats(f), ctl_skip_duo(f), ctl_skip_big(f), duo_limits(f, B), f->prim_dev, whole ? f->fh.pflag : nullptr,
                        whole ? ctl_n_flagged(f) : nullptr);
    f->flag_folded = whole;
    cand->rec = f->cand_dev;
    cand->far = far;
    cand->over = ctl_cand_over(f);
    cand->slots = 2 * kCandSlots;
    f->pub_ecap = ecap;
    return PK_OK;
  }
  if ((al.regs && f->cand_lists && regs_cand_lds_bytes(f->d.lay.Lp, B) <= kMaxDynLds) || small_pub) {
    if (!f->cand_dev && (rc = dev_alloc(f, &f->cand_dev, ((size_t)f->d.lay.Lp + kCandSpare) * 3))) return rc;
    int ecap = !f->pub_step ? 0 : al.fused ? step_pub_entry_capacity_small(B) : step_pub_entry_capacity(B);
    if (f->pub_entry_limit > 0 && ecap > f->pub_entry_limit) ecap = f->pub_entry_limit;
    if (ecap > 0) {
      if (!f->erec_dev && (rc = dev_alloc(f, &f->erec_dev, (size_t)f->d.lay.Lp + kCandSpare))) return rc;
      if (!f->npass_dev && (rc = dev_alloc(f, &f->npass_dev, (size_t)f->d.lay.Lp + kCandSpare))) return rc;
      if (!f->far_dev && (rc = dev_alloc(f, &f->far_dev, ((size_t)f->d.lay.Lp + kCandSpare) * 3))) return rc;
      if ((rc = ensure_inverse_lists(f, B))) return rc;
    }
    Span t(f, PK_T_ASSOC);
    const double* part = f->pose_part_ok ? f->pose_part : nullptr;  // (as above)
    if (!part) launch_summary_partials(f->stream, f->d, f->partial, f->out4);
    cand->rec = f->cand_dev;
    cand->over = ctl_cand_over(f);
    if (ecap > 0) {  // candidate lists both ways, and the publish table's layout
      uint4* far = f->far_prune ? f->far_dev : nullptr;
      launch_candidates(f->stream, f->d, B, al.exact, ref, f->cand_dev, ctl_cand_over(f), f->bcnt_dev, f->brec_dev, ctl_n_stray(f),
                        kCandSlots, f->out4, f->npass_dev, far, part);
      // (the flags are folded where no stand-by kernel takes a scan k_step_pub leaves: pruned lists, growing maps, maps of at most 512 landmarks)
      const bool fold = whole && (far != nullptr || f->grow_on || al.fused);
      launch_cand_entries(f->stream, f->d, B, f->cand_dev, f->erec_dev, f->bcnt_dev, f->brec_dev, f->binfo_dev, f->glist_dev, ctl_cand_over(f),
                          ctl_skip_pub(f), ctl_skip_cand(f), ecap, kCandSlots, nullptr, nullptr, f->npass_dev, far != nullptr, ctl_pub_stats(f),
                          nullptr, nullptr, DuoLimits(), nullptr, fold ? f->fh.pflag : nullptr, fold ? ctl_n_flagged(f) : nullptr);
      f->flag_folded = fold;
      cand->far = far;
      cand->skip_cand = ctl_skip_cand(f);
      f->pub_ecap = ecap;
    } else {
      launch_candidates(f->stream, f->d, B, al.exact, ref, f->cand_dev, ctl_cand_over(f), nullptr, nullptr, nullptr, kCandSlots, f->out4, nullptr, nullptr, part);
    }
  }
  return PK_OK;
}
// 2. the one-pass kernel on the particles [p0, p1) (the fused kernel: the whole range only)
static int onepass_launch(pk_filter* f, const AssocLaunch& al, int B, const ObserveExtras& ex, const CandTable& cand, int64_t p0,
                          int64_t p1, int reserve_cus = 0) {
  FastHandoff fh = f->fh;
  fh.n_flagged = ctl_n_flagged(f);
  fh.flags_only = true;
  Span t(f, PK_T_OBSERVE);
  ObserveExtras e1 = ex;
  e1.flip = false;
  if (al.big) {
    // the scan goes to ONE of the two instances (k_cand_entries decided which: the two-workgroups-per-CU instance when its share of
    // LDS holds the scan's publish table); both are launched, one returns at once
    const DuoLimits duo = duo_limits(f, B);
    if (duo.tbytes > 0)
      launch_step_pub_duo(f->stream, f->d, B, al.exact, al.order, fh, f->qt, e1, cand, f->erec_dev2, f->glist_dev, ctl_skip_duo(f), ctl_pub_stats(f), duo,
                          f->gate4_dev, f->prim_dev, p0, p1, reserve_cus);
    launch_step_pub_big(f->stream, f->d, B, al.exact, al.order, fh, f->qt, e1, cand, f->erec_dev2, f->glist_dev, ctl_skip_big(f), f->pub_ecap, f->gate4_dev,
                        p0, p1, reserve_cus, f->prim_dev, ctl_pub_stats(f));
    // a scan the kernel stood back from (a list overflowed, the table did not fit): every particle to the fall-back kernels
    if (!f->flag_folded) launch_flag_range_if(f->stream, ctl_skip_pub(f), fh.pflag, fh.n_flagged, p0, p1);
  } else if (al.regs) {
    if (f->pub_ecap > 0 && cand.rec)
      launch_step_pub(f->stream, f->d, B, al.exact, al.order, fh, f->qt, e1, cand, f->erec_dev, f->glist_dev, ctl_skip_pub(f), f->pub_ecap,
                      p0, p1, reserve_cus);
    // pruned lists (cand.far): k_step_regs' candidate-list instance never takes them (k_cand_entries: skip_cand) -- it is not even
    // launched then (5 us a step for a kernel that returns at once) --, so a scan the publish / subscribe kernel stood back from goes
    // to the fall-back kernels as a whole
    if ((cand.far || f->grow_on) && f->pub_ecap > 0) {  // (growing maps: only the publish / subscribe kernels leave the unmatched blobs' bit rows)
      if (!f->flag_folded) launch_flag_range_if(f->stream, ctl_skip_pub(f), fh.pflag, fh.n_flagged, p0, p1);
    } else
      launch_step_regs(f->stream, f->d, B, al.grid, al.n9, al.tables, al.exact, al.order, fh, f->qt, e1, f->regs_warm, cand, p0, p1,
                       reserve_cus);
  } else {
    if (f->pub_ecap > 0 && cand.rec)
      launch_step_pub(f->stream, f->d, B, al.exact, al.order, fh, f->qt, e1, cand, f->erec_dev, f->glist_dev, ctl_skip_pub(f), f->pub_ecap,
                      p0, p1, reserve_cus);
    // (round 6: no k_step_fused stand-by behind the publish / subscribe instance -- 10 000 workgroups that return at once were 5.7 us of a
    // 270-us step; a scan that kernel stands back from -- its table does not fit a third of a CU's LDS: a few hundred entries do -- goes
    // to the general kernels as a whole.  Growing maps need it that way: only the publish / subscribe kernels leave the unmatched blobs' rows)
    if (f->pub_ecap > 0 && cand.rec) {
      if (!f->flag_folded) launch_flag_range_if(f->stream, ctl_skip_pub(f), fh.pflag, fh.n_flagged, p0, p1);
    } else
      launch_step_fused(f->stream, f->d, B, al.grid, al.n9, al.tables, al.exact, al.order, fh, f->qt, e1,
                        (f->pub_ecap > 0 && cand.rec) ? ctl_skip_pub(f) : nullptr);
  }
  return PK_OK;
}
// 3. what the one-pass kernel flagged, over all particles: second chance, then the general kernels (which swap the map buffers)
static int onepass_finish(pk_filter* f, const AssocLaunch& al, int B, const ObserveExtras& ex, const CandTable& cand) {
  int rc;
  FastHandoff fh = f->fh;
  fh.n_flagged = ctl_n_flagged(f);
  fh.flags_only = true;
  if ((al.regs || al.big) && al.retry) {
    // second chance for what k_step_regs flagged (some landmark passes more than its four register slots -- 2 us per
    // particle in the general kernels, and up to 9 % of the particles at some poses of the bench's trajectory): the
    // hand-off instance with eight slots and k_observe_sweep, both on the flagged particles only (timed with the other
    // fallbacks in the association slot: the observe slot holds the one-pass kernel alone, one span per launch)
    Span t(f, PK_T_ASSOC);
    const SweepPlan plan = observe_sweep_plan(f->d, B);
    const size_t need = (size_t)plan.grid * plan.results_per_wg;
    if (need > f->sweep_cap) {
      PK_HIP(hipStreamSynchronize(f->stream));
      if (f->sweep_results) (void)hipFree(f->sweep_results);
      f->sweep_results = nullptr;
      f->sweep_cap = 0;
      if ((rc = dev_alloc(f, &f->sweep_results, need))) return rc;
      f->sweep_cap = need;
    }
    FastHandoff fr = f->fh;
    fr.slots = kSweepSlots;
    fr.retry = true;
    fr.n_flagged = ctl_n_flagged(f);
    fr.row_next = ctl_retry_rows(f);
    fr.row_cap = retry_rows(f);
    // (pruned candidate lists are the publish / subscribe kernels' alone: the hand-off instance walks the colour grid then)
    launch_assoc_grid(f->stream, f->d, B, al.grid, al.n9, al.tables, al.exact, f->ids_dev, false, fr, cand.far ? CandTable{} : cand);
    ObserveExtras e3 = ex;
    e3.flip = false;
    e3.sweep_only_value = 2;
    e3.n_flagged = ctl_n_flagged(f);
    launch_observe_sweep(f->stream, f->d, B, al.exact, al.order, fr, f->qt, e3, plan, f->sweep_results);
    if (!f->retry_seen && hipHostMalloc((void**)&f->retry_seen, 64, hipHostMallocDefault) == hipSuccess) *f->retry_seen = 0u;
    if (f->retry_seen) (void)hipMemcpyAsync(f->retry_seen, ctl_retry_rows(f), sizeof(unsigned), hipMemcpyDeviceToHost, f->stream);
    (void)hipGetLastError();
  }
  // the particles still flagged (a landmark passing more blobs than any slot count): general kernels, both timed in the
  // association slot
  Span t(f, PK_T_ASSOC);
  launch_assoc_grid(f->stream, f->d, B, al.grid, al.n9, al.tables, al.exact, f->ids_dev, false, fh);
  ObserveExtras e2 = ex;
  e2.only_flagged = f->fh.pflag;
  e2.n_flagged = ctl_n_flagged(f);
  launch_observe(f->stream, f->d, al.blobs, al.dir, B, nullptr, nullptr, 0, f->ids_dev, f->qt, e2);
  return PK_OK;
}

static int observe_impl(pk_filter* f, const double* blobs, int32_t B, const int32_t* ids, int32_t* ids_out,
                        bool reset) {
  if (!f) return fail(PK_ERR_INVALID, "pk_observe: NULL handle");
  if (B < 0 || (B > 0 && !blobs)) return fail(PK_ERR_INVALID, "pk_observe: bad blobs");
  if (!f->map_loaded) return fail(PK_ERR_STATE, "pk_observe: no map uploaded (pk_upload_map)");
  if (f->split.active) return fail(PK_ERR_STATE, "pk_observe: a split observe is in progress (pk_observe_staged_range with last = 1 ends it)");
  int rc;
  if ((rc = use_device(f))) return rc;
  const MapLayout& lay = f->d.lay;
  const int L = lay.L;
  for (int i = 0; i < 4 * B; ++i)
    if (!std::isfinite(blobs[i])) return fail(PK_ERR_INVALID, "pk_observe: blob %d is not finite", i / 4);
  static const int32_t no_ids = 0;
  if (!ids && B == 0) ids = &no_ids;  // an empty scan needs no association: every landmark keeps its state
  if (ids)
    for (int b = 0; b < B; ++b)
      if (ids[b] < 0 || ids[b] > L) return fail(PK_ERR_INVALID, "pk_observe: ids[%d] = %d outside 0..%d", b, ids[b], L);
  // section 8(f4) on the device: the unmatched blobs of every particle go through the new-landmark bookkeeping behind the observe
  const bool grow = f->grow_on && B > 0;
  if (grow && ids) return fail(PK_ERR_STATE, "pk_observe: the new-landmark bookkeeping (pk_grow_enable) follows the maximum-likelihood association: no ids");
  if (grow && f->dense)
    return fail(PK_ERR_STATE, "pk_observe: the new-landmark bookkeeping (pk_grow_enable) writes the compact landmark layout; this filter "
                              "is on the dense one (a covariance or Qt that couples position and colour)");
  if (f->dense) return dense_observe(f, blobs, B, ids, ids_out, reset, true);
  ObserveExtras ex;
  ex.reset = reset;
  if (ids) {
    f->staged.valid = false;  // a staged ML scan does not survive another observe
    // block: ctl | blobs (4B doubles) | first (Lp int32) | next (B int32)
    const size_t o_blobs = kCtlBytes;
    const size_t o_first = o_blobs + (size_t)B * 4 * sizeof(double);
    const size_t o_next = o_first + (size_t)lay.Lp * sizeof(int32_t);
    const size_t total = o_next + (size_t)B * sizeof(int32_t);
    unsigned char* st = nullptr;
    int slot = 0;
    if ((rc = take_stage(f, total, &st, &slot))) return rc;
    if ((rc = ensure_scan_capacity(f, total))) return rc;
    memset(st, 0, kCtlBytes);
    if (B > 0) memcpy(st + o_blobs, blobs, (size_t)B * 4 * sizeof(double));
    // landmark -> blob chains shared by all particles, in scan order (prkt_core_v2.py:88)
    int32_t* first = reinterpret_cast<int32_t*>(st + o_first);
    int32_t* next = reinterpret_cast<int32_t*>(st + o_next);
    std::vector<int32_t> last((size_t)lay.Lp, -1);
    for (int l = 0; l < lay.Lp; ++l) first[l] = -1;
    int n0 = 0;
    ex.single_sightings = true;
    for (int b = 0; b < B; ++b) {
      next[b] = -1;
      int id = ids[b];
      if (id == 0) {
        ++n0;
        continue;
      }
      if (first[id - 1] < 0) {
        first[id - 1] = b;
      } else {
        next[last[id - 1]] = b;
        ex.single_sightings = false;
      }
      last[id - 1] = b;
    }
    if ((rc = upload_scan(f, st, total))) return rc;
    if ((rc = note_upload(f, slot))) return rc;
    ex.gmax_key = ctl_gmax_key(f);
    {
      Span t(f, PK_T_OBSERVE);
      launch_observe(f->stream, f->d, reinterpret_cast<const double*>(f->scan_dev + o_blobs), nullptr, B,
                     reinterpret_cast<const int32_t*>(f->scan_dev + o_first),
                     reinterpret_cast<const int32_t*>(f->scan_dev + o_next), n0, nullptr, f->qt, ex);
    }
    PK_LAUNCH_CHECK("pk_observe");
    f->src_identity = true;
    f->d.alt = nullptr;  // every map slot was just rewritten from its source: the receive buffer of the last exchange is free
    f->gmax_fused = true;
    f->route = PK_ROUTE_KNOWN_IDS;
    if (ids_out)
      for (int64_t p = 0; p < f->d.P; ++p) memcpy(ids_out + (size_t)p * B, ids, (size_t)B * 4);
    return PK_OK;
  }
  // maximum-likelihood association on the device
  AssocLaunch al;
  // (ids wanted: the association kernel + k_observe; growing maps: a publish / subscribe kernel, or that)
  if ((rc = enqueue_association(f, blobs, B, false, ids_out == nullptr, &al, grow))) return rc;
  f->grow_bits = false;
  if (grow && (al.fused || al.regs || al.big)) {
    // round 6: the one-pass kernel leaves every particle's unmatched blobs as a bit row (scan order); no second chance -- what it
    // hands on goes to the general kernels, which leave ids
    al.retry = false;
    const int words = 2 * ((B + 63) / 64);
    if ((int64_t)f->d.P * words > f->unm_cap) {
      PK_HIP(hipStreamSynchronize(f->stream));
      if (f->unm_dev) (void)hipFree(f->unm_dev);
      f->unm_dev = nullptr;
      f->unm_cap = 0;
      const int64_t cap = (int64_t)f->d.P * (words + words / 4 + 2);
      if ((rc = dev_alloc(f, &f->unm_dev, (size_t)cap))) return rc;
      f->unm_cap = cap;
    }
    f->unm_words = words;
    ex.unm = f->unm_dev;
    ex.unm_words = words;
    f->grow_bits = true;
  }
  ex.gmax_key = ctl_gmax_key(f);
  f->route = al.big ? PK_ROUTE_ML_PUB_BIG
             : al.fused ? PK_ROUTE_ML_FUSED
             : al.regs ? PK_ROUTE_ML_REGS
             : !al.fast ? PK_ROUTE_ML_GENERAL
             : (f->d.lay.L > kFastMaxL || f->fast_observe >= 2) ? PK_ROUTE_ML_SWEEP : PK_ROUTE_ML_HANDOFF;
  if (al.fused || al.regs || al.big) {
    CandTable cand;
    if ((rc = onepass_prepare(f, al, B, &cand, 0, true))) return rc;
    if ((rc = onepass_launch(f, al, B, ex, cand, 0, f->d.P))) return rc;
    if ((rc = onepass_finish(f, al, B, ex, cand))) return rc;
  } else {
    Span t(f, PK_T_OBSERVE);
    if (al.fast) {
      ObserveExtras e1 = ex;
      e1.flip = false;
      if (f->d.lay.L > kFastMaxL || f->fast_observe >= 2) {
        const SweepPlan plan = observe_sweep_plan(f->d, B);
        const size_t need = (size_t)plan.grid * plan.results_per_wg;
        if (need > f->sweep_cap) {
          PK_HIP(hipStreamSynchronize(f->stream));
          if (f->sweep_results) (void)hipFree(f->sweep_results);
          f->sweep_results = nullptr;
          f->sweep_cap = 0;
          if ((rc = dev_alloc(f, &f->sweep_results, need))) return rc;
          f->sweep_cap = need;
        }
        launch_observe_sweep(f->stream, f->d, B, al.exact, al.order, f->fh, f->qt, e1, plan, f->sweep_results);
      } else {
        launch_observe_fast(f->stream, f->d, B, al.exact, al.order, f->fh, f->qt, e1);
      }
      ObserveExtras e2 = ex;
      e2.only_flagged = f->fh.pflag;
      e2.n_flagged = ctl_n_flagged(f);
      launch_observe(f->stream, f->d, al.blobs, al.dir, B, nullptr, nullptr, 0, f->ids_dev, f->qt, e2);
    } else {
      launch_observe(f->stream, f->d, al.blobs, al.dir, B, nullptr, nullptr, 0, f->ids_dev, f->qt, ex);
    }
  }
  PK_LAUNCH_CHECK("pk_observe");
  f->src_identity = true;
  f->d.alt = nullptr;  // every map slot was just rewritten from its source: the receive buffer of the last exchange is free
  f->gmax_fused = true;
  if (grow) {  // :92-95 for every particle, on the ids the association kernel left in HBM (every particle's slot is its own now)
    Span t(f, PK_T_OBSERVE);
    launch_new_landmarks(f->stream, f->d, f->grow, f->ids_dev, al.blobs, B, f->grow_bits ? f->unm_dev : nullptr, f->unm_words,
                         f->grow_bits ? f->fh.pflag : nullptr);
    PK_LAUNCH_CHECK("pk_observe (new landmarks)");
  }
  if (ids_out && B > 0) {
    PK_HIP(hipMemcpyAsync(ids_out, f->ids_dev, (size_t)f->d.P * B * 4, hipMemcpyDeviceToHost, f->stream));
    PK_HIP(hipStreamSynchronize(f->stream));
  }
  return PK_OK;
}

int pk_observe(pk_filter* f, const double* blobs, int32_t B, const int32_t* ids, int32_t* ids_out) {
  return observe_impl(f, blobs, B, ids, ids_out, false);
}

int pk_observe_fresh(pk_filter* f, const double* blobs, int32_t B, const int32_t* ids, int32_t* ids_out) {
  return observe_impl(f, blobs, B, ids, ids_out, true);
}

int pk_stage_scan(pk_filter* f, const double* blobs, int32_t B) {
  if (!f) return fail(PK_ERR_INVALID, "pk_stage_scan: NULL handle");
  if (B < 1 || !blobs) return fail(PK_ERR_INVALID, "pk_stage_scan: needs at least one blob");
  if (!f->map_loaded) return fail(PK_ERR_STATE, "pk_stage_scan: no map uploaded (pk_upload_map)");
  for (int i = 0; i < 4 * B; ++i)
    if (!std::isfinite(blobs[i])) return fail(PK_ERR_INVALID, "pk_stage_scan: blob %d is not finite", i / 4);
  int rc;
  if ((rc = use_device(f))) return rc;
  if (f->dense) {  // no tables to build: the blobs wait on the host
    f->dense_staged.assign(blobs, blobs + 4 * (size_t)B);
    return PK_OK;
  }
  f->dense_staged.clear();
  return stage_ml_scan(f, blobs, B);
}

int pk_observe_staged(pk_filter* f, int32_t fresh) {
  if (!f) return fail(PK_ERR_INVALID, "pk_observe_staged: NULL handle");
  if (f->dense && !f->dense_staged.empty()) {
    std::vector<double> b;
    b.swap(f->dense_staged);
    return observe_impl(f, b.data(), (int32_t)(b.size() / 4), nullptr, nullptr, fresh != 0);
  }
  if (!f->staged.valid) return fail(PK_ERR_STATE, "pk_observe_staged: no staged scan (pk_stage_scan)");
  const double* blobs = reinterpret_cast<const double*>(f->staged.st + kCtlBytes);
  return observe_impl(f, blobs, f->staged.B, nullptr, nullptr, fresh != 0);
}

int pk_staged_takes_regs(pk_filter* f) {
  if (!f || f->dense || f->grow_on || !f->staged.valid || !f->staged.use_grid) return 0;  // (the bookkeeping wants the ids in HBM: the general route)
  return regs_route_taken(f, f->staged.g, f->staged.B, f->staged.n9) ? 1 : 0;
}

int pk_observe_staged_range(pk_filter* f, int32_t fresh, int64_t p0, int64_t p1, int32_t first, int32_t last) {
  if (!f) return fail(PK_ERR_INVALID, "pk_observe_staged_range: NULL handle");
  if (p0 < 0 || p1 < p0 || p1 > f->d.P) return fail(PK_ERR_INVALID, "pk_observe_staged_range: bad particle range [%lld, %lld)", (long long)p0, (long long)p1);
  if (f->grow_on) return fail(PK_ERR_STATE, "pk_observe_staged_range: the new-landmark bookkeeping (pk_grow_enable) runs behind whole observes (pk_observe / pk_observe_staged)");
  int rc;
  if ((rc = use_device(f))) return rc;
  pk_filter::Split& sp = f->split;
  if (first) {
    if (sp.active) return fail(PK_ERR_STATE, "pk_observe_staged_range: the previous split observe was not finished (last = 1)");
    if (!pk_staged_takes_regs(f))
      return fail(PK_ERR_STATE, "pk_observe_staged_range: needs a staged scan that takes the register route (pk_staged_takes_regs)");
    const double* blobs = reinterpret_cast<const double*>(f->staged.st + kCtlBytes);
    sp.B = f->staged.B;
    sp.reset = fresh != 0;
    sp.al = AssocLaunch();
    if ((rc = enqueue_association(f, blobs, sp.B, false, true, &sp.al))) return rc;
    if (!sp.al.regs && !sp.al.big) return fail(PK_ERR_STATE, "pk_observe_staged_range: the scan did not take the register route");
    f->route = sp.al.big ? PK_ROUTE_ML_PUB_BIG : PK_ROUTE_ML_REGS;
    sp.cand = CandTable();
    if ((rc = onepass_prepare(f, sp.al, sp.B, &sp.cand, p1 > p0 ? p0 : 0))) return rc;
    sp.active = true;
  } else if (!sp.active) {
    return fail(PK_ERR_STATE, "pk_observe_staged_range: no split observe in progress (first = 1)");
  }
  ObserveExtras ex;
  ex.reset = sp.reset;
  ex.gmax_key = ctl_gmax_key(f);
  // the first part of a split step runs while the exchange is in flight: it leaves some CUs to the collective's kernels
  if (p1 > p0 && (rc = onepass_launch(f, sp.al, sp.B, ex, sp.cand, p0, p1, (first && !last) ? f->split_reserve_cus : 0))) {
    sp.active = false;  // a failed piece ends the split observe: the filter is not left waiting for a "last" that cannot come
    return rc;
  }
  if (last) {
    sp.active = false;
    if ((rc = onepass_finish(f, sp.al, sp.B, ex, sp.cand))) return rc;
    PK_LAUNCH_CHECK("pk_observe_staged_range");
    f->src_identity = true;
    f->d.alt = nullptr;
    f->gmax_fused = true;
  }
  return PK_OK;
}

int pk_motion_range(pk_filter* f, double v, double w, double dt, uint64_t seed, uint64_t draw, int64_t p0, int64_t p1) {
  if (f) f->pose_part_ok = false;  // (the poses change: the motion launch's pose sums are no longer theirs)
  if (!f) return fail(PK_ERR_INVALID, "pk_motion_range: NULL handle");
  if (!std::isfinite(v) || !std::isfinite(w) || !std::isfinite(dt)) return fail(PK_ERR_INVALID, "pk_motion_range: non-finite control");
  if (p0 < 0 || p1 < p0 || p1 > f->d.P) return fail(PK_ERR_INVALID, "pk_motion_range: bad particle range [%lld, %lld)", (long long)p0, (long long)p1);
  int rc;
  if ((rc = use_device(f))) return rc;
  Span t(f, PK_T_MOTION);
  launch_motion_range(f->stream, f->d, v, w, dt, seed, draw, p0, p1);
  PK_LAUNCH_CHECK("pk_motion_range");
  return PK_OK;
}

int pk_associate(pk_filter* f, const double* blobs, int32_t B, int32_t* ids_out) {
  if (!f || !ids_out) return fail(PK_ERR_INVALID, "pk_associate: NULL argument");
  if (B < 0 || (B > 0 && !blobs)) return fail(PK_ERR_INVALID, "pk_associate: bad blobs");
  if (!f->map_loaded) return fail(PK_ERR_STATE, "pk_associate: no map uploaded (pk_upload_map)");
  if (B == 0) return PK_OK;
  int rc;
  if ((rc = use_device(f))) return rc;
  for (int i = 0; i < 4 * B; ++i)
    if (!std::isfinite(blobs[i])) return fail(PK_ERR_INVALID, "pk_associate: blob %d is not finite", i / 4);
  if (f->dense) return dense_observe(f, blobs, B, nullptr, ids_out, false, false);
  if ((rc = enqueue_association(f, blobs, B, true, false, nullptr))) return rc;
  PK_LAUNCH_CHECK("pk_associate");
  PK_HIP(hipMemcpyAsync(ids_out, f->ids_dev, (size_t)f->d.P * B * 4, hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

int pk_set_option(pk_filter* f, const char* name, int64_t value) {
  if (!f || !name) return fail(PK_ERR_INVALID, "pk_set_option: NULL argument");
  if (!strcmp(name, "assoc_kernel")) {
    if (value != 0 && value != 1) return fail(PK_ERR_INVALID, "assoc_kernel: 0 (colour grid) or 1 (brute force)");
    f->assoc_kernel = (int)value;
    return PK_OK;
  }
  if (!strcmp(name, "fast_observe")) {
    if (value < 0 || value > 3)
      return fail(PK_ERR_INVALID, "fast_observe: 0 (general kernels), 1 (default), 2 (always the sweep kernel) or 3 (... with eight slots)");
    f->fast_observe = (int)value;
    return PK_OK;
  }
  if (!strcmp(name, "timing_stride")) {
    if (value < 1 || value > 1000000) return fail(PK_ERR_INVALID, "timing_stride: 1 .. 1000000");
    f->timing_stride = (int)value;
    return PK_OK;
  }
  if (!strcmp(name, "upload_kernel")) {
    f->upload_kernel = value != 0;
    return PK_OK;
  }
  if (!strcmp(name, "fused_step")) {
    f->fused_step = value != 0;
    return PK_OK;
  }
  if (!strcmp(name, "cand_lists")) {
    f->cand_lists = value != 0;
    return PK_OK;
  }
  if (!strcmp(name, "pub_step")) {
    f->pub_step = value != 0;
    return PK_OK;
  }
  if (!strcmp(name, "pub_small")) {
    f->pub_small = value < 0 ? -1 : (value != 0 ? 1 : 0);
    return PK_OK;
  }
  if (!strcmp(name, "far_prune")) {
    f->far_prune = value != 0;
    return PK_OK;
  }
  if (!strcmp(name, "pub_duo")) {
    if (value < 0 || value > 2) return fail(PK_ERR_INVALID, "pub_duo: 0 (off), 1 (two 512-lane workgroups per CU) or 2 (three 256-lane workgroups per CU)");
    f->duo_on = (int)value;
    return PK_OK;
  }
  if (!strcmp(name, "pub_duo_park_limit")) {
    if (value < -1 || value > 65535) return fail(PK_ERR_INVALID, "pub_duo_park_limit: -1 (what LDS holds) .. 65535");
    f->duo_park_limit = (int)value;
    return PK_OK;
  }
  if (!strcmp(name, "pub_entry_limit")) {
    if (value < 0 || value > 65534) return fail(PK_ERR_INVALID, "pub_entry_limit: 0 (what LDS holds) .. 65534");
    f->pub_entry_limit = (int)value;
    return PK_OK;
  }
  if (!strcmp(name, "regs_step")) {
    f->regs_step = value != 0;
    return PK_OK;
  }
  if (!strcmp(name, "split_loopback_lo") || !strcmp(name, "split_loopback_hi")) {
    // debug (one-rank tests of the exchange): the next split adoption fills only the LOCAL slots [lo, hi) of this shard with its
    // own particles directly; the slots outside are filled from records -- which the caller packs with pk_shard_pack_slots_dev
    // and sends through the all-to-all to itself.  Cleared by pk_shard_adopt_remote_dev.
    if (value < 0 || value > f->d.P) return fail(PK_ERR_INVALID, "%s: 0..P", name);
    (name[15] == 'l' ? f->loop_lo : f->loop_hi) = value;
    return PK_OK;
  }
  if (!strcmp(name, "balanced_loopback_keep")) {
    if (value < -1 || value > f->d.P) return fail(PK_ERR_INVALID, "balanced_loopback_keep: -1 (off) .. P");
    f->bal_loop_keep = value;
    return PK_OK;
  }
  if (!strcmp(name, "split_reserve_cus")) {
    if (value < 0 || value > 128) return fail(PK_ERR_INVALID, "split_reserve_cus: 0..128");
    f->split_reserve_cus = (int)value;
    return PK_OK;
  }
  if (!strcmp(name, "regs_retry")) {
    if (value != 0 && value != 1) return fail(PK_ERR_INVALID, "regs_retry: 0 or 1");
    f->regs_retry = (int)value;
    return PK_OK;
  }
  if (!strcmp(name, "regs_warm")) {
    if (value < 0 || value > 2) return fail(PK_ERR_INVALID, "regs_warm: 0 (off), 1 (mean rows) or 2 (whole slot)");
    f->regs_warm = (int)value;
    return PK_OK;
  }
  if (!strcmp(name, "assoc_dup")) {
    f->assoc_dup = value != 0;
    return PK_OK;
  }
  if (!strcmp(name, "observe_landmarks_per_lane")) {
    if (value < 0 || value > 2) return fail(PK_ERR_INVALID, "observe_landmarks_per_lane: 0 (default), 1 or 2");
    g_observe_nv = (int)value;
    return PK_OK;
  }
  return fail(PK_ERR_INVALID, "pk_set_option: unknown option '%s'", name);
}

int pk_resample(pk_filter* f, double u, int32_t weight_domain, int64_t* ancestors_out) {
  if (f) f->pose_part_ok = false;  // (the poses change: the motion launch's pose sums are no longer theirs)
  if (!f) return fail(PK_ERR_INVALID, "pk_resample: NULL handle");
  if (!(u >= 0.0 && u < 1.0)) return fail(PK_ERR_INVALID, "pk_resample: u = %g outside [0,1)", u);
  if (weight_domain != PK_WEIGHTS_LINEAR && weight_domain != PK_WEIGHTS_LOG)
    return fail(PK_ERR_INVALID, "pk_resample: weight_domain %d", weight_domain);
  if (f->split.active) return fail(PK_ERR_STATE, "pk_resample: a split observe is in progress (half the particles observed)");
  if (f->d.logical[0]) return fail(PK_ERR_STATE, "pk_resample: the balanced placement is active on this filter (its slots carry logical indices): resample through pk_shard_plan_balanced_dev / pk_shard_adopt_balanced_dev");
  int rc;
  if ((rc = use_device(f))) return rc;
  DeviceState& d = f->d;
  {
    Span t(f, PK_T_WEIGHTS);
    const unsigned long long* key = nullptr;
    if (weight_domain == PK_WEIGHTS_LOG) {
      if (f->gmax_fused)
        key = ctl_gmax_key(f);  // the observe kernels kept the max of the weights they wrote
      else
        launch_block_max(f->stream, d, f->partial, f->gmax);
    }
    launch_scan_local(f->stream, d, f->gmax, weight_domain, f->clocal, f->totals, key);
    if (f->nblocks <= kAncestorsScanMaxBlocks) {  // k_ancestors scans the few block totals itself: one launch less
      launch_ancestors(f->stream, f->clocal, f->totals, nullptr, nullptr, f->nblocks, d.P, d.P, u, 0, d.P, f->anc, &d);
    } else {
      launch_scan_blocks(f->stream, f->totals, f->nblocks, f->offsets, f->sum);
      launch_ancestors(f->stream, f->clocal, f->totals, f->offsets, f->sum, f->nblocks, d.P, d.P, u, 0, d.P, f->anc, &d);
    }
  }
  if (f->grow_on) launch_grow_gather(f->stream, f->grow, f->anc, d.P);  // the bookkeeping follows the particles (:243)
  PK_LAUNCH_CHECK("pk_resample");
  f->src_identity = false;
  f->gmax_fused = false;
  if (ancestors_out) {
    std::vector<int32_t> a((size_t)d.P);
    PK_HIP(hipMemcpyAsync(a.data(), f->anc, (size_t)d.P * 4, hipMemcpyDeviceToHost, f->stream));
    PK_HIP(hipStreamSynchronize(f->stream));
    for (int64_t i = 0; i < d.P; ++i) ancestors_out[i] = a[i];
  }
  return PK_OK;
}

int pk_summary(pk_filter* f, double out[3]) {
  if (!f || !out) return fail(PK_ERR_INVALID, "pk_summary: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  {
    Span t(f, PK_T_SUMMARY);
    launch_summary_partials(f->stream, f->d, f->partial, f->out4);
  }
  double s[4];
  PK_HIP(hipMemcpyAsync(s, f->out4, sizeof(s), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  const double n = (double)f->d.P;  // prkt_core_v2.py:262
  out[0] = s[0] / n;                // :273
  out[1] = s[1] / n;                // :274
  out[2] = std::atan2(s[2], s[3]);  // :275
  return PK_OK;
}

int pk_pose_sums(pk_filter* f, double out[4]) {
  if (!f || !out) return fail(PK_ERR_INVALID, "pk_pose_sums: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  {
    Span t(f, PK_T_SUMMARY);
    launch_summary_partials(f->stream, f->d, f->partial, f->out4);
  }
  PK_HIP(hipMemcpyAsync(out, f->out4, 4 * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

int pk_step(pk_filter* f, double v, double w, double dt, const double* z, uint64_t seed, uint64_t draw,
            const double* blobs, int32_t B, const int32_t* ids, double u, int32_t weight_domain) {
  int rc;
  // Throughput mode with ML association: the host half of the scan upload first, then ONE launch
  // for the motion update and the upload of the scan block, then the observe kernels.
  if (f && !f->dense && !f->grow_on && !z && !ids && B > 0 && blobs && f->map_loaded && f->upload_kernel && std::isfinite(v) && std::isfinite(w) &&
      std::isfinite(dt)) {
    bool finite = true;
    for (int i = 0; i < 4 * B && finite; ++i) finite = std::isfinite(blobs[i]);
    if (finite && B <= 65535) {
      if ((rc = use_device(f))) return rc;
      if ((rc = stage_ml_scan(f, blobs, B))) return rc;
      pk_filter::Staged& sg = f->staged;
      void* dev_view = nullptr;
      if (hipHostGetDevicePointer(&dev_view, sg.st, 0) == hipSuccess && dev_view) {
        {
          Span t(f, PK_T_MOTION);
          launch_motion(f->stream, f->d, v, w, dt, nullptr, seed, draw, 0, f->scan_dev, dev_view, staged_upload_bytes(sg), f->pose_part);
      f->pose_part_ok = true;
        }
        sg.uploaded = true;
        f->gmax_fused = false;
        if ((rc = note_upload(f, sg.slot))) return rc;
        const double* staged_blobs = reinterpret_cast<const double*>(sg.st + kCtlBytes);
        if ((rc = observe_impl(f, staged_blobs, B, nullptr, nullptr, true))) return rc;  // :73 (reset fused) + :82-124
        return pk_resample(f, u, weight_domain, nullptr);                                // :137
      }
      (void)hipGetLastError();
    }
  }
  if ((rc = pk_motion(f, v, w, dt, z, seed, draw))) return rc;              // :75-77
  if ((rc = observe_impl(f, blobs, B, ids, nullptr, true))) return rc;    // :73 (reset fused) + :82-124
  return pk_resample(f, u, weight_domain, nullptr);                         // :137
}

// ---- section 8(f4) on the device: the new-landmark bookkeeping (pk_k_grow.hip) -------
int pk_grow_enable(pk_filter* f, int32_t preset_landmarks, int32_t reading_capacity, double pair_threshold) {
  if (!f) return fail(PK_ERR_INVALID, "pk_grow_enable: NULL handle");
  const int L = f->d.lay.L;
  if (preset_landmarks < 0 || preset_landmarks >= L)
    return fail(PK_ERR_INVALID, "pk_grow_enable: %d preset landmarks leave no spare slot among the filter's %d", preset_landmarks, L);
  if (reading_capacity < 1 || reading_capacity > 65536) return fail(PK_ERR_INVALID, "pk_grow_enable: reading_capacity %d outside 1..65536", reading_capacity);
  if (!(pair_threshold >= 0.0)) return fail(PK_ERR_INVALID, "pk_grow_enable: pair_threshold %g", pair_threshold);
  if (f->grow_on) return fail(PK_ERR_STATE, "pk_grow_enable: already enabled on this filter");
  if (f->dense)
    return fail(PK_ERR_STATE, "pk_grow_enable: the bookkeeping kernel writes the compact landmark layout; this filter is on the dense one "
                              "(a covariance or Qt that couples position and colour)");
  int rc;
  if ((rc = use_device(f))) return rc;
  GrowState& g = f->grow;
  g.L0 = preset_landmarks;
  g.S = L - preset_landmarks;
  g.R = reading_capacity;
  g.pair_threshold = pair_threshold;
  g.cur = 0;
  const size_t P = (size_t)f->d.P;
  for (int i = 0; i < 2; ++i) {
    if ((rc = dev_alloc(f, &g.hyp[i], P * g.R * 8))) return rc;
    if ((rc = dev_alloc(f, &g.cnt[i], P * 4))) return rc;
    if ((rc = dev_alloc(f, &g.slot_id[i], P * g.S))) return rc;
  }
  std::vector<int32_t> c0(P * 4, 0);
  for (size_t p = 0; p < P; ++p) c0[4 * p + 2] = preset_landmarks + 1;  // FilterParticle.next_id (:298)
  PK_HIP(hipMemcpyAsync(g.cnt[0], c0.data(), P * 16, hipMemcpyHostToDevice, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  f->grow_on = true;
  return PK_OK;
}

static int grow_range(pk_filter* f, const char* who, int64_t p0, int64_t p1) {
  if (!f) return fail(PK_ERR_INVALID, "%s: NULL handle", who);
  if (!f->grow_on) return fail(PK_ERR_STATE, "%s: pk_grow_enable was not called on this filter", who);
  if (p0 < 0 || p1 < p0 || p1 > f->d.P) return fail(PK_ERR_INVALID, "%s: particles [%lld, %lld) of %lld", who, (long long)p0, (long long)p1, (long long)f->d.P);
  return use_device(f);
}

int pk_grow_download(pk_filter* f, int64_t p0, int64_t p1, int32_t* counters, double* readings, int32_t* slot_ids) {
  int rc;
  if ((rc = grow_range(f, "pk_grow_download", p0, p1))) return rc;
  const GrowState& g = f->grow;
  const size_t n = (size_t)(p1 - p0);
  if (n == 0) return PK_OK;
  if (counters) PK_HIP(hipMemcpyAsync(counters, g.cnt[g.cur] + 4 * p0, n * 16, hipMemcpyDeviceToHost, f->stream));
  if (readings) PK_HIP(hipMemcpyAsync(readings, g.hyp[g.cur] + (size_t)p0 * g.R * 8, n * g.R * 64, hipMemcpyDeviceToHost, f->stream));
  if (slot_ids) PK_HIP(hipMemcpyAsync(slot_ids, g.slot_id[g.cur] + (size_t)p0 * g.S, n * g.S * 4, hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

int pk_grow_upload(pk_filter* f, int64_t p0, int64_t p1, const int32_t* counters, const double* readings, const int32_t* slot_ids) {
  int rc;
  if ((rc = grow_range(f, "pk_grow_upload", p0, p1))) return rc;
  const GrowState& g = f->grow;
  const size_t n = (size_t)(p1 - p0);
  if (n == 0) return PK_OK;
  if (counters)
    for (size_t i = 0; i < n; ++i)
      if (counters[4 * i] < 0 || counters[4 * i] > g.R || counters[4 * i + 1] < 0 || counters[4 * i + 1] > g.S || counters[4 * i + 3] < 0)
        return fail(PK_ERR_INVALID, "pk_grow_upload: particle %lld: %d readings of %d, %d spare slots of %d in use", (long long)(p0 + (int64_t)i),
                    counters[4 * i], g.R, counters[4 * i + 1], g.S);
  if (counters) PK_HIP(hipMemcpyAsync(g.cnt[g.cur] + 4 * p0, counters, n * 16, hipMemcpyHostToDevice, f->stream));
  if (readings) PK_HIP(hipMemcpyAsync(g.hyp[g.cur] + (size_t)p0 * g.R * 8, readings, n * g.R * 64, hipMemcpyHostToDevice, f->stream));
  if (slot_ids) PK_HIP(hipMemcpyAsync(g.slot_id[g.cur] + (size_t)p0 * g.S, slot_ids, n * g.S * 4, hipMemcpyHostToDevice, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

int pk_grow_shape(const pk_filter* f, int32_t* preset_landmarks, int32_t* spare_slots, int32_t* reading_capacity) {
  if (!f) return fail(PK_ERR_INVALID, "pk_grow_shape: NULL handle");
  if (preset_landmarks) *preset_landmarks = f->grow_on ? f->grow.L0 : 0;
  if (spare_slots) *spare_slots = f->grow_on ? f->grow.S : 0;
  if (reading_capacity) *reading_capacity = f->grow_on ? f->grow.R : 0;
  return PK_OK;
}

// ---- sharded resampling ---------------------------------------
int pk_shard_max_logw(pk_filter* f, double* max_logw) {
  if (!f || !max_logw) return fail(PK_ERR_INVALID, "pk_shard_max_logw: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  launch_block_max(f->stream, f->d, f->partial, f->gmax);
  PK_HIP(hipMemcpyAsync(max_logw, f->gmax, sizeof(double), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

int64_t pk_shard_num_blocks(const pk_filter* f) { return f ? f->nblocks : -1; }

int pk_shard_block_totals(pk_filter* f, double gmax, int32_t weight_domain, double* totals) {
  if (!f || !totals) return fail(PK_ERR_INVALID, "pk_shard_block_totals: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  PK_HIP(hipMemcpyAsync(f->gmax, &gmax, sizeof(double), hipMemcpyHostToDevice, f->stream));
  launch_scan_local(f->stream, f->d, f->gmax, weight_domain, f->clocal, f->totals);
  PK_HIP(hipMemcpyAsync(totals, f->totals, (size_t)f->nblocks * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

int pk_set_shard(pk_filter* f, int64_t global_offset) {
  if (!f || global_offset < 0) return fail(PK_ERR_INVALID, "pk_set_shard: bad argument");
  f->d.global_offset = global_offset;
  return PK_OK;
}

int pk_shard_offspring(pk_filter* f, const double* global_totals, int64_t n_global_blocks, int64_t first_block,
                       int64_t global_particles, double u, int32_t last_shard, int64_t* slot_hi) {
  if (!f || !global_totals || !slot_hi) return fail(PK_ERR_INVALID, "pk_shard_offspring: NULL argument");
  if (!(u >= 0.0 && u < 1.0)) return fail(PK_ERR_INVALID, "pk_shard_offspring: u = %g outside [0,1)", u);
  if (first_block < 0 || first_block + f->nblocks > n_global_blocks || global_particles < f->d.P)
    return fail(PK_ERR_INVALID, "pk_shard_offspring: shard [%lld, +%lld) does not fit %lld global blocks",
                (long long)first_block, (long long)f->nblocks, (long long)n_global_blocks);
  int rc;
  if ((rc = use_device(f))) return rc;
  if (n_global_blocks > f->gblocks_cap) {
    PK_HIP(hipStreamSynchronize(f->stream));
    if (f->g_totals) (void)hipFree(f->g_totals);
    if (f->g_offsets) (void)hipFree(f->g_offsets);
    f->g_totals = f->g_offsets = nullptr;
    f->gblocks_cap = 0;
    if ((rc = dev_alloc(f, &f->g_totals, (size_t)n_global_blocks))) return rc;
    if ((rc = dev_alloc(f, &f->g_offsets, (size_t)n_global_blocks + 1))) return rc;
    f->gblocks_cap = n_global_blocks;
  }
  if (!f->hi_dev && (rc = dev_alloc(f, &f->hi_dev, (size_t)f->d.P + 1))) return rc;
  PK_HIP(hipMemcpyAsync(f->g_totals, global_totals, (size_t)n_global_blocks * sizeof(double), hipMemcpyHostToDevice,
                        f->stream));
  {
    Span t(f, PK_T_WEIGHTS);
    launch_scan_blocks(f->stream, f->g_totals, n_global_blocks, f->g_offsets, f->sum);
    launch_offspring(f->stream, f->clocal, f->g_offsets, f->sum, first_block, f->d.P, global_particles, u,
                     last_shard ? 1 : 0, f->hi_dev);
  }
  PK_HIP(hipMemcpyAsync(slot_hi, f->hi_dev, ((size_t)f->d.P + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

// one particle in the sharded exchange: header | map slot | the new-landmark bookkeeping when it is on (pk_grow_enable)
static size_t record_stride(const pk_filter* f) { return kPoseRecordBytes + f->d.lay.slot_bytes + (f->grow_on ? grow_tail_bytes(f->grow) : 0); }
int64_t pk_particle_bytes(const pk_filter* f) { return f ? (int64_t)record_stride(f) : -1; }

int pk_pack_particles(pk_filter* f, const int64_t* local_idx, int64_t n, void* dev_buf) {
  if (!f || n < 0 || (n > 0 && (!local_idx || !dev_buf))) return fail(PK_ERR_INVALID, "pk_pack_particles: bad argument");
  // (pk_particle_bytes() counts the bookkeeping's tail, k_pack writes records without one -- and pk_adopt_particles
  // would leave readings and id counters on the wrong particles)
  if (f->grow_on) return fail(PK_ERR_STATE, "pk_pack_particles: the new-landmark bookkeeping (pk_grow_enable) travels with the balanced placement only (pk_shard_*_balanced_dev)");
  if (n > f->d.P) return fail(PK_ERR_INVALID, "pk_pack_particles: %lld records from %lld particles", (long long)n, (long long)f->d.P);
  for (int64_t i = 0; i < n; ++i)
    if (local_idx[i] < 0 || local_idx[i] >= f->d.P) return fail(PK_ERR_INVALID, "pk_pack_particles: index %lld out of range", (long long)local_idx[i]);
  if (n == 0) return PK_OK;
  int rc;
  if ((rc = use_device(f))) return rc;
  if (!f->idx_dev && (rc = dev_alloc(f, &f->idx_dev, (size_t)f->d.P))) return rc;
  PK_HIP(hipMemcpyAsync(f->idx_dev, local_idx, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, f->stream));
  {
    Span t(f, PK_T_RESAMPLE);
    launch_pack(f->stream, f->d, f->idx_dev, n, static_cast<unsigned char*>(dev_buf));
  }
  PK_HIP(hipStreamSynchronize(f->stream));  // the collective that ships dev_buf runs on another stream
  return PK_OK;
}

int pk_adopt_particles(pk_filter* f, const int64_t* src, const void* dev_buf, int64_t n_received) {
  if (f) f->pose_part_ok = false;  // (the poses change: the motion launch's pose sums are no longer theirs)
  if (!f || !src || n_received < 0 || (n_received > 0 && !dev_buf)) return fail(PK_ERR_INVALID, "pk_adopt_particles: bad argument");
  if (f->grow_on) return fail(PK_ERR_STATE, "pk_adopt_particles: the new-landmark bookkeeping (pk_grow_enable) travels with the balanced placement only (pk_shard_*_balanced_dev)");
  const int64_t P = f->d.P;
  for (int64_t k = 0; k < P; ++k)
    if (src[k] >= P || src[k] < -n_received) return fail(PK_ERR_INVALID, "pk_adopt_particles: src[%lld] = %lld out of range", (long long)k, (long long)src[k]);
  if (f->d.logical[0]) return fail(PK_ERR_STATE, "pk_adopt_particles: the balanced placement is active on this filter (its slots carry logical indices): resample through pk_shard_plan_balanced_dev / pk_shard_adopt_balanced_dev");
  int rc;
  if ((rc = use_device(f))) return rc;
  if (f->d.alt) {  // an earlier adoption is still referenced: fold it into the map buffer first
    f->src_identity = false;
    if ((rc = materialise(f))) return rc;
  }
  if (!f->srcs_dev && (rc = dev_alloc(f, &f->srcs_dev, (size_t)P))) return rc;
  PK_HIP(hipMemcpyAsync(f->srcs_dev, src, (size_t)P * sizeof(int64_t), hipMemcpyHostToDevice, f->stream));
  {
    Span t(f, PK_T_RESAMPLE);
    launch_adopt(f->stream, f->d, f->srcs_dev, static_cast<const unsigned char*>(dev_buf));
  }
  if (n_received == 0) f->d.alt = nullptr;
  f->src_identity = false;
  f->gmax_fused = false;
  PK_HIP(hipStreamSynchronize(f->stream));  // src was pageable host memory
  return PK_OK;
}

// ---- device-resident variants: every buffer is a device pointer owned by the caller (torch
// tensors), nothing synchronises the stream; see DESIGN.md section 6 --------------------------
int pk_shard_max_logw_dev(pk_filter* f, double* dev_out) {
  if (!f || !dev_out) return fail(PK_ERR_INVALID, "pk_shard_max_logw_dev: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  Span t(f, PK_T_WEIGHTS);
  if (f->gmax_fused)
    launch_keys_max(f->stream, ctl_gmax_key(f), dev_out);  // the observe kernels kept the running max
  else
    launch_block_max(f->stream, f->d, f->partial, dev_out);
  return PK_OK;
}

int pk_shard_block_totals_dev(pk_filter* f, const double* dev_gmax, int32_t weight_domain, double* dev_totals) {
  if (!f || !dev_totals || (weight_domain == PK_WEIGHTS_LOG && !dev_gmax))
    return fail(PK_ERR_INVALID, "pk_shard_block_totals_dev: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  Span t(f, PK_T_WEIGHTS);
  launch_scan_local(f->stream, f->d, dev_gmax ? dev_gmax : f->gmax, weight_domain, f->clocal, dev_totals);
  return PK_OK;
}

int pk_shard_plan_dev(pk_filter* f, const double* dev_global_totals, int64_t n_global_blocks, int64_t first_block,
                      int64_t global_particles, double u, int32_t last_shard, int32_t world, int64_t* dev_ranges) {
  if (!f || !dev_global_totals || !dev_ranges || world < 1) return fail(PK_ERR_INVALID, "pk_shard_plan_dev: bad argument");
  if (!(u >= 0.0 && u < 1.0)) return fail(PK_ERR_INVALID, "pk_shard_plan_dev: u = %g outside [0,1)", u);
  if (first_block < 0 || first_block + f->nblocks > n_global_blocks || global_particles != (int64_t)world * f->d.P)
    return fail(PK_ERR_INVALID, "pk_shard_plan_dev: shard geometry (%lld blocks from %lld of %lld, %lld particles x %d)",
                (long long)f->nblocks, (long long)first_block, (long long)n_global_blocks, (long long)f->d.P, world);
  int rc;
  if ((rc = use_device(f))) return rc;
  if (n_global_blocks > f->gblocks_cap) {
    PK_HIP(hipStreamSynchronize(f->stream));
    if (f->g_totals) (void)hipFree(f->g_totals);
    if (f->g_offsets) (void)hipFree(f->g_offsets);
    f->g_totals = f->g_offsets = nullptr;
    f->gblocks_cap = 0;
    if ((rc = dev_alloc(f, &f->g_totals, (size_t)n_global_blocks))) return rc;
    if ((rc = dev_alloc(f, &f->g_offsets, (size_t)n_global_blocks + 1))) return rc;
    f->gblocks_cap = n_global_blocks;
  }
  if (!f->hi_dev && (rc = dev_alloc(f, &f->hi_dev, (size_t)f->d.P + 1))) return rc;
  if (!f->plan_ticket) {
    if ((rc = dev_alloc(f, &f->plan_ticket, (size_t)1))) return rc;
    PK_HIP(hipMemsetAsync(f->plan_ticket, 0, sizeof(unsigned), f->stream));
  }
  Span t(f, PK_T_WEIGHTS);
  if (n_global_blocks <= kAncestorsScanMaxBlocks) {
    // one launch: every workgroup scans the few global block totals itself, the last one to finish
    // writes the per-destination ranges
    launch_offspring_plan(f->stream, f->clocal, dev_global_totals, n_global_blocks, first_block, f->d.P, global_particles, u,
                          last_shard ? 1 : 0, f->hi_dev, world, dev_ranges, f->plan_ticket);
    return PK_OK;
  }
  launch_scan_blocks(f->stream, dev_global_totals, n_global_blocks, f->g_offsets, f->sum);
  launch_offspring(f->stream, f->clocal, f->g_offsets, f->sum, first_block, f->d.P, global_particles, u,
                   last_shard ? 1 : 0, f->hi_dev);
  launch_shard_ranges(f->stream, f->hi_dev, f->d.P, world, dev_ranges);
  return PK_OK;
}

// ---- the same plan from the scan of the WHOLE filter's weights: any shard size ----------------------------------------
int pk_shard_logw_dev(pk_filter* f, double* dev_out) {
  if (!f || !dev_out) return fail(PK_ERR_INVALID, "pk_shard_logw_dev: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  PK_HIP(hipMemcpyAsync(dev_out, f->d.logw[f->d.cur], (size_t)f->d.P * sizeof(double), hipMemcpyDeviceToDevice, f->stream));
  return PK_OK;
}

int pk_shard_plan_global_dev(pk_filter* f, const double* dev_global_logw, int64_t global_particles, const double* dev_gmax,
                             int32_t weight_domain, double u, int32_t last_shard, int32_t world, int64_t* dev_ranges) {
  if (!f || !dev_global_logw || !dev_ranges || world < 1) return fail(PK_ERR_INVALID, "pk_shard_plan_global_dev: bad argument");
  if (weight_domain == PK_WEIGHTS_LOG && !dev_gmax) return fail(PK_ERR_INVALID, "pk_shard_plan_global_dev: NULL maximum");
  if (!(u >= 0.0 && u < 1.0)) return fail(PK_ERR_INVALID, "pk_shard_plan_global_dev: u = %g outside [0,1)", u);
  const int64_t P = f->d.P, goff = f->d.global_offset;
  if (global_particles != (int64_t)world * P || goff < 0 || goff + P > global_particles)
    return fail(PK_ERR_INVALID, "pk_shard_plan_global_dev: shard [%lld, +%lld) of %lld particles, world %d", (long long)goff,
                (long long)P, (long long)global_particles, world);
  int rc;
  if ((rc = use_device(f))) return rc;
  const int64_t nbg = (global_particles + kScanBlock - 1) / kScanBlock;
  if (global_particles > f->gl_cap) {
    PK_HIP(hipStreamSynchronize(f->stream));
    for (double** q : {&f->gl_clocal, &f->gl_totals, &f->gl_offsets}) {
      if (*q) (void)hipFree(*q);
      *q = nullptr;
    }
    f->gl_cap = 0;
    if ((rc = dev_alloc(f, &f->gl_clocal, (size_t)global_particles))) return rc;
    if ((rc = dev_alloc(f, &f->gl_totals, (size_t)nbg))) return rc;
    if ((rc = dev_alloc(f, &f->gl_offsets, (size_t)nbg + 1))) return rc;
    f->gl_cap = global_particles;
  }
  if (!f->hi_dev && (rc = dev_alloc(f, &f->hi_dev, (size_t)P + 1))) return rc;
  Span t(f, PK_T_WEIGHTS);
  // exactly the kernels of the 1-GPU resample on the whole filter's log-weights: same blocks, same additions, same bits
  launch_scan_local_of(f->stream, dev_global_logw, global_particles, dev_gmax ? dev_gmax : f->gmax, weight_domain, f->gl_clocal,
                       f->gl_totals);
  launch_scan_blocks(f->stream, f->gl_totals, nbg, f->gl_offsets, f->sum);
  launch_offspring_global(f->stream, f->gl_clocal, f->gl_offsets, f->sum, goff, P, global_particles, u, last_shard ? 1 : 0,
                          f->hi_dev);
  launch_shard_ranges(f->stream, f->hi_dev, P, world, dev_ranges);
  PK_LAUNCH_CHECK("pk_shard_plan_global_dev");
  return PK_OK;
}

int pk_shard_download_offspring(pk_filter* f, int64_t* slot_hi) {
  if (!f || !slot_hi) return fail(PK_ERR_INVALID, "pk_shard_download_offspring: NULL argument");
  if (!f->hi_dev) return fail(PK_ERR_STATE, "pk_shard_download_offspring: no plan yet");
  int rc;
  if ((rc = use_device(f))) return rc;
  PK_HIP(hipMemcpyAsync(slot_hi, f->hi_dev, ((size_t)f->d.P + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

int pk_shard_pack_dev(pk_filter* f, const int64_t* ranges, int32_t world, int32_t rank, void* dev_buf) {
  if (f && f->grow_on) return fail(PK_ERR_STATE, "pk_shard_pack_dev: the new-landmark bookkeeping (pk_grow_enable) travels with the balanced placement only (pk_shard_*_balanced_dev)");
  if (!f || !ranges || world < 1 || rank < 0 || rank >= world) return fail(PK_ERR_INVALID, "pk_shard_pack_dev: bad argument");
  if (!f->hi_dev) return fail(PK_ERR_STATE, "pk_shard_pack_dev: call pk_shard_plan_dev first");
  int rc;
  if ((rc = use_device(f))) return rc;
  const int64_t P = f->d.P;
  const size_t stride = kPoseRecordBytes + f->d.lay.slot_bytes;
  int64_t rec = 0;
  Span t(f, PK_T_RESAMPLE);
  for (int d = 0; d < world; ++d) {
    if (d == rank) continue;
    const int64_t j0 = ranges[2 * d], j1 = ranges[2 * d + 1];
    if (j0 < 0 || j1 < j0 || j1 > P) return fail(PK_ERR_INVALID, "pk_shard_pack_dev: range [%lld, %lld) for rank %d", (long long)j0, (long long)j1, d);
    if (j1 > j0 && !dev_buf) return fail(PK_ERR_INVALID, "pk_shard_pack_dev: NULL buffer");
    launch_pack_range(f->stream, f->d, f->hi_dev, j0, j1 - j0, (int64_t)d * P, (int64_t)(d + 1) * P,
                      static_cast<unsigned char*>(dev_buf) + (size_t)rec * stride);
    rec += j1 - j0;
  }
  return PK_OK;
}

int pk_shard_pack_slots_dev(pk_filter* f, int64_t j0, int64_t j1, int64_t slot_lo, int64_t slot_hi, void* dev_buf) {
  if (f && f->grow_on) return fail(PK_ERR_STATE, "pk_shard_pack_slots_dev: the new-landmark bookkeeping (pk_grow_enable) travels with the balanced placement only (pk_shard_*_balanced_dev)");
  if (!f || j0 < 0 || j1 < j0 || j1 > f->d.P || slot_hi < slot_lo) return fail(PK_ERR_INVALID, "pk_shard_pack_slots_dev: bad argument");
  if (!f->hi_dev) return fail(PK_ERR_STATE, "pk_shard_pack_slots_dev: call pk_shard_plan_dev first");
  if (j1 > j0 && !dev_buf) return fail(PK_ERR_INVALID, "pk_shard_pack_slots_dev: NULL buffer");
  int rc;
  if ((rc = use_device(f))) return rc;
  Span t(f, PK_T_RESAMPLE);
  launch_pack_range(f->stream, f->d, f->hi_dev, j0, j1 - j0, slot_lo, slot_hi, static_cast<unsigned char*>(dev_buf));
  PK_LAUNCH_CHECK("pk_shard_pack_slots_dev");
  return PK_OK;
}

int pk_shard_adopt_dev(pk_filter* f, int32_t rank, const void* dev_recv, int64_t n_received) {
  if (f && f->grow_on) return fail(PK_ERR_STATE, "pk_shard_adopt_dev: the new-landmark bookkeeping (pk_grow_enable) travels with the balanced placement only (pk_shard_*_balanced_dev)");
  if (f) f->pose_part_ok = false;  // (the poses change: the motion launch's pose sums are no longer theirs)
  if (!f || rank < 0 || n_received < 0 || (n_received > 0 && !dev_recv)) return fail(PK_ERR_INVALID, "pk_shard_adopt_dev: bad argument");
  if (!f->hi_dev) return fail(PK_ERR_STATE, "pk_shard_adopt_dev: call pk_shard_plan_dev first");
  if (f->d.logical[0]) return fail(PK_ERR_STATE, "pk_shard_adopt_dev: the balanced placement is active on this filter (its slots carry logical indices): resample through pk_shard_plan_balanced_dev / pk_shard_adopt_balanced_dev");
  int rc;
  if ((rc = use_device(f))) return rc;
  if (f->d.alt) {  // an earlier adoption is still referenced: fold it into the map buffer first
    f->src_identity = false;
    if ((rc = materialise(f))) return rc;
  }
  if (n_received > f->rlohi_cap) {
    PK_HIP(hipStreamSynchronize(f->stream));
    if (f->rlohi_dev) (void)hipFree(f->rlohi_dev);
    f->rlohi_dev = nullptr;
    f->rlohi_cap = 0;
    if ((rc = dev_alloc(f, &f->rlohi_dev, (size_t)(2 * n_received + 2 * n_received / 4 + 16)))) return rc;
    f->rlohi_cap = n_received + n_received / 4;
  }
  Span t(f, PK_T_RESAMPLE);
  launch_adopt_dev(f->stream, f->d, f->hi_dev, (int64_t)rank * f->d.P, static_cast<const unsigned char*>(dev_recv),
                   n_received, f->rlohi_dev);
  f->src_identity = false;
  f->gmax_fused = false;
  return PK_OK;
}

int pk_shard_adopt_local_dev(pk_filter* f, int32_t rank) {
  if (f && f->grow_on) return fail(PK_ERR_STATE, "pk_shard_adopt_local_dev: the new-landmark bookkeeping (pk_grow_enable) travels with the balanced placement only (pk_shard_*_balanced_dev)");
  if (f) f->pose_part_ok = false;  // (the poses change: the motion launch's pose sums are no longer theirs)
  if (!f || rank < 0) return fail(PK_ERR_INVALID, "pk_shard_adopt_local_dev: bad argument");
  if (!f->hi_dev) return fail(PK_ERR_STATE, "pk_shard_adopt_local_dev: call pk_shard_plan_dev first");
  if (f->d.logical[0]) return fail(PK_ERR_STATE, "pk_shard_adopt_local_dev: the balanced placement is active on this filter (its slots carry logical indices): resample through pk_shard_plan_balanced_dev / pk_shard_adopt_balanced_dev");
  int rc;
  if ((rc = use_device(f))) return rc;
  if (f->d.alt) {  // an earlier adoption is still referenced: fold it into the map buffer first
    f->src_identity = false;
    if ((rc = materialise(f))) return rc;
  }
  Span t(f, PK_T_RESAMPLE);
  const int64_t base = (int64_t)rank * f->d.P;
  launch_adopt_dev(f->stream, f->d, f->hi_dev, base, nullptr, 0, nullptr, 1, f->loop_lo == INT64_MIN ? INT64_MIN : base + f->loop_lo,
                   f->loop_hi == INT64_MAX ? INT64_MAX : base + f->loop_hi);
  f->src_identity = false;
  f->gmax_fused = false;
  f->adopt_local_done = true;
  return PK_OK;
}

int pk_shard_adopt_remote_dev(pk_filter* f, int32_t rank, const void* dev_recv, int64_t n_received) {
  if (f && f->grow_on) return fail(PK_ERR_STATE, "pk_shard_adopt_remote_dev: the new-landmark bookkeeping (pk_grow_enable) travels with the balanced placement only (pk_shard_*_balanced_dev)");
  if (f) f->pose_part_ok = false;  // (the poses change: the motion launch's pose sums are no longer theirs)
  if (!f || rank < 0 || n_received < 0 || (n_received > 0 && !dev_recv)) return fail(PK_ERR_INVALID, "pk_shard_adopt_remote_dev: bad argument");
  if (!f->hi_dev) return fail(PK_ERR_STATE, "pk_shard_adopt_remote_dev: call pk_shard_plan_dev first");
  if (!f->adopt_local_done)
    return fail(PK_ERR_STATE, "pk_shard_adopt_remote_dev: pk_shard_adopt_local_dev first (it makes the new generation current)");
  f->adopt_local_done = false;
  int rc;
  if ((rc = use_device(f))) return rc;
  if (n_received > f->rlohi_cap) {
    PK_HIP(hipStreamSynchronize(f->stream));
    if (f->rlohi_dev) (void)hipFree(f->rlohi_dev);
    f->rlohi_dev = nullptr;
    f->rlohi_cap = 0;
    if ((rc = dev_alloc(f, &f->rlohi_dev, (size_t)(2 * n_received + 2 * n_received / 4 + 16)))) return rc;
    f->rlohi_cap = n_received + n_received / 4;
  }
  Span t(f, PK_T_RESAMPLE);
  const int64_t base = (int64_t)rank * f->d.P;
  launch_adopt_dev(f->stream, f->d, f->hi_dev, base, static_cast<const unsigned char*>(dev_recv), n_received, f->rlohi_dev, 2,
                   f->loop_lo == INT64_MIN ? INT64_MIN : base + f->loop_lo, f->loop_hi == INT64_MAX ? INT64_MAX : base + f->loop_hi);
  f->loop_lo = INT64_MIN;
  f->loop_hi = INT64_MAX;
  return PK_OK;
}

int pk_shard_local_span_dev(pk_filter* f, int64_t* dev_out2) {
  if (!f || !dev_out2) return fail(PK_ERR_INVALID, "pk_shard_local_span_dev: NULL argument");
  if (!f->hi_dev) return fail(PK_ERR_STATE, "pk_shard_local_span_dev: no plan yet");
  int rc;
  if ((rc = use_device(f))) return rc;
  PK_HIP(hipMemcpyAsync(dev_out2, f->hi_dev, sizeof(int64_t), hipMemcpyDeviceToDevice, f->stream));
  PK_HIP(hipMemcpyAsync(dev_out2 + 1, f->hi_dev + f->d.P, sizeof(int64_t), hipMemcpyDeviceToDevice, f->stream));
  return PK_OK;
}

// ---- balanced placement: minimum migration (sharded.py::plan_balanced is the readable reference) ----
static int ensure_logical(pk_filter* f) {
  DeviceState& d = f->d;
  if (d.logical[0]) return PK_OK;
  int rc;
  int64_t *a = nullptr, *b = nullptr;
  if ((rc = dev_alloc(f, &a, (size_t)d.P))) return rc;
  if ((rc = dev_alloc(f, &b, (size_t)d.P))) {
    (void)hipFree(a);
    return rc;
  }
  launch_iota64(f->stream, a, d.P, d.global_offset);
  launch_iota64(f->stream, b, d.P, d.global_offset);
  d.logical[0] = a;
  d.logical[1] = b;
  return PK_OK;
}

int pk_shard_reset_placement(pk_filter* f) {
  if (!f) return fail(PK_ERR_INVALID, "pk_shard_reset_placement: NULL handle");
  int rc;
  if ((rc = use_device(f))) return rc;
  if (f->d.logical[0]) launch_iota64(f->stream, f->d.logical[f->d.cur], f->d.P, f->d.global_offset);
  return PK_OK;
}

int pk_shard_download_logical(pk_filter* f, int64_t* logical) {
  if (!f || !logical) return fail(PK_ERR_INVALID, "pk_shard_download_logical: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  if (!f->d.logical[0]) {
    for (int64_t j = 0; j < f->d.P; ++j) logical[j] = f->d.global_offset + j;
    return PK_OK;
  }
  PK_HIP(hipMemcpyAsync(logical, f->d.logical[f->d.cur], (size_t)f->d.P * sizeof(int64_t), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

/* The placement set from the host (tests of the planner in isolation; restoring a snapshot taken in physical order): slot j holds
 * logical particle logical[j].  Every index must lie in [0, 2^31). */
int pk_shard_upload_logical(pk_filter* f, const int64_t* logical) {
  if (!f || !logical) return fail(PK_ERR_INVALID, "pk_shard_upload_logical: NULL argument");
  for (int64_t j = 0; j < f->d.P; ++j)
    if (logical[j] < 0 || logical[j] >= ((int64_t)1 << 31)) return fail(PK_ERR_INVALID, "pk_shard_upload_logical: logical[%lld] = %lld", (long long)j, (long long)logical[j]);
  if (f->split.active) return fail(PK_ERR_STATE, "pk_shard_upload_logical: a split observe is in progress");
  int rc;
  if ((rc = use_device(f))) return rc;
  if ((rc = ensure_logical(f))) return rc;
  PK_HIP(hipMemcpyAsync(f->d.logical[f->d.cur], logical, (size_t)f->d.P * sizeof(int64_t), hipMemcpyHostToDevice, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

/* This rank's tables of the last balanced plan (tests): rel[P + 1] the children of its particles [0, j), Hl[P] the first output slot
 * of every particle's children, alive[P] its particles with children, ascending -- the first rel-derived count of them is valid,
 * the rest is -1. */
int pk_shard_download_balanced_plan(pk_filter* f, int64_t* rel, int64_t* Hl, int32_t* alive) {
  if (!f || !rel || !Hl || !alive) return fail(PK_ERR_INVALID, "pk_shard_download_balanced_plan: NULL argument");
  if (!f->bal.rel || !f->bal.Hl || !f->bal.alive) return fail(PK_ERR_STATE, "pk_shard_download_balanced_plan: no balanced plan yet (pk_shard_plan_balanced_dev)");
  int rc;
  if ((rc = use_device(f))) return rc;
  const int64_t P = f->d.P;
  PK_HIP(hipMemcpyAsync(rel, f->bal.rel, ((size_t)P + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipMemcpyAsync(Hl, f->bal.Hl, (size_t)P * sizeof(int64_t), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipMemcpyAsync(alive, f->bal.alive, (size_t)P * sizeof(int32_t), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  int64_t n = 0;  // particles with children: what k_bal_own wrote of alive[]
  for (int64_t j = 0; j < P; ++j) n += rel[j + 1] > rel[j] ? 1 : 0;
  for (int64_t j = n; j < P; ++j) alive[j] = -1;
  return PK_OK;
}

int pk_shard_state_dev(pk_filter* f, double* dev_out) {
  if (!f || !dev_out) return fail(PK_ERR_INVALID, "pk_shard_state_dev: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  if ((rc = ensure_logical(f))) return rc;
  launch_bal_state(f->stream, f->d, dev_out);
  PK_LAUNCH_CHECK("pk_shard_state_dev");
  return PK_OK;
}

int pk_shard_plan_balanced_dev(pk_filter* f, const double* dev_global_state, int64_t global_particles, const double* dev_gmax,
                               int32_t weight_domain, double u, int32_t world, int32_t rank, int64_t* dev_table) {
  if (!f || !dev_global_state || !dev_table || world < 1 || world > 64 || rank < 0 || rank >= world)
    return fail(PK_ERR_INVALID, "pk_shard_plan_balanced_dev: bad argument (world 1 .. 64)");
  if (weight_domain != PK_WEIGHTS_LINEAR && weight_domain != PK_WEIGHTS_LOG)
    return fail(PK_ERR_INVALID, "pk_shard_plan_balanced_dev: weight_domain %d", weight_domain);
  if (weight_domain == PK_WEIGHTS_LOG && !dev_gmax) return fail(PK_ERR_INVALID, "pk_shard_plan_balanced_dev: NULL maximum");
  if (!(u >= 0.0 && u < 1.0)) return fail(PK_ERR_INVALID, "pk_shard_plan_balanced_dev: u = %g outside [0,1)", u);
  const int64_t P = f->d.P, Pg = global_particles;
  if (Pg != (int64_t)world * P || Pg >= ((int64_t)1 << 31))
    return fail(PK_ERR_INVALID, "pk_shard_plan_balanced_dev: %lld particles are not %d shards of %lld (or more than 2^31)", (long long)Pg, world, (long long)P);
  if (f->split.active) return fail(PK_ERR_STATE, "pk_shard_plan_balanced_dev: a split observe is in progress");
  int rc;
  if ((rc = use_device(f))) return rc;
  if ((rc = ensure_logical(f))) return rc;
  BalancedBuffers& b = f->bal;
  const int64_t nbg = (Pg + kScanBlock - 1) / kScanBlock;
  if (Pg > b.cap) {
    PK_HIP(hipStreamSynchronize(f->stream));
    for (void* q : {(void*)b.glogw, (void*)b.clocal, (void*)b.totals, (void*)b.offsets, (void*)b.H, (void*)b.cloc, (void*)b.ctot, (void*)b.coff})
      if (q) (void)hipFree(q);
    b.glogw = b.clocal = b.totals = b.offsets = nullptr;
    b.H = nullptr;
    b.cloc = b.ctot = b.coff = nullptr;
    b.cap = 0;
    if ((rc = dev_alloc(f, &b.glogw, (size_t)Pg))) return rc;
    if ((rc = dev_alloc(f, &b.clocal, (size_t)Pg))) return rc;
    if ((rc = dev_alloc(f, &b.totals, (size_t)nbg))) return rc;
    if ((rc = dev_alloc(f, &b.offsets, (size_t)nbg + 1))) return rc;
    if ((rc = dev_alloc(f, &b.H, (size_t)Pg + 1))) return rc;
    if ((rc = dev_alloc(f, &b.cloc, (size_t)Pg))) return rc;
    if ((rc = dev_alloc(f, &b.ctot, (size_t)nbg))) return rc;
    if ((rc = dev_alloc(f, &b.coff, (size_t)nbg + 1))) return rc;
    if (!b.sum && (rc = dev_alloc(f, &b.sum, (size_t)1))) return rc;
    if (!b.rel && (rc = dev_alloc(f, &b.rel, (size_t)P + 1))) return rc;
    if (!b.Hl && (rc = dev_alloc(f, &b.Hl, (size_t)P))) return rc;
    if (!b.alive && (rc = dev_alloc(f, &b.alive, (size_t)P))) return rc;
    if (!b.bad) {
      if ((rc = dev_alloc(f, &b.bad, (size_t)1))) return rc;
      PK_HIP(hipMemsetAsync(b.bad, 0, sizeof(int), f->stream));
    }
    b.cap = Pg;
  }
  Span t(f, PK_T_WEIGHTS);
  launch_bal_plan(f->stream, f->d, dev_global_state, Pg, world, rank, dev_gmax ? dev_gmax : f->gmax, weight_domain, u, b, dev_table);
  PK_LAUNCH_CHECK("pk_shard_plan_balanced_dev");
  f->bal_m = -1;
  return PK_OK;
}

int pk_shard_download_balanced_offspring(pk_filter* f, int64_t global_particles, int64_t* H) {
  if (!f || !H) return fail(PK_ERR_INVALID, "pk_shard_download_balanced_offspring: NULL argument");
  if (!f->bal.H || global_particles != f->bal.cap) return fail(PK_ERR_STATE, "pk_shard_download_balanced_offspring: no balanced plan of %lld particles yet", (long long)global_particles);
  int rc;
  if ((rc = use_device(f))) return rc;
  PK_HIP(hipMemcpyAsync(H, f->bal.H, ((size_t)global_particles + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}

// the plan's table as the host read it: world rows of (a0, a1) x world, n, m, ebase, dbase
static int balanced_row_check(const int64_t* table, int world, int64_t P, const char* who) {
  const int row = 2 * world + 4;
  int64_t e_sum = 0, d_sum = 0;
  for (int r = 0; r < world; ++r) {
    const int64_t n = table[(size_t)r * row + 2 * world], m = table[(size_t)r * row + 2 * world + 1];
    if (n < 0 || m != (n < P ? n : P) || table[(size_t)r * row + 2 * world + 2] != e_sum || table[(size_t)r * row + 2 * world + 3] != d_sum)
      return fail(PK_ERR_INVALID, "%s: the plan table is not one pk_shard_plan_balanced_dev wrote (rank %d)", who, r);
    e_sum += n - m;
    d_sum += P - m;
    for (int d = 0; d < world; ++d) {
      const int64_t a0 = table[(size_t)r * row + 2 * d], a1 = table[(size_t)r * row + 2 * d + 1];
      if (a0 < 0 || a1 < a0 || a1 > P) return fail(PK_ERR_INVALID, "%s: range [%lld, %lld) from rank %d to rank %d", who, (long long)a0, (long long)a1, r, d);
    }
  }
  if (e_sum != d_sum) return fail(PK_ERR_INVALID, "%s: excess %lld and free slots %lld differ", who, (long long)e_sum, (long long)d_sum);
  return PK_OK;
}

int pk_shard_pack_balanced_dev(pk_filter* f, const int64_t* table, int32_t world, int32_t rank, void* dev_buf) {
  if (!f || !table || world < 1 || world > 64 || rank < 0 || rank >= world) return fail(PK_ERR_INVALID, "pk_shard_pack_balanced_dev: bad argument");
  if (!f->bal.rel || !f->d.logical[0]) return fail(PK_ERR_STATE, "pk_shard_pack_balanced_dev: call pk_shard_plan_balanced_dev first");
  int rc;
  const int64_t P = f->d.P;
  if ((rc = balanced_row_check(table, world, P, "pk_shard_pack_balanced_dev"))) return rc;
  if ((rc = use_device(f))) return rc;
  const int row = 2 * world + 4;
  const size_t stride = record_stride(f);
  const int64_t ebase_s = table[(size_t)rank * row + 2 * world + 2];
  int64_t rec = 0;
  Span t(f, PK_T_RESAMPLE);
  for (int d = 0; d < world; ++d) {
    if (d == rank) continue;
    const int64_t a0 = table[(size_t)rank * row + 2 * d], a1 = table[(size_t)rank * row + 2 * d + 1];
    if (a1 <= a0) continue;
    if (!dev_buf) return fail(PK_ERR_INVALID, "pk_shard_pack_balanced_dev: NULL buffer");
    const int64_t m_d = table[(size_t)d * row + 2 * world + 1], dbase_d = table[(size_t)d * row + 2 * world + 3];
    launch_bal_pack(f->stream, f->d, f->bal, a0, a1 - a0, ebase_s, dbase_d, P - m_d, m_d,
                    static_cast<unsigned char*>(dev_buf) + (size_t)rec * stride, stride, f->grow_on ? &f->grow : nullptr);
    rec += a1 - a0;
  }
  PK_LAUNCH_CHECK("pk_shard_pack_balanced_dev");
  return PK_OK;
}

/* debug (one-rank tests of the balanced exchange, option "balanced_loopback_keep"): records of this rank's particles alive[a0, a1) for
 * ITSELF -- the children from position `keep` on, destined for its own slots [keep, P): what k_bal_pack writes for another rank whose
 * free slots are [keep, P), with the 64-byte balanced header (and the new-landmark bookkeeping behind the map while that is on). */
int pk_shard_pack_balanced_loop_dev(pk_filter* f, int64_t keep, int64_t a0, int64_t a1, void* dev_buf) {
  if (!f || keep < 0 || keep > f->d.P || a0 < 0 || a1 < a0 || a1 > f->d.P || (a1 > a0 && !dev_buf))
    return fail(PK_ERR_INVALID, "pk_shard_pack_balanced_loop_dev: bad argument");
  if (!f->bal.rel || !f->d.logical[0]) return fail(PK_ERR_STATE, "pk_shard_pack_balanced_loop_dev: call pk_shard_plan_balanced_dev first");
  int rc;
  if ((rc = use_device(f))) return rc;
  Span t(f, PK_T_RESAMPLE);
  // (k_bal_pack's "P" is the number of children a rank keeps: here keep; excess positions count from 0, the destination's free slots
  // are [keep, P): ebase = dbase = 0, dd = P - keep, m_d = keep)
  launch_bal_pack(f->stream, f->d, f->bal, a0, a1 - a0, 0, 0, f->d.P - keep, keep, static_cast<unsigned char*>(dev_buf), record_stride(f),
                  f->grow_on ? &f->grow : nullptr, keep);
  PK_LAUNCH_CHECK("pk_shard_pack_balanced_loop_dev");
  return PK_OK;
}

int pk_shard_adopt_balanced_dev(pk_filter* f, const int64_t* table, int32_t world, int32_t rank, const void* dev_recv,
                                int64_t n_received, int32_t mode) {
  if (f) f->pose_part_ok = false;
  if (!f || !table || world < 1 || world > 64 || rank < 0 || rank >= world || n_received < 0 || mode < 0 || mode > 2 ||
      (n_received > 0 && !dev_recv))
    return fail(PK_ERR_INVALID, "pk_shard_adopt_balanced_dev: bad argument");
  if (!f->bal.rel || !f->d.logical[0]) return fail(PK_ERR_STATE, "pk_shard_adopt_balanced_dev: call pk_shard_plan_balanced_dev first");
  if (f->grow_on && mode != 0) return fail(PK_ERR_STATE, "pk_shard_adopt_balanced_dev: the new-landmark bookkeeping (pk_grow_enable) is adopted whole (mode 0)");
  int rc;
  const int64_t P = f->d.P;
  if ((rc = balanced_row_check(table, world, P, "pk_shard_adopt_balanced_dev"))) return rc;
  const int row = 2 * world + 4;
  int64_t m = table[(size_t)rank * row + 2 * world + 1];
  // debug loopback (one rank): the rank's own children fill [0, keep) only, its children from position keep on arrive as records
  const bool loop = f->bal_loop_keep >= 0 && world == 1;
  if (loop) m = f->bal_loop_keep < m ? f->bal_loop_keep : m;
  if (mode == 2) {
    if (!f->adopt_local_done) return fail(PK_ERR_STATE, "pk_shard_adopt_balanced_dev: mode 1 first (it makes the new generation current)");
    f->adopt_local_done = false;
  }
  if (mode != 1) {
    int64_t expect = 0;
    for (int s = 0; s < world; ++s)
      if (s != rank) expect += table[(size_t)s * row + 2 * rank + 1] - table[(size_t)s * row + 2 * rank];
    if (!loop && expect != n_received) return fail(PK_ERR_INVALID, "pk_shard_adopt_balanced_dev: %lld records received, the plan sends %lld", (long long)n_received, (long long)expect);
    if (m < P && n_received == 0) return fail(PK_ERR_INVALID, "pk_shard_adopt_balanced_dev: %lld free slots and no records", (long long)(P - m));
    f->bal_loop_keep = -1;  // (one adoption: cleared behind mode 0 or mode 2)
  }
  if ((rc = use_device(f))) return rc;
  if (mode != 2 && f->d.alt) {  // an earlier adoption is still referenced: fold it into the map buffer first
    f->src_identity = false;
    if ((rc = materialise(f))) return rc;
  }
  if (3 * n_received > 2 * f->rlohi_cap) {
    PK_HIP(hipStreamSynchronize(f->stream));
    if (f->rlohi_dev) (void)hipFree(f->rlohi_dev);
    f->rlohi_dev = nullptr;
    f->rlohi_cap = 0;
    if ((rc = dev_alloc(f, &f->rlohi_dev, (size_t)(3 * n_received + 3 * n_received / 4 + 16)))) return rc;
    f->rlohi_cap = (3 * n_received + 3 * n_received / 4) / 2;
  }
  Span t(f, PK_T_RESAMPLE);
  const size_t stride = record_stride(f);
  launch_bal_adopt(f->stream, f->d, f->bal, m, static_cast<const unsigned char*>(dev_recv), n_received, f->rlohi_dev, mode, stride,
                   f->grow_on ? f->anc : nullptr);
  if (f->grow_on)  // the bookkeeping follows: own children from this filter's arrays, adopted ones from their records' tails
    launch_grow_gather(f->stream, f->grow, f->anc, P, static_cast<const unsigned char*>(dev_recv), stride, kPoseRecordBytes + f->d.lay.slot_bytes);
  PK_LAUNCH_CHECK("pk_shard_adopt_balanced_dev");
  f->src_identity = false;
  f->gmax_fused = false;
  if (mode == 1) f->adopt_local_done = true;
  return PK_OK;
}

/* consistency failures the balanced kernels counted since the filter was made (a logical index out of range, received
 * records that do not tile the free slots): 0 unless the plan and the exchange disagree */
int pk_shard_balanced_errors(pk_filter* f, int64_t* count) {
  if (!f || !count) return fail(PK_ERR_INVALID, "pk_shard_balanced_errors: NULL argument");
  *count = 0;
  if (!f->bal.bad) return PK_OK;
  int rc;
  if ((rc = use_device(f))) return rc;
  int v = 0;
  PK_HIP(hipMemcpyAsync(&v, f->bal.bad, sizeof(int), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  *count = v;
  return PK_OK;
}

// ---- probe ------------------------------------------------------------------------------
int pk_probe(int32_t device, const double pose[3], const double mean[5], const double cov[25],
             const double blob[4], const double Qt[16], double* out) {
  if (!pose || !mean || !cov || !blob || !Qt || !out) return fail(PK_ERR_INVALID, "pk_probe: NULL argument");
  int ndev = pk_device_count();
  if (ndev <= 0) return fail(PK_ERR_HIP, "pk_probe: no HIP device visible (the HIP path has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(PK_ERR_INVALID, "pk_probe: device %d of %d", device, ndev);
  const int cls = classify_covariance(cov, 0);
  if (cls < 0) return cls;
  for (int i = 0; i < 16; ++i)
    if (!std::isfinite(Qt[i])) return fail(PK_ERR_INVALID, "pk_probe: Qt is not finite");
  bool dense = cls == 1;
  {
    double scale = 0;
    for (int i = 0; i < 4; ++i) scale = fmax(scale, fabs(Qt[i * 5]));
    for (int j = 1; j < 4; ++j)
      if (fabs(Qt[j]) > 1e-14 * scale || fabs(Qt[j * 4]) > 1e-14 * scale) dense = true;
    for (int i = 1; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j)
        if (fabs(Qt[i * 4 + j] - Qt[j * 4 + i]) > 1e-9 * (scale > 0 ? scale : 1.0)) dense = true;
  }
  PK_HIP(hipSetDevice(device));
  double in[55];
  memcpy(in, pose, 3 * 8);
  memcpy(in + 3, mean, 5 * 8);
  memcpy(in + 8, cov, 25 * 8);
  memcpy(in + 33, blob, 4 * 8);
  memcpy(in + 37, Qt, 16 * 8);
  blob_directions(blob, 1, in + 53);
  double* dev = nullptr;
  PK_HIP(hipMalloc((void**)&dev, (55 + PK_PROBE_LEN) * sizeof(double)));
  hipError_t e = hipMemcpy(dev, in, sizeof(in), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    if (dense)
      launch_probe_dense(nullptr, dev, dev + 55);  // xy-rgb coupling or a coupled Qt: the dense formulas
    else
      launch_probe(nullptr, dev, dev + 55);
    e = hipMemcpy(out, dev + 55, PK_PROBE_LEN * sizeof(double), hipMemcpyDeviceToHost);
  }
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(PK_ERR_HIP, "pk_probe: %s", hipGetErrorString(e));
  return PK_OK;
}

#ifdef PK_STAMPS
__attribute__((visibility("default"))) int pk_debug_pub_wave_stamps(unsigned long long* out, int reset) {
  pk::debug_read_pub_wave_stamps(out, reset != 0);  // out[8][12]: k_step_pub's sums per wave of the workgroup
  return PK_OK;
}
__attribute__((visibility("default"))) int pk_debug_stamps(unsigned long long* out, int reset) {
  pk::debug_read_stamps(out, reset != 0);            // out[0..15]: k_assoc_grid
  pk::debug_read_fused_stamps(out + 16, reset != 0);  // out[16..31]: k_step_fused
  pk::debug_read_regs_stamps(out + 32, reset != 0);   // out[32..47]: k_step_regs
  pk::debug_read_pub_stamps(out + 48, reset != 0);    // out[48..63]: k_step_pub
  return PK_OK;
}
#endif

// ---- instrumentation ----------------------------------------------------------------------
int pk_enable_timing(pk_filter* f, int32_t on) {
  if (!f) return fail(PK_ERR_INVALID, "pk_enable_timing: NULL handle");
  int rc;
  if ((rc = use_device(f))) return rc;
  if ((rc = drain_timings(f))) return rc;
  f->timing_mask = on < 0 ? 0xffffffffu : (uint32_t)on;
  return PK_OK;
}
int pk_reset_timings(pk_filter* f) {
  if (!f) return fail(PK_ERR_INVALID, "pk_reset_timings: NULL handle");
  int rc;
  if ((rc = use_device(f))) return rc;
  if ((rc = drain_timings(f))) return rc;
  for (int i = 0; i < PK_T_COUNT; ++i) {
    f->ms[i] = 0;
    f->launches[i] = 0;
    f->timing_seen[i] = 0;
  }
  return PK_OK;
}
int pk_timings(pk_filter* f, double ms[PK_T_COUNT], int64_t launches[PK_T_COUNT]) {
  if (!f || !ms || !launches) return fail(PK_ERR_INVALID, "pk_timings: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  if ((rc = drain_timings(f))) return rc;
  for (int i = 0; i < PK_T_COUNT; ++i) {
    ms[i] = f->ms[i];
    launches[i] = f->launches[i];
  }
  return PK_OK;
}
int pk_observe_route(const pk_filter* f) { return f ? f->route : PK_ROUTE_NONE; }
int pk_observe_flagged(pk_filter* f, int64_t* flagged, int64_t* cand_overflow) {
  if (!f) return fail(PK_ERR_INVALID, "pk_observe_flagged: NULL handle");
  int rc;
  if ((rc = use_device(f))) return rc;
  unsigned w[2] = {0u, 0u};
  if (f->scan_dev && f->route != PK_ROUTE_NONE && f->route != PK_ROUTE_KNOWN_IDS) {
    PK_HIP(hipMemcpyAsync(w, ctl_n_flagged(f), sizeof(w), hipMemcpyDeviceToHost, f->stream));
    PK_HIP(hipStreamSynchronize(f->stream));
  }
  if (flagged) *flagged = w[0];
  if (cand_overflow) *cand_overflow = w[1];
  return PK_OK;
}
/* second-chance rows: how many the last scan wanted (particles the one-pass kernel flagged) and how many there are */
int pk_observe_retry_rows(pk_filter* f, int64_t* wanted, int64_t* capacity) {
  if (!f) return fail(PK_ERR_INVALID, "pk_observe_retry_rows: NULL handle");
  int rc;
  if ((rc = use_device(f))) return rc;
  PK_HIP(hipStreamSynchronize(f->stream));
  if (wanted) *wanted = f->retry_seen ? (int64_t)*f->retry_seen : 0;
  if (capacity) *capacity = retry_rows(f);
  return PK_OK;
}
int pk_observe_flags(pk_filter* f, uint8_t* flags) {
  if (!f || !flags) return fail(PK_ERR_INVALID, "pk_observe_flags: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  const bool onepass = f->route == PK_ROUTE_ML_REGS || f->route == PK_ROUTE_ML_FUSED || f->route == PK_ROUTE_ML_PUB_BIG;
  if (onepass && f->fh.pflag) {
    PK_HIP(hipMemcpyAsync(flags, f->fh.pflag, (size_t)f->d.P, hipMemcpyDeviceToHost, f->stream));
    PK_HIP(hipStreamSynchronize(f->stream));
  } else {
    memset(flags, 0, (size_t)f->d.P);
  }
  return PK_OK;
}
int pk_observe_published(pk_filter* f, int32_t* published) {
  if (!f || !published) return fail(PK_ERR_INVALID, "pk_observe_published: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  *published = 0;
  if (f->scan_dev && (f->route == PK_ROUTE_ML_REGS || f->route == PK_ROUTE_ML_FUSED || f->route == PK_ROUTE_ML_PUB_BIG) && f->pub_ecap > 0) {
    unsigned w = 1u;
    PK_HIP(hipMemcpyAsync(&w, ctl_skip_pub(f), sizeof(w), hipMemcpyDeviceToHost, f->stream));
    PK_HIP(hipStreamSynchronize(f->stream));
    *published = w == 0u ? 1 : 0;
  }
  return PK_OK;
}
int pk_observe_pub_stats(pk_filter* f, int64_t stats[6]) {
  if (!f || !stats) return fail(PK_ERR_INVALID, "pk_observe_pub_stats: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  for (int i = 0; i < 6; ++i) stats[i] = 0;
  if (f->scan_dev && (f->route == PK_ROUTE_ML_REGS || f->route == PK_ROUTE_ML_FUSED || f->route == PK_ROUTE_ML_PUB_BIG) && f->pub_ecap > 0) {
    unsigned w[4] = {0u, 0u, 0u, 0u}, sk[2] = {1u, 1u};
    PK_HIP(hipMemcpyAsync(w, ctl_pub_stats(f), sizeof(w), hipMemcpyDeviceToHost, f->stream));
    PK_HIP(hipMemcpyAsync(&sk[0], ctl_skip_pub(f), sizeof(unsigned), hipMemcpyDeviceToHost, f->stream));
    PK_HIP(hipMemcpyAsync(&sk[1], ctl_skip_duo(f), sizeof(unsigned), hipMemcpyDeviceToHost, f->stream));
    PK_HIP(hipStreamSynchronize(f->stream));
    for (int i = 0; i < 4; ++i) stats[i] = w[i];
    stats[4] = f->pub_ecap;
    // which instance worked on the scan: 0 none of the publish / subscribe kernels, 1 the one-workgroup-per-CU instance, 2 k_step_pub_duo
    stats[5] = sk[0] != 0u ? 0 : (f->route == PK_ROUTE_ML_PUB_BIG && f->duo_on && sk[1] == 0u) ? 1 + f->duo_on : 1;
  }
  return PK_OK;
}
int pk_download_sources(pk_filter* f, int32_t* src) {
  if (!f || !src) return fail(PK_ERR_INVALID, "pk_download_sources: NULL argument");
  int rc;
  if ((rc = use_device(f))) return rc;
  PK_HIP(hipMemcpyAsync(src, f->d.src[f->d.cur], (size_t)f->d.P * sizeof(int32_t), hipMemcpyDeviceToHost, f->stream));
  PK_HIP(hipStreamSynchronize(f->stream));
  return PK_OK;
}
int pk_observe_bytes(const pk_filter* f, int32_t B, int64_t* algorithmic, int64_t* moved) {
  if (!f) return fail(PK_ERR_INVALID, "pk_observe_bytes: NULL handle");
  const int64_t P = f->d.P, L = f->d.lay.L;
  if (algorithmic) *algorithmic = P * L * 28 * (int64_t)sizeof(double);  // 14 read + 14 written
  if (moved) *moved = P * (L * (28 * (int64_t)sizeof(double) + 8) + (int64_t)B * 32);
  return PK_OK;
}

}  // extern "C"
