/*
 * parakeet_slam.h -- C ABI of the MI355X-native FastSLAM-1.0 particle update.
 *
 * This is the drop-in boundary for the hot path of buckbaskin/parakeet_slam:
 * the per-timestep particle update in src/prkt_core_v2.py (+ src/matrix.py).
 * The reference has no FFI of its own -- its boundary is the Python class
 * surface FastSLAM / FilterParticle / Feature -- so each entry point below
 * names the reference method (file:line) whose arithmetic it replaces; the
 * Python facade in parakeet_slam_amd/core.py keeps the class surface and calls
 * these through ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 (PK_OK) or a negative pk_status; the message of
 *     the most recent failure on the calling thread is pk_last_error().
 *   - the caller owns every host buffer; the library owns every device buffer.
 *   - one caller thread per handle (the facade serialises with a mutex; the
 *     reference itself races here, prkt_ros.py:113-121 vs prkt_core_v2.py:162).
 *   - all host-side numbers are float64, the reference's arithmetic type.
 *   - kernels are enqueued on the handle's HIP stream; entry points that return
 *     data to the host synchronise that stream, the others do not.
 *   - landmark ids are the reference's: 1..L in preset order, 0 = "no match"
 *     (prkt_core_v2.py:294-299, :369-381).
 *
 * State layout in HBM: see DESIGN.md section 3.
 */
#ifndef PARAKEET_SLAM_H_
#define PARAKEET_SLAM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */
#endif

#define PK_ABI_VERSION 1

typedef enum pk_status {
  PK_OK = 0,
  PK_ERR_INVALID = -1,     /* bad argument (null pointer, size, non-finite input) */
  PK_ERR_HIP = -2,         /* a HIP runtime call failed; pk_last_error() has the text */
  PK_ERR_STATE = -3,       /* call order violated (e.g. observe before a map upload) */
  PK_ERR_UNSUPPORTED = -4, /* valid request the device path does not implement */
  PK_ERR_NOMEM = -5        /* host or device allocation failed */
} pk_status;

/* weight_domain of pk_resample */
/* Flag in a landmark's count word (pk_upload_landmarks / pk_download_landmarks): a POTENTIAL feature -- the reference's
 * negative ids, prkt_core_v2.py:109-118: matched and updated like any landmark, but a match multiplies the particle's
 * weight by 0.1 instead of the importance factor; the kernels clear the flag when the update count passes 5 (:113-117). */
#define PK_LANDMARK_POTENTIAL 0x40000000
#define PK_WEIGHTS_LINEAR 0 /* w = exp(logw): the reference's quantity, underflows like it */
#define PK_WEIGHTS_LOG 1    /* w = exp(logw - max logw): same ancestors unless the former underflows */

/* kernel slots of pk_timings */
enum {
  PK_T_MOTION = 0,
  PK_T_ASSOC = 1,
  PK_T_OBSERVE = 2,
  PK_T_WEIGHTS = 3,
  PK_T_RESAMPLE = 4,
  PK_T_SUMMARY = 5,
  PK_T_MATERIALISE = 6,
  PK_T_COUNT = 7
};

typedef struct pk_filter pk_filter; /* opaque */

int pk_abi_version(void);
const char* pk_status_string(int status);
const char* pk_last_error(void);

/* Number of HIP devices visible to this process (0 when there is none). */
int pk_device_count(void);

/* ---- life cycle -------------------------------------------------------------
 * FastSLAM.__init__ (prkt_core_v2.py:38-57) + FilterParticle.__init__ (:279-292):
 * P particles at pose (0,0,0), weight 1, Qt = 0.1*I4, empty map of capacity L.
 * The reference hard-codes P = 50 (:41); here it is a parameter. */
int pk_create(int64_t num_particles, int32_t num_landmarks, int32_t device, pk_filter** out);
int pk_destroy(pk_filter* f);

/* Use a caller-provided hipStream_t (e.g. torch's current stream); NULL restores the
 * handle's own stream. */
int pk_set_stream(pk_filter* f, void* hip_stream);
int pk_synchronize(pk_filter* f);
int64_t pk_num_particles(const pk_filter* f);
int32_t pk_num_landmarks(const pk_filter* f);
/* Bytes of HBM the handle holds (maps are double-buffered). */
int64_t pk_device_bytes(const pk_filter* f);

/* Tuning knobs that never change results:
 *   "assoc_kernel" = 0 (colour-grid association kernel, default) or 1 (brute-force reference
 *                    kernel that gate-tests every landmark x blob pair);
 *   "assoc_dup"    = 1 (default: 9x column-duplicated blob index list when it fits in LDS) or 0;
 *   "fast_observe" = 1 (default: contested associations are settled inside the EKF kernel --
 *                    k_observe_fast with the landmark state in registers for L <= 512,
 *                    k_observe_sweep in two sweeps over landmark chunks above that), 0 (general
 *                    path: association kernel writes ids), 2 (k_observe_sweep for every L) or 3 (the same with
 *                    eight hand-off slots per landmark, the default only for scans of >= 3000 blobs);
 *   "timing_stride" = 1 (default) .. n: with pk_enable_timing, bracket only every n-th launch of a
 *                    slot (two event records cost the stream a few microseconds each time);
 *   "upload_kernel" = 1 (default: the per-scan block is read from pinned host memory by a small
 *                    kernel in stream order) or 0 (hipMemcpyAsync);
 *   "fused_step"   = 1 (default: with "fast_observe" = 1, L <= 512 and scan tables that fit LDS twice
 *                    per CU, gates + settling + EKF update of a particle run in ONE kernel,
 *                    k_step_fused, without the hand-off through HBM) or 0;
 *   "regs_step"    = 1 (default: with "fast_observe" = 1, 512 < L <= 2048 and scan tables that fit LDS,
 *                    gates + settling + EKF update of a particle run in ONE pass, k_step_regs, with the
 *                    particle's whole map in registers) or 0 (k_assoc_grid hand-off + k_observe_sweep);
 *   "pub_step"     = 1 (default: the register route settles contested blobs by static publish / subscribe -- k_step_pub,
 *                    the inverse candidate lists lay a per-scan table out in LDS, every landmark publishes its verdicts
 *                    there, one lane per contested blob picks the winner; two landmark pairs per lane, 512 lanes -- while
 *                    the table fits LDS and no candidate list overflows; decided per scan on the device,
 *                    pk_observe_published) or 0 (k_step_regs: per-blob counters, probability queue, bids);
 *   "pub_duo"      = 0 (default), 1 or 2: maps of 2 049 .. 5 120 landmarks: a scan whose publish table fits its share of a CU's LDS is
 *                    worked on by k_step_pub_duo -- the two-pass kernel k_step_pub_big with ONE landmark at a time and one carried
 *                    word per landmark: 1 = 512 lanes, at most 128 VGPRs, TWO workgroups per CU; 2 = 256 lanes with a pair of
 *                    landmarks each, at most 168 VGPRs, THREE workgroups per CU -- so that one workgroup's memory waits are the
 *                    others' arithmetic; decided per scan on the device (pk_observe_pub_stats).  Measured slower than one
 *                    workgroup per CU (twice / three times the particles in flight no longer find their second read of the
 *                    map in the Infinity Cache: DESIGN.md section 4), hence off.  "pub_duo_park_limit" >= 0 (tests) treats its
 *                    overflow area -- where a landmark with several blobs of probability > 0 parks its slots between the
 *                    passes -- as that many places;
 *   "far_prune"    = 1 (default) or 0: once per scan, the look-alikes whose match probability is certainly 0 for every particle
 *                    (a key beyond the float64 underflow edge by the reference particle's bound with margins) leave the candidate
 *                    lists of the publish / subscribe kernels; a landmark whose own bound is weaker re-checks them itself
 *                    (prkt_core_v2.py:369: probability 0 never matches).  0: every particle tests and judges them (round 4).
 *   "pub_small"    = -1 (default), 0 or 1: maps of at most 512 landmarks through k_step_pub's 256-lane instance (three workgroups per
 *                    CU, candidate lists made once per scan) instead of k_step_fused.  Its kernel is 11-18 % faster, its per-scan
 *                    kernels cost 27 us whatever the number of particles: -1 takes it where the whole step was measured no slower --
 *                    filters of at least 5e6 particle.landmarks and 128 landmarks (BASELINE configs[1], 10 000 x 500, is the tie;
 *                    DESIGN.md section 10.3) --, 1 always, 0 never;
 *   "pub_entry_limit" = 0 (default: what LDS holds) or n: treat the publish table as n entries small (tests: scans
 *                    whose table does not fit fall back to k_step_regs);
 *   "cand_lists"   = 1 (default: k_step_regs tests each landmark against the candidate list of a reference particle
 *                    -- k_candidates, once per scan -- instead of walking the colour grid; particles outside the
 *                    list's margins go the general way) or 0 (grid walk);
 *   "split_reserve_cus" = 0..128 (default 16): CUs the first part of a split step (pk_observe_staged_range, first = 1,
 *                    last = 0) leaves without a workgroup -- k_step_regs holds a CU's whole register file for the whole
 *                    launch, and the all-to-all that is to run meanwhile needs CUs of its own;
 *   "regs_retry"   = 1 (default: particles k_step_regs flags -- a landmark passing more than four blobs -- get a second
 *                    chance on the eight-slot hand-off + k_observe_sweep before the general kernels) or 0;
 *   "regs_warm"    = 0..2: how much of the NEXT particle's map slot k_step_regs touches ahead of time
 *                    (0 nothing, 1 the mean rows -- the default --, 2 the whole slot: measured slower) so that it waits in L2;
 *   "observe_landmarks_per_lane" = 0 (default), 1 or 2  (process-wide). */
int pk_set_option(pk_filter* f, const char* name, int64_t value);

/* ---- configuration ----------------------------------------------------------
 * FastSLAM.Qt (prkt_core_v2.py:50-53), row-major 4x4.  Qt = [q00] (+) [3x3 symmetric] runs on the compact layout and
 * the fast kernels; any other finite Qt (bearing-colour coupling, asymmetry) switches the filter to the dense layout and
 * the general dense kernel (PK_ROUTE_DENSE), converting maps that are already loaded. */
int pk_set_measurement_noise(pk_filter* f, const double Qt[16]);

/* FilterParticle.load_feature_list (prkt_core_v2.py:294-299) for every particle:
 * means[L*5] (x,y,r,g,b), covs[L*25] row-major 5x5, immutable[L] = Feature.__immutable__
 * (:883; NULL = all mutable).  update_count starts at 0.  Symmetric, block-diagonal covariances
 * (xy 2x2 (+) rgb 3x3 -- the structure the reference's update preserves) take the compact
 * 14-row layout and the fast kernels; a map in which ANY covariance couples position and colour or is not
 * symmetric takes the dense 30-row layout (the reference's full 5 + 25 state) and the general dense kernel
 * (PK_ROUTE_DENSE: the reference's 4x4 / 5x4 / 5x5 algebra entry by entry, :804-833, :897-930) -- correct, slow. */
int pk_upload_map(pk_filter* f, const double* means, const double* covs, const uint8_t* immutable);

/* Poses as rows (x, y, heading, weight); weight is the linear particle weight
 * (FilterParticle.weight, :288), stored on the device as its natural log. */
int pk_upload_poses(pk_filter* f, const double* xyhw);
int pk_download_poses(pk_filter* f, double* xyhw);
/* One particle's pose and weight -- FastSLAM.particles[i] = p, prkt_core_v2.py:162 -- without touching the other particles'
 * log-weights (which a download / upload of all poses sends through exp and log: every weight that underflows comes back as 0). */
int pk_upload_pose(pk_filter* f, int64_t particle, const double xyhw[4]);
/* The natural logarithms of the particle weights (the quantity the filter keeps: with B >~ 1000
 * blobs per scan the weights themselves underflow float64, prkt_core_v2.py:95,124). */
int pk_download_log_weights(pk_filter* f, double* logw);

/* Landmark state of particles [p0, p1): means (n*L*5), covs (n*L*25 dense 5x5),
 * counts (n*L) = Feature.update_count.  Any output/input pointer may be NULL. */
int pk_download_landmarks(pk_filter* f, int64_t p0, int64_t p1, double* means, double* covs,
                          int32_t* counts);
int pk_upload_landmarks(pk_filter* f, int64_t p0, int64_t p1, const double* means,
                        const double* covs, const int32_t* counts);

/* ---- the step ---------------------------------------------------------------
 * particles[i].weight = 1 (prkt_core_v2.py:73). */
int pk_reset_weights(pk_filter* f);

/* FastSLAM.motion_update/motion_model (prkt_core_v2.py:148-208) for every particle.
 * z != NULL: host array P*3 of standard normals in the reference's draw order
 *            (drive, heading-1, heading-2 per particle) -- fixed-seed parity mode;
 * z == NULL: device counter-based generator (Philox4x32-10) keyed by (seed, draw). */
int pk_motion(pk_filter* f, double v, double w, double dt, const double* z, uint64_t seed,
              uint64_t draw);

/* The per-particle body of FastSLAM.cam_cb (prkt_core_v2.py:82-124): data association
 * (match_features_to_scan :317-381, probability_of_match :383-455), then per blob in scan
 * order generate_measurement :859, measurement_jacobian :748, measurement_covariance :804,
 * inverse matrix.py:11, kalman_gain :821, Feature.update_mean :897 / update_covar :916,
 * importance_factor :835 or no_match_weight :851, multiplied into the particle weight.
 *   blobs: B rows (bearing, r, g, b)  (matrix.blob_to_matrix, matrix.py:35-39)
 *   ids:   NULL -> maximum-likelihood association on the device, per particle;
 *          else B landmark ids (0 = unmatched) applied to every particle.
 *   ids_out: NULL or P*B int32 receiving the ids each particle used. */
int pk_observe(pk_filter* f, const double* blobs, int32_t num_blobs, const int32_t* ids,
               int32_t* ids_out);
/* pk_reset_weights + pk_observe in one pass over the particles: every weight restarts from 1
 * (prkt_core_v2.py:73) before the scan is applied -- what cam_cb does, and what pk_step uses. */
int pk_observe_fresh(pk_filter* f, const double* blobs, int32_t num_blobs, const int32_t* ids,
                     int32_t* ids_out);
/* The same in two halves, for callers that have host work to hide: pk_stage_scan does the HOST
 * part of a maximum-likelihood observe (checks, ray directions, association tables, all into a
 * pinned staging slot; no kernel is launched), pk_observe_staged enqueues the upload and the
 * kernels for the scan staged last (fresh != 0: weights restart from 1 first).  Any other
 * observe / associate call in between discards the staged scan. */
int pk_stage_scan(pk_filter* f, const double* blobs, int32_t num_blobs);
int pk_observe_staged(pk_filter* f, int32_t fresh);

/* Data association alone (FilterParticle.match_features_to_scan, prkt_core_v2.py:317-351):
 * ids_out[P*B] receives, per particle, the landmark id each blob matches (0 = none).
 * No state changes. */
int pk_associate(pk_filter* f, const double* blobs, int32_t num_blobs, int32_t* ids_out);

/* ---- new landmarks: FilterParticle.add_hypothesis / find_nearest_reading / ray_intersect /
 * color_distance / add_new_feature / cross_readings / add_orphaned_reading (prkt_core_v2.py:546-746) for every particle on the
 * device.  The filter's landmarks beyond the first `preset_landmarks` are SPARE slots (upload them with a colour no blob can gate
 * against).  From then on every maximum-likelihood observe (pk_observe / pk_observe_fresh / pk_step without ids) ends with one
 * kernel that walks each particle's unmatched blobs in scan order (:92-95): a blob pairs with the particle's nearest stored reading
 * -- rays crossing, colour distance below pair_threshold; the one change to the reference, whose find_nearest_reading (:579) walks
 * potential_features where the readings are in hypothesis_set and so never pairs anything -- and becomes a potential feature at
 * the crossing in the next spare slot (mean colour, identity covariance, count word PK_LANDMARK_POTENTIAL, :656-686); else it is
 * stored as an orphaned reading (:739-746).  Each particle keeps up to reading_capacity readings (the reference's dict grows
 * without bound; what does not fit is counted, see below).  pk_resample carries the bookkeeping along with the particles (:243).
 * Supplied ids and the dense layout are refused while it is on (PK_ERR_STATE). */
int pk_grow_enable(pk_filter* f, int32_t preset_landmarks, int32_t reading_capacity, double pair_threshold);
/* The bookkeeping of particles [p0, p1) (any pointer may be NULL): counters[n][4] = readings stored, spare slots in use, next_id
 * (:298), readings dropped because the ring was full; readings[n][reading_capacity][8] = id, x, y, heading, bearing, r, g, b of each
 * stored reading (the rows beyond `readings stored` are undefined); slot_ids[n][spare] = the feature id of each spare slot in use
 * (potential: the reference's -id, promoted: +id). */
int pk_grow_download(pk_filter* f, int64_t p0, int64_t p1, int32_t* counters, double* readings, int32_t* slot_ids);
int pk_grow_upload(pk_filter* f, int64_t p0, int64_t p1, const int32_t* counters, const double* readings, const int32_t* slot_ids);
int pk_grow_shape(const pk_filter* f, int32_t* preset_landmarks, int32_t* spare_slots, int32_t* reading_capacity);

/* FastSLAM.low_variance_resample (prkt_core_v2.py:210-252) with step = u * sum/P, u in
 * [0,1) standing for random.random() (:226).  ancestors_out: NULL or P int64. */
int pk_resample(pk_filter* f, double u, int32_t weight_domain, int64_t* ancestors_out);

/* FastSLAM.summary (prkt_core_v2.py:254-276): (mean x, mean y, circular mean heading). */
int pk_summary(pk_filter* f, double out[3]);

/* The four sums FastSLAM.summary reduces (prkt_core_v2.py:267-271): sum x, sum y, sum sin h,
 * sum cos h -- what a sharded filter all-reduces before dividing / atan2. */
int pk_pose_sums(pk_filter* f, double out[4]);

/* One whole cam_cb without host synchronisation: reset weights, motion, observe,
 * resample.  Arguments as in the calls above. */
int pk_step(pk_filter* f, double v, double w, double dt, const double* z, uint64_t seed,
            uint64_t draw, const double* blobs, int32_t num_blobs, const int32_t* ids, double u,
            int32_t weight_domain);

/* ---- sharded (multi-GPU) resampling: see DESIGN.md section 6 -------------------
 * Particles are sharded contiguously over one process per GPU; global particle index =
 * global_offset + local index.  The collectives themselves are the caller's
 * (torch.distributed over RCCL); the entry points below only read/write host scalars and
 * arrays, plus one caller-owned device buffer for the particles that migrate.
 * pk_set_shard: index of local particle 0 in the whole filter (keys the Philox counters so
 * the motion noise does not depend on the shard count). */
int pk_set_shard(pk_filter* f, int64_t global_offset);
/* max over the shard of log(weight). */
int pk_shard_max_logw(pk_filter* f, double* max_logw);
/* Number of weight-scan blocks of the shard (1024 particles each) and their totals of
 * exp(logw - gmax) (PK_WEIGHTS_LOG) or exp(logw) (PK_WEIGHTS_LINEAR); also leaves the
 * block-local inclusive scans on the device for pk_shard_offspring. */
int64_t pk_shard_num_blocks(const pk_filter* f);
int pk_shard_block_totals(pk_filter* f, double gmax, int32_t weight_domain, double* totals);
/* Given every shard's block totals concatenated in rank order, this shard's first block and
 * the global particle count: slot_hi[0] = number of output slots filled by all earlier
 * shards, slot_hi[1 + j] = that number after local particle j; i.e. local particle j is the
 * ancestor of the global output slots [slot_hi[j], slot_hi[j + 1]).  Same arithmetic as
 * pk_resample, so G shards reproduce the 1-GPU ancestors when shards are multiples of 1024.
 * last_shard: this is the highest rank (its last particle absorbs the clamped tail). */
int pk_shard_offspring(pk_filter* f, const double* global_totals, int64_t n_global_blocks,
                       int64_t first_block, int64_t global_particles, double u, int32_t last_shard,
                       int64_t* slot_hi);
/* Bytes of one migrating particle record: a 64-byte header (x, y, heading, log weight, slot_lo, slot_hi, the logical index of the
 * child in slot_lo, 0) + its landmark slot + -- while the new-landmark bookkeeping is on (pk_grow_enable) -- the particle's
 * counters, spare-slot ids and stored readings behind the slot. */
int64_t pk_particle_bytes(const pk_filter* f);
/* Pack the listed local particles into dev_buf (device pointer owned by the caller, e.g. a
 * torch tensor), n records of pk_particle_bytes. */
int pk_pack_particles(pk_filter* f, const int64_t* local_idx, int64_t n, void* dev_buf);
/* New generation of the shard: slot k takes local particle src[k] (>= 0) or received record
 * -(src[k]) - 1 of dev_buf.  Poses are gathered now; adopted landmark slots are read in place
 * from dev_buf by the next pk_observe / pk_associate / download, so dev_buf must stay alive
 * and unchanged until one of those has run. */
int pk_adopt_particles(pk_filter* f, const int64_t* src, const void* dev_buf, int64_t n_received);

/* Device-resident variants of the same protocol: every buffer is a DEVICE pointer owned by the
 * caller (torch tensors the collectives run on), nothing is copied to the host and nothing
 * synchronises the stream.  plan: leaves the offspring table on the device and writes
 * dev_ranges[2 d], [2 d + 1] = the contiguous local particles [j0, j1) whose offspring overlap
 * rank d's output slots.  pack: records of those particles for every rank but `rank`, in rank
 * order (`ranges` is the host copy of dev_ranges); each record header carries the destination
 * slots [lo, hi) it fills, so no per-particle metadata travels separately.  adopt: builds the
 * new generation from the local table and the received records (source-rank order). */
int pk_shard_max_logw_dev(pk_filter* f, double* dev_out);
int pk_shard_block_totals_dev(pk_filter* f, const double* dev_gmax, int32_t weight_domain,
                              double* dev_totals);
int pk_shard_plan_dev(pk_filter* f, const double* dev_global_totals, int64_t n_global_blocks,
                      int64_t first_block, int64_t global_particles, double u, int32_t last_shard,
                      int32_t world, int64_t* dev_ranges);
/* Debug / test entry (round 4; nothing in the reference: the exchange belongs to the multi-GPU resample of prkt_core_v2.py:210-252):
 * pack the local particles [j0, j1) of the planned resample as exchange records whose slot ranges are clipped to the GLOBAL slots
 * [slot_lo, slot_hi) -- what pk_shard_pack_dev does per destination rank, for an arbitrary slot interval.  With the options
 * "split_loopback_lo" / "split_loopback_hi" (local slot bounds; pk_set_option) a single rank can send records of its OWN
 * particles through the all-to-all to itself and adopt them with pk_shard_adopt_remote_dev: pack -> all_to_all_single(async) ->
 * wait -> adopt with records that really travelled, on one device (tests/test_gpu_sharded.py). */
int pk_shard_pack_slots_dev(pk_filter* f, int64_t j0, int64_t j1, int64_t slot_lo, int64_t slot_hi, void* dev_buf);
/* The same plan for shards of ANY size (the block-total plan above reproduces the 1-GPU ancestors only when shards are
 * multiples of the 1024-particle scan block): the ranks all-gather their log-weights (pk_shard_logw_dev copies the shard's
 * into a caller buffer), and every rank runs the 1-GPU scan kernels on the whole array -- same blocks, same additions,
 * same bits on every rank and as on one GPU -- then derives its own particles' output slots.  8 B per particle of the
 * whole filter travel per resample (the migrating particles' maps are 10^4 times that).
 * pk_shard_download_offspring: the plan's table slot_hi[P + 1] (as pk_shard_offspring returns it), for tests. */
int pk_shard_logw_dev(pk_filter* f, double* dev_out);
int pk_shard_plan_global_dev(pk_filter* f, const double* dev_global_logw, int64_t global_particles, const double* dev_gmax,
                             int32_t weight_domain, double u, int32_t last_shard, int32_t world, int64_t* dev_ranges);
int pk_shard_download_offspring(pk_filter* f, int64_t* slot_hi);
int pk_shard_pack_dev(pk_filter* f, const int64_t* ranges, int32_t world, int32_t rank, void* dev_buf);
int pk_shard_adopt_dev(pk_filter* f, int32_t rank, const void* dev_recv, int64_t n_received);

/* The split step of the sharded filter: the exchange of the migrating particles (each a whole map) runs while the
 * particles that stay on this rank are already being worked on.  The output slots a shard fills with its OWN particles form
 * one contiguous run (ancestors are monotone in the slot index): pk_shard_local_span_dev copies its two global bounds
 * (slot_hi[0], slot_hi[P] of the plan) into a caller's device buffer; pk_shard_adopt_local_dev makes the new generation
 * current with those slots filled, pk_shard_adopt_remote_dev fills the others from the received records;
 * pk_motion_range / pk_observe_staged_range are pk_motion (device noise) / pk_observe_staged on the particles [p0, p1) only:
 * `first` consumes the staged scan (it must take the register route: pk_staged_takes_regs), `last` runs what the one-pass
 * kernel flagged over ALL particles and closes the step.  The reference has one process (prkt_core_v2.py:59-137). */
int pk_shard_local_span_dev(pk_filter* f, int64_t* dev_out2);
int pk_shard_adopt_local_dev(pk_filter* f, int32_t rank);
int pk_shard_adopt_remote_dev(pk_filter* f, int32_t rank, const void* dev_recv, int64_t n_received);
int pk_motion_range(pk_filter* f, double v, double w, double dt, uint64_t seed, uint64_t draw, int64_t p0, int64_t p1);
int pk_staged_takes_regs(pk_filter* f); /* 1: the staged scan will take k_step_regs, 0: not (or nothing staged) */
int pk_observe_staged_range(pk_filter* f, int32_t fresh, int64_t p0, int64_t p1, int32_t first, int32_t last);

/* ---- balanced placement of the sharded filter: minimum migration (round 5) ----------------------
 * The exchange above gives rank r the output slots [r P, (r + 1) P) of prkt_core_v2.py:233-250's ordered walk, so every rank
 * boundary moves by the cumulative imbalance of the ranks below it, and it packs every particle of a contiguous index
 * range, with or without children.  Here the ORDER of that walk is decoupled from where a particle lives: every physical
 * slot carries the LOGICAL index of its particle -- its index in one filter holding all of them; the Philox counters of
 * pk_motion and the weight scan are keyed by it, so the results stay those of one filter, bit for bit -- a rank keeps its
 * own children (in its slots [0, m)), and only a rank's EXCESS children travel, to whichever rank has free slots; only
 * particles that HAVE children there are packed.
 *   pk_shard_state_dev          this rank's row for the ONE all-gather of a resample: [logw(P) | logical(P) as int64 bits]
 *   pk_shard_plan_balanced_dev  from all ranks' rows (rank-major, 2 P words each): the 1-GPU weight scan in logical order, the
 *                               offspring table, and the whole plan -- every rank derives it by itself.  dev_table: world rows
 *                               of 2 world + 4 int64: per destination the range [a0, a1) of the row's rank's particles WITH
 *                               children that it sends there, then n (children), m = min(n, P), ebase, dbase (running sums of
 *                               the excess n - m and of the free slots P - m).  The caller reads the table on the host (it
 *                               sizes the all-to-all) and hands it back to the two calls below.
 *   pk_shard_pack_balanced_dev  the records for every other rank, destination-major; header = x, y, h, logw, lo, up, klo:
 *                               the copy fills the destination's slots [lo, up), the child in slot k is logical klo + k - lo
 *   pk_shard_adopt_balanced_dev mode 0: the whole new generation; 1: the slots [0, m) from this rank's own particles (the
 *                               generation becomes current); 2: the slots [m, P) from the received records (after mode 1:
 *                               the split step works on [0, m) while the records travel)
 *   pk_shard_download_logical / pk_shard_reset_placement (slot j <- logical pk_set_shard's offset + j) /
 *   pk_shard_download_balanced_offspring (the plan's table H[P_global + 1], tests) / pk_shard_balanced_errors (kernel-side
 *   consistency failures: must stay 0).
 * Once pk_shard_state_dev has run on a filter, pk_resample and the contiguous adoptions refuse it (PK_ERR_STATE). */
int pk_shard_state_dev(pk_filter* f, double* dev_out);
int pk_shard_plan_balanced_dev(pk_filter* f, const double* dev_global_state, int64_t global_particles, const double* dev_gmax,
                               int32_t weight_domain, double u, int32_t world, int32_t rank, int64_t* dev_table);
int pk_shard_pack_balanced_dev(pk_filter* f, const int64_t* table, int32_t world, int32_t rank, void* dev_buf);
int pk_shard_adopt_balanced_dev(pk_filter* f, const int64_t* table, int32_t world, int32_t rank, const void* dev_recv,
                                int64_t n_received, int32_t mode);
/* debug (one-rank tests of the balanced exchange over RCCL): with the option "balanced_loopback_keep" = keep set, the next balanced
 * adoption fills only the slots [0, keep) with the rank's own children; its children from position keep on travel as records -- packed
 * here for the particles alive[a0, a1) with the same 64-byte header (and bookkeeping tail) another rank would get -- through the
 * all-to-all to the rank itself and are adopted into [keep, P) from the receive buffer. */
int pk_shard_pack_balanced_loop_dev(pk_filter* f, int64_t keep, int64_t a0, int64_t a1, void* dev_buf);
int pk_shard_download_logical(pk_filter* f, int64_t* logical);
/* tests of the planner in isolation (tests/test_gpu_sharded.py): the placement set from the host -- slot j holds logical particle
 * logical[j] --, and this rank's tables of the last plan: rel[P + 1] (children of its particles [0, j)), Hl[P] (first output slot of
 * particle j's children), alive[P] (its particles with children, ascending; -1 behind the last) */
int pk_shard_upload_logical(pk_filter* f, const int64_t* logical);
int pk_shard_download_balanced_plan(pk_filter* f, int64_t* rel, int64_t* Hl, int32_t* alive);
int pk_shard_reset_placement(pk_filter* f);
int pk_shard_download_balanced_offspring(pk_filter* f, int64_t global_particles, int64_t* H);
int pk_shard_balanced_errors(pk_filter* f, int64_t* count);

/* ---- single-triple probe ------------------------------------------------------
 * Runs the device functions the kernels are built from on ONE (pose, landmark, blob):
 * the scalar methods of FilterParticle the reference's unit tests call.
 * out[PK_PROBE_LEN]:
 *   [0] probability_of_match :383      [1] prob_position_match :457
 *   [2..3] closest_point :496          [4] prob_color_match :524
 *   [5..8] generate_measurement :859   [9..10] measurement_jacobian H[0][0], H[0][1] :748
 *   [11..26] measurement_covariance Q 4x4 :804
 *   [27..46] kalman_gain K 5x4 :821    [47] importance_factor :835
 *   [48..52] updated mean :897         [53..77] updated covariance 5x5 :916
 *   [78] log importance factor */
#define PK_PROBE_LEN 79
int pk_probe(int32_t device, const double pose[3], const double mean[5], const double cov[25],
             const double blob[4], const double Qt[16], double* out);

/* ---- instrumentation ----------------------------------------------------------
 * Kernel launches of the enabled PK_T_* slots are bracketed by hipEvents on the handle's
 * stream: `mask` bit i enables slot i, a negative mask enables all, 0 disables.  pk_timings synchronises, then returns accumulated milliseconds and launch
 * counts per PK_T_* slot since the last pk_reset_timings. */
int pk_enable_timing(pk_filter* f, int32_t mask);
int pk_reset_timings(pk_filter* f);
int pk_timings(pk_filter* f, double ms[PK_T_COUNT], int64_t launches[PK_T_COUNT]);
/* Algorithmic HBM bytes of one observe launch (14 scalars read + 14 written per
 * particle.landmark) and the bytes the layout actually moves (adds update_count, ids). */
int pk_observe_bytes(const pk_filter* f, int32_t num_blobs, int64_t* algorithmic, int64_t* moved);
/* Which kernels the last pk_observe / pk_step used for association + EKF update (instrumentation):
 * PK_ROUTE_NONE before the first call. */
enum {
  PK_ROUTE_NONE = 0,
  PK_ROUTE_KNOWN_IDS = 1,    /* ids supplied: k_observe */
  PK_ROUTE_ML_GENERAL = 2,   /* association kernel writes ids, k_observe builds chains from them */
  PK_ROUTE_ML_HANDOFF = 3,   /* k_assoc_grid hand-off + k_observe_fast (L <= 512) */
  PK_ROUTE_ML_SWEEP = 4,     /* k_assoc_grid hand-off + k_observe_sweep (L > 512) */
  PK_ROUTE_ML_FUSED = 5,     /* k_step_fused: gates + settling + EKF update in one kernel (L <= 512) */
  PK_ROUTE_ML_REGS = 6,      /* k_step_regs: the same in one pass for 512 < L <= 2048, two landmarks per lane */
  PK_ROUTE_DENSE = 8,        /* k_observe_dense: the general dense path (covariances that couple position and colour, or a
                                coupled Qt): association, 4x4 / 5x5 update and weight in one slow, general kernel */
  PK_ROUTE_ML_PUB_BIG = 9,   /* k_step_pub_big: maps of 2 049 ... 6 144 landmarks -- publish / subscribe settling in two passes over the map
                              * (the second one from L2 / Infinity Cache); what it leaves goes through k_observe_sweep */
  /* (7 was k_step_owner, removed in round 3) */
};
int pk_observe_route(const pk_filter* f);
/* The map indirection of the live generation (instrumentation): src[P], the map slot each particle's
 * landmarks currently live in.  After pk_resample slot k holds src[ancestor(k)] (the lazy copy of
 * prkt_core_v2.py:236,:246: maps are not moved until the next observe rewrites them); the number of
 * DISTINCT values is the number of slots the next observe launch really has to fetch.  Negative
 * values name records of a sharded adoption buffer. */
int pk_download_sources(pk_filter* f, int32_t* src);
/* How the last maximum-likelihood observe went (instrumentation): flagged = particles the one-pass kernels
 * (k_step_fused / k_step_regs) handed to the general kernels (a landmark passing more blobs than it has slots, a
 * full probability queue, a particle outside the margins of the reference particle's candidate lists);
 * cand_overflow = landmarks of the reference particle with more candidate blobs than list slots (non-zero: the
 * scan took the grid walk instead of the lists).  Synchronises the stream. */
int pk_observe_flagged(pk_filter* f, int64_t* flagged, int64_t* cand_overflow);
/* The second chance of flagged particles (eight-slot hand-off + k_observe_sweep) has list rows for P / 16 particles (>= 1 024);
 * a scan that wanted more -- the one-pass kernel stood back from it as a whole -- sends the rest through the general kernels
 * ONCE: the rows grow to what the last scan wanted.  wanted: of the last finished scan; capacity: rows the next scan will find. */
int pk_observe_retry_rows(pk_filter* f, int64_t* wanted, int64_t* capacity);
/* Which instance of the register route (PK_ROUTE_ML_REGS) worked on the last scan (instrumentation; the choice is made
 * on the device, per scan): *published = 1 when k_step_pub did -- contested blobs (prkt_core_v2.py:353-381) settled by
 * static publish / subscribe through LDS, three barriers per particle -- 0 when k_step_regs did (the publish table did
 * not fit LDS, a candidate list overflowed, or "pub_step" is off).  Synchronises the stream. */
int pk_observe_published(pk_filter* f, int32_t* published);
/* What the last scan's publish table came to (instrumentation; written by k_cand_entries on the device, once per scan):
 * stats[0] entries of the table (every (landmark, blob) pair several landmarks contend for, prkt_core_v2.py:353-381),
 * [1] contested blobs, [2] landmarks of the reference particle with two or more blobs inside their own gates (:433, :441),
 * [3] the longest candidate list, [4] entries the LDS table was given, [5] which instance worked on the scan: 0 none of the
 * publish / subscribe kernels (the table did not fit, a list overflowed), 1 the one-workgroup-per-CU instance, 2 / 3 the
 * two- / three-workgroups-per-CU instance of the two-pass kernel (k_step_pub_duo, option "pub_duo" = 1 / 2).  All zero when the
 * last observe took another route.  Synchronises the stream. */
int pk_observe_pub_stats(pk_filter* f, int64_t stats[6]);
/* Per particle, how the last one-pass maximum-likelihood observe (k_step_fused / k_step_pub / k_step_pub_big / k_step_regs)
 * dealt with it (instrumentation; what the full-size oracle audits of tests/test_gpu_audit.py pick their samples by):
 * flags[P], 0 = settled by the one-pass kernel itself, 2 = redone by the second-chance route (eight-slot hand-off +
 * k_observe_sweep), 1 = redone by the general kernels.  The particles are match_features_to_scan's, prkt_core_v2.py:317-351,
 * whatever the route.  All zeros when the last observe took another route.  Synchronises the stream. */
int pk_observe_flags(pk_filter* f, uint8_t* flags);

/* ---- host-side reproductions of the reference's RNG streams (no GPU needed) ------
 * numpy.random.seed(s); numpy.random.normal(0,1,n)  (legacy MT19937 + polar method), the
 * stream prkt_core_v2.py:185-193 draws from; and random.seed(s); random.random() (:226). */
typedef struct pk_rng pk_rng;
int pk_rng_create_numpy(uint32_t seed, pk_rng** out);
int pk_rng_create_python(uint32_t seed, pk_rng** out);
int pk_rng_destroy(pk_rng* r);
int pk_rng_standard_normal(pk_rng* r, int64_t n, double* out);
int pk_rng_random(pk_rng* r, int64_t n, double* out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* PARAKEET_SLAM_H_ */
