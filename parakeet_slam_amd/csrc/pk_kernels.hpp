// Host-visible launchers of the gfx950 kernels (defined in pk_kernels.hip).
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
};

constexpr int kScanBlock = 1024;  // particles per weight-scan block (256 threads x 4)

// K1
void launch_motion(hipStream_t s, DeviceState& d, double v, double w, double dt, const double* z_dev,
                   uint64_t seed, uint64_t draw, int64_t global_offset);
void launch_reset_weights(hipStream_t s, DeviceState& d);
// K2: maximum-likelihood association -> ids[P*B]
void launch_assoc(hipStream_t s, DeviceState& d, const double* blobs_dev, const double* blobdir_dev,
                  int B, int32_t* ids_dev);
// K3: EKF update + log-weight.  known: first/next chains shared by all particles (device
// arrays, built on the host); otherwise built per particle in LDS from ids_dev.
void launch_observe(hipStream_t s, DeviceState& d, const double* blobs_dev, int B,
                    const int32_t* first_dev, const int32_t* next_dev, int n_unmatched,
                    const int32_t* ids_dev, const NoiseD& qt);
// K4: weights -> block totals / local scans
void launch_block_max(hipStream_t s, DeviceState& d, double* partial_dev, double* gmax_dev);
void launch_scan_local(hipStream_t s, DeviceState& d, const double* gmax_dev, int domain,
                       double* clocal_dev, double* totals_dev);
// exclusive scan of the (global) block totals, sequential in block order: offsets[nb], sum[1]
void launch_scan_blocks(hipStream_t s, const double* totals_dev, int64_t nb, double* offsets_dev,
                        double* sum_dev);
// ancestors of the local output slots [slot0, slot0 + n)
void launch_ancestors(hipStream_t s, const double* clocal_dev, const double* totals_dev,
                      const double* offsets_dev, const double* sum_dev, int64_t nb, int64_t P_global,
                      int64_t P_scan, double u, int64_t slot0, int64_t n, int32_t* anc_dev);
// K5: pose gather + map indirection
void launch_gather_poses(hipStream_t s, DeviceState& d, const int32_t* anc_dev);
// K6
void launch_summary_partials(hipStream_t s, DeviceState& d, double* partial_dev, double* out4_dev);
// map maintenance
void launch_materialise(hipStream_t s, DeviceState& d);
void launch_broadcast_slot(hipStream_t s, DeviceState& d, const unsigned char* slot_dev);
void launch_probe(hipStream_t s, const double* in_dev, double* out_dev);
void launch_iota(hipStream_t s, int32_t* p, int64_t n);

}  // namespace pk
