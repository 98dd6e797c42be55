// K1 motion sample, K6 pose summary, map maintenance (copy / broadcast), the single-triple probe.
//
// Hand-written gfx950 (CDNA4, wave64) kernels of the FastSLAM particle update; see DESIGN.md
// section 4.  No MFMA: the algebra is 2x2 / 3x3 and register resident (pk_math.hpp).
#include "pk_device.hpp"
#include "pk_philox.hpp"

namespace pk {

// ------------------------------------------------------------------ K1 motion
__global__ void __launch_bounds__(256) k_motion(double* __restrict__ x, double* __restrict__ y,
                                                double* __restrict__ h, int64_t P, double v, double w,
                                                double dt, double sd, double sh,
                                                const double* __restrict__ z, uint64_t seed,
                                                uint64_t draw, int64_t goff, int motion_blocks,
                                                uint4* __restrict__ up_dst, const uint4* __restrict__ up_src,
                                                int64_t up_n16, const int64_t* __restrict__ logical,
                                                double* __restrict__ pose_part) {
  __shared__ double red[4];
  if ((int)blockIdx.x >= motion_blocks) {
    // the extra workgroups of a combined launch: the per-scan upload (see k_upload)
    const int64_t nb = (int64_t)gridDim.x - motion_blocks;
    for (int64_t i = (int64_t)(blockIdx.x - motion_blocks) * blockDim.x + threadIdx.x; i < up_n16; i += nb * blockDim.x)
      up_dst[i] = up_src[i];
    return;
  }
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P && pose_part == nullptr) return;
  double px = 0.0, py = 0.0, ps = 0.0, pc = 0.0;  // this particle's share of the block's pose sums
  if (i < P) {
  double z0, z1, z2;
  if (z) {
    z0 = z[3 * i];
    z1 = z[3 * i + 1];
    z2 = z[3 * i + 2];
  } else {
    // the counter is the particle's index in the WHOLE filter: its slot plus the shard's offset, or -- balanced placement of
    // the sharded filter -- the logical index its slot carries
    uint64_t g = logical ? (uint64_t)logical[i] : (uint64_t)(i + goff);
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
  if (pose_part) {
    px = xi;
    py = yi;
    sincos(hi, &ps, &pc);
  }
  }
  // The block's sums of x, y, sin h, cos h of the MOVED particles (fixed order): k_candidates takes the particles' mean pose from
  // them (the reference the candidate lists are made around), which used to cost two launches of its own per scan
  if (pose_part) {  // (kernel-uniform)
    px = block_sum<4>(px, red);
    py = block_sum<4>(py, red);
    ps = block_sum<4>(ps, red);
    pc = block_sum<4>(pc, red);
    if (threadIdx.x == 0) {
      pose_part[4 * blockIdx.x + 0] = px;
      pose_part[4 * blockIdx.x + 1] = py;
      pose_part[4 * blockIdx.x + 2] = ps;
      pose_part[4 * blockIdx.x + 3] = pc;
    }
  }
}

void launch_motion(hipStream_t s, DeviceState& d, double v, double w, double dt, const double* z_dev,
                   uint64_t seed, uint64_t draw, int64_t global_offset, void* up_dst_dev,
                   const void* up_src_host_mapped, size_t up_bytes, double* pose_part_dev) {
  const int64_t n16 = up_dst_dev ? (int64_t)((up_bytes + 15) / 16) : 0;
  if (d.P == 0 && n16 == 0) return;
  double sd = fabs(.05 * v) + fabs(.005 * w) + .0005;  // :185
  double sh = fabs(.025 * w) + fabs(.005 * v) + .0005;  // :190,:193
  int blocks = (int)((d.P + 255) / 256);
  int64_t ub = (n16 + 255) / 256;
  if (ub > 256) ub = 256;
  hipLaunchKernelGGL(k_motion, dim3((unsigned)(blocks + ub)), dim3(256), 0, s, d.x[d.cur], d.y[d.cur], d.h[d.cur], d.P,
                     v, w, dt, sd, sh, z_dev, seed, draw, global_offset + d.global_offset, blocks,
                     static_cast<uint4*>(up_dst_dev), static_cast<const uint4*>(up_src_host_mapped), n16,
                     (const int64_t*)d.logical[d.cur], pose_part_dev);
}

// The same for the particles [p0, p1) only (device noise; the Philox counters use the global particle index, so the draws
// are those of the whole-range launch).
void launch_motion_range(hipStream_t s, DeviceState& d, double v, double w, double dt, uint64_t seed, uint64_t draw,
                         int64_t p0, int64_t p1) {
  if (p1 <= p0) return;
  double sd = fabs(.05 * v) + fabs(.005 * w) + .0005;  // :185
  double sh = fabs(.025 * w) + fabs(.005 * v) + .0005;  // :190,:193
  const int64_t n = p1 - p0;
  int blocks = (int)((n + 255) / 256);
  hipLaunchKernelGGL(k_motion, dim3((unsigned)blocks), dim3(256), 0, s, d.x[d.cur] + p0, d.y[d.cur] + p0, d.h[d.cur] + p0, n, v, w,
                     dt, sd, sh, (const double*)nullptr, seed, draw, d.global_offset + p0, blocks, (uint4*)nullptr,
                     (const uint4*)nullptr, (int64_t)0, d.logical[d.cur] ? (const int64_t*)(d.logical[d.cur] + p0) : (const int64_t*)nullptr,
                     (double*)nullptr);
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

// Per-scan upload: a few workgroups read the pinned host block over PCIe and write it to HBM.
// For the <= 100 KB of a scan this takes a few microseconds in stream order, where a
// hipMemcpyAsync (copy engine hand-over on both sides) left the kernels of the step waiting
// for ~25 us.  n16 = number of 16-byte words.
__global__ void __launch_bounds__(256) k_upload(uint4* __restrict__ dst, const uint4* __restrict__ src, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
void launch_upload(hipStream_t s, void* dst_dev, const void* src_host_mapped, size_t bytes) {
  const int64_t n16 = (int64_t)((bytes + 15) / 16);
  if (n16 == 0) return;
  int64_t nb = (n16 + 255) / 256;
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(k_upload, dim3((unsigned)nb), dim3(256), 0, s, static_cast<uint4*>(dst_dev),
                     static_cast<const uint4*>(src_host_mapped), n16);
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
  const Noise<double> qt = make_noise(Qt[0], Qt[5], Qt[6], Qt[7], Qt[10], Qt[11], Qt[15]);
  const double ux = in[53], uy = in[54];  // unit((cos b, sin b, 0)), host side like the kernels' input
  for (int i = 0; i < 79; ++i) out[i] = 0.0;
  out[0] = probability_of_match(f, sx, sy, sh, z, ux, uy);
  double pse = pk_atan2(f.my - sy, f.mx - sx);
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
