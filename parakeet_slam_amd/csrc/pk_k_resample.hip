// K4 weights -> scans -> ancestors (+ pose gather), and the sharded (multi-GPU) resample kernels.
//
// Hand-written gfx950 (CDNA4, wave64) kernels of the FastSLAM particle update; see DESIGN.md
// section 4.  No MFMA: the algebra is 2x2 / 3x3 and register resident (pk_math.hpp).
#include "pk_device.hpp"

namespace pk {

// ------------------------------------------------------------------ K4 weights
__global__ void __launch_bounds__(256) k_block_max(const double* __restrict__ logw, int64_t P,
                                                   double* __restrict__ partial) {
  __shared__ double red[4];
  double m = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x)
    m = fmax(m, logw[i]);
  m = block_max<4>(m, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = m;
}
__global__ void __launch_bounds__(256) k_final_max(const double* __restrict__ partial, int n,
                                                   double* __restrict__ out) {
  __shared__ double red[4];
  double m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmax(m, partial[i]);
  m = block_max<4>(m, red);
  if (threadIdx.x == 0) out[0] = m;
}
// max over the sharded running-max keys the observe kernels kept (kGmaxKeys == blockDim.x)
__global__ void __launch_bounds__(256) k_keys_max(const unsigned long long* __restrict__ keys,
                                                  double* __restrict__ out) {
  __shared__ double red[4];
  const double m = block_max<4>(key_to_double(keys[threadIdx.x]), red);
  if (threadIdx.x == 0) out[0] = m;
}
void launch_keys_max(hipStream_t s, const unsigned long long* keys_dev, double* gmax_dev) {
  hipLaunchKernelGGL(k_keys_max, dim3(1), dim3(kGmaxKeys), 0, s, keys_dev, gmax_dev);
}

void launch_block_max(hipStream_t s, DeviceState& d, double* partial_dev, double* gmax_dev) {
  int nb = (int)((d.P + 255) / 256);
  if (nb > kRedBlocks) nb = kRedBlocks;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_block_max, dim3(nb), dim3(256), 0, s, d.logw[d.cur], d.P, partial_dev);
  hipLaunchKernelGGL(k_final_max, dim3(1), dim3(256), 0, s, partial_dev, nb, gmax_dev);
}

// Block-local inclusive scan of w = exp(logw - shift) over kScanBlock particles:
// 4 consecutive particles per thread (sequential), Kogge-Stone over the wave with
// __shfl_up, sequential over the 4 waves.  The association order is fixed by the block
// size alone, so shards that are multiples of kScanBlock reproduce the 1-GPU bits.
__global__ void __launch_bounds__(256) k_scan_local(const double* __restrict__ logw, int64_t P,
                                                    const double* __restrict__ gmax, int domain,
                                                    double* __restrict__ clocal, double* __restrict__ totals,
                                                    const unsigned long long* __restrict__ gmax_key) {
  __shared__ double wtot[4];
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  double shift = 0.0;
  if (domain == 1) {
    if (gmax_key) {  // max over the sharded running-max keys (kGmaxKeys == blockDim.x)
      const double mine = key_to_double(gmax_key[tid]);
      shift = block_max<4>(mine, wtot);
      __syncthreads();
    } else {
      shift = gmax[0];
    }
    if (!(shift > -INFINITY)) shift = 0.0;  // all weights zero: keep exp(-inf) = 0, not NaN
  }
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + 4 * tid;
  double w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (base + i < P) ? exp(logw[base + i] - shift) : 0.0;
  double s0 = w[0], s1 = s0 + w[1], s2 = s1 + w[2], s3 = s2 + w[3];
  double val = s3;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    double t = __shfl_up(val, off, kWave);
    if (lane >= off) val += t;
  }
  if (lane == kWave - 1) wtot[wave] = val;
  double prev = __shfl_up(val, 1, kWave);  // exclusive prefix inside the wave
  if (lane == 0) prev = 0.0;
  __syncthreads();
  double woff = 0.0;
  for (int i = 0; i < wave; ++i) woff += wtot[i];
  const double excl = woff + prev;
  if (base < P) clocal[base] = excl + s0;
  if (base + 1 < P) clocal[base + 1] = excl + s1;
  if (base + 2 < P) clocal[base + 2] = excl + s2;
  if (base + 3 < P) clocal[base + 3] = excl + s3;
  // the block total IS the inclusive value of its last particle (bit for bit): shards hand
  // over at block boundaries and both sides must see the same cumulative weight there
  if (tid == 255) totals[blockIdx.x] = excl + s3;
}
void launch_scan_local(hipStream_t s, DeviceState& d, const double* gmax_dev, int domain, double* clocal_dev,
                       double* totals_dev, const unsigned long long* gmax_key_dev) {
  if (d.P == 0) return;
  int nb = (int)((d.P + kScanBlock - 1) / kScanBlock);
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, s, d.logw[d.cur], d.P, gmax_dev, domain,
                     clocal_dev, totals_dev, gmax_key_dev);
}

// Exclusive scan of the block totals in block order by ONE thread: the canonical
// (shard-count independent) association of the global prefix sum.
__global__ void __launch_bounds__(256) k_scan_blocks(const double* __restrict__ totals, int64_t nb,
                                                     double* __restrict__ offsets, double* __restrict__ sum) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* t = reinterpret_cast<double*>(smem);
  double run = 0.0;
  for (int64_t c0 = 0; c0 < nb; c0 += 2048) {
    int n = (int)((nb - c0 < 2048) ? (nb - c0) : 2048);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) t[i] = totals[c0 + i];
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 0; i < n; ++i) {
        double v = t[i];
        t[i] = run;
        run += v;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) offsets[c0 + i] = t[i];
  }
  if (threadIdx.x == 0) sum[0] = run;
}
void launch_scan_blocks(hipStream_t s, const double* totals_dev, int64_t nb, double* offsets_dev,
                        double* sum_dev) {
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 2048 * sizeof(double), s, totals_dev, nb, offsets_dev,
                     sum_dev);
}

constexpr int kAncMaxBlocks = 256;  // up to this many scan blocks k_ancestors scans the totals itself
// Ancestor of output slot k: first particle j whose inclusive cumulative weight C_j is
// >= u*r + k*r, r = sum/P  (equivalent to the walk at prkt_core_v2.py:233-250, '<=' at :239).
__global__ void __launch_bounds__(256) k_ancestors(const double* __restrict__ clocal,
                                                   const double* __restrict__ totals,
                                                   const double* offsets,
                                                   const double* sum, int64_t nb, int64_t Pg,
                                                   int64_t Pscan, double u, int64_t slot0, int64_t n,
                                                   int32_t* __restrict__ anc, const double* __restrict__ gx,
                                                   const double* __restrict__ gy, const double* __restrict__ gh,
                                                   const double* __restrict__ glw, const int32_t* __restrict__ gsrc,
                                                   double* __restrict__ gx2, double* __restrict__ gy2,
                                                   double* __restrict__ gh2, double* __restrict__ glw2,
                                                   int32_t* __restrict__ gsrc2) {
  // offsets == NULL (few blocks): the exclusive scan of the block totals is done here, by one
  // thread per workgroup, sequentially in block order -- the same additions in the same order as
  // k_scan_blocks, without its launch
  __shared__ double s_off[kAncMaxBlocks + 1];
  if (offsets == nullptr) {
    if (threadIdx.x == 0) {
      double run = 0.0;
      for (int64_t b = 0; b < nb; ++b) {
        s_off[b] = run;
        run += totals[b];
      }
      s_off[nb] = run;
    }
    __syncthreads();
    offsets = s_off;
    sum = s_off + nb;
  }
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double r = __ddiv_rn(sum[0], (double)Pg);                  // range_ :225
  const double t = __dadd_rn(__dmul_rn(u, r), __dmul_rn((double)(slot0 + k), r));  // step :226 + k*range_
  // block: first b with offsets[b] + totals[b] >= t
  int64_t lo = 0, hi = nb - 1;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (__dadd_rn(offsets[mid], totals[mid]) >= t)
      hi = mid;
    else
      lo = mid + 1;
  }
  const int64_t b = lo;
  const double off = offsets[b];
  int64_t j0 = b * kScanBlock, j1 = j0 + kScanBlock - 1;
  if (j1 > Pscan - 1) j1 = Pscan - 1;
  while (j0 < j1) {
    int64_t mid = (j0 + j1) >> 1;
    if (__dadd_rn(off, clocal[mid]) >= t)
      j1 = mid;
    else
      j0 = mid + 1;
  }
  anc[k] = (int32_t)j0;
  if (gx) {  // fused k_gather_poses (single-GPU resample)
    const int32_t a = (int32_t)j0;
    gx2[k] = gx[a];
    gy2[k] = gy[a];
    gh2[k] = gh[a];
    glw2[k] = glw[a];
    gsrc2[k] = gsrc[a];
  }
}
void launch_ancestors(hipStream_t s, const double* clocal_dev, const double* totals_dev,
                      const double* offsets_dev, const double* sum_dev, int64_t nb, int64_t P_global,
                      int64_t P_scan, double u, int64_t slot0, int64_t n, int32_t* anc_dev, DeviceState* gather) {
  if (n == 0) return;
  if (gather) {
    DeviceState& d = *gather;
    const int c = d.cur, m = c ^ 1;
    hipLaunchKernelGGL(k_ancestors, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, clocal_dev, totals_dev,
                       offsets_dev, sum_dev, nb, P_global, P_scan, u, slot0, n, anc_dev, d.x[c], d.y[c], d.h[c],
                       d.logw[c], d.src[c], d.x[m], d.y[m], d.h[m], d.logw[m], d.src[m]);
    d.cur = m;
    return;
  }
  hipLaunchKernelGGL(k_ancestors, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, clocal_dev, totals_dev,
                     offsets_dev, sum_dev, nb, P_global, P_scan, u, slot0, n, anc_dev, (const double*)nullptr,
                     (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (const int32_t*)nullptr,
                     (double*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr, (int32_t*)nullptr);
}

// ------------------------------------------------------------------ sharded resample
// Owner-computes offspring: with C_j the global inclusive cumulative weight of local
// particle j and t_k = u r + k r the comb (same expressions as k_ancestors), particle j fills
// the output slots [hi_{j-1}, hi_j), hi_j = #{k : t_k <= C_j}.  hi[0] is the count at the
// shard's lower boundary, hi[1 + j] that of particle j.  Every shard evaluates the same
// formula on the same block offsets, so the slot ranges tile [0, P) without communication.
//
// Two launches are folded in when the caller asks for them: with offsets == NULL (<= kAncMaxBlocks global
// scan blocks) every workgroup scans the global block totals itself, sequentially in block order -- the
// same additions in the same order as k_scan_blocks, as k_ancestors does; and with ranges != NULL the
// LAST workgroup to finish (ticket counter, reset for the next call) derives the per-destination
// particle ranges that k_shard_ranges would: 2 x world binary searches over the finished hi[].
__global__ void __launch_bounds__(256) k_offspring(const double* __restrict__ clocal,
                                                   const double* __restrict__ totals, const double* offsets,
                                                   const double* sum, int64_t nb_global, int64_t first_block,
                                                   int64_t Pl, int64_t Pg, double u, int last_shard,
                                                   int64_t* __restrict__ hi, int world, int64_t* __restrict__ ranges,
                                                   unsigned* __restrict__ ticket, int64_t goff) {
  // goff >= 0: clocal and the blocks are those of the WHOLE filter (every rank scanned the all-gathered log-weights:
  // exactly the 1-GPU scan whatever the shard size), this shard's particles are [goff, goff + Pl); first_block unused
  __shared__ double s_off[kAncMaxBlocks + 1];
  __shared__ int s_last;
  if (offsets == nullptr) {
    // the totals are fetched by all lanes at once (one L2 round trip), then scanned by one thread
    for (int b = threadIdx.x; b < nb_global; b += blockDim.x) s_off[b] = totals[b];
    __syncthreads();
    if (threadIdx.x == 0) {
      double run = 0.0;
      for (int64_t b = 0; b < nb_global; ++b) {
        const double v = s_off[b];
        s_off[b] = run;
        run += v;
      }
      s_off[nb_global] = run;
    }
    __syncthreads();
    offsets = s_off;
    sum = s_off + nb_global;
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. Pl
  if (i <= Pl) {
    const double r = __ddiv_rn(sum[0], (double)Pg);
    const double ur = __dmul_rn(u, r);
    double C = 0.0;
    bool search = true;
    int64_t val = 0;
    if (i == 0) {
      if (goff >= 0 ? goff == 0 : first_block == 0) {
        val = 0;
        search = false;
      } else if (goff >= 0) {
        C = __dadd_rn(offsets[(goff - 1) / kScanBlock], clocal[goff - 1]);  // cumulative weight of everybody before this shard
      } else {
        C = offsets[first_block];
      }
    } else {
      const int64_t j = i - 1;
      if (last_shard && j == Pl - 1) {  // the tail is clamped to the last particle, as k_ancestors does
        val = Pg;
        search = false;
      } else if (goff >= 0) {
        C = __dadd_rn(offsets[(goff + j) / kScanBlock], clocal[goff + j]);
      } else {
        C = __dadd_rn(offsets[first_block + j / kScanBlock], clocal[j]);
      }
    }
    if (search) {
      int64_t lo = 0, up = Pg;  // first k with t_k > C
      while (lo < up) {
        const int64_t mid = (lo + up) >> 1;
        const double t = __dadd_rn(ur, __dmul_rn((double)mid, r));
        if (t > C)
          up = mid;
        else
          lo = mid + 1;
      }
      val = lo;
    }
    hi[i] = val;
  }
  if (ranges == nullptr) return;
  // ---- the last workgroup to get here writes the ranges (what k_shard_ranges does) --------------
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) *ticket = 0u;  // ready for the next call (stream-ordered)
  __threadfence();
  // Two predicates per destination d, both monotone in j: A_d(j) = hi[j + 1] > d Pl and
  // B_d(j) = hi[j] >= (d + 1) Pl; wanted: the first j in [0, Pl) where each holds (Pl if none).
  // A 256-ary search by the whole workgroup -- two L2 round trips (coarse probes, then the one
  // segment), where a binary search per destination costs fourteen dependent ones.
  const volatile int64_t* vhi = hi;  // written by other workgroups: not through this CU's L1
  __shared__ int s_first[2];
  const int64_t step = (Pl + blockDim.x - 1) / blockDim.x;  // coarse probe j_t = min(t step, Pl - 1)
  for (int d = 0; d < world; ++d) {
    const int64_t start = (int64_t)d * Pl, end = start + Pl;
    if (threadIdx.x < 2) s_first[threadIdx.x] = INT_MAX;
    __syncthreads();
    {
      int64_t j = (int64_t)threadIdx.x * step;
      if (j > Pl - 1) j = Pl - 1;
      const int64_t h0 = vhi[j], h1 = vhi[j + 1];
      if (h1 > start) atomicMin(&s_first[0], (int)threadIdx.x);
      if (h0 >= end) atomicMin(&s_first[1], (int)threadIdx.x);
    }
    __syncthreads();
    const int ta = s_first[0], tb = s_first[1];
    __syncthreads();
    if (threadIdx.x < 2) s_first[threadIdx.x] = INT_MAX;
    __syncthreads();
    // fine: the segment before the first coarse hit (or the tail when no coarse probe holds)
    {
      const int64_t baseA = ta == INT_MAX ? (int64_t)(blockDim.x - 1) * step : (ta == 0 ? 0 : (int64_t)(ta - 1) * step);
      const int64_t baseB = tb == INT_MAX ? (int64_t)(blockDim.x - 1) * step : (tb == 0 ? 0 : (int64_t)(tb - 1) * step);
      for (int64_t o = threadIdx.x; o <= step; o += blockDim.x) {  // one pass unless Pl > 65 280
        const int64_t ja = baseA + o, jb = baseB + o;
        if (ja < Pl && vhi[ja + 1] > start) atomicMin(&s_first[0], (int)o);
        if (jb < Pl && vhi[jb] >= end) atomicMin(&s_first[1], (int)o);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        const int64_t j0 = s_first[0] == INT_MAX ? Pl : baseA + s_first[0];
        const int64_t j1 = s_first[1] == INT_MAX ? Pl : baseB + s_first[1];
        ranges[2 * d] = j0;
        ranges[2 * d + 1] = j1 < j0 ? j0 : j1;
      }
    }
    __syncthreads();
  }
}
void launch_offspring(hipStream_t s, const double* clocal_dev, const double* offsets_dev, const double* sum_dev,
                      int64_t first_block, int64_t P_local, int64_t P_global, double u, int last_shard,
                      int64_t* hi_dev) {
  hipLaunchKernelGGL(k_offspring, dim3((unsigned)((P_local + 1 + 255) / 256)), dim3(256), 0, s, clocal_dev,
                     (const double*)nullptr, offsets_dev, sum_dev, (int64_t)0, first_block, P_local, P_global, u, last_shard,
                     hi_dev, 0, (int64_t*)nullptr, (unsigned*)nullptr, (int64_t)-1);
}
// offspring + block-total scan + per-destination ranges in ONE launch (n_global_blocks <= kAncestorsScanMaxBlocks)
void launch_offspring_plan(hipStream_t s, const double* clocal_dev, const double* global_totals_dev,
                           int64_t n_global_blocks, int64_t first_block, int64_t P_local, int64_t P_global, double u,
                           int last_shard, int64_t* hi_dev, int world, int64_t* ranges_dev, unsigned* ticket_dev) {
  hipLaunchKernelGGL(k_offspring, dim3((unsigned)((P_local + 1 + 255) / 256)), dim3(256), 0, s, clocal_dev,
                     global_totals_dev, (const double*)nullptr, (const double*)nullptr, n_global_blocks, first_block,
                     P_local, P_global, u, last_shard, hi_dev, world, ranges_dev, ticket_dev, (int64_t)-1);
}
// The same on the scan of the WHOLE filter's weights (any shard size; see pk_shard_plan_global_dev): clocal_global /
// the block offsets cover all P_global particles, this shard's are [goff, goff + P_local).
void launch_offspring_global(hipStream_t s, const double* clocal_global_dev, const double* offsets_dev, const double* sum_dev,
                             int64_t goff, int64_t P_local, int64_t P_global, double u, int last_shard, int64_t* hi_dev) {
  hipLaunchKernelGGL(k_offspring, dim3((unsigned)((P_local + 1 + 255) / 256)), dim3(256), 0, s, clocal_global_dev,
                     (const double*)nullptr, offsets_dev, sum_dev, (int64_t)0, (int64_t)0, P_local, P_global, u, last_shard,
                     hi_dev, 0, (int64_t*)nullptr, (unsigned*)nullptr, goff);
}
// block-local scans of an arbitrary log-weight array (not the filter's own)
void launch_scan_local_of(hipStream_t s, const double* logw_dev, int64_t n, const double* gmax_dev, int domain,
                          double* clocal_dev, double* totals_dev) {
  if (n == 0) return;
  int nb = (int)((n + kScanBlock - 1) / kScanBlock);
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, s, logw_dev, n, gmax_dev, domain, clocal_dev, totals_dev,
                     (const unsigned long long*)nullptr);
}

// record = (x, y, h, logw) + map slot
__global__ void __launch_bounds__(256) k_pack(SlotSource ss, const int32_t* __restrict__ src,
                                              const double* __restrict__ x, const double* __restrict__ y,
                                              const double* __restrict__ h, const double* __restrict__ lw,
                                              const int64_t* __restrict__ idx, unsigned char* __restrict__ buf) {
  const int64_t i = blockIdx.x;
  const int64_t j = idx[i];
  unsigned char* rec = buf + (size_t)i * (kPoseRecordBytes + ss.slot_bytes);
  if (threadIdx.x == 0) {
    double* hd = reinterpret_cast<double*>(rec);
    hd[0] = x[j];
    hd[1] = y[j];
    hd[2] = h[j];
    hd[3] = lw[j];
    hd[4] = 0.0;
    hd[5] = 0.0;
  }
  const uint4* s = reinterpret_cast<const uint4*>(ss.at(src[j]));
  uint4* d = reinterpret_cast<uint4*>(rec + kPoseRecordBytes);
  const size_t n = ss.slot_bytes / 16;
  for (size_t k = threadIdx.x; k < n; k += blockDim.x) d[k] = s[k];
}
void launch_pack(hipStream_t s, DeviceState& d, const int64_t* idx_dev, int64_t n, unsigned char* buf_dev) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_pack, dim3((unsigned)n), dim3(256), 0, s, slot_source(d), d.src[d.cur], d.x[d.cur],
                     d.y[d.cur], d.h[d.cur], d.logw[d.cur], idx_dev, buf_dev);
}

// New generation of the shard: slot k <- local particle srcs[k] (>= 0) or received record
// -(srcs[k]) - 1.  Poses are gathered now; maps stay where they are (own buffer / record)
// until the next observe rewrites them.
__global__ void __launch_bounds__(256) k_adopt(const double* __restrict__ x, const double* __restrict__ y,
                                               const double* __restrict__ h, const double* __restrict__ lw,
                                               const int32_t* __restrict__ src, double* __restrict__ x2,
                                               double* __restrict__ y2, double* __restrict__ h2,
                                               double* __restrict__ lw2, int32_t* __restrict__ src2,
                                               const int64_t* __restrict__ srcs, const unsigned char* __restrict__ buf,
                                               size_t stride, int64_t P) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const int64_t a = srcs[k];
  if (a >= 0) {
    x2[k] = x[a];
    y2[k] = y[a];
    h2[k] = h[a];
    lw2[k] = lw[a];
    src2[k] = src[a];
  } else {
    const double* hd = reinterpret_cast<const double*>(buf + (size_t)(-(a + 1)) * stride);
    x2[k] = hd[0];
    y2[k] = hd[1];
    h2[k] = hd[2];
    lw2[k] = hd[3];
    src2[k] = (int32_t)a;
  }
}
void launch_adopt(hipStream_t s, DeviceState& d, const int64_t* src_dev, const unsigned char* buf_dev) {
  if (d.P == 0) return;
  const int c = d.cur, n = c ^ 1;
  const size_t stride = kPoseRecordBytes + d.lay.slot_bytes;
  hipLaunchKernelGGL(k_adopt, dim3((unsigned)((d.P + 255) / 256)), dim3(256), 0, s, d.x[c], d.y[c], d.h[c],
                     d.logw[c], d.src[c], d.x[n], d.y[n], d.h[n], d.logw[n], d.src[n], src_dev, buf_dev, stride,
                     d.P);
  d.cur = n;
  d.alt = buf_dev;
  d.alt_stride = stride;
  d.alt_off = kPoseRecordBytes;
}


// ---- device-resident exchange ------------------------------------------------------------
// ranges[2 d], ranges[2 d + 1] = [j0, j1): the local particles whose offspring overlap the
// output slots [d P, (d + 1) P) of rank d (hi from k_offspring, monotone).
__global__ void k_shard_ranges(const int64_t* __restrict__ hi, int64_t P, int world, int64_t* __restrict__ ranges) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= world) return;
  const int64_t start = (int64_t)d * P, end = start + P;
  int64_t lo = 0, up = P;  // first j with hi[j + 1] > start
  while (lo < up) {
    const int64_t mid = (lo + up) >> 1;
    if (hi[mid + 1] > start)
      up = mid;
    else
      lo = mid + 1;
  }
  const int64_t j0 = lo;
  lo = 0;
  up = P;  // first j with hi[j] >= end
  while (lo < up) {
    const int64_t mid = (lo + up) >> 1;
    if (hi[mid] >= end)
      up = mid;
    else
      lo = mid + 1;
  }
  ranges[2 * d] = j0;
  ranges[2 * d + 1] = lo < j0 ? j0 : lo;
}
void launch_shard_ranges(hipStream_t s, const int64_t* hi_dev, int64_t P_local, int world, int64_t* ranges_dev) {
  hipLaunchKernelGGL(k_shard_ranges, dim3((world + 63) / 64), dim3(64), 0, s, hi_dev, P_local, world, ranges_dev);
}

// Records of the contiguous local particles [j0, j0 + n) for one destination; the header
// carries the destination slots [lo, hi) each copy fills (empty for particles without
// offspring there: they ride along, the receiver skips them).
__global__ void __launch_bounds__(256) k_pack_range(SlotSource ss, const int32_t* __restrict__ src,
                                                    const double* __restrict__ x, const double* __restrict__ y,
                                                    const double* __restrict__ h, const double* __restrict__ lw,
                                                    const int64_t* __restrict__ hi, int64_t j0, int64_t slot_start,
                                                    int64_t slot_end, unsigned char* __restrict__ buf) {
  const int64_t i = blockIdx.x;
  const int64_t j = j0 + i;
  unsigned char* rec = buf + (size_t)i * (kPoseRecordBytes + ss.slot_bytes);
  if (threadIdx.x == 0) {
    double* hd = reinterpret_cast<double*>(rec);
    hd[0] = x[j];
    hd[1] = y[j];
    hd[2] = h[j];
    hd[3] = lw[j];
    int64_t lo = hi[j] > slot_start ? hi[j] : slot_start;
    int64_t up = hi[j + 1] < slot_end ? hi[j + 1] : slot_end;
    if (up < lo) up = lo;
    reinterpret_cast<int64_t*>(rec)[4] = lo;
    reinterpret_cast<int64_t*>(rec)[5] = up;
  }
  const uint4* s = reinterpret_cast<const uint4*>(ss.at(src[j]));
  uint4* d = reinterpret_cast<uint4*>(rec + kPoseRecordBytes);
  const size_t n = ss.slot_bytes / 16;
  for (size_t k = threadIdx.x; k < n; k += blockDim.x) d[k] = s[k];
}
void launch_pack_range(hipStream_t s, DeviceState& d, const int64_t* hi_dev, int64_t j0, int64_t n, int64_t slot_start,
                       int64_t slot_end, unsigned char* buf_dev) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_pack_range, dim3((unsigned)n), dim3(256), 0, s, slot_source(d), d.src[d.cur], d.x[d.cur],
                     d.y[d.cur], d.h[d.cur], d.logw[d.cur], hi_dev, j0, slot_start, slot_end, buf_dev);
}

__global__ void __launch_bounds__(256) k_extract_lohi(const unsigned char* __restrict__ buf, size_t stride, int64_t n,
                                                      int64_t* __restrict__ rlohi) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t* hd = reinterpret_cast<const int64_t*>(buf + (size_t)r * stride);
  rlohi[2 * r] = hd[4];
  rlohi[2 * r + 1] = hd[5];
}

// New generation from the device-side plan: local slot k (global slot K) takes the local
// particle j with hi[j] <= K < hi[j + 1] if K lies in this shard's offspring range, else the
// received record whose [lo, hi) holds K (records arrive in slot order).
__global__ void __launch_bounds__(256) k_adopt_dev(const double* __restrict__ x, const double* __restrict__ y,
                                                   const double* __restrict__ h, const double* __restrict__ lw,
                                                   const int32_t* __restrict__ src, double* __restrict__ x2,
                                                   double* __restrict__ y2, double* __restrict__ h2,
                                                   double* __restrict__ lw2, int32_t* __restrict__ src2,
                                                   const int64_t* __restrict__ hi, int64_t slot_start,
                                                   const unsigned char* __restrict__ buf, size_t stride,
                                                   const int64_t* __restrict__ rlohi, int64_t n_recv, int64_t P, int mode,
                                                   int64_t span_lo, int64_t span_hi) {
  // mode 0: every slot; 1: only the slots filled by this shard's own particles; 2: only those filled by received records
  // [span_lo, span_hi): the "split_loopback" debug option narrows what counts as filled locally, so that the slots at either
  // end are filled from records of this shard's OWN particles that went through the exchange (one-rank RCCL tests)
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const int64_t K = slot_start + k;
  const bool local = K >= hi[0] && K < hi[P] && K >= span_lo && K < span_hi;
  if ((mode == 1 && !local) || (mode == 2 && local)) return;
  if (local) {
    int64_t lo = 0, up = P - 1;  // first j with hi[j + 1] > K
    while (lo < up) {
      const int64_t mid = (lo + up) >> 1;
      if (hi[mid + 1] > K)
        up = mid;
      else
        lo = mid + 1;
    }
    x2[k] = x[lo];
    y2[k] = y[lo];
    h2[k] = h[lo];
    lw2[k] = lw[lo];
    src2[k] = src[lo];
  } else {
    int64_t lo = 0, up = n_recv - 1;  // first record with hi_r > K
    while (lo < up) {
      const int64_t mid = (lo + up) >> 1;
      if (rlohi[2 * mid + 1] > K)
        up = mid;
      else
        lo = mid + 1;
    }
    const double* hd = reinterpret_cast<const double*>(buf + (size_t)lo * stride);
    x2[k] = hd[0];
    y2[k] = hd[1];
    h2[k] = hd[2];
    lw2[k] = hd[3];
    src2[k] = (int32_t)(-(lo + 1));
  }
}
void launch_adopt_dev(hipStream_t s, DeviceState& d, const int64_t* hi_dev, int64_t slot_start,
                      const unsigned char* buf_dev, int64_t n_recv, int64_t* rlohi_dev, int mode, int64_t span_lo, int64_t span_hi) {
  if (d.P == 0) return;
  // mode 2 (the received part of a split adoption) writes into the generation mode 1 has already made current
  const int n = mode == 2 ? d.cur : d.cur ^ 1, c = n ^ 1;
  const size_t stride = kPoseRecordBytes + d.lay.slot_bytes;
  if (n_recv > 0 && mode != 1)
    hipLaunchKernelGGL(k_extract_lohi, dim3((unsigned)((n_recv + 255) / 256)), dim3(256), 0, s, buf_dev, stride, n_recv,
                       rlohi_dev);
  hipLaunchKernelGGL(k_adopt_dev, dim3((unsigned)((d.P + 255) / 256)), dim3(256), 0, s, d.x[c], d.y[c], d.h[c],
                     d.logw[c], d.src[c], d.x[n], d.y[n], d.h[n], d.logw[n], d.src[n], hi_dev, slot_start, buf_dev,
                     stride, rlohi_dev, n_recv, d.P, mode, span_lo, span_hi);
  d.cur = n;
  if (mode != 1) {
    d.alt = n_recv > 0 ? buf_dev : nullptr;
    d.alt_stride = stride;
    d.alt_off = kPoseRecordBytes;
  }
}

// ---- balanced placement (minimum migration; DESIGN.md section 6, sharded.py::plan_balanced is the readable reference) ----
// Every physical slot g = rank * P + j carries the LOGICAL index of its particle (its index in one filter holding all of
// them: what keys the Philox streams and orders the weight scan of prkt_core_v2.py:216-250).  Every rank holds the
// all-gathered state rows [logw(P) | logical(P)] of all ranks and derives the whole plan by itself.
__device__ __forceinline__ int64_t bal_logical(const double* __restrict__ gstate, int64_t P, int64_t g) {
  const int64_t r = g / P, j = g - r * P;
  return __double_as_longlong(gstate[(size_t)r * 2 * P + P + j]);
}
// the log-weights put into the single filter's order
__global__ void __launch_bounds__(256) k_bal_scatter(const double* __restrict__ gstate, int64_t P, int64_t Pg,
                                                     double* __restrict__ glogw, int* __restrict__ bad) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= Pg) return;
  const int64_t r = g / P, j = g - r * P;
  const int64_t l = __double_as_longlong(gstate[(size_t)r * 2 * P + P + j]);
  if (l < 0 || l >= Pg) {  // a corrupted placement must not become an out-of-bounds store
    atomicAdd(bad, 1);
    return;
  }
  glogw[l] = gstate[(size_t)r * 2 * P + j];
}
// Children count of the particle in physical place g, c = H[l + 1] - H[l], as the packed scan element (c << 32) | (c > 0);
// block-local inclusive scan over kScanBlock places (integers: any association gives the same sums).
__global__ void __launch_bounds__(256) k_bal_counts(const double* __restrict__ gstate, const int64_t* __restrict__ H, int64_t P,
                                                    int64_t Pg, long long* __restrict__ cloc, long long* __restrict__ ctot) {
  __shared__ long long wtot[4];
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + 4 * tid;
  long long e[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    e[i] = 0;
    if (base + i < Pg) {
      int64_t l = bal_logical(gstate, P, base + i);
      if (l < 0 || l >= Pg) l = 0;  // (reported by k_bal_scatter)
      const long long c = H[l + 1] - H[l];
      e[i] = (c << 32) | (c > 0 ? 1 : 0);
    }
  }
  const long long s0 = e[0], s1 = s0 + e[1], s2 = s1 + e[2], s3 = s2 + e[3];
  long long val = s3;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const long long t = __shfl_up(val, off, kWave);
    if (lane >= off) val += t;
  }
  if (lane == kWave - 1) wtot[wave] = val;
  long long prev = __shfl_up(val, 1, kWave);
  if (lane == 0) prev = 0;
  __syncthreads();
  long long woff = 0;
  for (int i = 0; i < wave; ++i) woff += wtot[i];
  const long long excl = woff + prev;
  if (base < Pg) cloc[base] = excl + s0;
  if (base + 1 < Pg) cloc[base + 1] = excl + s1;
  if (base + 2 < Pg) cloc[base + 2] = excl + s2;
  if (base + 3 < Pg) cloc[base + 3] = excl + s3;
  if (tid == 255) ctot[blockIdx.x] = excl + s3;
}
// exclusive packed prefix (children << 32 | particles with children) of the physical places [0, g)
__device__ __forceinline__ long long bal_prefix(const long long* __restrict__ cloc, const long long* __restrict__ coff, int64_t g) {
  return g == 0 ? 0 : coff[(g - 1) / kScanBlock] + cloc[g - 1];
}
constexpr int kBalMaxWorld = 64;
// One workgroup: the block offsets, the per-rank tables (n, m, ebase, dbase) and who sends which of its particles to whom.
// table: world rows of 2 world + 4 words -- (a0, a1) per destination (indices into the sender's list of particles WITH
// children), then n, m, ebase, dbase of the row's rank.
__global__ void __launch_bounds__(256) k_bal_plan(const long long* __restrict__ cloc, const long long* __restrict__ ctot,
                                                  long long* __restrict__ coff, int64_t nbg, int64_t P, int world,
                                                  int64_t* __restrict__ table) {
  __shared__ long long s_bound[kBalMaxWorld + 1];
  __shared__ int64_t s_n[kBalMaxWorld], s_m[kBalMaxWorld], s_eb[kBalMaxWorld], s_db[kBalMaxWorld];
  __shared__ long long s_t[2048];
  {
    long long run = 0;  // (thread 0's)
    for (int64_t c0 = 0; c0 < nbg; c0 += 2048) {
      const int n = (int)((nbg - c0 < 2048) ? (nbg - c0) : 2048);
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += blockDim.x) s_t[i] = ctot[c0 + i];
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int i = 0; i < n; ++i) {
          const long long v = s_t[i];
          s_t[i] = run;
          run += v;
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += blockDim.x) coff[c0 + i] = s_t[i];
    }
    if (threadIdx.x == 0) coff[nbg] = run;
  }
  __syncthreads();
  for (int r = threadIdx.x; r <= world; r += blockDim.x) s_bound[r] = bal_prefix(cloc, coff, (int64_t)r * P);
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t eb = 0, db = 0;
    for (int r = 0; r < world; ++r) {
      const int64_t n = (s_bound[r + 1] >> 32) - (s_bound[r] >> 32);
      const int64_t m = n < P ? n : P;
      s_n[r] = n;
      s_m[r] = m;
      s_eb[r] = eb;
      s_db[r] = db;
      eb += n - m;
      db += P - m;
    }
  }
  __syncthreads();
  const int row = 2 * world + 4;
  for (int r = threadIdx.x; r < world; r += blockDim.x) {
    table[(size_t)r * row + 2 * world + 0] = s_n[r];
    table[(size_t)r * row + 2 * world + 1] = s_m[r];
    table[(size_t)r * row + 2 * world + 2] = s_eb[r];
    table[(size_t)r * row + 2 * world + 3] = s_db[r];
  }
  for (int pr = threadIdx.x; pr < world * world; pr += blockDim.x) {
    const int s = pr / world, d = pr - s * world;
    int64_t a0 = 0, a1 = 0;
    const int64_t es = s_n[s] - s_m[s], dd = P - s_m[d];
    const int64_t lo = s_eb[s] > s_db[d] ? s_eb[s] : s_db[d];
    const int64_t up = (s_eb[s] + es < s_db[d] + dd) ? s_eb[s] + es : s_db[d] + dd;
    if (s != d && up > lo) {
      const int64_t q0 = P + lo - s_eb[s], q1 = P + up - s_eb[s];  // rank-relative child positions [q0, q1)
      const int64_t gb = (int64_t)s * P;
      const long long base = s_bound[s];
      int64_t l = 0, u = P;  // first j with rel[j + 1] > q0
      while (l < u) {
        const int64_t mid = (l + u) >> 1;
        if (((bal_prefix(cloc, coff, gb + mid + 1) - base) >> 32) > q0)
          u = mid;
        else
          l = mid + 1;
      }
      const int64_t j0 = l;
      l = 0;
      u = P;  // first j with rel[j] >= q1
      while (l < u) {
        const int64_t mid = (l + u) >> 1;
        if (((bal_prefix(cloc, coff, gb + mid) - base) >> 32) >= q1)
          u = mid;
        else
          l = mid + 1;
      }
      const int64_t j1 = l < j0 ? j0 : l;
      a0 = (bal_prefix(cloc, coff, gb + j0) - base) & 0xffffffffll;
      a1 = (bal_prefix(cloc, coff, gb + j1) - base) & 0xffffffffll;
    }
    table[(size_t)s * row + 2 * d] = a0;
    table[(size_t)s * row + 2 * d + 1] = a1;
  }
}
// This rank's own tables: rel[j] = children of its particles [0, j) (P + 1), Hl[j] = first output slot of particle j's
// children, alive[] = its particles with children, ascending.
__global__ void __launch_bounds__(256) k_bal_own(const long long* __restrict__ cloc, const long long* __restrict__ coff,
                                                 const int64_t* __restrict__ H, const int64_t* __restrict__ logical, int64_t P,
                                                 int64_t gbase, int64_t* __restrict__ rel, int64_t* __restrict__ Hl,
                                                 int32_t* __restrict__ alive) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > P) return;
  const long long base = bal_prefix(cloc, coff, gbase);
  const long long ex = bal_prefix(cloc, coff, gbase + j) - base;
  rel[j] = ex >> 32;
  if (j == P) return;
  const long long in = bal_prefix(cloc, coff, gbase + j + 1) - base;
  Hl[j] = H[logical[j]];
  if ((in >> 32) > (ex >> 32)) alive[ex & 0xffffffffll] = (int32_t)j;
}
__global__ void __launch_bounds__(256) k_iota64(int64_t* p, int64_t n, int64_t off) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = off + i;
}
void launch_iota64(hipStream_t s, int64_t* p, int64_t n, int64_t off) {
  if (n > 0) hipLaunchKernelGGL(k_iota64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, off);
}
// state row of this rank for the all-gather: [logw(P) | logical(P) as int64 bits]
__global__ void __launch_bounds__(256) k_bal_state(const double* __restrict__ logw, const int64_t* __restrict__ logical, int64_t P,
                                                   double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= P) return;
  out[j] = logw[j];
  out[P + j] = __longlong_as_double(logical[j]);
}
void launch_bal_state(hipStream_t s, const DeviceState& d, double* out_dev) {
  if (d.P > 0)
    hipLaunchKernelGGL(k_bal_state, dim3((unsigned)((d.P + 255) / 256)), dim3(256), 0, s, d.logw[d.cur], d.logical[d.cur], d.P,
                       out_dev);
}
void launch_bal_plan(hipStream_t s, const DeviceState& d, const double* gstate_dev, int64_t Pg, int world, int rank,
                     const double* gmax_dev, int domain, double u, BalancedBuffers& b, int64_t* table_dev) {
  const int64_t P = d.P;
  const int64_t nbg = (Pg + kScanBlock - 1) / kScanBlock;
  hipLaunchKernelGGL(k_bal_scatter, dim3((unsigned)((Pg + 255) / 256)), dim3(256), 0, s, gstate_dev, P, Pg, b.glogw, b.bad);
  // exactly the kernels of the 1-GPU resample on the whole filter's log-weights in logical order: same blocks, same bits
  launch_scan_local_of(s, b.glogw, Pg, gmax_dev, domain, b.clocal, b.totals);
  launch_scan_blocks(s, b.totals, nbg, b.offsets, b.sum);
  launch_offspring_global(s, b.clocal, b.offsets, b.sum, 0, Pg, Pg, u, 1, b.H);
  hipLaunchKernelGGL(k_bal_counts, dim3((unsigned)nbg), dim3(256), 0, s, gstate_dev, b.H, P, Pg, b.cloc, b.ctot);
  hipLaunchKernelGGL(k_bal_plan, dim3(1), dim3(256), 0, s, b.cloc, b.ctot, b.coff, nbg, P, world, table_dev);
  hipLaunchKernelGGL(k_bal_own, dim3((unsigned)((P + 1 + 255) / 256)), dim3(256), 0, s, b.cloc, b.coff, b.H, d.logical[d.cur], P,
                     (int64_t)rank * P, b.rel, b.Hl, b.alive);
}

// Records of this rank's particles alive[a0 .. a0 + n) for one destination: header (x, y, h, logw, lo, up, klo, 0) -- the
// free slots [lo, up) of the destination the particle's excess children fill, klo the logical index of the child in slot lo.
__global__ void __launch_bounds__(256) k_bal_pack(SlotSource ss, const int32_t* __restrict__ src, const double* __restrict__ x,
                                                  const double* __restrict__ y, const double* __restrict__ h,
                                                  const double* __restrict__ lw, const int64_t* __restrict__ rel,
                                                  const int64_t* __restrict__ Hl, const int32_t* __restrict__ alive, int64_t a0,
                                                  int64_t P, int64_t ebase_s, int64_t dbase_d, int64_t dd_d, int64_t m_d,
                                                  unsigned char* __restrict__ buf, size_t stride, GrowState g) {
  const int64_t i = blockIdx.x;
  const int64_t j = alive[a0 + i];
  unsigned char* rec = buf + (size_t)i * stride;
  if (threadIdx.x == 0) {
    double* hd = reinterpret_cast<double*>(rec);
    hd[0] = x[j];
    hd[1] = y[j];
    hd[2] = h[j];
    hd[3] = lw[j];
    const int64_t r0 = rel[j] > P ? rel[j] : P, r1 = rel[j + 1] > P ? rel[j + 1] : P;
    const int64_t e0 = ebase_s + r0 - P, e1 = ebase_s + r1 - P;
    const int64_t elo = e0 > dbase_d ? e0 : dbase_d;
    int64_t eup = e1 < dbase_d + dd_d ? e1 : dbase_d + dd_d;
    if (eup < elo) eup = elo;
    int64_t* hl = reinterpret_cast<int64_t*>(rec);
    hl[4] = m_d + elo - dbase_d;
    hl[5] = m_d + eup - dbase_d;
    hl[6] = Hl[j] + (elo - ebase_s + P - rel[j]);
    hl[7] = 0;
  }
  const uint4* s = reinterpret_cast<const uint4*>(ss.at(src[j]));
  uint4* d = reinterpret_cast<uint4*>(rec + kPoseRecordBytes);
  const size_t n = ss.slot_bytes / 16;
  for (size_t k = threadIdx.x; k < n; k += blockDim.x) d[k] = s[k];
  if (g.R > 0) {  // the particle's new-landmark bookkeeping travels behind its map (GrowState: counters | slot ids | readings)
    unsigned char* tail = rec + kPoseRecordBytes + ss.slot_bytes;
    const int32_t* sc = g.cnt[g.cur] + 4 * j;
    int32_t* tc = reinterpret_cast<int32_t*>(tail);
    if (threadIdx.x < 4) tc[threadIdx.x] = sc[threadIdx.x];
    const int nrd = sc[0], used = sc[1];
    const int32_t* ss_ = g.slot_id[g.cur] + (size_t)j * g.S;
    int32_t* ts = tc + 4;
    for (int k = threadIdx.x; k < used; k += blockDim.x) ts[k] = ss_[k];
    const double* sr = g.hyp[g.cur] + (size_t)j * g.R * 8;
    double* tr = reinterpret_cast<double*>(tail + grow_tail_readings_off(g.S));
    for (int k = threadIdx.x; k < 8 * nrd; k += blockDim.x) tr[k] = sr[k];
  }
}
// keep: the children a rank keeps for itself (its own slots) -- P, or less in the one-rank loopback of the tests
void launch_bal_pack(hipStream_t s, DeviceState& d, const BalancedBuffers& b, int64_t a0, int64_t n, int64_t ebase_s,
                     int64_t dbase_d, int64_t dd_d, int64_t m_d, unsigned char* buf_dev, size_t stride, const GrowState* g, int64_t keep) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_bal_pack, dim3((unsigned)n), dim3(256), 0, s, slot_source(d), d.src[d.cur], d.x[d.cur], d.y[d.cur],
                     d.h[d.cur], d.logw[d.cur], b.rel, b.Hl, b.alive, a0, keep < 0 ? d.P : keep, ebase_s, dbase_d, dd_d, m_d, buf_dev, stride,
                     g ? *g : GrowState{});
}

__global__ void __launch_bounds__(256) k_bal_extract(const unsigned char* __restrict__ buf, size_t stride, int64_t n,
                                                     int64_t* __restrict__ rh) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t* hd = reinterpret_cast<const int64_t*>(buf + (size_t)r * stride);
  rh[3 * r] = hd[4];
  rh[3 * r + 1] = hd[5];
  rh[3 * r + 2] = hd[6];
}
// New generation: slot k < m takes the child at position k of this rank's own particles, slot k >= m the received record
// whose [lo, up) holds k (records arrive in slot order); every slot gets its logical index.
__global__ void __launch_bounds__(256) k_bal_adopt(const double* __restrict__ x, const double* __restrict__ y,
                                                   const double* __restrict__ h, const double* __restrict__ lw,
                                                   const int32_t* __restrict__ src, double* __restrict__ x2,
                                                   double* __restrict__ y2, double* __restrict__ h2, double* __restrict__ lw2,
                                                   int32_t* __restrict__ src2, int64_t* __restrict__ logical2,
                                                   const int64_t* __restrict__ rel, const int64_t* __restrict__ Hl, int64_t m,
                                                   const unsigned char* __restrict__ buf, size_t stride,
                                                   const int64_t* __restrict__ rh, int64_t n_recv, int64_t P, int mode,
                                                   int* __restrict__ bad, int32_t* __restrict__ anc) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const bool local = k < m;
  if ((mode == 1 && !local) || (mode == 2 && local)) return;
  if (local) {
    int64_t lo = 0, up = P - 1;  // first j with rel[j + 1] > k
    while (lo < up) {
      const int64_t mid = (lo + up) >> 1;
      if (rel[mid + 1] > k)
        up = mid;
      else
        lo = mid + 1;
    }
    x2[k] = x[lo];
    y2[k] = y[lo];
    h2[k] = h[lo];
    lw2[k] = lw[lo];
    src2[k] = src[lo];
    logical2[k] = Hl[lo] + (k - rel[lo]);
    if (anc) anc[k] = (int32_t)lo;
  } else {
    if (n_recv <= 0) {
      atomicAdd(bad, 1);
      return;
    }
    int64_t lo = 0, up = n_recv - 1;  // first record with up_r > k
    while (lo < up) {
      const int64_t mid = (lo + up) >> 1;
      if (rh[3 * mid + 1] > k)
        up = mid;
      else
        lo = mid + 1;
    }
    if (!(rh[3 * lo] <= k && k < rh[3 * lo + 1])) atomicAdd(bad, 1);  // the records do not tile [m, P): plan and exchange disagree
    const double* hd = reinterpret_cast<const double*>(buf + (size_t)lo * stride);
    x2[k] = hd[0];
    y2[k] = hd[1];
    h2[k] = hd[2];
    lw2[k] = hd[3];
    src2[k] = (int32_t)(-(lo + 1));
    logical2[k] = rh[3 * lo + 2] + (k - rh[3 * lo]);
    if (anc) anc[k] = (int32_t)(-(lo + 1));
  }
}
// anc_dev (or NULL): where every slot's particle came from -- a local particle (>= 0) or received record -anc - 1
void launch_bal_adopt(hipStream_t s, DeviceState& d, const BalancedBuffers& b, int64_t m, const unsigned char* buf_dev,
                      int64_t n_recv, int64_t* rh_dev, int mode, size_t stride, int32_t* anc_dev) {
  if (d.P == 0) return;
  // mode 2 (the received part of a split adoption) writes into the generation mode 1 has already made current
  const int n = mode == 2 ? d.cur : d.cur ^ 1, c = n ^ 1;
  if (n_recv > 0 && mode != 1)
    hipLaunchKernelGGL(k_bal_extract, dim3((unsigned)((n_recv + 255) / 256)), dim3(256), 0, s, buf_dev, stride, n_recv, rh_dev);
  hipLaunchKernelGGL(k_bal_adopt, dim3((unsigned)((d.P + 255) / 256)), dim3(256), 0, s, d.x[c], d.y[c], d.h[c], d.logw[c],
                     d.src[c], d.x[n], d.y[n], d.h[n], d.logw[n], d.src[n], d.logical[n], b.rel, b.Hl, m, buf_dev, stride, rh_dev,
                     n_recv, d.P, mode, b.bad, anc_dev);
  d.cur = n;
  if (mode != 1) {
    d.alt = n_recv > 0 ? buf_dev : nullptr;
    d.alt_stride = stride;
    d.alt_off = kPoseRecordBytes;
  }
}

}  // namespace pk
