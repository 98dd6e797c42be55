// Section 8(f4) on the device: the per-unmatched-blob bookkeeping of the new-landmark machinery (prkt_core_v2.py:546-746) --
// add_hypothesis / find_nearest_reading / reading_distance_function / ray_intersect / color_distance / add_new_feature /
// cross_readings / add_orphaned_reading -- for every particle in ONE launch, with the one change that makes it work (DESIGN.md
// section 9): the nearest-reading search walks the particle's stored readings.  The tests check it particle by particle against a CPU restatement.
//
// Per particle, in HBM: a ring of stored (orphaned) readings, 8 doubles each -- id, x, y, heading, bearing, r, g, b (:739-746) --,
// the counters (readings stored, spare slots in use, next_id :298, readings dropped because the ring was full), the feature id of
// every spare slot in use.  They follow the particles through the resample like the poses do (k_grow_gather).
//
// Hand-written gfx950 (CDNA4, wave64); one wave per particle: the blobs of a scan are taken in order, each seeing what the ones
// before it stored (:92-95) -- that order is sequential by nature; the search over the stored readings is not.
#include "pk_device.hpp"

namespace pk {

// The reference's arithmetic, operation by operation (no contraction into fused multiply-adds: the checker evaluates the same
// expressions in the same order; cos / sin are the device library's, within an ulp or two of the host's).
#pragma clang fp contract(off)
// :611-643 -- do the half-lines (x1, y1, b1) and (x3, y3, b3) meet (both ray parameters >= 0)
__device__ __forceinline__ bool grow_ray_intersect(double x1, double y1, double b1, double x3, double y3, double b3) {
  double ax, ay, bx, by;
  sincos(b1, &ay, &ax);
  sincos(b3, &by, &bx);
  const double den = ay * bx - ax * by;
  if (den == 0.0) return false;
  const double v = (ax * y3 - ay * x3 + ay * x1 - ax * y1) / den;
  double u;
  if (fabs(ay) < fabs(ax))
    u = (x3 + bx * v - x1) / ax;
  else
    u = (y3 + by * v - y1) / ay;
  return u >= 0.0 && v >= 0.0;
}
// :688-737 -- intersection of the two LINES (world bearings h1, h3); false when parallel
__device__ __forceinline__ bool grow_cross_readings(double x1, double y1, double h1, double x3, double y3, double h3, double& ox,
                                                    double& oy) {
  double c1, s1, c3, s3;
  sincos(h1, &s1, &c1);
  sincos(h3, &s3, &c3);
  const double x2 = x1 + c1, y2 = y1 + s1, x4 = x3 + c3, y4 = y3 + s3;
  const double t0 = x1 * y2 - y1 * x2, t3 = x3 * y4 - x4 * y3;
  const double den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
  if (den == 0.0) return false;
  ox = (t0 * (x3 - x4) - (x1 - x2) * t3) / den;
  oy = (t0 * (y3 - y4) - (y1 - y2) * t3) / den;
  return true;
}

// One WAVE per particle: the lanes read the particle's row of ids side by side (64 blobs a turn) and vote the unmatched ones out;
// those are taken one after the other, in scan order (each sees what the ones before it stored: the loop over the vote is
// wave-uniform); the nearest-reading search of one blob runs over the stored readings lane-parallel, the first minimum wins as in the
// reference's loop (strict <: the smallest distance, then the smallest index); lane 0 writes.
__global__ void __launch_bounds__(256) k_new_landmarks(GrowState g, const double* __restrict__ x, const double* __restrict__ y,
                                                       const double* __restrict__ h, const int32_t* __restrict__ ids,
                                                       const double* __restrict__ blobs, int B, unsigned char* __restrict__ map,
                                                       size_t slot_bytes, size_t count_off, int Lp, int64_t P,
                                                       const unsigned* __restrict__ unm, int unm_words,
                                                       const unsigned char* __restrict__ pflag) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t p = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  if (p >= P) return;  // wave-uniform
  const int c = g.cur;
  int32_t* cnt = g.cnt[c] + 4 * p;
  double* ring = g.hyp[c] + (size_t)p * g.R * 8;
  int32_t* sid = g.slot_id[c] + (size_t)p * g.S;
  int n = cnt[0], used = cnt[1], next_id = cnt[2], dropped = cnt[3];
  const double px = x[p], py = y[p], ph = h[p];
  double* f = reinterpret_cast<double*>(map + (size_t)p * slot_bytes);
  int32_t* fc = reinterpret_cast<int32_t*>(map + (size_t)p * slot_bytes + count_off);
  const int32_t* row = ids + (size_t)p * B;
  // (round 6) behind a one-pass kernel the unmatched blobs are a bit row the kernel left (scan order); the particles it handed to the
  // fall-back kernels have their ids in HBM as before
  const bool bits = unm != nullptr && (pflag == nullptr || pflag[p] == 0);  // wave-uniform
  const unsigned* urow = unm + (size_t)p * (bits ? unm_words : 0);
  for (int b0 = 0; b0 < B; b0 += kWave) {  // the unmatched blobs of the scan, in scan order (:88-95)
    const int bl = b0 + lane;
    unsigned long long vote;
    if (bits) {
      const int w = b0 >> 5;
      const unsigned lo = urow[w], hi = (w + 1 < unm_words) ? urow[w + 1] : 0u;
      vote = (unsigned long long)lo | ((unsigned long long)hi << 32);
      if (B - b0 < 64) vote &= (1ull << (B - b0)) - 1ull;
    } else {
      vote = __ballot(bl < B && row[bl] == 0);
    }
    while (vote != 0ull) {  // wave-uniform
      const int b = b0 + __builtin_ctzll(vote);
      vote &= vote - 1ull;
      const double zb = blobs[4 * b], zr = blobs[4 * b + 1], zg = blobs[4 * b + 2], zc = blobs[4 * b + 3];
      int best = -1;
      double best_d = INFINITY;
      for (int r = lane; r < n; r += kWave) {  // find_nearest_reading :566-590 over the stored readings
        const double* rd = ring + 8 * r;
        if (!grow_ray_intersect(rd[1], rd[2], rd[4] + rd[3], px, py, zb + ph)) continue;  // :592-609
        const double d = sqrt((rd[5] - zr) * (rd[5] - zr) + (rd[6] - zg) * (rd[6] - zg) + (rd[7] - zc) * (rd[7] - zc));  // :645-654
        if (d < best_d) {
          best = r;
          best_d = d;
        }
      }
#pragma unroll
      for (int off = kWave / 2; off >= 1; off >>= 1) {  // the smallest distance, then the smallest index: the reference's first minimum
        const double od = __shfl_xor(best_d, off);
        const int orr = __shfl_xor(best, off);
        if (orr >= 0 && (best < 0 || od < best_d || (od == best_d && orr < best))) {
          best = orr;
          best_d = od;
        }
      }
      bool made = false;
      if (best >= 0 && best_d < g.pair_threshold && used < g.S) {  // wave-uniform
        const double* rd = ring + 8 * best;
        double ox, oy;
        if (grow_cross_readings(rd[1], rd[2], rd[3] + rd[4], px, py, ph + zb, ox, oy)) {  // add_new_feature :656-686
          if (lane == 0) {
            const int l = g.L0 + used;
            f[(size_t)F_MX * Lp + l] = ox;
            f[(size_t)F_MY * Lp + l] = oy;
            f[(size_t)F_MR * Lp + l] = (rd[5] + zr) / 2;
            f[(size_t)F_MG * Lp + l] = (rd[6] + zg) / 2;
            f[(size_t)F_MB * Lp + l] = (rd[7] + zc) / 2;
            f[(size_t)F_PXX * Lp + l] = 1.0;  // identity covariance (:676)
            f[(size_t)F_PXY * Lp + l] = 0.0;
            f[(size_t)F_PYY * Lp + l] = 1.0;
            f[(size_t)F_CRR * Lp + l] = 1.0;
            f[(size_t)F_CRG * Lp + l] = 0.0;
            f[(size_t)F_CRB * Lp + l] = 0.0;
            f[(size_t)F_CGG * Lp + l] = 1.0;
            f[(size_t)F_CGB * Lp + l] = 0.0;
            f[(size_t)F_CBB * Lp + l] = 1.0;
            fc[l] = kPotentialBit;  // update_count 0, potential (the reference's negative id)
            sid[used] = next_id;
          }
          ++used;
          made = true;
        }
      }
      if (!made) {  // add_orphaned_reading :739-746
        if (n < g.R) {
          if (lane < 8) {
            const double v = lane == 0 ? (double)next_id : lane == 1 ? px : lane == 2 ? py : lane == 3 ? ph : lane == 4 ? zb : lane == 5 ? zr : lane == 6 ? zg : zc;
            ring[8 * n + lane] = v;
          }
          ++n;
          __threadfence_block();  // the next blob's search reads it (other lanes of this wave)
        } else {
          ++dropped;  // (the reference's dict grows without bound; the ring holds g.R readings: counted, never silent)
        }
      }
      ++next_id;
    }
  }
  if (lane == 0) {
    cnt[0] = n;
    cnt[1] = used;
    cnt[2] = next_id;
    cnt[3] = dropped;
  }
}
#pragma clang fp contract(on)

void launch_new_landmarks(hipStream_t s, DeviceState& d, GrowState& g, const int32_t* ids_dev, const double* blobs_dev, int B,
                          const unsigned* unm_dev, int unm_words, const unsigned char* pflag_dev) {
  if (d.P == 0 || B == 0) return;
  hipLaunchKernelGGL(k_new_landmarks, dim3((unsigned)((d.P + 3) / 4)), dim3(256), 0, s, g, d.x[d.cur], d.y[d.cur], d.h[d.cur], ids_dev,
                     blobs_dev, B, d.map[d.mcur], d.lay.slot_bytes, d.lay.count_off, d.lay.Lp, d.P, unm_dev, unm_words, pflag_dev);
}

// the bookkeeping follows the particles through the resample (:243: the deepcopy of the whole particle): slot k takes ancestor
// anc[k]'s -- a particle of this filter, or (sharded filter, anc < 0) the tail of received record -anc - 1
__global__ void __launch_bounds__(64) k_grow_gather(GrowState g, const int32_t* __restrict__ anc, int64_t P,
                                                    const unsigned char* __restrict__ buf, size_t stride, size_t tail_off) {
  const int64_t k = blockIdx.x;
  if (k >= P) return;
  const int c = g.cur, n = c ^ 1;
  const int64_t a = anc[k];
  const int32_t *sc, *ss;
  const double* sr;
  if (a >= 0) {
    sc = g.cnt[c] + 4 * a;
    sr = g.hyp[c] + (size_t)a * g.R * 8;
    ss = g.slot_id[c] + (size_t)a * g.S;
  } else {
    const unsigned char* tail = buf + (size_t)(-(a + 1)) * stride + tail_off;
    sc = reinterpret_cast<const int32_t*>(tail);
    ss = sc + 4;
    sr = reinterpret_cast<const double*>(tail + grow_tail_readings_off(g.S));
  }
  int m = sc[0], used = sc[1];
  m = m < 0 ? 0 : m > g.R ? g.R : m;  // (a record is foreign data: never past the ring)
  used = used < 0 ? 0 : used > g.S ? g.S : used;
  if (threadIdx.x < 4) g.cnt[n][4 * k + threadIdx.x] = threadIdx.x == 0 ? m : threadIdx.x == 1 ? used : sc[threadIdx.x];
  double* dr = g.hyp[n] + (size_t)k * g.R * 8;
  for (int i = threadIdx.x; i < 8 * m; i += blockDim.x) dr[i] = sr[i];
  int32_t* ds = g.slot_id[n] + (size_t)k * g.S;
  for (int i = threadIdx.x; i < used; i += blockDim.x) ds[i] = ss[i];
}
void launch_grow_gather(hipStream_t s, GrowState& g, const int32_t* anc_dev, int64_t P, const unsigned char* buf_dev, size_t stride,
                        size_t tail_off) {
  if (P == 0) return;
  hipLaunchKernelGGL(k_grow_gather, dim3((unsigned)P), dim3(64), 0, s, g, anc_dev, P, buf_dev, stride, tail_off);
  g.cur ^= 1;
}

}  // namespace pk
