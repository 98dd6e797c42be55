// Device-side helpers shared by the kernel translation units (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cstring>
#include <type_traits>

#include "pk_kernels.hpp"
#include "pk_math.hpp"

namespace pk {

constexpr int kWave = 64;
constexpr int kObsThreads = 256;
constexpr int kRedBlocks = 1024;

// Workgroup barrier that orders LDS traffic only: global loads that are still in flight stay in
// flight (__syncthreads() carries a workgroup-scope fence and waits for them, vmcnt(0)).  Only for
// phases that exchange data through LDS alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Diagnostic build only (-DPK_STAMPS, never shipped): shader-clock stamps for per-phase cycle sums.
#ifdef PK_STAMPS
#define PK_STAMP(var) \
  unsigned long long var; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");
#else
#define PK_STAMP(var)
#endif

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;  // identical in every lane; butterfly order is fixed => deterministic
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, kWave));
  return v;
}

// Sum over a workgroup of NW waves; result valid in every thread.  Fixed order.
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* lds /* >= NW doubles */) {
  v = wave_sum(v);
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  __syncthreads();
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double t = lds[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t += lds[i];
  return t;
}
// The same sum with barriers that order LDS only: the workgroup's stores stay in flight (a
// __syncthreads() waits for vmcnt(0), i.e. until every store of the wave has been acknowledged).
template <int NW>
__device__ __forceinline__ double block_sum_lds_only(double v, double* lds /* >= NW doubles, not in use */, int tid) {
  v = wave_sum(v);
  const int wave = tid / kWave, lane = tid % kWave;
  if (lane == 0) lds[wave] = v;
  lds_barrier();
  double t = lds[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t += lds[i];
  return t;
}
template <int NW>
__device__ __forceinline__ double block_max(double v, double* lds) {
  v = wave_max(v);
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  __syncthreads();
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double t = lds[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmax(t, lds[i]);
  return t;
}

// Order-preserving map double -> uint64 (max of keys == max of doubles), for the running max
// of the log-weights that the observe kernels keep with one atomicMax per particle.
__host__ __device__ inline unsigned long long double_to_key(double x) {
  unsigned long long b;
  memcpy(&b, &x, 8);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double key_to_double(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double x;
  memcpy(&x, &b, 8);
  return x;
}

// ------------------------------------------------------------------ landmark slot access
__device__ __forceinline__ Landmark<double> load_landmark(const double* f, const int* cnt, int Lp, int l) {
  Landmark<double> m;
  m.mx = f[F_MX * Lp + l];
  m.my = f[F_MY * Lp + l];
  m.mr = f[F_MR * Lp + l];
  m.mg = f[F_MG * Lp + l];
  m.mb = f[F_MB * Lp + l];
  m.pxx = f[F_PXX * Lp + l];
  m.pxy = f[F_PXY * Lp + l];
  m.pyy = f[F_PYY * Lp + l];
  m.crr = f[F_CRR * Lp + l];
  m.crg = f[F_CRG * Lp + l];
  m.crb = f[F_CRB * Lp + l];
  m.cgg = f[F_CGG * Lp + l];
  m.cgb = f[F_CGB * Lp + l];
  m.cbb = f[F_CBB * Lp + l];
  m.count = cnt[l];
  return m;
}

// The same with the five mean rows requested FIRST (vmcnt retires in order: a kernel whose first phase
// needs only the means then waits for five loads, not for wherever the scheduler put them).
__device__ __forceinline__ Landmark<double> load_landmark_means_first(const double* f, const int* cnt, int Lp, int l) {
  Landmark<double> m;
  m.mx = f[F_MX * Lp + l];
  m.my = f[F_MY * Lp + l];
  m.mr = f[F_MR * Lp + l];
  m.mg = f[F_MG * Lp + l];
  m.mb = f[F_MB * Lp + l];
  asm volatile("" ::: "memory");
  m.pxx = f[F_PXX * Lp + l];
  m.pxy = f[F_PXY * Lp + l];
  m.pyy = f[F_PYY * Lp + l];
  m.crr = f[F_CRR * Lp + l];
  m.crg = f[F_CRG * Lp + l];
  m.crb = f[F_CRB * Lp + l];
  m.cgg = f[F_CGG * Lp + l];
  m.cgb = f[F_CGB * Lp + l];
  m.cbb = f[F_CBB * Lp + l];
  m.count = cnt[l];
  return m;
}

__device__ __forceinline__ Landmark<double> load_landmark_nocount(const double* f, int Lp, int l) {
  Landmark<double> m;
  m.mx = f[F_MX * Lp + l];
  m.my = f[F_MY * Lp + l];
  m.mr = f[F_MR * Lp + l];
  m.mg = f[F_MG * Lp + l];
  m.mb = f[F_MB * Lp + l];
  m.pxx = f[F_PXX * Lp + l];
  m.pxy = f[F_PXY * Lp + l];
  m.pyy = f[F_PYY * Lp + l];
  m.crr = f[F_CRR * Lp + l];
  m.crg = f[F_CRG * Lp + l];
  m.crb = f[F_CRB * Lp + l];
  m.cgg = f[F_CGG * Lp + l];
  m.cgb = f[F_CGB * Lp + l];
  m.cbb = f[F_CBB * Lp + l];
  m.count = 0;
  return m;
}

}  // namespace pk
