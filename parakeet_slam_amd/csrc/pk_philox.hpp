// Philox4x32-10 counter-based generator + Box-Muller, for the throughput-mode motion
// noise (pk_motion with z == NULL).  Counter = (global particle index, draw index, pair),
// key = seed, so the stream is independent of how particles are sharded over GPUs.
// Works on host and device (tests restate it in NumPy).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pk {

struct Philox4 {
  uint32_t v[4];
};

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  return Philox4{{c0, c1, c2, c3}};
}

// 53-bit uniform in (0, 1): ((a >> 5) * 2^26 + (b >> 6) + 0.5) / 2^53
__host__ __device__ inline double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

}  // namespace pk
