// Host-side reproductions of the two random streams the reference draws from, so that
// fixed-seed parity does not need Python on the hot path:
//
//  * numpy.random.seed(s) + numpy.random.normal(0, sigma, 1)   (prkt_core_v2.py:27,185-193)
//      legacy MT19937 seeded by Knuth's LCG, 53-bit doubles, Marsaglia polar method with
//      the second variate cached (NumPy's legacy_gauss).  Third-party: NumPy, not vendored
//      under /root/reference and not version-pinned by package.xml:51,59; the legacy
//      stream is frozen by NumPy's compatibility policy.  Pinned here by
//      tests/test_host_rng.py against numpy.random.RandomState.
//  * random.seed(s) + random.random()                           (prkt_core_v2.py:28,226)
//      CPython's MT19937 seeded with init_by_array([s]), same 53-bit construction.
//
// Pure host code: usable (and tested) without a GPU.
#include <cmath>
#include <cstdint>
#include <new>

#include "../../include/parakeet_slam.h"

namespace {

struct MT {
  uint32_t key[624];
  int pos;
  void init_genrand(uint32_t s) {
    for (int i = 0; i < 624; ++i) {
      key[i] = s;
      s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
    pos = 624;
  }
  void init_by_array(const uint32_t* init_key, int key_length) {
    init_genrand(19650218u);
    // init_genrand above leaves key[i] = f(i) with key[0] = seed: same recurrence as the
    // reference implementation (mt[i] = 1812433253 * (mt[i-1] ^ (mt[i-1] >> 30)) + i).
    int i = 1, j = 0;
    int k = 624 > key_length ? 624 : key_length;
    for (; k; --k) {
      key[i] = (key[i] ^ ((key[i - 1] ^ (key[i - 1] >> 30)) * 1664525u)) + init_key[j] + (uint32_t)j;
      ++i;
      ++j;
      if (i >= 624) {
        key[0] = key[623];
        i = 1;
      }
      if (j >= key_length) j = 0;
    }
    for (k = 623; k; --k) {
      key[i] = (key[i] ^ ((key[i - 1] ^ (key[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      ++i;
      if (i >= 624) {
        key[0] = key[623];
        i = 1;
      }
    }
    key[0] = 0x80000000u;
    pos = 624;
  }
  void gen() {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    int i;
    uint32_t y;
    for (i = 0; i < 624 - 397; ++i) {
      y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    }
    for (; i < 623; ++i) {
      y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    }
    y = (key[623] & UPPER) | (key[0] & LOWER);
    key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    pos = 0;
  }
  uint32_t next32() {
    if (pos == 624) gen();
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  double next_double() {
    uint32_t a = next32() >> 5, b = next32() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
};

}  // namespace

struct pk_rng {
  MT mt;
  bool has_gauss;
  double gauss;
};

extern "C" {

int pk_rng_create_numpy(uint32_t seed, pk_rng** out) {
  if (!out) return PK_ERR_INVALID;
  pk_rng* r = new (std::nothrow) pk_rng;
  if (!r) return PK_ERR_NOMEM;
  r->mt.init_genrand(seed);
  r->has_gauss = false;
  r->gauss = 0.0;
  *out = r;
  return PK_OK;
}

int pk_rng_create_python(uint32_t seed, pk_rng** out) {
  if (!out) return PK_ERR_INVALID;
  pk_rng* r = new (std::nothrow) pk_rng;
  if (!r) return PK_ERR_NOMEM;
  uint32_t key[1] = {seed};
  r->mt.init_by_array(key, 1);
  r->has_gauss = false;
  r->gauss = 0.0;
  *out = r;
  return PK_OK;
}

int pk_rng_destroy(pk_rng* r) {
  delete r;
  return PK_OK;
}

int pk_rng_standard_normal(pk_rng* r, int64_t n, double* out) {
  if (!r || (n > 0 && !out) || n < 0) return PK_ERR_INVALID;
  for (int64_t i = 0; i < n; ++i) {
    if (r->has_gauss) {
      out[i] = r->gauss;
      r->has_gauss = false;
      r->gauss = 0.0;
    } else {
      double f, x1, x2, r2;
      do {
        x1 = 2.0 * r->mt.next_double() - 1.0;
        x2 = 2.0 * r->mt.next_double() - 1.0;
        r2 = x1 * x1 + x2 * x2;
      } while (r2 >= 1.0 || r2 == 0.0);
      f = std::sqrt(-2.0 * std::log(r2) / r2);
      r->gauss = f * x1;
      r->has_gauss = true;
      out[i] = f * x2;
    }
  }
  return PK_OK;
}

int pk_rng_random(pk_rng* r, int64_t n, double* out) {
  if (!r || (n > 0 && !out) || n < 0) return PK_ERR_INVALID;
  for (int64_t i = 0; i < n; ++i) out[i] = r->mt.next_double();
  return PK_OK;
}

}  // extern "C"
