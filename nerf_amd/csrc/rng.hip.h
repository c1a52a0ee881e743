// rng.hip.h -- counter-based Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3").
// Used for givePerturb / givePerturbPDF / addNoise (nerfBlock.scala:21,55,71) and parameter init when
// the caller does not pass explicit random tensors.  Stateless: value = f(seed, stream, step, index).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NH_HD __host__ __device__ __forceinline__
#else
#define NH_HD inline
#endif

namespace nh {

struct U4 { uint32_t x, y, z, w; };

NH_HD void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
  const uint64_t p = (uint64_t)a * (uint64_t)b;
  hi = (uint32_t)(p >> 32);
  lo = (uint32_t)p;
}

NH_HD U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1) {
  for (int i = 0; i < 10; ++i) {
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo(0xD2511F53u, ctr.x, hi0, lo0);
    mulhilo(0xCD9E8D57u, ctr.z, hi1, lo1);
    U4 n;
    n.x = hi1 ^ ctr.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ ctr.w ^ k1;
    n.w = lo0;
    ctr = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return ctr;
}

// uniform in (0,1): 24 random bits + half-ulp offset, never 0 or 1
NH_HD float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

enum { kStreamPerturb = 1, kStreamPdf = 2, kStreamNoiseC = 3, kStreamNoiseF = 4, kStreamInit = 5 };

NH_HD U4 philox_at(uint64_t seed, uint32_t stream, uint32_t step, uint64_t index) {
  U4 c;
  c.x = (uint32_t)index;
  c.y = (uint32_t)(index >> 32);
  c.z = stream;
  c.w = step;
  return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

NH_HD float philox_uniform(uint64_t seed, uint32_t stream, uint32_t step, uint64_t index) {
  return u01(philox_at(seed, stream, step, index).x);
}

#if defined(__HIPCC__)
__device__ __forceinline__ float philox_normal(uint64_t seed, uint32_t stream, uint32_t step, uint64_t index) {
  const U4 r = philox_at(seed, stream, step, index);
  const float u1 = u01(r.x), u2 = u01(r.y);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}
#endif

}  // namespace nh
