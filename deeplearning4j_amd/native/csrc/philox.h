// Counter-based RNG shared by the host runtime and the HIP kernels (same bits on both sides).
//
// Plays the role of sd::graph::RandomGenerator (libnd4j/include/graph/RandomGenerator.h: root seed +
// per-node state, stateless "relativeT(index)" draws so every element can be generated
// independently by any thread). Own design: Philox4x32-10 keyed by (rootSeed, nodeSeed) with the
// element index as the counter — one 128-bit block yields 4 uniforms, so a wave64 fills 256
// elements per round with no shared state and no atomics.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define PHX_HD __host__ __device__ __forceinline__
#else
#define PHX_HD inline
#endif

namespace philox {

struct u4 {
  uint32_t x, y, z, w;
};

PHX_HD void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
  uint64_t p = (uint64_t)a * (uint64_t)b;
  hi = (uint32_t)(p >> 32);
  lo = (uint32_t)p;
}

PHX_HD u4 block(uint64_t counter, uint64_t stream, uint64_t key) {
  uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = (uint32_t)stream, c3 = (uint32_t)(stream >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t h0, l0, h1, l1;
    mulhilo(0xD2511F53u, c0, h0, l0);
    mulhilo(0xCD9E8D57u, c2, h1, l1);
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u4{c0, c1, c2, c3};
}

// uniform in (0,1): never returns exactly 0 or 1 (safe for log / inverse-CDF transforms)
PHX_HD float u01(uint32_t v) { return ((float)(v >> 8) + 0.5f) * (1.0f / 16777216.0f); }
PHX_HD double u01d(uint32_t hi, uint32_t lo) {
  uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
  return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

// element-addressed draws: element i uses block i/4, lane i%4
PHX_HD uint32_t bits_at(uint64_t key, uint64_t stream, uint64_t i) {
  u4 b = block(i >> 2, stream, key);
  switch (i & 3) {
    case 0: return b.x;
    case 1: return b.y;
    case 2: return b.z;
    default: return b.w;
  }
}

}  // namespace philox
