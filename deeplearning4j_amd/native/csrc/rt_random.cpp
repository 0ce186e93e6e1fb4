// Host side of the counter RNG: generator handles (root/node state) and host fill routines that
// produce bit-identical streams to the device kernels in random.hip.
// Reference role: libnd4j graph/RandomGenerator.h (root/node state, relativeT), NativeOps
// createRandomGenerator/getRandomGeneratorRelative* (legacy/NativeOps.h:1648-1661), and the legacy
// RandomBuffer API initRandom/refreshBuffer/reSeedBuffer (NativeOps.h:1005-1027).
#include "rt_common.h"
#include "philox.h"

#include <cmath>

namespace {
struct Generator {
  uint64_t root;
  uint64_t node;
  uint64_t offset;   // consumed draws; advanced by rewind
};
}  // namespace

RT_API void* rng_create(uint64_t root_seed, uint64_t node_seed) { return new Generator{root_seed, node_seed, 0}; }
RT_API void rng_destroy(void* g) { delete (Generator*)g; }
RT_API void rng_set_states(void* g, uint64_t root_seed, uint64_t node_seed) {
  auto* r = (Generator*)g;
  r->root = root_seed;
  r->node = node_seed;
  r->offset = 0;
}
RT_API uint64_t rng_root_state(void* g) { return ((Generator*)g)->root; }
RT_API uint64_t rng_node_state(void* g) { return ((Generator*)g)->node; }
RT_API uint64_t rng_offset(void* g) { return ((Generator*)g)->offset; }
// advance the stream by n draws (callers rewind after handing `offset` to a kernel)
RT_API void rng_rewind(void* g, uint64_t n) { ((Generator*)g)->offset += n; }

RT_API uint32_t rng_relative_int(void* g, uint64_t index) {
  auto* r = (Generator*)g;
  return philox::bits_at(r->root, r->node, r->offset + index);
}
RT_API uint64_t rng_relative_long(void* g, uint64_t index) {
  auto* r = (Generator*)g;
  uint64_t i = r->offset + 2 * index;
  return ((uint64_t)philox::bits_at(r->root, r->node, i) << 32) | philox::bits_at(r->root, r->node, i + 1);
}
RT_API float rng_relative_float(void* g, uint64_t index) { return philox::u01(rng_relative_int(g, index)); }
RT_API double rng_relative_double(void* g, uint64_t index) {
  uint64_t v = rng_relative_long(g, index);
  return philox::u01d((uint32_t)(v >> 32), (uint32_t)v);
}

// ---- host fills (stateless: key/stream/offset passed explicitly; mirror random.hip) ----
RT_API void rng_fill_uniform_f32(float* z, LongType n, uint64_t key, uint64_t stream, uint64_t offset, float lo, float hi) {
  rt::parallel_for(0, n, 1 << 16, [&](LongType b, LongType e) {
    for (LongType i = b; i < e; ++i) z[i] = lo + (hi - lo) * philox::u01(philox::bits_at(key, stream, offset + (uint64_t)i));
  });
}

// Box-Muller on element pairs (2k, 2k+1) -> element 2k gets the cos branch, 2k+1 the sin branch
RT_API void rng_fill_gaussian_f32(float* z, LongType n, uint64_t key, uint64_t stream, uint64_t offset, float mean, float stddev) {
  rt::parallel_for(0, (n + 1) / 2, 1 << 15, [&](LongType b, LongType e) {
    for (LongType p = b; p < e; ++p) {
      float u1 = philox::u01(philox::bits_at(key, stream, offset + 2 * (uint64_t)p));
      float u2 = philox::u01(philox::bits_at(key, stream, offset + 2 * (uint64_t)p + 1));
      float r = std::sqrt(-2.0f * std::log(u1)), t = 6.28318530717958647692f * u2;
      z[2 * p] = mean + stddev * r * std::cos(t);
      if (2 * p + 1 < n) z[2 * p + 1] = mean + stddev * r * std::sin(t);
    }
  });
}

RT_API void rng_fill_bernoulli_f32(float* z, LongType n, uint64_t key, uint64_t stream, uint64_t offset, float p) {
  rt::parallel_for(0, n, 1 << 16, [&](LongType b, LongType e) {
    for (LongType i = b; i < e; ++i) z[i] = philox::u01(philox::bits_at(key, stream, offset + (uint64_t)i)) < p ? 1.0f : 0.0f;
  });
}
