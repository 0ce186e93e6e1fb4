// Launch-dimension overrides honoured by the kernels of this library (reference: execution/cuda/LaunchDims.cu:346-400 — a registry
// of {grid, block, shared memory} per algorithm name with GRID_SIZE_<NAME> / BLOCK_SIZE_<NAME> environment overrides).
// Every grid-stride kernel family asks launch_grid(site, computed) before it launches: an override caps the number of workgroups
// (the kernels stride over the remaining work), which is what tuning a launch on a shared or partitioned GPU needs.
#include "common.h"

#include <atomic>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {
struct Site { const char* names[5]; };
// reference algorithm names that map onto each kernel family here
const Site kSites[LS_COUNT] = {
    {{"pairwiseTransforms", "scalarSame", "broadcast", nullptr}},                    // LS_PAIRWISE
    {{"transformScan", "lambda", nullptr}},                                          // LS_TRANSFORM
    {{"reduceFloat", "reduceSame", "reduceBool", "reduceLong", nullptr}},            // LS_REDUCE
    {{"batch_norm", nullptr}},                                                       // LS_BATCHNORM
    {{"pooling", nullptr}},                                                          // LS_POOLING
};
std::atomic<int> g_grid[LS_COUNT], g_block[LS_COUNT], g_last[LS_COUNT];
std::atomic<int> g_env_read{0};

int env_int(const std::string& key) {
  const char* v = std::getenv(key.c_str());
  return v && *v ? std::atoi(v) : 0;
}
void read_env_once() {
  if (g_env_read.exchange(1)) return;
  for (int s = 0; s < LS_COUNT; ++s)
    for (int k = 0; k < 5 && kSites[s].names[k]; ++k) {
      std::string up = kSites[s].names[k];
      for (auto& c : up) c = (char)std::toupper((unsigned char)c);
      const int g = env_int("GRID_SIZE_" + up), b = env_int("BLOCK_SIZE_" + up);
      if (g > 0) g_grid[s].store(g);
      if (b > 0) g_block[s].store(b);
    }
}
int site_of(const char* name) {
  for (int s = 0; s < LS_COUNT; ++s)
    for (int k = 0; k < 5 && kSites[s].names[k]; ++k)
      if (std::strcmp(kSites[s].names[k], name) == 0) return s;
  return -1;
}
}  // namespace

int launch_grid(int site, int computed) {
  read_env_once();
  const int o = g_grid[site].load(std::memory_order_relaxed);
  const int g = (o > 0 && o < computed) ? o : computed;
  g_last[site].store(g, std::memory_order_relaxed);
  return g;
}
int launch_block(int site, int dflt) {
  read_env_once();
  const int o = g_block[site].load(std::memory_order_relaxed);
  return (o >= 64 && o <= 1024 && (o & 63) == 0) ? o : dflt;
}

// grid / block <= 0 remove the override. Returns 0 when the name belongs to a kernel family of this library, 1 otherwise.
DL4J_API int dl4j_launch_dims_set(const char* name, int grid, int block, int /*smem*/) {
  read_env_once();
  const int s = site_of(name);
  if (s < 0) return 1;
  g_grid[s].store(grid > 0 ? grid : 0);
  g_block[s].store(block > 0 ? block : 0);
  return 0;
}
// out = {grid override (0 = none), block override, workgroups of the most recent launch of the family}
DL4J_API int dl4j_launch_dims_get(const char* name, int* out) {
  read_env_once();
  const int s = site_of(name);
  if (s < 0) return 1;
  out[0] = g_grid[s].load(); out[1] = g_block[s].load(); out[2] = g_last[s].load();
  return 0;
}
