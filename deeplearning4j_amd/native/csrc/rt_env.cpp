// Process-wide Environment flags, the named launch-dims registry and the op-trace ring.
//
// Reference roles: sd::Environment (libnd4j/include/system/Environment.h:41-163), LaunchDims
// registry with GRID_SIZE_*/BLOCK_SIZE_*/SHARED_MEM_SIZE_* env overrides
// (execution/cuda/LaunchDims.cu:11-160,346-400) and the op-trace API
// (NativeOps.h:109-125 printOpTrace/listOpTraces/toggleOpTrace/purgeOpTrace).
// Defaults are chosen for MI355X (256 CUs, wave64): grid = a multiple of the CU count, 256-thread
// blocks (4 waves) — not the reference's sm_60-era 256x256 everywhere.
#include "rt_common.h"

#include <atomic>
#include <cstdlib>
#include <deque>
#include <map>
#include <mutex>

namespace {
struct Env {
  std::atomic<int> debug{0}, verbose{0}, profiling{0}, leak_detect{0}, helpers_allowed{1}, precision_boost{0},
      check_input_change{0}, log_native_ndarray_creation{0}, delete_shape_info{1}, delete_primary{1}, delete_special{1};
  std::atomic<int> tad_threshold{1}, elementwise_threshold{1024}, max_threads{16}, max_master_threads{16};
  std::atomic<int> default_float_dtype{5};   // sd::DataType FLOAT32
  std::atomic<LongType> max_primary_memory{0}, max_special_memory{0}, max_device_memory{0};
  std::atomic<int> func_trace{0};
} g_env;

struct Dims {
  int grid, block, smem;
};
std::mutex g_dims_mu;
std::map<std::string, Dims> g_dims;

const Dims kDefault = {1024, 256, 0};
const struct {
  const char* name;
  Dims d;
} kBuiltins[] = {
    {"pairwiseTransforms", {2048, 256, 0}}, {"scalarSame", {2048, 256, 0}},   {"transformScan", {2048, 256, 0}},
    {"broadcast", {2048, 256, 0}},          {"reduceFloat", {1024, 256, 2048}}, {"reduceSame", {1024, 256, 2048}},
    {"reduceBool", {1024, 256, 2048}},      {"reduceLong", {1024, 256, 2048}},  {"reduce3", {1024, 256, 4096}},
    {"summaryStats", {1024, 256, 6144}},    {"indexReduce", {1024, 256, 4096}}, {"random", {2048, 256, 0}},
    {"average", {2048, 256, 0}},            {"accumulate", {2048, 256, 0}},     {"updater", {2048, 256, 0}},
    {"batch_norm", {2048, 256, 4096}},      {"pooling", {2048, 256, 0}},        {"im2col", {2048, 256, 0}},
    {"col2im", {2048, 256, 0}},             {"concat", {2048, 256, 0}},         {"gather", {2048, 256, 0}},
    {"scatter", {2048, 256, 0}},            {"lambda", {2048, 256, 0}},         {"dropout", {2048, 256, 0}},
    {"merge", {2048, 256, 0}},              {"clip", {1024, 256, 2048}},        {"mmul", {1024, 256, 65536}},
    {"matrixMultiply", {1024, 256, 65536}}, {"sortTensorByDimKey", {512, 256, 8192}}, {"pullRows", {1024, 256, 0}},
    {"shuffle", {1024, 256, 0}},            {"tear", {1024, 256, 0}},           {"tile", {2048, 256, 0}},
    {"flatten", {2048, 256, 0}},            {"lrn", {1024, 256, 8192}},         {"prefix", {1024, 256, 4096}},
    {"segment", {1024, 256, 0}},            {"top_k_mover", {1024, 256, 0}},    {"one_hot", {2048, 256, 0}},
    {"pad", {2048, 256, 0}},                {"upsampling", {2048, 256, 0}},     {"histogram", {1024, 256, 4096}},
    {"image_resize", {2048, 256, 0}},       {"addBias", {2048, 256, 0}},        {"encoder", {2048, 256, 1024}},
};

int env_int(const std::string& key, int fallback) {
  const char* v = std::getenv(key.c_str());
  return v && *v ? std::atoi(v) : fallback;
}

struct Trace {
  std::string op;
  double ms;
};
std::mutex g_trace_mu;
std::deque<Trace> g_traces;
std::atomic<int> g_trace_on{0};
}  // namespace

#define ENV_FLAG(name, field)                                  \
  RT_API void env_set_##name(int v) { g_env.field.store(v); }  \
  RT_API int env_get_##name() { return g_env.field.load(); }

ENV_FLAG(debug, debug)
ENV_FLAG(verbose, verbose)
ENV_FLAG(profiling, profiling)
ENV_FLAG(leak_detect, leak_detect)
ENV_FLAG(helpers_allowed, helpers_allowed)
ENV_FLAG(precision_boost, precision_boost)
ENV_FLAG(check_input_change, check_input_change)
ENV_FLAG(tad_threshold, tad_threshold)
ENV_FLAG(elementwise_threshold, elementwise_threshold)
ENV_FLAG(max_threads, max_threads)
ENV_FLAG(max_master_threads, max_master_threads)
ENV_FLAG(default_float_dtype, default_float_dtype)
ENV_FLAG(func_trace, func_trace)

RT_API void env_set_max_primary_memory(LongType v) { g_env.max_primary_memory.store(v); }
RT_API LongType env_get_max_primary_memory() { return g_env.max_primary_memory.load(); }
RT_API void env_set_max_special_memory(LongType v) { g_env.max_special_memory.store(v); }
RT_API LongType env_get_max_special_memory() { return g_env.max_special_memory.load(); }
RT_API void env_set_max_device_memory(LongType v) { g_env.max_device_memory.store(v); }
RT_API LongType env_get_max_device_memory() { return g_env.max_device_memory.load(); }

// out = {grid, block, smemBytes}. Env override: GRID_SIZE_<NAME>, BLOCK_SIZE_<NAME>, SHARED_MEM_SIZE_<NAME>
// with <NAME> upper-cased.
RT_API void launch_dims_get(const char* name, int* out) {
  std::string key(name);
  Dims d = kDefault;
  {
    std::lock_guard<std::mutex> lk(g_dims_mu);
    auto it = g_dims.find(key);
    if (it != g_dims.end()) d = it->second;
    else
      for (auto& b : kBuiltins)
        if (key == b.name) {
          d = b.d;
          break;
        }
  }
  std::string up = key;
  for (auto& c : up) c = (char)std::toupper((unsigned char)c);
  out[0] = env_int("GRID_SIZE_" + up, d.grid);
  out[1] = env_int("BLOCK_SIZE_" + up, d.block);
  out[2] = env_int("SHARED_MEM_SIZE_" + up, d.smem);
}

RT_API void launch_dims_set(const char* name, int grid, int block, int smem) {
  std::lock_guard<std::mutex> lk(g_dims_mu);
  g_dims[name] = Dims{grid, block, smem};
}

RT_API int launch_dims_count() { return (int)(sizeof(kBuiltins) / sizeof(kBuiltins[0])); }
RT_API const char* launch_dims_name(int i) { return kBuiltins[i].name; }

// ---- op trace ring (bounded) ----
RT_API void op_trace_toggle(int on) { g_trace_on.store(on); }
RT_API int op_trace_enabled() { return g_trace_on.load(); }
RT_API void op_trace_record(const char* op, double ms) {
  if (!g_trace_on.load()) return;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_traces.push_back(Trace{op, ms});
  if (g_traces.size() > 65536) g_traces.pop_front();
}
RT_API int op_trace_count() {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  return (int)g_traces.size();
}
RT_API void op_trace_purge() {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_traces.clear();
}
// copies "op\tms\n" lines into buf; returns bytes needed
RT_API LongType op_trace_list(char* buf, LongType cap) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  std::string s;
  for (auto& t : g_traces) s += t.op + "\t" + std::to_string(t.ms) + "\n";
  if (buf && cap > 0) {
    LongType n = std::min<LongType>(cap - 1, (LongType)s.size());
    std::memcpy(buf, s.data(), (size_t)n);
    buf[n] = 0;
  }
  return (LongType)s.size() + 1;
}
