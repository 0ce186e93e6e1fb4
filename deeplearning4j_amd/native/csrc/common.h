// Common device helpers for the gfx950 (MI355X / CDNA4) kernel library.
// Wave size is 64 on CDNA; every cross-lane idiom here is written for 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DL4J_API extern "C" __attribute__((visibility("default")))

typedef uint16_t bf16_t;   // raw bfloat16 bits
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAVE 64
// number of CUs on MI355X; used for grid sizing of memory-bound kernels
#define DL4J_NUM_CU 256
#define DL4J_MAX_GRID (DL4J_NUM_CU * 8)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// plain cast -> v_cvt_pk_bf16_f32 on gfx950 (RNE, keeps NaN a NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float ld(const float* p, int64_t i) { return p[i]; }
  static __device__ __forceinline__ void st(float* p, int64_t i, float v) { p[i] = v; }
};
template <> struct Cvt<double> {
  static __device__ __forceinline__ double ld(const double* p, int64_t i) { return p[i]; }
  static __device__ __forceinline__ void st(double* p, int64_t i, double v) { p[i] = v; }
};
template <> struct Cvt<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p, int64_t i) { return bf2f(p[i]); }
  static __device__ __forceinline__ void st(bf16_t* p, int64_t i, float v) { p[i] = f2bf(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (<=16 waves). `red` must hold 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}

static inline int grid_for(int64_t work_items, int per_block) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > DL4J_MAX_GRID) g = DL4J_MAX_GRID;
  return (int)g;
}

// launch-dimension overrides (launch_dims.hip; LaunchDims.cu of the reference): kernel families consult these before launching
enum LaunchSite { LS_PAIRWISE = 0, LS_TRANSFORM, LS_REDUCE, LS_BATCHNORM, LS_POOLING, LS_COUNT };
int launch_grid(int site, int computed);
int launch_block(int site, int dflt);

#define DL4J_RET_LAST() return (int)hipGetLastError()

// hipFuncSetAttribute is PER DEVICE: a process that drives several GPUs from worker threads (ParallelInference replicas)
// must raise the dynamic-LDS limit on each of them. One flag per (call site, device), set under a mutex so two threads
// that race on the first launch both see the attribute applied before they launch.
#include <mutex>
struct DeviceOnceLds {
  std::mutex mu;
  int bytes[64] = {0};
  void ensure(const void* fn, int want) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> g(mu);
    if (bytes[dev] >= want) return;
    hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, want);
    bytes[dev] = want;
  }
};
