// Input staging for channels-last networks: NCHW fp32 (what a DataSetIterator delivers) -> NHWC in the compute type, in ONE pass.
// The composition it replaces (cast kernel, then a permuting copy) reads the image twice and writes it twice; for the 3-channel
// ImageNet input of the ResNet-50 benchmark that was 0.15 ms per step.
//
// One thread per pixel: the C plane values are read coalesced across the wave (C <= 4), the interleaved pixel is written as one
// contiguous run (a wave writes 64 * C consecutive elements).
#include "common.h"

namespace {

template <typename T, int C>
__global__ void __launch_bounds__(256) nchw_f32_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t HW, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / HW, p = i - n * HW;
    const float* src = x + n * C * HW + p;
    T* dst = y + i * C;
#pragma unroll
    for (int c = 0; c < C; ++c) Cvt<T>::st(dst, c, src[c * HW]);
  }
}

template <typename T>
int launch(const float* x, void* y, int64_t N, int C, int64_t HW, hipStream_t st) {
  const int64_t total = N * HW;
  const int grid = grid_for(total, 256);
  switch (C) {
    case 1: hipLaunchKernelGGL((nchw_f32_to_nhwc_kernel<T, 1>), dim3(grid), dim3(256), 0, st, x, (T*)y, HW, total); break;
    case 2: hipLaunchKernelGGL((nchw_f32_to_nhwc_kernel<T, 2>), dim3(grid), dim3(256), 0, st, x, (T*)y, HW, total); break;
    case 3: hipLaunchKernelGGL((nchw_f32_to_nhwc_kernel<T, 3>), dim3(grid), dim3(256), 0, st, x, (T*)y, HW, total); break;
    case 4: hipLaunchKernelGGL((nchw_f32_to_nhwc_kernel<T, 4>), dim3(grid), dim3(256), 0, st, x, (T*)y, HW, total); break;
    default: return -2;
  }
  DL4J_RET_LAST();
}

}  // namespace

// x: contiguous NCHW fp32; y: N*HW*C elements of dtype (0 fp32, 1 bf16) in NHWC order. C in 1..4.
DL4J_API int dl4j_nchw_f32_to_nhwc(const float* x, void* y, int64_t N, int C, int64_t HW, int dtype, hipStream_t st) {
  if (N <= 0 || HW <= 0) return 0;
  if (!x || !y || C < 1 || C > 4) return -2;
  if (dtype == 1) return launch<bf16_t>(x, y, N, C, HW, st);
  if (dtype == 0) return launch<float>(x, y, N, C, HW, st);
  return -2;
}
