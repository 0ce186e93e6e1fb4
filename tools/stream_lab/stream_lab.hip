// Streaming-kernel lab: y = max(x * s + b, 0) over bf16 [M, C] (the BatchNorm apply pattern) in several launch / loop shapes, to find
// what reaches the HBM rate on MI355X. Build: hipcc --offload-arch=gfx950 -O3 -o stream_lab stream_lab.hip ; run: ./stream_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) { __bf16 b = (__bf16)f; return __builtin_bit_cast(uint16_t, b); }
__device__ __forceinline__ s16x8 apply8(s16x8 v, const float* sc, const float* sh) {
  s16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(fmaxf(bf2f((uint16_t)v[j]) * sc[j] + sh[j], 0.f));
  return o;
}
// V0: grid-stride, one vector per iteration (the current bn_apply shape), hoisted coefficients
__global__ void __launch_bounds__(256) k_stride(const s16x8* x, s16x8* y, const float* scale, const float* shift, int64_t nvec, int cv) {
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, st = (int64_t)gridDim.x * 256;
  float sc[8], sh[8];
  const int c0 = (int)(t0 % cv) * 8;
  for (int j = 0; j < 8; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
  for (int64_t i = t0; i < nvec; i += st) y[i] = apply8(x[i], sc, sh);
}
// V1: grid-stride, U vectors per iteration with all loads issued first
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_unroll(const s16x8* x, s16x8* y, const float* scale, const float* shift, int64_t nvec, int cv) {
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, st = (int64_t)gridDim.x * 256;
  float sc[8], sh[8];
  const int c0 = (int)(t0 % cv) * 8;
  for (int j = 0; j < 8; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
  int64_t i = t0;
  for (; i + (U - 1) * st < nvec; i += U * st) {
    s16x8 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * st) : x[i + u * st];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const s16x8 o = apply8(v[u], sc, sh);
      if (NT) __builtin_nontemporal_store(o, y + i + u * st); else y[i + u * st] = o;
    }
  }
  for (; i < nvec; i += st) y[i] = apply8(x[i], sc, sh);
}
// V2: contiguous chunk per block (block b owns vectors [b*chunk, (b+1)*chunk)), U-unrolled
template <int U>
__global__ void __launch_bounds__(256) k_chunk(const s16x8* x, s16x8* y, const float* scale, const float* shift, int64_t nvec, int cv, int64_t chunk) {
  const int64_t b0 = (int64_t)blockIdx.x * chunk;
  int64_t e = b0 + chunk; if (e > nvec) e = nvec;
  for (int64_t i = b0 + threadIdx.x; i < e; i += 256 * U) {
    s16x8 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < e) v[u] = x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < e) {
      float sc[8], sh[8];
      const int c0 = (int)((i + u * 256) % cv) * 8;
      const f32x4 a = *(const f32x4*)(scale + c0), b = *(const f32x4*)(scale + c0 + 4), c = *(const f32x4*)(shift + c0), d = *(const f32x4*)(shift + c0 + 4);
      for (int j = 0; j < 4; ++j) { sc[j] = a[j]; sc[4 + j] = b[j]; sh[j] = c[j]; sh[4 + j] = d[j]; }
      y[i + u * 256] = apply8(v[u], sc, sh);
    }
  }
}
__global__ void __launch_bounds__(256) k_copy(const s16x8* x, s16x8* y, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) y[i] = x[i];
}
template <class F> float timeit(F f, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main() {
  const int shapes[][2] = {{200704, 256}, {50176, 512}, {200704, 64}, {12544, 1024}, {4096, 2048}};
  // rotate through several buffers so the 256 MB Infinity Cache does not serve the reads
  for (auto& s : shapes) {
    const int64_t M = s[0]; const int C = s[1];
    const int64_t nvec = M * C / 8; const int cv = C / 8;
    const size_t bytes = (size_t)nvec * 16;
    const int NB = (int)(1200000000ull / (2 * bytes)) + 1;
    std::vector<s16x8*> xs(NB), ys(NB);
    for (int i = 0; i < NB; ++i) { hipMalloc(&xs[i], bytes); hipMalloc(&ys[i], bytes); hipMemset(xs[i], 0x3c, bytes); }
    float *sc, *sh; hipMalloc(&sc, C * 4); hipMalloc(&sh, C * 4); hipMemset(sc, 0, C * 4); hipMemset(sh, 0, C * 4);
    int k = 0;
    auto nx = [&]() { k = (k + 1) % NB; };
    const double gb = 2.0 * bytes / 1e9;
    printf("[%ld x %d] %.0f MB in + out, %d buffer pairs\n", (long)M, C, 2.0 * bytes / 1e6, NB);
    for (int g : {2048, 4096, 8192, 16384, 65536}) {
      const int gg = (int)((nvec + 255) / 256) < g ? (int)((nvec + 255) / 256) : g;
      int gh = gg; if (cv <= 256 && 256 % cv == 0) {} else if (cv % 256 == 0) { const int m = cv / 256; gh = (gg + m - 1) / m * m; }
      float t;
      t = timeit([&] { nx(); hipLaunchKernelGGL(k_copy, dim3(gg), dim3(256), 0, 0, xs[k], ys[k], nvec); });
      printf("  grid %6d  copy %6.1f us %5.2f TB/s |", gg, t, gb / t * 1e3);
      t = timeit([&] { nx(); hipLaunchKernelGGL(k_stride, dim3(gh), dim3(256), 0, 0, xs[k], ys[k], sc, sh, nvec, cv); });
      printf(" stride %6.1f %5.2f |", t, gb / t * 1e3);
      t = timeit([&] { nx(); hipLaunchKernelGGL((k_unroll<2, false>), dim3(gh), dim3(256), 0, 0, xs[k], ys[k], sc, sh, nvec, cv); });
      printf(" u2 %6.1f %5.2f |", t, gb / t * 1e3);
      t = timeit([&] { nx(); hipLaunchKernelGGL((k_unroll<4, false>), dim3(gh), dim3(256), 0, 0, xs[k], ys[k], sc, sh, nvec, cv); });
      printf(" u4 %6.1f %5.2f |", t, gb / t * 1e3);
      t = timeit([&] { nx(); hipLaunchKernelGGL((k_unroll<4, true>), dim3(gh), dim3(256), 0, 0, xs[k], ys[k], sc, sh, nvec, cv); });
      printf(" u4nt %6.1f %5.2f |", t, gb / t * 1e3);
      const int64_t chunk = (nvec + gg - 1) / gg;
      t = timeit([&] { nx(); hipLaunchKernelGGL((k_chunk<4>), dim3(gg), dim3(256), 0, 0, xs[k], ys[k], sc, sh, nvec, cv, chunk); });
      printf(" chunk4 %6.1f %5.2f\n", t, gb / t * 1e3);
    }
    for (int i = 0; i < NB; ++i) { hipFree(xs[i]); hipFree(ys[i]); }
    hipFree(sc); hipFree(sh);
  }
  return 0;
}
