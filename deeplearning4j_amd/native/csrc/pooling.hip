// 2-D max / average pooling forward + backward for channels-last (NHWC) activations.
// Reference semantics: libnd4j maxpool2d / avgpool2d (+_bp) custom ops
// (libnd4j/include/ops/declarable/helpers/cuda/convolutions_pooling2d.cu, _pooling2dBP.cu);
// iArgs contract [kH,kW,sH,sW,pH,pW,dH,dW,same,extra,format]. The reference backward scatters
// with atomicAdd per output element; here the forward records a 1-byte window-local argmax and
// the backward is a deterministic gather per input pixel (no atomics), 16-byte vector accesses.
#include "common.h"

struct PoolP {
  int N, H, W, C, OH, OW;
  int kH, kW, sH, sW, pH, pW, dH, dW;
  int avg_exclude_pad;
};

template <typename T, int V> struct PV;
template <> struct PV<bf16_t, 8> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&o)[8]) {
    s16x8 v = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bf2f((bf16_t)v[j]);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&o)[8]) {
    s16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(o[j]);
    *reinterpret_cast<s16x8*>(p) = v;
  }
};
template <> struct PV<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j];
  }
  static __device__ __forceinline__ void st(float* p, const float (&o)[4]) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = o[j];
    *reinterpret_cast<f32x4*>(p) = v;
  }
};
template <> struct PV<bf16_t, 1> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&o)[1]) { o[0] = bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&o)[1]) { *p = f2bf(o[0]); }
};
template <> struct PV<float, 1> {
  static __device__ __forceinline__ void ld(const float* p, float (&o)[1]) { o[0] = *p; }
  static __device__ __forceinline__ void st(float* p, const float (&o)[1]) { *p = o[0]; }
};

// mode 0 = max, 1 = avg
template <typename T, int V, int MODE>
__global__ void __launch_bounds__(256) pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                       uint8_t* __restrict__ amax, PoolP p) {
  const int cv = p.C / V;
  const int64_t total = (int64_t)p.N * p.OH * p.OW * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c0 = (int)(i % cv) * V;
    int64_t r = i / cv;
    const int ow = (int)(r % p.OW); r /= p.OW;
    const int oh = (int)(r % p.OH);
    const int n = (int)(r / p.OH);
    float acc[V];
    int arg[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { acc[j] = MODE == 0 ? -INFINITY : 0.f; arg[j] = 0; }
    int cnt = 0;
    for (int kh = 0; kh < p.kH; ++kh) {
      const int h = oh * p.sH - p.pH + kh * p.dH;
      if (h < 0 || h >= p.H) continue;
      for (int kw = 0; kw < p.kW; ++kw) {
        const int w = ow * p.sW - p.pW + kw * p.dW;
        if (w < 0 || w >= p.W) continue;
        float v[V];
        PV<T, V>::ld(x + (((int64_t)n * p.H + h) * p.W + w) * p.C + c0, v);
        ++cnt;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          if (MODE == 0) { if (v[j] > acc[j]) { acc[j] = v[j]; arg[j] = kh * p.kW + kw; } }
          else acc[j] += v[j];
        }
      }
    }
    if (MODE == 1) {
      const float d = p.avg_exclude_pad ? (float)(cnt > 0 ? cnt : 1) : (float)(p.kH * p.kW);
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] /= d;
    }
    const int64_t o = (((int64_t)n * p.OH + oh) * p.OW + ow) * p.C + c0;
    PV<T, V>::st(y + o, acc);
    if (MODE == 0 && amax != nullptr) {
#pragma unroll
      for (int j = 0; j < V; ++j) amax[o + j] = (uint8_t)arg[j];
    }
  }
}

// Gather-style backward: every input position sums the contributions of the (few) output windows that cover it.
// The covering windows are enumerated directly (oh in [ceil((h+pH-(kH-1)dH)/sH), floor((h+pH)/sH)]) instead of testing all
// kH*kW taps with div/mod, and the index decomposition runs in 32-bit arithmetic: the original tap loop with 64-bit
// div/mod was ALU-bound at ~1.1 TB/s on the ResNet stem pool.
template <typename T, int V, int MODE>
__global__ void __launch_bounds__(256) pool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ amax,
                                                       T* __restrict__ dx, PoolP p) {
  const unsigned cv = p.C / V;
  const int64_t total = (int64_t)p.N * p.H * p.W * cv;
  const int spanH = (p.kH - 1) * p.dH, spanW = (p.kW - 1) * p.dW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c0, w, h, n;
    if (total < (int64_t)0xFFFFFFFFll) {
      unsigned r = (unsigned)i;
      c0 = (int)(r % cv) * V; r /= cv;
      w = (int)(r % (unsigned)p.W); r /= (unsigned)p.W;
      h = (int)(r % (unsigned)p.H);
      n = (int)(r / (unsigned)p.H);
    } else {
      int64_t r = i;
      c0 = (int)(r % cv) * V; r /= cv;
      w = (int)(r % p.W); r /= p.W;
      h = (int)(r % p.H);
      n = (int)(r / p.H);
    }
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    const int hp = h + p.pH, wp = w + p.pW;
    int oh_hi = hp / p.sH; if (oh_hi > p.OH - 1) oh_hi = p.OH - 1;
    int ow_hi = wp / p.sW; if (ow_hi > p.OW - 1) ow_hi = p.OW - 1;
    int oh_lo = hp - spanH; oh_lo = oh_lo <= 0 ? 0 : (oh_lo + p.sH - 1) / p.sH;
    int ow_lo = wp - spanW; ow_lo = ow_lo <= 0 ? 0 : (ow_lo + p.sW - 1) / p.sW;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int rh = hp - oh * p.sH;
      int kh = rh;
      if (p.dH != 1) { if (rh % p.dH) continue; kh = rh / p.dH; }
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int rw = wp - ow * p.sW;
        int kw = rw;
        if (p.dW != 1) { if (rw % p.dW) continue; kw = rw / p.dW; }
        const int64_t o = (((int64_t)n * p.OH + oh) * p.OW + ow) * p.C + c0;
        float d[V];
        PV<T, V>::ld(dy + o, d);
        if (MODE == 0) {
          const int code = kh * p.kW + kw;
          if constexpr (V == 8) {
            const uint2 am = *reinterpret_cast<const uint2*>(amax + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[j] += (int)((am.x >> (8 * j)) & 0xFF) == code ? d[j] : 0.f;
              acc[4 + j] += (int)((am.y >> (8 * j)) & 0xFF) == code ? d[4 + j] : 0.f;
            }
          } else {
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += (amax[o + j] == code) ? d[j] : 0.f;
          }
        } else {
          float dv = (float)(p.kH * p.kW);
          if (p.avg_exclude_pad) {
            int cnt = 0;
            for (int a = 0; a < p.kH; ++a) {
              const int hh = oh * p.sH - p.pH + a * p.dH;
              if (hh < 0 || hh >= p.H) continue;
              for (int b = 0; b < p.kW; ++b) {
                const int ww = ow * p.sW - p.pW + b * p.dW;
                if (ww >= 0 && ww < p.W) ++cnt;
              }
            }
            dv = (float)(cnt > 0 ? cnt : 1);
          }
          const float inv = 1.f / dv;
#pragma unroll
          for (int j = 0; j < V; ++j) acc[j] += d[j] * inv;
        }
      }
    }
    PV<T, V>::st(dx + (((int64_t)n * p.H + h) * p.W + w) * p.C + c0, acc);
  }
}

template <typename T, int MODE>
static int pool_fwd_launch(const void* x, void* y, uint8_t* amax, const PoolP& p, hipStream_t st) {
  constexpr int VF = sizeof(T) == 2 ? 8 : 4;
  if (p.C % VF == 0) {
    const int64_t total = (int64_t)p.N * p.OH * p.OW * (p.C / VF);
    hipLaunchKernelGGL((pool_fwd_kernel<T, VF, MODE>), dim3(launch_grid(LS_POOLING, grid_for(total, 256))), dim3(256), 0, st, (const T*)x, (T*)y,
                       amax, p);
  } else {
    const int64_t total = (int64_t)p.N * p.OH * p.OW * p.C;
    hipLaunchKernelGGL((pool_fwd_kernel<T, 1, MODE>), dim3(launch_grid(LS_POOLING, grid_for(total, 256))), dim3(256), 0, st, (const T*)x, (T*)y,
                       amax, p);
  }
  DL4J_RET_LAST();
}
template <typename T, int MODE>
static int pool_bwd_launch(const void* dy, const uint8_t* amax, void* dx, const PoolP& p, hipStream_t st) {
  constexpr int VF = sizeof(T) == 2 ? 8 : 4;
  if (p.C % VF == 0) {
    const int64_t total = (int64_t)p.N * p.H * p.W * (p.C / VF);
    hipLaunchKernelGGL((pool_bwd_kernel<T, VF, MODE>), dim3(launch_grid(LS_POOLING, grid_for(total, 256))), dim3(256), 0, st, (const T*)dy, amax,
                       (T*)dx, p);
  } else {
    const int64_t total = (int64_t)p.N * p.H * p.W * p.C;
    hipLaunchKernelGGL((pool_bwd_kernel<T, 1, MODE>), dim3(launch_grid(LS_POOLING, grid_for(total, 256))), dim3(256), 0, st, (const T*)dy, amax,
                       (T*)dx, p);
  }
  DL4J_RET_LAST();
}

// geom = [N,H,W,C,OH,OW,kH,kW,sH,sW,pH,pW,dH,dW,avg_exclude_pad]
static PoolP mkp(const int* g) {
  PoolP p;
  p.N = g[0]; p.H = g[1]; p.W = g[2]; p.C = g[3]; p.OH = g[4]; p.OW = g[5];
  p.kH = g[6]; p.kW = g[7]; p.sH = g[8]; p.sW = g[9]; p.pH = g[10]; p.pW = g[11]; p.dH = g[12]; p.dW = g[13];
  p.avg_exclude_pad = g[14];
  return p;
}

DL4J_API int dl4j_pool2d_fwd(const void* x, void* y, uint8_t* amax, const int* geom, int mode, int dtype,
                             hipStream_t st) {
  const PoolP p = mkp(geom);
  if (p.kH * p.kW > 255) return -1;
  if (dtype == 1) return mode == 0 ? pool_fwd_launch<bf16_t, 0>(x, y, amax, p, st) : pool_fwd_launch<bf16_t, 1>(x, y, amax, p, st);
  return mode == 0 ? pool_fwd_launch<float, 0>(x, y, amax, p, st) : pool_fwd_launch<float, 1>(x, y, amax, p, st);
}

DL4J_API int dl4j_pool2d_bwd(const void* dy, const uint8_t* amax, void* dx, const int* geom, int mode, int dtype,
                             hipStream_t st) {
  const PoolP p = mkp(geom);
  if (dtype == 1) return mode == 0 ? pool_bwd_launch<bf16_t, 0>(dy, amax, dx, p, st) : pool_bwd_launch<bf16_t, 1>(dy, amax, dx, p, st);
  return mode == 0 ? pool_bwd_launch<float, 0>(dy, amax, dx, p, st) : pool_bwd_launch<float, 1>(dy, amax, dx, p, st);
}
