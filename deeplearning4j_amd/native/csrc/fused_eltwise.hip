// Fused element-wise companions of the GEMM kernels for transformer-style graphs (xw_plus_b -> activation -> xw_plus_b ...):
//
//   act_fwd          y = act(x)                                    one 16-byte-vector pass
//   act_bwd_colsum   dz = dy * act'(z)  and  db[n] += sum_m dz     the activation backward and the bias gradient of the producing
//                                                                  linear op in ONE pass over dy (the reference runs the derivative
//                                                                  transform, a pairwise multiply and a reduce_sum along dimension 0:
//                                                                  three passes, ops/declarable/generic/nn/xw_plus_b.cpp _bp)
//   (z == null: plain column sum of dy, the bias gradient of a linear op with no activation)
//
// Memory-bound: every thread owns 8 consecutive columns (one 16-byte bf16/fp16 chunk, two for fp32) and walks rows; column sums stay
// in registers, are combined across the block's row lanes through LDS and leave as one fp32 atomic per column per block.
#include "common.h"

namespace {

enum { A_NONE = 0, A_RELU = 1, A_GELU_TANH = 2, A_GELU_ERF = 3, A_TANH = 4, A_SIGMOID = 5, A_GELU_SIGMOID = 6, A_SWISH = 7 };

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

__device__ __forceinline__ float act_f(float v, int act) {
  switch (act) {
    case A_RELU: return fmaxf(v, 0.f);
    case A_GELU_TANH: { const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v); return 0.5f * v * (1.f + tanhf(u)); }
    case A_GELU_ERF: return 0.5f * v * (1.f + erff(v * 0.7071067811865476f));
    case A_TANH: return tanhf(v);
    case A_SIGMOID: return sigm(v);
    case A_GELU_SIGMOID: return v * sigm(1.702f * v);                  // libnd4j GELU (templatemath.h: x * sigmoid(1.702 x))
    case A_SWISH: return v * sigm(v);
    default: return v;
  }
}

__device__ __forceinline__ float act_d(float v, int act) {
  switch (act) {
    case A_RELU: return v > 0.f ? 1.f : 0.f;
    case A_GELU_TANH: {
      const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v), t = tanhf(u);
      return 0.5f * (1.f + t) + 0.5f * v * (1.f - t * t) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * v * v);
    }
    case A_GELU_ERF: return 0.5f * (1.f + erff(v * 0.7071067811865476f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
    case A_TANH: { const float t = tanhf(v); return 1.f - t * t; }
    case A_SIGMOID: { const float s = sigm(v); return s * (1.f - s); }
    case A_GELU_SIGMOID: { const float s = sigm(1.702f * v); return s + 1.702f * v * s * (1.f - s); }
    case A_SWISH: { const float s = sigm(v); return s + v * s * (1.f - s); }
    default: return 1.f;
  }
}

// 8 consecutive elements <-> 8 floats
template <int DT> struct V8;
template <> struct V8<0> {                                               // bf16
  typedef bf16_t T;
  static __device__ __forceinline__ void ld(const T* p, float (&f)[8]) {
    const s16x8 v = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = bf2f((bf16_t)v[i]);
  }
  static __device__ __forceinline__ void st(T* p, const float (&f)[8]) {
    s16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (short)f2bf(f[i]);
    *reinterpret_cast<s16x8*>(p) = v;
  }
};
template <> struct V8<1> {                                               // fp16
  typedef uint16_t T;
  static __device__ __forceinline__ void ld(const T* p, float (&f)[8]) {
    const s16x8 v = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)__builtin_bit_cast(_Float16, (uint16_t)v[i]);
  }
  static __device__ __forceinline__ void st(T* p, const float (&f)[8]) {
    s16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const _Float16 h = (_Float16)f[i]; v[i] = (short)__builtin_bit_cast(uint16_t, h); }
    *reinterpret_cast<s16x8*>(p) = v;
  }
};
template <> struct V8<2> {                                               // fp32
  typedef float T;
  static __device__ __forceinline__ void ld(const T* p, float (&f)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void st(T* p, const float (&f)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = f[i]; b[i] = f[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
};

template <int DT>
__global__ void __launch_bounds__(256) act_fwd_kernel(const typename V8<DT>::T* __restrict__ x, typename V8<DT>::T* __restrict__ y,
                                                      int64_t n8, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float f[8];
    V8<DT>::ld(x + i * 8, f);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = act_f(f[k], act);
    V8<DT>::st(y + i * 8, f);
  }
}

// grid (ceil(N / 256), row blocks); 256 threads = 32 column chunks x 8 row lanes. The pass is short (a [4096, 768] bf16 operand is 6 MB),
// so what matters is memory-level parallelism, not the instruction count: a thread issues the loads of FOUR rows before it touches any of
// them (the one-row-per-trip loop of round 2 paid one memory latency per row: 12.8 us for 16 rows per thread at 4096 x 768, against
// ~3 us of traffic), and the grid is sized from N in 256-column slices so narrow operands do not leave half a block idle.
template <int DT>
__global__ void __launch_bounds__(256) act_bwd_colsum_kernel(const typename V8<DT>::T* __restrict__ z,
                                                             const typename V8<DT>::T* __restrict__ dy,
                                                             typename V8<DT>::T* __restrict__ dz, float* __restrict__ db, int64_t M, int N,
                                                             int64_t rows_per_block, int act) {
  __shared__ float red[8][256];
  const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + cc * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < N) {
    int64_t r = r0 + rl;
    for (; r + 24 < r1; r += 32) {
      float g0[8], g1[8], g2[8], g3[8];
      V8<DT>::ld(dy + r * N + col, g0);
      V8<DT>::ld(dy + (r + 8) * N + col, g1);
      V8<DT>::ld(dy + (r + 16) * N + col, g2);
      V8<DT>::ld(dy + (r + 24) * N + col, g3);
      if (z != nullptr) {
        float v0[8], v1[8], v2[8], v3[8];
        V8<DT>::ld(z + r * N + col, v0);
        V8<DT>::ld(z + (r + 8) * N + col, v1);
        V8<DT>::ld(z + (r + 16) * N + col, v2);
        V8<DT>::ld(z + (r + 24) * N + col, v3);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          g0[k] *= act_d(v0[k], act); g1[k] *= act_d(v1[k], act); g2[k] *= act_d(v2[k], act); g3[k] *= act_d(v3[k], act);
        }
      }
      if (dz != nullptr) {
        V8<DT>::st(dz + r * N + col, g0);
        V8<DT>::st(dz + (r + 8) * N + col, g1);
        V8<DT>::st(dz + (r + 16) * N + col, g2);
        V8<DT>::st(dz + (r + 24) * N + col, g3);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += (g0[k] + g1[k]) + (g2[k] + g3[k]);
    }
    for (; r < r1; r += 8) {
      float g[8];
      V8<DT>::ld(dy + r * N + col, g);
      if (z != nullptr) {
        float v[8];
        V8<DT>::ld(z + r * N + col, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] *= act_d(v[k], act);
      }
      if (dz != nullptr) V8<DT>::st(dz + r * N + col, g);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += g[k];
    }
  }
  if (db == nullptr) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl][cc * 8 + k] = s[k];
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][threadIdx.x];
    atomicAdd(db + n, t);
  }
}

}  // namespace

#define DISPATCH_DT(...)                                   \
  if (dtype == 0) { constexpr int DT = 0; __VA_ARGS__ }    \
  else if (dtype == 1) { constexpr int DT = 1; __VA_ARGS__ } \
  else if (dtype == 2) { constexpr int DT = 2; __VA_ARGS__ } \
  else return -2;

// n must be a multiple of 8, x / y 16-byte aligned. dtype: 0 bf16, 1 fp16, 2 fp32 (the GEMM entry points' codes).
DL4J_API int dl4j_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  if (n % 8) return -3;
  const int64_t n8 = n / 8;
  DISPATCH_DT(typedef typename V8<DT>::T T;
              hipLaunchKernelGGL(act_fwd_kernel<DT>, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const T*)x, (T*)y, n8, act);)
  DL4J_RET_LAST();
}

// dz[M, N] = dy * act'(z) (z null: dz = dy; dz null: not written); db[N] (fp32, pre-initialised: the sums are ADDED) += column sums.
// N must be a multiple of 8, all operands row-contiguous with row stride N and 16-byte aligned.
DL4J_API int dl4j_act_bwd_colsum(const void* z, const void* dy, void* dz, float* db, int64_t M, int N, int act, int dtype,
                                 hipStream_t st) {
  if (M <= 0 || N <= 0) return 0;
  if (N % 8) return -3;
  const int gx = (N + 255) / 256;
  // 32 rows per block (4 per thread, all in flight at once) until that makes more than ~8 blocks per CU; then longer row ranges
  int64_t gy = (M + 31) / 32;
  const int64_t want = (2048 + gx - 1) / gx;
  if (gy > want) gy = want;
  const int64_t cap = (DL4J_MAX_GRID + gx - 1) / gx;
  if (gy > cap) gy = cap;
  const int64_t rpb = (((M + gy - 1) / gy) + 31) / 32 * 32;
  gy = (M + rpb - 1) / rpb;
  DISPATCH_DT(typedef typename V8<DT>::T T;
              hipLaunchKernelGGL(act_bwd_colsum_kernel<DT>, dim3(gx, (unsigned)gy), dim3(256), 0, st, (const T*)z, (const T*)dy, (T*)dz,
                                 db, M, N, rpb, act);)
  DL4J_RET_LAST();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Transposed compute-precision replicas of the 2-D weights of a flat fp32 variable vector, ALL in one launch: entry e copies
// src[src_off .. + rows*cols] viewed as [rows][cols] to dst[dst_off ..] as [cols][rows] (the "NT" B operand of y = x . W for a
// W declared [K, N]). One workgroup per 64x64 tile through a padded LDS tile, so reads and writes are both coalesced.
namespace {
struct __attribute__((packed, aligned(8))) TrSeg {
  int64_t src_off, dst_off;
  int32_t rows, cols, tiles_c, pad;
};
static_assert(sizeof(TrSeg) == 32, "TrSeg layout is part of the ctypes ABI");

template <int DT>
__global__ void __launch_bounds__(256) transpose_cast_table_kernel(const float* __restrict__ src, typename V8<DT>::T* __restrict__ dst,
                                                                   const TrSeg* __restrict__ segs, const int32_t* __restrict__ blk_seg,
                                                                   const int32_t* __restrict__ blk_tile) {
  __shared__ float tile[64][65];
  const TrSeg s = segs[blk_seg[blockIdx.x]];
  const int t = blk_tile[blockIdx.x];
  const int tr = t / s.tiles_c, tc = t - tr * s.tiles_c;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  const float* sp = src + s.src_off;
  typename V8<DT>::T* dp = dst + s.dst_off;
#pragma unroll
  for (int r = r4; r < 64; r += 4) {
    const int gr = tr * 64 + r, gc = tc * 64 + c;
    tile[r][c] = (gr < s.rows && gc < s.cols) ? sp[(int64_t)gr * s.cols + gc] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = r4; r < 64; r += 4) {
    const int gc = tc * 64 + r, gr = tr * 64 + c;           // destination row = source column
    if (gc < s.cols && gr < s.rows) {
      const float v = tile[c][r];
      if (DT == 0) dp[(int64_t)gc * s.rows + gr] = (typename V8<DT>::T)f2bf(v);
      else if (DT == 1) { const _Float16 h = (_Float16)v; dp[(int64_t)gc * s.rows + gr] = (typename V8<DT>::T)__builtin_bit_cast(uint16_t, h); }
      else dp[(int64_t)gc * s.rows + gr] = (typename V8<DT>::T)v;
    }
  }
}
}  // namespace

DL4J_API int dl4j_transpose_cast_table(const float* src, void* dst, const void* segs, const int32_t* blk_seg, const int32_t* blk_tile,
                                       int num_blocks, int dtype, hipStream_t st) {
  if (num_blocks <= 0) return 0;
  DISPATCH_DT(typedef typename V8<DT>::T T;
              hipLaunchKernelGGL(transpose_cast_table_kernel<DT>, dim3(num_blocks), dim3(256), 0, st, src, (T*)dst, (const TrSeg*)segs,
                                 blk_seg, blk_tile);)
  DL4J_RET_LAST();
}
