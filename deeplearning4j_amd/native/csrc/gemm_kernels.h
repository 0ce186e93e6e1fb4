// Plain NT GEMM kernels on the LDS-DMA core (gemm_core.h) with fused epilogues:
//   C = act(A . B^T + bias) [+ residual]      act in {identity, relu, gelu(tanh), gelu(erf), tanh, sigmoid}
// bf16 in, fp32 accumulate, bf16 (or fp32) out. Batched form: blockIdx.y = batch, element strides per operand.
#pragma once
#include "gemm_core.h"

namespace gemmcore {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU_TANH = 2, ACT_GELU_ERF = 3, ACT_TANH = 4, ACT_SIGMOID = 5 };

struct GemmP {
  const bf16_t* A; const bf16_t* B; void* C;
  const float* bias;          // [N] or null
  const bf16_t* residual;     // [M, ldc] added after the activation, or null
  int M, N, K;
  int64_t lda, ldb, ldc;
  int64_t sA, sB, sC;         // batch strides (elements)
  int mtiles, ntiles, gm;
  int act;
  int out_f32;                // 1: C is fp32
  float alpha;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_GELU_TANH: { const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v); return 0.5f * v * (1.f + tanhf(u)); }
    case ACT_GELU_ERF: return 0.5f * v * (1.f + erff(v * 0.7071067811865476f));
    case ACT_TANH: return tanhf(v);
    case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

template <class C> struct PlainSrc {
  const bf16_t* ap[C::PA];
  const bf16_t* bp[C::PB];
  int kchunk8;     // first k of this lane's 16-byte chunk inside a k-tile
  int K;
  __device__ __forceinline__ const void* a(int i, int t) const {
    const int k = t * 64 + kchunk8;
    return k < K ? (const void*)(ap[i] + k) : (const void*)g_zero_page;
  }
  __device__ __forceinline__ const void* b(int i, int t) const {
    const int k = t * 64 + kchunk8;
    return k < K ? (const void*)(bp[i] + k) : (const void*)g_zero_page;
  }
};

template <class C, bool PP = false>
__global__ void __launch_bounds__(C::NT) gemm_nt_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int mt, nt;
  tile_coords(lid, p.mtiles, p.ntiles, p.gm, mt, nt);
  const int m0 = mt * C::BM, n0 = nt * C::BN;
  const bf16_t* A = p.A + (int64_t)blockIdx.y * p.sA;
  const bf16_t* B = p.B + (int64_t)blockIdx.y * p.sB;

  PlainSrc<C> src;
  src.K = p.K;
  {
    const int r0 = stage_row<C>(0);
    src.kchunk8 = stage_chunk(r0) * 8;
  }
#pragma unroll
  for (int i = 0; i < C::PA; ++i) {
    int m = m0 + stage_row<C>(i);
    if (m > p.M - 1) m = p.M - 1;                      // tail rows read a valid row, their results are never stored
    src.ap[i] = A + (int64_t)m * p.lda;
  }
#pragma unroll
  for (int i = 0; i < C::PB; ++i) {
    int n = n0 + stage_row<C>(i);
    if (n > p.N - 1) n = p.N - 1;
    src.bp[i] = B + (int64_t)n * p.ldb;
  }
  typename C::acc_t acc[C::TM][C::TN];
  if constexpr (PP) mainloop_pingpong<C>(smem, src, (p.K + 63) >> 6, acc);
  else mainloop<C>(smem, src, (p.K + 63) >> 6, acc);

  // ---- epilogue: alpha, bias, activation, residual, rounding; LDS transpose so every row leaves as whole 16-byte chunks
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / C::WN, wn = wid - wm * C::WN;
  const int lc = C::acc_col(lane);
  if (p.out_f32) {
    float* Cf = (float*)p.C + (int64_t)blockIdx.y * p.sC;
#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
      const int n = n0 + wn * (C::BN / C::WN) + j * C::MF + lc;
      const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int e = 0; e < C::ACC; ++e) {
          const int m = m0 + wm * (C::BM / C::WM) + i * C::MF + C::acc_row(e, lane);
          if (m < p.M && n < p.N) {
            float v = apply_act(acc[i][j][e] * p.alpha + bv, p.act);
            if (p.residual) v += bf2f(p.residual[(int64_t)blockIdx.y * p.sC + (int64_t)m * p.ldc + n]);
            Cf[(int64_t)m * p.ldc + n] = v;
          }
        }
    }
    return;
  }
  char* sC = smem;
#pragma unroll
  for (int j = 0; j < C::TN; ++j) {
    const int col = wn * (C::BN / C::WN) + j * C::MF + lc;
    const float bv = (p.bias && n0 + col < p.N) ? p.bias[n0 + col] : 0.f;
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int e = 0; e < C::ACC; ++e) {
        const int row = wm * (C::BM / C::WM) + i * C::MF + C::acc_row(e, lane);
        *reinterpret_cast<bf16_t*>(sC + row * C::PITCH_C + col * 2) = f2bf(apply_act(acc[i][j][e] * p.alpha + bv, p.act));
      }
  }
  __syncthreads();
  bf16_t* Cb = (bf16_t*)p.C + (int64_t)blockIdx.y * p.sC;
  constexpr int CPR = C::BN / 8;
  const bool vec_ok = (p.ldc & 7) == 0;
  for (int idx = tid; idx < C::BM * CPR; idx += C::NT) {
    const int row = idx / CPR, ch = idx - row * CPR;
    const int m = m0 + row, n = n0 + ch * 8;
    if (m >= p.M || n >= p.N) continue;
    s16x8 v = *reinterpret_cast<const s16x8*>(sC + row * C::PITCH_C + ch * 16);
    const int64_t o = (int64_t)m * p.ldc + n;
    if (vec_ok && n + 8 <= p.N) {
      if (p.residual) {
        const s16x8 r = *reinterpret_cast<const s16x8*>(p.residual + (int64_t)blockIdx.y * p.sC + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(bf2f((bf16_t)v[j]) + bf2f((bf16_t)r[j]));
      }
      *reinterpret_cast<s16x8*>(Cb + o) = v;
    } else {
      for (int j = 0; j < 8 && n + j < p.N; ++j) {
        float f = bf2f((bf16_t)v[j]);
        if (p.residual) f += bf2f(p.residual[(int64_t)blockIdx.y * p.sC + o + j]);
        Cb[o + j] = f2bf(f);
      }
    }
  }
}

template <class C, bool PP = false>
static int launch_gemm(GemmP& p, int batch, hipStream_t st) {
  p.mtiles = (p.M + C::BM - 1) / C::BM;
  p.ntiles = (p.N + C::BN - 1) / C::BN;
  if (p.gm <= 0) p.gm = 4;
  static DeviceOnceLds lds_attr;
  lds_attr.ensure(reinterpret_cast<const void*>(&gemm_nt_kernel<C, PP>), C::LDS_BYTES);
  hipLaunchKernelGGL((gemm_nt_kernel<C, PP>), dim3(p.mtiles * p.ntiles, batch), dim3(C::NT), C::LDS_BYTES, st, p);
  return (int)hipGetLastError();
}

}  // namespace gemmcore
