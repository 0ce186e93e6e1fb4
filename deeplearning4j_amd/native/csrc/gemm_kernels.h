// Plain NT GEMM kernels on the LDS-DMA core (gemm_core.h) with the shared fused epilogue (gemm_epilogue.h):
//   C = act(alpha * A . B^T + bias) [+ residual]   A [M, K], B [N, K] K-contiguous; bf16 / fp16 / fp32 operands, fp32 accumulation.
// Batched form: blockIdx.y = batch with element strides per operand (attention scores, batched_gemm).
#pragma once
#include "gemm_epilogue.h"

namespace gemmcore {

struct GemmP {
  const void* A; const void* B;
  EpiP epi;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int64_t sA, sB, sC;         // batch strides (elements)
  int mtiles, ntiles, gm;
};

template <class C> struct PlainSrc {
  const char* ap[C::PA];
  const char* bp[C::PB];
  int kchunk;      // first k of this lane's 16-byte chunk inside a k-tile
  int K;
  __device__ __forceinline__ int tile(int t) const { return t * C::BK; }
  __device__ __forceinline__ const void* a(int i, int k0) const {
    const int k = k0 + kchunk;
    return k < K ? (const void*)(ap[i] + (int64_t)k * C::ESZ) : (const void*)g_zero_page;
  }
  __device__ __forceinline__ const void* b(int i, int k0) const {
    const int k = k0 + kchunk;
    return k < K ? (const void*)(bp[i] + (int64_t)k * C::ESZ) : (const void*)g_zero_page;
  }
};

// operand rows of the deep ping-pong loop (mainloop_pingpong_deep): index h*2+s = half h, piece s of this wave
template <class C> struct PlainSrcPP {
  const char* ap[4];
  const char* bp[4];
  int kch[2];      // first k of this lane's 16-byte chunk inside a k-tile, per piece parity s
  int K;
  __device__ __forceinline__ const void* a_half(int h, int s, int t) const {
    const int k = t * C::BK + kch[s];
    return k < K ? (const void*)(ap[h * 2 + s] + (int64_t)k * C::ESZ) : (const void*)g_zero_page;
  }
  __device__ __forceinline__ const void* b_half(int h, int s, int t) const {
    const int k = t * C::BK + kch[s];
    return k < K ? (const void*)(bp[h * 2 + s] + (int64_t)k * C::ESZ) : (const void*)g_zero_page;
  }
};

// PP: 0 = barrier-synchronous loop, 1 = ping-pong wave groups, 2 = ping-pong with half-tile staging and counted waits
template <class C, int OT, int PP = 0>
__global__ void __launch_bounds__(C::NT) gemm_nt_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int mt, nt;
  tile_coords(lid, p.mtiles, p.ntiles, p.gm, mt, nt);
  const int m0 = mt * C::BM, n0 = nt * C::BN;
  const char* A = (const char*)p.A + (int64_t)blockIdx.y * p.sA * C::ESZ;
  const char* B = (const char*)p.B + (int64_t)blockIdx.y * p.sB * C::ESZ;

  typename C::acc_t acc[C::TM][C::TN];
  const int KT = (p.K + C::BK - 1) / C::BK;
  if constexpr (PP == 2) {
    PlainSrcPP<C> src;
    src.K = p.K;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 0; s < 2; ++s) src.kch[s] = ((lane & 7) ^ (s * 4 + (lane >> 4))) * C::CH;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        int m = m0 + pp_piece_a<C>(wid, h, s) * 8 + (lane >> 3);
        if (m > p.M - 1) m = p.M - 1;
        src.ap[h * 2 + s] = A + (int64_t)m * p.lda * C::ESZ;
        int n = n0 + pp_piece_b<C>(wid, h, s) * 8 + (lane >> 3);
        if (n > p.N - 1) n = p.N - 1;
        src.bp[h * 2 + s] = B + (int64_t)n * p.ldb * C::ESZ;
      }
    mainloop_pingpong_deep<C>(smem, src, KT, acc);
  } else {
  PlainSrc<C> src;
  src.K = p.K;
  src.kchunk = stage_chunk(stage_row<C>(0)) * C::CH;
#pragma unroll
  for (int i = 0; i < C::PA; ++i) {
    int m = m0 + stage_row<C>(i);
    if (m > p.M - 1) m = p.M - 1;                      // tail rows read a valid row, their results are never stored
    src.ap[i] = A + (int64_t)m * p.lda * C::ESZ;
  }
#pragma unroll
  for (int i = 0; i < C::PB; ++i) {
    int n = n0 + stage_row<C>(i);
    if (n > p.N - 1) n = p.N - 1;
    src.bp[i] = B + (int64_t)n * p.ldb * C::ESZ;
  }
  if constexpr (PP == 1) mainloop_pingpong<C>(smem, src, KT, acc);
  else mainloop<C>(smem, src, KT, acc);
  }

  EpiP e = p.epi;
  const int64_t boff = (int64_t)blockIdx.y * p.sC;
  e.out = (char*)e.out + boff * Out<OT>::SZ;
  if (e.residual) e.residual = (const char*)e.residual + boff * Out<OT>::SZ;
  const int M = p.M;
  const int64_t ldc = p.ldc;
  epilogue<C, OT>(smem, acc, mt, n0, p.N, M - m0, e, [=](int r) -> int64_t {
    const int m = m0 + r;
    return m < M ? (int64_t)m * ldc + n0 : (int64_t)-1;
  });
}

template <class C, int OT, int PP = 0>
static int launch_gemm(GemmP& p, int batch, hipStream_t st) {
  p.mtiles = (p.M + C::BM - 1) / C::BM;
  p.ntiles = (p.N + C::BN - 1) / C::BN;
  if (p.gm <= 0) p.gm = 8;
  constexpr int LDS = C::lds_bytes(Out<OT>::SZ);
  static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  static DeviceOnceLds lds_attr;
  lds_attr.ensure(reinterpret_cast<const void*>(&gemm_nt_kernel<C, OT, PP>), LDS);
  hipLaunchKernelGGL((gemm_nt_kernel<C, OT, PP>), dim3(p.mtiles * p.ntiles, batch), dim3(C::NT), LDS, st, p);
  return (int)hipGetLastError();
}

}  // namespace gemmcore
