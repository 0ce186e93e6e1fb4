// Transposed-operand loaders for the LDS-DMA GEMM core: every op(A) / op(B) combination without a transpose pass.
//
//   C[M, N] = op(A) . op(B)        op(A): A [M, K] K-contiguous ("N")  or  A [K, M] M-contiguous ("T")
//                                  op(B): B [N, K] K-contiguous (the core's native NT form)  or  B [K, N] N-contiguous
//
// Reference: libnd4j/include/helpers/cuda_off/MmulHelper.cu:255-341 hands every transpose combination to cuBLAS; the core of
// gemm_core.h only had the NT form, so NN callers paid a transpose kernel (profiles/r2_op_microbench.md: matmul 0.45x) and the dense
// weight gradient dW = dz^T . x (TN on both sides) ran on the round-1 staging-register kernel.
//
// A K-major operand tile is staged by the same LDS-DMA pieces as a K-contiguous one, but its LDS image is [BK k-rows][BX elements]
// (row pitch BX * ESZ bytes), and the MFMA fragment - 8 consecutive k for one row / column per lane - is gathered by the transposing
// LDS read ds_read_b64_tr_b16: 16 lanes read a [4 k][16 x] block and lane x receives the 4 k values of column x. Two reads give the
// 8 k values of a 16x16x32 fragment. fp32 operands have no transposing read; their fragment (4 k per lane) is four ds_read_b32.
// Bank conflicts: one fragment instruction touches 16 k-rows x 32 bytes; rows are 128 / 256 bytes apart, so 16-byte chunk c of row r
// is stored at chunk c ^ tr_swz(r), which spreads the 16 rows over all bank groups (2 passes for 512 bytes = the LDS minimum). The
// swizzle is applied to the per-lane SOURCE address of the DMA (the LDS side of a DMA piece is lane-linear) and again on the read.
#pragma once
#include "gemm_epilogue.h"

namespace gemmcore {

template <class C, int BX> __device__ __forceinline__ int tr_swz(int r) {
  constexpr int PITCH = BX * C::ESZ;
  if constexpr (C::DT == DT_F32) return ((r >> 2) & 3) << 2;
  else if constexpr (PITCH == 256) return ((r & 3) | (((r >> 3) & 1) << 2)) << 1;
  else return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1;
}

// fragment of the 16 rows / columns starting at x0 for k-step ks of a K-major tile image (MF = 16 geometry: lane = (x = lane & 15,
// k group g = lane >> 4); bf16 / fp16: k = 32 ks + 8 g + 0..7; fp32: k = 16 ks + 4 g + 0..3 - the same k sets as the K-contiguous path)
template <class C, int BX>
__device__ __forceinline__ bf16x8_t tr_frag16(const char* tile, int x0, int ks, int lane) {
  constexpr int PITCH = BX * C::ESZ;
  static_assert(C::MF == 16, "transposed loaders use the 16x16 MFMA geometry");
  static_assert(PITCH == 128 || PITCH == 256 || (C::DT == DT_F32 && PITCH == 512), "tile extents 64 / 128");
  const int g = lane >> 4;
  if constexpr (C::DT == DT_F32) {
    const int x = x0 + (lane & 15);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = ks * 16 + 4 * g + j;
      r[j] = *reinterpret_cast<const float*>(tile + row * PITCH + ((((x >> 2) ^ tr_swz<C, BX>(row))) << 4) + (x & 3) * 4);
    }
    return __builtin_bit_cast(bf16x8_t, r);
  } else {
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const int x = x0 + 4 * pp;
    const int p0 = ks * 32 + 8 * g + q, p1 = p0 + 4;
    const int a0 = p0 * PITCH + (((x >> 3) ^ tr_swz<C, BX>(p0)) << 4) + (x & 7) * 2;
    const int a1 = p1 * PITCH + (((x >> 3) ^ tr_swz<C, BX>(p1)) << 4) + (x & 7) * 2;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr_t;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + a0));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + a1));
    s16x8 r;
    r[0] = v0[0]; r[1] = v0[1]; r[2] = v0[2]; r[3] = v0[3];
    r[4] = v1[0]; r[5] = v1[1]; r[6] = v1[2]; r[7] = v1[3];
    return __builtin_bit_cast(bf16x8_t, r);
  }
}

// per-lane staging constants of DMA piece i of a K-major operand tile: k-row inside the tile and first element of the LOGICAL chunk
// this lane fetches (its LDS position is lane-linear = the swizzled place of that chunk)
template <class C, int BX> __device__ __forceinline__ void tr_stage_pos(int i, int& row, int& x) {
  constexpr int NC = BX * C::ESZ / 16;           // chunks per row
  constexpr int RPP = 64 / NC;                   // rows per 1-KiB piece
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  row = (wid + i * C::NW) * RPP + lane / NC;
  x = ((lane % NC) ^ tr_swz<C, BX>(row)) * C::CH;
}

// Source of the main loop for any transpose combination. K range [kb, ke) (split-K), tile t covers k = kb + t * BK ...
template <class C, bool TA, bool TB> struct FlexSrc {
  const char* ap[C::PA];
  const char* bp[C::PB];
  int arow[TA ? C::PA : 1], brow[TB ? C::PB : 1];   // k-row of the piece (K-major operands)
  int kchunk;                                       // first k of this lane's chunk inside a k-tile (K-contiguous operands)
  int kb, ke;
  int64_t a_kstride, b_kstride;                     // bytes per k step of a K-major operand (ld * ESZ)
  __device__ __forceinline__ int tile(int t) const { return kb + t * C::BK; }
  __device__ __forceinline__ const void* a(int i, int k0) const {
    if constexpr (TA) {
      const int k = k0 + arow[i];
      return (k < ke && ap[i]) ? (const void*)(ap[i] + (int64_t)k * a_kstride) : (const void*)g_zero_page;
    } else {
      const int k = k0 + kchunk;
      return k < ke ? (const void*)(ap[i] + (int64_t)k * C::ESZ) : (const void*)g_zero_page;
    }
  }
  __device__ __forceinline__ const void* b(int i, int k0) const {
    if constexpr (TB) {
      const int k = k0 + brow[i];
      return (k < ke && bp[i]) ? (const void*)(bp[i] + (int64_t)k * b_kstride) : (const void*)g_zero_page;
    } else {
      const int k = k0 + kchunk;
      return k < ke ? (const void*)(bp[i] + (int64_t)k * C::ESZ) : (const void*)g_zero_page;
    }
  }
};

// Main loop with selectable fragment paths (same pipeline as gemm_core.h mainloop: STAGES buffers, counted waits, raw barriers).
template <class C, class Src, bool TA, bool TB>
__device__ __forceinline__ void mainloop_flex(char* smem, const Src& src, int KT, typename C::acc_t (&acc)[C::TM][C::TN]) {
  static_assert(C::MF == 16, "16x16 MFMA geometry");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / C::WN, wn = wid - wm * C::WN;
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int e = 0; e < C::ACC; ++e) acc[i][j][e] = 0.f;

  auto stage = [&](int t) {
    char* sb = smem + (t % C::STAGES) * C::STAGE_BYTES;
    const auto ts = src.tile(t);
#pragma unroll
    for (int i = 0; i < C::PA; ++i) glds16(src.a(i, ts), sb + (wid + i * C::NW) * 1024);
#pragma unroll
    for (int i = 0; i < C::PB; ++i) glds16(src.b(i, ts), sb + C::A_BYTES + (wid + i * C::NW) * 1024);
  };
  constexpr int KS = 2;
  const int lr = lane & 15, lq = lane >> 4, sw = (lr >> 1) & 7;
  int foff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) foff[ks] = lr * 128 + (((ks * 4 + lq) ^ sw) << 4);
  const int a_row0 = wm * (C::BM / C::WM), b_row0 = wn * (C::BN / C::WN);
  constexpr int BLK = 16 * 128;

  if (C::STAGES == 1) stage(0);
#pragma unroll
  for (int s = 0; s < C::STAGES - 1; ++s)
    if (s < KT) stage(s);
  for (int t = 0; t < KT; ++t) {
    const int ahead = KT - 1 - t;
    if (C::STAGES >= 4 && ahead >= 2) wait_vmcnt<2 * C::LPT>();
    else if (C::STAGES >= 3 && ahead >= 1) wait_vmcnt<C::LPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (C::STAGES > 1 && t + C::STAGES - 1 < KT) stage(t + C::STAGES - 1);
    const char* cs = smem + (t % C::STAGES) * C::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8_t af[C::TM], bfr[C::TN];
#pragma unroll
      for (int i = 0; i < C::TM; ++i) {
        if constexpr (TA) af[i] = tr_frag16<C, C::BM>(cs, a_row0 + i * 16, ks, lane);
        else af[i] = *reinterpret_cast<const bf16x8_t*>(cs + a_row0 * 128 + i * BLK + foff[ks]);
      }
#pragma unroll
      for (int j = 0; j < C::TN; ++j) {
        if constexpr (TB) bfr[j] = tr_frag16<C, C::BN>(cs + C::A_BYTES, b_row0 + j * 16, ks, lane);
        else bfr[j] = *reinterpret_cast<const bf16x8_t*>(cs + C::A_BYTES + b_row0 * 128 + j * BLK + foff[ks]);
      }
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j) mma<C>(af[i], bfr[j], acc[i][j]);
    }
    if (C::STAGES == 1 && t + 1 < KT) { __builtin_amdgcn_s_barrier(); stage(t + 1); }
  }
  __builtin_amdgcn_s_barrier();
}

struct GemmTP {
  const void* A; const void* B;
  EpiP epi;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int64_t sA, sB, sC;
  int mtiles, ntiles, gm;
  int kper;                   // k extent of one split (multiple of BK); gridDim.z splits
};

template <class C, int OT, bool TA, bool TB>
__global__ void __launch_bounds__(C::NT) gemm_flex_kernel(const GemmTP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int mt, nt;
  tile_coords(lid, p.mtiles, p.ntiles, p.gm, mt, nt);
  const int m0 = mt * C::BM, n0 = nt * C::BN;
  const char* A = (const char*)p.A + (int64_t)blockIdx.y * p.sA * C::ESZ;
  const char* B = (const char*)p.B + (int64_t)blockIdx.y * p.sB * C::ESZ;

  FlexSrc<C, TA, TB> src;
  src.kb = blockIdx.z * p.kper;
  src.ke = min(p.K, src.kb + p.kper);
  src.kchunk = stage_chunk(stage_row<C>(0)) * C::CH;
  src.a_kstride = p.lda * C::ESZ;
  src.b_kstride = p.ldb * C::ESZ;
#pragma unroll
  for (int i = 0; i < C::PA; ++i) {
    if constexpr (TA) {
      int row, x;
      tr_stage_pos<C, C::BM>(i, row, x);
      src.arow[i] = row;
      src.ap[i] = (m0 + x < p.M) ? A + (int64_t)(m0 + x) * C::ESZ : nullptr;        // M % CH == 0: a chunk is inside or outside as a whole
    } else {
      int m = m0 + stage_row<C>(i);
      if (m > p.M - 1) m = p.M - 1;
      src.ap[i] = A + (int64_t)m * p.lda * C::ESZ;
    }
  }
#pragma unroll
  for (int i = 0; i < C::PB; ++i) {
    if constexpr (TB) {
      int row, x;
      tr_stage_pos<C, C::BN>(i, row, x);
      src.brow[i] = row;
      src.bp[i] = (n0 + x < p.N) ? B + (int64_t)(n0 + x) * C::ESZ : nullptr;
    } else {
      int n = n0 + stage_row<C>(i);
      if (n > p.N - 1) n = p.N - 1;
      src.bp[i] = B + (int64_t)n * p.ldb * C::ESZ;
    }
  }
  typename C::acc_t acc[C::TM][C::TN];
  const int KT = (src.ke - src.kb + C::BK - 1) / C::BK;
  mainloop_flex<C, FlexSrc<C, TA, TB>, TA, TB>(smem, src, KT, acc);

  EpiP e = p.epi;
  const int64_t boff = (int64_t)blockIdx.y * p.sC;
  e.out = (char*)e.out + boff * Out<OT>::SZ;
  if (e.residual) e.residual = (const char*)e.residual + boff * Out<OT>::SZ;
  if (blockIdx.z > 0) { e.bias = nullptr; e.residual = nullptr; }     // split-K: the additive terms enter once
  const int M = p.M;
  const int64_t ldc = p.ldc;
  epilogue<C, OT>(smem, acc, mt, n0, p.N, M - m0, e, [=](int r) -> int64_t {
    const int m = m0 + r;
    return m < M ? (int64_t)m * ldc + n0 : (int64_t)-1;
  });
}

template <class C, int OT, bool TA, bool TB>
static int launch_gemm_flex(GemmTP& p, int batch, int splits, hipStream_t st) {
  p.mtiles = (p.M + C::BM - 1) / C::BM;
  p.ntiles = (p.N + C::BN - 1) / C::BN;
  if (p.gm <= 0) p.gm = 8;
  const int ktiles = (p.K + C::BK - 1) / C::BK;
  if (splits < 1) splits = 1;
  if (splits > ktiles) splits = ktiles;
  p.kper = ((ktiles + splits - 1) / splits) * C::BK;
  splits = (p.K + p.kper - 1) / p.kper;
  constexpr int LDS = C::lds_bytes(Out<OT>::SZ);
  static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  static DeviceOnceLds lds_attr;
  lds_attr.ensure(reinterpret_cast<const void*>(&gemm_flex_kernel<C, OT, TA, TB>), LDS);
  hipLaunchKernelGGL((gemm_flex_kernel<C, OT, TA, TB>), dim3(p.mtiles * p.ntiles, batch, splits), dim3(C::NT), LDS, st, p);
  return (int)hipGetLastError();
}

}  // namespace gemmcore
