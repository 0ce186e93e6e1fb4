// LDS-DMA pipelined NT GEMM core for gfx950 (MI355X / CDNA4).
//
//   C[M, N] = A[M, K] . B[N, K]^T   bf16 operands (both K-contiguous), fp32 accumulation on the matrix cores, fused epilogue.
//
// Why a second kernel next to conv_igemm.hip: round 1's implicit-GEMM loop staged tiles global -> VGPR -> ds_write -> barrier and
// reached ~50 % of hipBLASLt (profiles/r1_gemm_microbench.md). This core follows the CDNA4 playbook instead:
//   * operand tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging VGPRs, no
//     ds_write pass); the LDS image is lane-linear, so the XOR swizzle that makes ds_read_b128 conflict free is applied to the
//     per-lane SOURCE address and again on the read (both or neither);
//   * STAGES LDS buffers, loads for tile t+STAGES-1 issued right after the barrier that opens tile t and left in flight across
//     barriers behind a COUNTED s_waitcnt vmcnt(N) + raw s_barrier (never vmcnt(0) in the steady state when STAGES > 2);
//   * every row of the tile has its own 64-bit source pointer, so the same loop serves plain GEMM (row m -> A + m*lda) and
//     implicit-GEMM convolution (row m -> a pixel of the NHWC input, padding -> a zero page), and tensors are not limited to 2^31
//     elements;
//   * 32x32x16 bf16 MFMA, wave tile (BM/WM)x(BN/WN), XCD-aware workgroup -> tile map.
// Reference for the dispatch this replaces: libnd4j/include/helpers/cuda_off/MmulHelper.cu:255-341 (cuBLAS per dtype, naive
// 1-thread-per-element kernel for bf16) and helpers/cuda/batched_gemm.cu:145-193.
#pragma once
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

namespace gemmcore {

typedef __attribute__((address_space(3))) void* lds_vptr_t;
typedef const __attribute__((address_space(1))) void* glb_vptr_t;

// one LDS-DMA piece: 64 lanes x 16 bytes, lane l lands at lds_wave_base + 16 * l
__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_vptr_t)src, (lds_vptr_t)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
}

// 256 bytes of zeros every out-of-range 16-byte piece is redirected to (conv padding, K tail)
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

template <int BM_, int BN_, int WM_, int WN_, int STAGES_>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, STAGES = STAGES_;
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;      // 32x32 MFMA blocks per wave
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;    // BK = 64 bf16 = 128-byte rows
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;        // 1-KiB pieces per wave per k-tile
  static constexpr int LPT = PA + PB;                             // LDS-DMA instructions per wave per k-tile
  static constexpr int PITCH_C = BN * 2 + 16;
  static constexpr int EPI_BYTES = BM * PITCH_C;
  static constexpr int LDS_BYTES = (STAGES * STAGE_BYTES > EPI_BYTES ? STAGES * STAGE_BYTES : EPI_BYTES);
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave stages whole 8-row pieces");
  static_assert(NW % 2 == 0 && STAGES >= 2 && STAGES <= 4, "even wave count (lane-constant swizzle), 2..4 stages");
  static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "wave tile is a multiple of the 32x32 MFMA block");
};

// Main loop shared by GEMM and implicit-GEMM convolution. `Src` provides, per k-tile t, the source address of this lane's 16-byte
// chunk for staged row i of A / B:  const void* a(int i, int t),  const void* b(int i, int t).
template <class C, class Src>
__device__ __forceinline__ void mainloop(char* smem, const Src& src, int KT, f32x16 (&acc)[C::TM][C::TN]) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / C::WN, wn = wid - wm * C::WN;
  const int l31 = lane & 31, lh = lane >> 5;

#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto stage = [&](int t) {
    char* sb = smem + (t % C::STAGES) * C::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < C::PA; ++i) glds16(src.a(i, t), sb + (wid + i * C::NW) * 1024);
#pragma unroll
    for (int i = 0; i < C::PB; ++i) glds16(src.b(i, t), sb + C::A_BYTES + (wid + i * C::NW) * 1024);
  };

  // per-lane fragment read offsets: row l31 of a 32-row block, 16-byte chunk q = 2 ks + lh, swizzled by (row >> 1) & 7
  const int sw = (l31 >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = l31 * 128 + (((ks * 2 + lh) ^ sw) << 4);
  const int a_wave = wm * (C::BM / C::WM) * 128;
  const int b_wave = C::A_BYTES + wn * (C::BN / C::WN) * 128;

#pragma unroll
  for (int s = 0; s < C::STAGES - 1; ++s)
    if (s < KT) stage(s);

  for (int t = 0; t < KT; ++t) {
    // tile t landed (this wave's pieces), everybody's did once the barrier is passed; the barrier also closes the reads of the
    // buffer that the stage() below overwrites (it was consumed in iteration t-1)
    // tiles issued after t may stay in flight: min(STAGES - 2, KT - 1 - t) of them, LPT instructions each (wave-uniform branch)
    const int ahead = KT - 1 - t;
    if (C::STAGES >= 4 && ahead >= 2) wait_vmcnt<2 * C::LPT>();
    else if (C::STAGES >= 3 && ahead >= 1) wait_vmcnt<C::LPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + C::STAGES - 1 < KT) stage(t + C::STAGES - 1);
    const char* cs = smem + (t % C::STAGES) * C::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t af[C::TM], bfr[C::TN];
#pragma unroll
      for (int i = 0; i < C::TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(cs + a_wave + i * 4096 + foff[ks]);
#pragma unroll
      for (int j = 0; j < C::TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(cs + b_wave + j * 4096 + foff[ks]);
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  __builtin_amdgcn_s_barrier();     // all fragment reads done: the epilogue reuses the staging memory
}

// staging geometry of this lane: row within the tile and source chunk for piece i
template <class C> __device__ __forceinline__ int stage_row(int i) {
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  return (wid + i * C::NW) * 8 + (lane >> 3);
}
__device__ __forceinline__ int stage_chunk(int row) { return (threadIdx.x & 7) ^ ((row >> 1) & 7); }

}  // namespace gemmcore
