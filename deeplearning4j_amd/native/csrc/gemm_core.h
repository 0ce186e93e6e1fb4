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
static __device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

// MF_: MFMA shape, 32 = v_mfma_f32_32x32x16_bf16, 16 = v_mfma_f32_16x16x32_bf16 (same LDS bytes per FLOP for the same wave tile;
// the chip holds a higher clock on the 16x16 shape under load, MI355X_MICROARCH "DVFS give-back" item 7)
// DT_: operand type, 0 = bf16, 1 = fp16 (same lane maps, f16 MFMA), 2 = fp32 (v_mfma_f32_16x16x4_f32: exact fp32, 1/16 of the bf16
// rate; a K-tile is then 32 floats, so rows stay 128 bytes and staging / swizzle / fragment addresses are shared; MF must be 16).
enum { DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2 };
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

template <int BM_, int BN_, int WM_, int WN_, int STAGES_, int MF_ = 32, int DT_ = DT_BF16>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, STAGES = STAGES_, MF = MF_, DT = DT_;
  static constexpr int ESZ = DT_ == DT_F32 ? 4 : 2;               // operand element bytes
  static constexpr int BK = 128 / ESZ;                            // K extent of a tile (elements): 128-byte rows
  static constexpr int CH = 16 / ESZ;                             // elements per 16-byte chunk
  static_assert(DT_ != DT_F32 || MF_ == 16, "fp32 operands use the 16x16x4 MFMA");
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int TM = BM / WM / MF, TN = BN / WN / MF;      // MFMA blocks per wave
  static constexpr int ACC = MF == 32 ? 16 : 4;                   // accumulator registers per block
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;    // BK = 64 bf16 = 128-byte rows
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;        // 1-KiB pieces per wave per k-tile
  static constexpr int LPT = PA + PB;                             // LDS-DMA instructions per wave per k-tile
  // epilogue staging: output tile in the OUTPUT element size (row pitch padded by 16 bytes), one int64 row address per tile row,
  // and the per-slice column sums of the fused BatchNorm statistics
  static constexpr int pitch_c(int osz) { return BN * osz + 16; }
  static constexpr int epi_bytes(int osz) { return BM * pitch_c(osz) + BM * 8 + WM * 2 * BN * 4; }
  static constexpr int lds_bytes(int osz) {
    return STAGES * STAGE_BYTES > epi_bytes(osz) ? STAGES * STAGE_BYTES : epi_bytes(osz);
  }
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave stages whole 8-row pieces");
  static_assert(NW % 2 == 0 && STAGES >= 1 && STAGES <= 4, "even wave count (lane-constant swizzle), 1..4 stages");
  static_assert(MF == 32 || MF == 16, "MFMA block is 32x32x16 or 16x16x32");
  typedef float acc_t __attribute__((ext_vector_type(MF == 32 ? 16 : 4)));
  // accumulator element e of a block -> (row, col) inside the block
  static __device__ __forceinline__ int acc_row(int e, int lane) {
    return MF == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : (lane >> 4) * 4 + e;
  }
  // the same split into a lane part and a compile-time part (epilogue: one base address per lane + immediate offsets)
  static __device__ __forceinline__ int acc_row_lane(int lane) { return MF == 32 ? 4 * (lane >> 5) : (lane >> 4) * 4; }
  static constexpr int acc_row_const(int e) { return MF == 32 ? (e & 3) + 8 * (e >> 2) : e; }
  static __device__ __forceinline__ int acc_col(int lane) { return MF == 32 ? (lane & 31) : (lane & 15); }
  static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "wave tile is a multiple of the 32x32 MFMA block");
};

// one 16-byte A fragment x one 16-byte B fragment -> accumulate (K = 16 or 32 bf16/fp16 values, or 4 MFMAs over 4 fp32 values)
template <class C>
__device__ __forceinline__ void mma(const bf16x8_t& a, const bf16x8_t& b, typename C::acc_t& acc) {
  if constexpr (C::DT == DT_BF16) {
    if constexpr (C::MF == 32) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  } else if constexpr (C::DT == DT_F16) {
    const f16x8_t ah = __builtin_bit_cast(f16x8_t, a), bh = __builtin_bit_cast(f16x8_t, b);
    if constexpr (C::MF == 32) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  } else {
    // lane (row l & 15, group g = l >> 4) holds k = 4 g + j, j = 0..3 of this 16-wide k slab; MFMA step j sums k = {j, 4+j, 8+j,
    // 12+j}: a permutation of the slab's k values, identical for A and B
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc, 0, 0, 0);
  }
}

// Main loop shared by GEMM and implicit-GEMM convolution. `Src` provides, per k-tile t, the source address of this lane's 16-byte
// chunk for staged row i of A / B:  auto ts = tile(t) (wave-uniform scalars of the tile), const void* a(int i, ts), const void* b(int i, ts).
template <class C, class Src>
__device__ __forceinline__ void mainloop(char* smem, const Src& src, int KT, typename C::acc_t (&acc)[C::TM][C::TN]) {
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
    const auto ts = src.tile(t);          // wave-uniform per-tile scalars (tap, channel offset ...) computed ONCE per k-tile
#pragma unroll
    for (int i = 0; i < C::PA; ++i) glds16(src.a(i, ts), sb + (wid + i * C::NW) * 1024);
#pragma unroll
    for (int i = 0; i < C::PB; ++i) glds16(src.b(i, ts), sb + C::A_BYTES + (wid + i * C::NW) * 1024);
  };

  // per-lane fragment read offsets inside an MF-row block: row lr, 16-byte chunk q, swizzled by (row >> 1) & 7.
  //   32x32x16: lr = lane & 31, q = 2 ks + (lane >> 5), ks = 0..3;   16x16x32: lr = lane & 15, q = 4 ks + (lane >> 4), ks = 0..1
  constexpr int KS = C::MF == 32 ? 4 : 2;
  const int lr = C::MF == 32 ? (lane & 31) : (lane & 15);
  const int lq = C::MF == 32 ? (lane >> 5) : (lane >> 4);
  const int sw = (lr >> 1) & 7;
  int foff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) foff[ks] = lr * 128 + (((ks * (8 / KS) + lq) ^ sw) << 4);
  const int a_wave = wm * (C::BM / C::WM) * 128;
  const int b_wave = C::A_BYTES + wn * (C::BN / C::WN) * 128;
  constexpr int BLK = C::MF * 128;        // bytes between consecutive MFMA row blocks

  if (C::STAGES == 1) stage(0);
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
    if (C::STAGES > 1 && t + C::STAGES - 1 < KT) stage(t + C::STAGES - 1);
    const char* cs = smem + (t % C::STAGES) * C::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8_t af[C::TM], bfr[C::TN];
#pragma unroll
      for (int i = 0; i < C::TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(cs + a_wave + i * BLK + foff[ks]);
#pragma unroll
      for (int j = 0; j < C::TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(cs + b_wave + j * BLK + foff[ks]);
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j) mma<C>(af[i], bfr[j], acc[i][j]);
    }
    if (C::STAGES == 1 && t + 1 < KT) { __builtin_amdgcn_s_barrier(); stage(t + 1); }
  }
  __builtin_amdgcn_s_barrier();     // all fragment reads done: the epilogue reuses the staging memory
}

// Ping-pong main loop for the 256x256x64 tile, 8 waves as 2 (M) x 4 (N), 16x16x32 MFMA, two K-tile buffers (128 KiB LDS).
//
// A barrier-synchronous loop leaves the matrix pipe idle while every wave of the workgroup issues its LDS-DMA pieces and waits for
// its first fragments (profiles/r2_gemm_lab_v2.txt: 41 % MFMA busy, 46 % of wave time in s_waitcnt / s_barrier). Here the two wave
// groups (waves 0-3 = rows 0..127, waves 4-7 = rows 128..255; one wave of each group per SIMD) run ONE BARRIER APART: a K-tile is
// four phases  { LOAD segment | barrier | 16 MFMAs | barrier }, and while group 0 is in its MFMA segment group 1 is in its LOAD
// segment (fragment ds_reads + the LDS-DMA issue for the next K-tile) and vice versa, so each SIMD's matrix pipe always has one wave
// feeding it. Phases of K-tile t for a wave (output 128 x 64 = quadrants of 64 x 32):
//   0: read A rows 0..63, B cols 0..31; issue DMA of A(t+1)   -> quadrant (0,0)
//   1: read B cols 32..63;             issue DMA of B(t+1)   -> quadrant (0,1)
//   2: read A rows 64..127                                    -> quadrant (1,1)
//   3: counted wait for tile t+1 (issued two phases earlier)  -> quadrant (1,0)   (B cols 0..31 stayed in registers)
// Hazards (barrier indices in the comments of the code): a buffer is restaged only after the last fragment read of both groups has
// been waited for and a barrier passed; DMA data is read only after every issuing wave's vmcnt wait and a barrier behind it.
template <class C, class Src>
__device__ __forceinline__ void mainloop_pingpong(char* smem, const Src& src, int KT, typename C::acc_t (&acc)[C::TM][C::TN]) {
  static_assert(C::BM == 256 && C::BN == 256 && C::WM == 2 && C::WN == 4 && C::MF == 16 && C::STAGES == 2, "ping-pong geometry");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wn = wid & 3;
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  auto stage_a = [&](int t) {
    char* sb = smem + (t & 1) * C::STAGE_BYTES;
    const auto ts = src.tile(t);
#pragma unroll
    for (int i = 0; i < C::PA; ++i) glds16(src.a(i, ts), sb + (wid + i * C::NW) * 1024);
  };
  auto stage_b = [&](int t) {
    char* sb = smem + (t & 1) * C::STAGE_BYTES + C::A_BYTES;
    const auto ts = src.tile(t);
#pragma unroll
    for (int i = 0; i < C::PB; ++i) glds16(src.b(i, ts), sb + (wid + i * C::NW) * 1024);
  };
  const int lr = lane & 15, lq = lane >> 4, sw = (lr >> 1) & 7;
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = lr * 128 + (((ks * 4 + lq) ^ sw) << 4);
  const int a_wave = grp * 128 * 128;
  const int b_wave = C::A_BYTES + wn * 64 * 128;
  constexpr int BLK = 16 * 128;

#define PP_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
// end of a LOAD segment: the fragments must be IN REGISTERS before the barrier, so their LDS latency overlaps the other group's MFMA
// segment instead of opening this group's own (measured: with the wait behind the barrier a phase took ~580 cycles instead of ~256)
#define PP_LOAD_END() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PP_BAR(); } while (0)
  stage_a(0); stage_b(0);
  wait_vmcnt<0>();
  PP_BAR();
  if (grp == 1) PP_BAR();                 // stagger: group 1 runs one barrier behind group 0

  bf16x8_t af[4][2], b0[2][2], b1[2][2];
  for (int t = 0; t < KT; ++t) {
    const char* cs = smem + (t & 1) * C::STAGE_BYTES;
    const bool more = t + 1 < KT;
    // ---- phase 0
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8_t*>(cs + a_wave + i * BLK + foff[ks]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b0[j][ks] = *reinterpret_cast<const bf16x8_t*>(cs + b_wave + j * BLK + foff[ks]);
    if (more) stage_a(t + 1);
    PP_LOAD_END();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], b0[j][ks], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PP_BAR();
    // ---- phase 1
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b1[j][ks] = *reinterpret_cast<const bf16x8_t*>(cs + b_wave + (2 + j) * BLK + foff[ks]);
    if (more) stage_b(t + 1);
    PP_LOAD_END();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], b1[j][ks], acc[i][2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PP_BAR();
    // ---- phase 2
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8_t*>(cs + a_wave + (4 + i) * BLK + foff[ks]);
    PP_LOAD_END();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], b1[j][ks], acc[4 + i][2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PP_BAR();
    // ---- phase 3: tile t+1 (DMA issued in phases 0 and 1) must have landed before the barrier that opens its first read
    wait_vmcnt<0>();
    PP_LOAD_END();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], b0[j][ks], acc[4 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PP_BAR();
  }
  if (grp == 0) PP_BAR();                 // re-align the two groups: every LDS read of the main loop is complete behind this
#undef PP_BAR
#undef PP_LOAD_END
}

// Deep-prefetch form of the ping-pong loop: same tile, wave groups, phases and fragment reads, but the K-tile is staged as four
// HALF-TILES that are re-staged as soon as their last fragment read is behind a barrier, so every LDS-DMA has ~1.5 K-tiles (five to
// six phases) to land and the loop never drains vmcnt to 0 (the drain in phase 3 of mainloop_pingpong exposed the L2 / HBM round trip
// once per K-tile: ~600 cycles per phase against 256 of MFMA work):
//   A-lo = rows g*128 + 0..63 (read in phase 0), A-hi = rows g*128 + 64..127 (phase 2), B-lo = cols wn*64 + 0..31 (phase 0),
//   B-hi = cols wn*64 + 32..63 (phase 1); every wave stages two 1-KiB pieces of each half-tile.
//   issue order (the order the counted waits assume):  phase 0: A-hi(t+1)   1: A-lo(t+2)   2: B-lo(t+2)   3: B-hi(t+2)
//   waits (after the phase's issue, before its first barrier; the data is read one phase later):
//     phase 0: vmcnt(10) -> B-hi(t) landed     phase 1: vmcnt(10) -> A-hi(t)     phase 3: vmcnt(10) -> A-lo(t+1), B-lo(t+1)
//   (five half-tiles x 2 instructions stay in flight behind each wait).
// WAR: a slot is re-staged at least one barrier after the lgkmcnt(0) + barrier that closes its last read by BOTH groups (group 1 runs
// one barrier behind): A-lo/B-lo(t) are read in phase 0 (closed by barrier 8t+1) and re-staged in phases 1 / 2 (intervals >= 8t+2).
// Tiles past the end are staged from the zero page, so the counts hold in the tail without branches.
// Src: a_half(h, s, ktile) / b_half(h, s, ktile) -> this lane's source address for piece s (0..1) of half h of k-tile `ktile`.
template <class C>
__device__ __forceinline__ int pp_piece_a(int wid, int h, int s) { const int idx = wid * 2 + s; return (idx >> 3) * 16 + h * 8 + (idx & 7); }
template <class C>
__device__ __forceinline__ int pp_piece_b(int wid, int h, int s) { const int idx = wid * 2 + s; return (idx >> 2) * 8 + h * 4 + (idx & 3); }

template <class C, class Src>
__device__ __forceinline__ void mainloop_pingpong_deep(char* smem, const Src& src, int KT, typename C::acc_t (&acc)[C::TM][C::TN]) {
  static_assert(C::BM == 256 && C::BN == 256 && C::WM == 2 && C::WN == 4 && C::MF == 16 && C::STAGES == 2, "ping-pong geometry");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wn = wid & 3;
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // LDS byte offsets (inside a stage buffer) of this wave's two pieces of every half-tile
  int oa[2][2], ob[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      oa[h][s] = pp_piece_a<C>(wid, h, s) * 1024;
      ob[h][s] = C::A_BYTES + pp_piece_b<C>(wid, h, s) * 1024;
    }
  auto stage_ah = [&](int h, int t) {
    char* sb = smem + (t & 1) * C::STAGE_BYTES;
    glds16(src.a_half(h, 0, t), sb + oa[h][0]);
    glds16(src.a_half(h, 1, t), sb + oa[h][1]);
  };
  auto stage_bh = [&](int h, int t) {
    char* sb = smem + (t & 1) * C::STAGE_BYTES;
    glds16(src.b_half(h, 0, t), sb + ob[h][0]);
    glds16(src.b_half(h, 1, t), sb + ob[h][1]);
  };
  const int lr = lane & 15, lq = lane >> 4, sw = (lr >> 1) & 7;
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = lr * 128 + (((ks * 4 + lq) ^ sw) << 4);
  const int a_wave = grp * 128 * 128;
  const int b_wave = C::A_BYTES + wn * 64 * 128;
  constexpr int BLK = 16 * 128;

#define PP_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_LOAD_END() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PP_BAR(); } while (0)
  stage_ah(0, 0); stage_bh(0, 0); stage_bh(1, 0); stage_ah(1, 0);
  stage_ah(0, 1); stage_bh(0, 1); stage_bh(1, 1);
  wait_vmcnt<6>();                        // k-tile 0 landed
  PP_BAR();
  if (grp == 1) PP_BAR();                 // stagger: group 1 runs one barrier behind group 0

  bf16x8_t af[4][2], b0[2][2], b1[2][2];
  for (int t = 0; t < KT; ++t) {
    const char* cs = smem + (t & 1) * C::STAGE_BYTES;
    // ---- phase 0
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8_t*>(cs + a_wave + i * BLK + foff[ks]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b0[j][ks] = *reinterpret_cast<const bf16x8_t*>(cs + b_wave + j * BLK + foff[ks]);
    stage_ah(1, t + 1);
    wait_vmcnt<10>();
    PP_LOAD_END();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], b0[j][ks], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PP_BAR();
    // ---- phase 1
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b1[j][ks] = *reinterpret_cast<const bf16x8_t*>(cs + b_wave + (2 + j) * BLK + foff[ks]);
    stage_ah(0, t + 2);
    wait_vmcnt<10>();
    PP_LOAD_END();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], b1[j][ks], acc[i][2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PP_BAR();
    // ---- phase 2
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8_t*>(cs + a_wave + (4 + i) * BLK + foff[ks]);
    stage_bh(0, t + 2);
    PP_LOAD_END();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], b1[j][ks], acc[4 + i][2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PP_BAR();
    // ---- phase 3
    stage_bh(1, t + 2);
    wait_vmcnt<10>();
    PP_LOAD_END();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], b0[j][ks], acc[4 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PP_BAR();
  }
  wait_vmcnt<0>();                        // the zero-page stages of the tiles past the end: the epilogue reuses the staging memory
  if (grp == 0) PP_BAR();                 // re-align the two groups: every LDS read of the main loop is complete behind this
  PP_BAR();
#undef PP_BAR
#undef PP_LOAD_END
}

// Tile order inside the (XCD-contiguous) id space: GM consecutive ids walk down GM row tiles before moving to the next column
// tile, so the ~32 workgroups an XCD runs at a time cover a GM x (32 / GM) block of tiles and share their A / B panels in L2.
__device__ __forceinline__ void tile_coords(int lid, int mtiles, int ntiles, int gm, int& mt, int& nt) {
  const int per_group = gm * ntiles;
  const int g = lid / per_group, first_m = g * gm;
  const int rows = (mtiles - first_m < gm) ? (mtiles - first_m) : gm;
  const int r = lid - g * per_group;
  mt = first_m + r % rows;
  nt = r / rows;
}

// staging geometry of this lane: row within the tile and source chunk for piece i
template <class C> __device__ __forceinline__ int stage_row(int i) {
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  return (wid + i * C::NW) * 8 + (lane >> 3);
}
__device__ __forceinline__ int stage_chunk(int row) { return (threadIdx.x & 7) ^ ((row >> 1) & 7); }

}  // namespace gemmcore
