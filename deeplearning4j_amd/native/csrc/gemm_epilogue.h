// Shared epilogue of the LDS-DMA GEMM / implicit-GEMM kernels: alpha, bias, activation, rounding to the output type, LDS transpose
// so every output row leaves as whole 16-byte chunks, optional residual add / in-place accumulation, optional per-column
// (sum, sum of squares) partials of the ROUNDED values for a following BatchNorm (one pass over y saved; reference computes the
// statistics in a separate pass, nn/layers/normalization/BatchNormalization.java:270-360).
#pragma once
#include "gemm_core.h"

namespace gemmcore {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU_TANH = 2, ACT_GELU_ERF = 3, ACT_TANH = 4, ACT_SIGMOID = 5 };
enum { OT_BF16 = 0, OT_F16 = 1, OT_F32 = 2 };

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

template <int OT> struct Out;
template <> struct Out<OT_BF16> {
  typedef bf16_t T; static constexpr int SZ = 2;
  static __device__ __forceinline__ T cv(float f) { return f2bf(f); }
  static __device__ __forceinline__ float fl(T v) { return bf2f(v); }
};
template <> struct Out<OT_F16> {
  typedef uint16_t T; static constexpr int SZ = 2;
  static __device__ __forceinline__ T cv(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
  static __device__ __forceinline__ float fl(T v) { return (float)__builtin_bit_cast(_Float16, v); }
};
template <> struct Out<OT_F32> {
  typedef float T; static constexpr int SZ = 4;
  static __device__ __forceinline__ T cv(float f) { return f; }
  static __device__ __forceinline__ float fl(T v) { return v; }
};

struct EpiP {
  void* out;                 // output tensor
  const void* residual;      // same addressing as out, added after the activation (or null)
  const float* bias;         // [N] or null
  float* stats;              // [mtiles][2][N] partial column sums of the rounded outputs (or null)
  const float* stat_shift;   // [N] per-column shift subtracted before summing (running mean; null = 0): keeps E[x^2]-E[x]^2 stable
  float alpha;
  int act;
  int accum;                 // 1: out += result (read-modify-write), 2: fp32 atomic add (split-K partials)
};

// RowOff: int64 element offset of tile row r's first column (n0) in `out`, or -1 when the row is out of range.
// vrows: number of valid rows of this tile (rows >= vrows are padding: excluded from the statistics, never stored).
template <class C, int OT, int ACTK, int STATS, class RowOff>   // ACTK: ACT_NONE = linear / ReLU, else that activation; STATS: 0 never, 1 always, 2 run time (e.stats)
__device__ __forceinline__ void epilogue_impl(char* smem, typename C::acc_t (&acc)[C::TM][C::TN], int mt, int n0, int N, int vrows,
                                              const EpiP& e, RowOff rowoff) {
  typedef Out<OT> O;
  typedef typename O::T OTy;
  constexpr int PITCH = C::pitch_c(O::SZ);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / C::WN, wn = wid - wm * C::WN;
  const int lc = C::acc_col(lane);
  char* sC = smem;
  int64_t* sRow = reinterpret_cast<int64_t*>(smem + C::BM * PITCH);
  float* sS = reinterpret_cast<float*>(smem + C::BM * PITCH + C::BM * 8);
  // statistics straight from the accumulator registers: a lane owns one column of every block, so the column sums are a sum over
  // its registers and row blocks, then over the lanes that share the column (xor 16 / 32), then over the wave rows through LDS
  const bool want_stats = STATS == 2 ? e.stats != nullptr : STATS == 1;
  const int row0 = wm * (C::BM / C::WM) + C::acc_row_lane(lane);
  char* wbase = sC + row0 * PITCH + (wn * (C::BN / C::WN) + lc) * O::SZ;
  // ACTK == ACT_NONE (no activation or ReLU): branch-free, v = max(alpha * acc + bias, lo) with lo = 0 or -inf. Inlined per element (128
  // accumulators per lane on the 256 x 256 tile) the six-way activation switch makes the epilogue ~30k instructions / ~2000
  // branches, far beyond the instruction cache (+10 us per output tile, measured): only the transcendental activations pay that.
  const float lo = e.act == ACT_RELU ? 0.f : -__builtin_inff();
#pragma unroll
  for (int j = 0; j < C::TN; ++j) {
    const int col = wn * (C::BN / C::WN) + j * C::MF + lc;
    const bool cok = n0 + col < N;
    const float bv = (e.bias && cok) ? e.bias[n0 + col] : 0.f;
    const float sh = (want_stats && e.stat_shift && cok) ? e.stat_shift[n0 + col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int q = 0; q < C::ACC; ++q) {
        const int row = row0 + i * C::MF + C::acc_row_const(q);
        float f = acc[i][j][q] * e.alpha + bv;
        if constexpr (ACTK == ACT_NONE) f = fmaxf(f, lo); else f = apply_act(f, ACTK);
        const OTy o = O::cv(f);
        *reinterpret_cast<OTy*>(wbase + (i * C::MF + C::acc_row_const(q)) * PITCH + j * C::MF * O::SZ) = o;
        if (want_stats) {
          const float v = row < vrows ? O::fl(o) - sh : 0.f;
          s1 += v; s2 += v * v;
        }
      }
    if (want_stats) {
      if (C::MF == 16) { s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64); }
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (lane < C::MF) {
        sS[(wm * 2 + 0) * C::BN + col] = s1;
        sS[(wm * 2 + 1) * C::BN + col] = s2;
      }
    }
  }
  for (int r = tid; r < C::BM; r += C::NT) sRow[r] = rowoff(r);
  __syncthreads();
  if (want_stats) {
    for (int idx = tid; idx < 2 * C::BN; idx += C::NT) {
      const int qs = idx / C::BN, c = idx - qs * C::BN;
      if (n0 + c < N) {
        float s = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < C::WM; ++w_) s += sS[(w_ * 2 + qs) * C::BN + c];
        e.stats[((int64_t)mt * 2 + qs) * N + n0 + c] = s;
      }
    }
  }
  constexpr int EPC = 16 / O::SZ;               // elements per 16-byte chunk
  constexpr int CPR = C::BN / EPC;
  OTy* out = reinterpret_cast<OTy*>(e.out);
  const OTy* res = reinterpret_cast<const OTy*>(e.residual);
  const bool full_cols = n0 + C::BN <= N;
  for (int idx = tid; idx < C::BM * CPR; idx += C::NT) {
    const int row = idx / CPR, ch = idx - row * CPR;
    const int64_t ro = sRow[row];
    const int n = n0 + ch * EPC;
    if (ro < 0 || n >= N) continue;
    const int64_t o = ro + ch * EPC;
    const char* sp = sC + row * PITCH + ch * 16;
    if constexpr (OT == OT_F32) {
      if (e.accum == 2) {                       // split-K partial: fp32 atomics into a pre-zeroed (or beta-scaled) output
        const float* se = reinterpret_cast<const float*>(sp);
        for (int j = 0; j < EPC && n + j < N; ++j) {
          float f = se[j];
          if (res != nullptr) f += res[o + j];
          atomicAdd(out + o + j, f);
        }
        continue;
      }
    }
    if ((full_cols || n + EPC <= N) && ((reinterpret_cast<uintptr_t>(out + o) & 15) == 0)) {
      f32x4 v = *reinterpret_cast<const f32x4*>(sp);
      if (res != nullptr || e.accum) {
        OTy* ve = reinterpret_cast<OTy*>(&v);
        if (res != nullptr) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(res + o);
          const OTy* re = reinterpret_cast<const OTy*>(&rv);
#pragma unroll
          for (int j = 0; j < EPC; ++j) ve[j] = O::cv(O::fl(ve[j]) + O::fl(re[j]));
        }
        if (e.accum) {
          const f32x4 pv = *reinterpret_cast<const f32x4*>(out + o);
          const OTy* pe = reinterpret_cast<const OTy*>(&pv);
#pragma unroll
          for (int j = 0; j < EPC; ++j) ve[j] = O::cv(O::fl(ve[j]) + O::fl(pe[j]));
        }
      }
      *reinterpret_cast<f32x4*>(out + o) = v;
    } else {
      const OTy* se = reinterpret_cast<const OTy*>(sp);
      for (int j = 0; j < EPC && n + j < N; ++j) {
        float f = O::fl(se[j]);
        if (res != nullptr) f += O::fl(res[o + j]);
        if (e.accum) f += O::fl(out[o + j]);
        out[o + j] = O::cv(f);
      }
    }
  }
}

// one complete straight-line body per activation kind (nothing is joined behind the branches: each ends the kernel), chosen by
// workgroup-uniform tests
template <class C, int OT, class RowOff>
__device__ __forceinline__ void epilogue(char* smem, typename C::acc_t (&acc)[C::TM][C::TN], int mt, int n0, int N, int vrows,
                                         const EpiP& e, RowOff rowoff) {
  if (e.act == ACT_NONE || e.act == ACT_RELU) {
    if (e.stats == nullptr) epilogue_impl<C, OT, ACT_NONE, 0>(smem, acc, mt, n0, N, vrows, e, rowoff);
    else epilogue_impl<C, OT, ACT_NONE, 1>(smem, acc, mt, n0, N, vrows, e, rowoff);
  }
  else if (e.act == ACT_GELU_TANH) epilogue_impl<C, OT, ACT_GELU_TANH, 2>(smem, acc, mt, n0, N, vrows, e, rowoff);
  else if (e.act == ACT_GELU_ERF) epilogue_impl<C, OT, ACT_GELU_ERF, 2>(smem, acc, mt, n0, N, vrows, e, rowoff);
  else if (e.act == ACT_TANH) epilogue_impl<C, OT, ACT_TANH, 2>(smem, acc, mt, n0, N, vrows, e, rowoff);
  else epilogue_impl<C, OT, ACT_SIGMOID, 2>(smem, acc, mt, n0, N, vrows, e, rowoff);
}

}  // namespace gemmcore
