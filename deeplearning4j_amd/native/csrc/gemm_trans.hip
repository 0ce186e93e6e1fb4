// Entry points of the transposed-operand GEMM kernels (gemm_trans.h): NN, TN and TT forms next to the NT form of conv_mfma.hip, batched
// (blockIdx.y) and split-K (blockIdx.z, fp32 atomics) - MmulHelper.cu:255-341 / batched_gemm.cu:145-193 operand coverage.
#include "gemm_trans.h"

using namespace gemmcore;

namespace {

template <int DT, int OT, bool TA, bool TB>
int dispatch_tile(GemmTP& p, int tile, int batch, int splits, hipStream_t st) {
  switch (tile) {
    case 0: return launch_gemm_flex<Cfg<128, 128, 2, 2, 2, 16, DT>, OT, TA, TB>(p, batch, splits, st);
    case 1: return launch_gemm_flex<Cfg<128, 64, 2, 2, 2, 16, DT>, OT, TA, TB>(p, batch, splits, st);
    case 2: return launch_gemm_flex<Cfg<64, 64, 2, 2, 2, 16, DT>, OT, TA, TB>(p, batch, splits, st);
    case 3: return launch_gemm_flex<Cfg<64, 128, 2, 2, 2, 16, DT>, OT, TA, TB>(p, batch, splits, st);
    default: return -2;
  }
}

template <int DT, int OT>
int dispatch_trans(GemmTP& p, int ta, int tb, int tile, int batch, int splits, hipStream_t st) {
  if (ta && tb) return dispatch_tile<DT, OT, true, true>(p, tile, batch, splits, st);
  if (ta) return dispatch_tile<DT, OT, true, false>(p, tile, batch, splits, st);
  if (tb) return dispatch_tile<DT, OT, false, true>(p, tile, batch, splits, st);
  return dispatch_tile<DT, OT, false, false>(p, tile, batch, splits, st);
}

}  // namespace

// C[b] = act(alpha * op(A[b]) . op(B[b]) + bias) (+ residual[b]) (+= with accum)
// dims (int64): M N K lda ldb ldc sA sB sC batch tile dtype out_f32 act transA kmajorB splits accum
//   transA  = 1: A is [K, M] with M contiguous (lda = elements between consecutive k);   0: A is [M, K] K-contiguous
//   kmajorB = 1: B is [K, N] with N contiguous (the "N" operand of a row-major NN GEMM); 0: B is [N, K] K-contiguous (the core's NT form)
//   splits > 1: K is cut into `splits` ranges whose partial products are added with fp32 atomics (fp32 output, caller pre-zeroes C
//   unless accum says the old contents stay)
DL4J_API int dl4j_gemm_ex2(const void* A, const void* B, void* Cout, const float* bias, const void* residual, float alpha, const int64_t* d,
                           hipStream_t st) {
  GemmTP p;
  p.A = A; p.B = B;
  p.epi.out = Cout; p.epi.residual = residual; p.epi.bias = bias; p.epi.stats = nullptr; p.epi.stat_shift = nullptr;
  p.epi.alpha = alpha; p.epi.act = (int)d[13]; p.epi.accum = (int)d[17];
  p.M = (int)d[0]; p.N = (int)d[1]; p.K = (int)d[2];
  p.lda = d[3]; p.ldb = d[4]; p.ldc = d[5]; p.sA = d[6]; p.sB = d[7]; p.sC = d[8];
  p.gm = 0;
  const int batch = (int)d[9], tile = (int)d[10], dt = (int)d[11], of32 = (int)d[12], ta = (int)d[14], tb = (int)d[15];
  int splits = (int)d[16];
  const int ch = dt == DT_F32 ? 4 : 8;
  if (!A || !B || !Cout) return -3;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || batch <= 0 || batch > 65535 || splits > 65535) return -2;
  if (p.lda % ch || p.ldb % ch || p.sA % ch || p.sB % ch) return -2;                      // 16-byte aligned operand chunks
  if ((!ta || !tb) && p.K % ch) return -2;                                               // a K-contiguous operand: whole chunks along K
  if ((ta && p.M % ch) || (tb && p.N % ch)) return -2;                                    // a K-major operand: whole chunks along M / N
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return -2;
  const bool f32out = of32 || dt == DT_F32;
  if (splits > 1) {
    if (!f32out || p.epi.act != 0) return -2;                                             // partial sums: linear epilogue, fp32 atomics
    p.epi.accum = 2;
  }
  if (dt == DT_BF16) return f32out ? dispatch_trans<DT_BF16, OT_F32>(p, ta, tb, tile, batch, splits, st) : dispatch_trans<DT_BF16, OT_BF16>(p, ta, tb, tile, batch, splits, st);
  if (dt == DT_F16) return f32out ? dispatch_trans<DT_F16, OT_F32>(p, ta, tb, tile, batch, splits, st) : dispatch_trans<DT_F16, OT_F16>(p, ta, tb, tile, batch, splits, st);
  if (dt == DT_F32) return dispatch_trans<DT_F32, OT_F32>(p, ta, tb, tile, batch, splits, st);
  return -2;
}
