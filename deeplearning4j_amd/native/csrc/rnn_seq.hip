// Recurrent layers as NATIVE time loops: the whole sequence of a layer is issued from C++ — per step one recurrent GEMM on the
// in-tree MFMA kernel (h_{t-1} . RW^T with the pre-computed input projection added in the GEMM epilogue) and one fused pointwise cell
// kernel — instead of a Python loop around library GEMMs. The reference runs the same loop in Java with 2 GEMMs + ~20 elementwise ops
// per step (nn/layers/recurrent/LSTMHelpers.java:211-369; libnd4j helpers/impl/lstmLayer.cpp:200-420, helpers/impl/gru.cpp:60-180).
// Also here: the fused GRU cell (forward / backward pointwise math in one kernel each) for the gruCell / gru custom ops.
#include "gemm_kernels.h"

using namespace gemmcore;

extern "C" int dl4j_lstm_cell_fwd(const void* gates, const void* c_prev, const float* pf, const float* po, const float* pg,
                                  const float* mask, void* h, void* c, void* sv, int mb, int n, int dtype, hipStream_t st);
extern "C" int dl4j_lstm_cell_bwd(const void* dh, const void* dc_next, const void* c_prev, const void* sv, const float* pf,
                                  const float* po, const float* pg, const float* mask, void* dgates, void* dc_prev, float* dpeep,
                                  int mb, int n, int dtype, hipStream_t st);

// y[M, N] = a[M, K] . b[N, K]^T (+ residual[M, N]); dtype 0 fp32, 1 bf16 (the cell kernels' code space)
static int rnn_gemm(const void* a, const void* b, void* y, const void* residual, int M, int N, int K, int64_t ldb, int dtype,
                    hipStream_t st) {
  GemmP p;
  p.A = a; p.B = b;
  p.epi.out = y; p.epi.residual = residual; p.epi.bias = nullptr; p.epi.stats = nullptr; p.epi.stat_shift = nullptr;
  p.epi.alpha = 1.f; p.epi.act = ACT_NONE; p.epi.accum = 0;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = ldb; p.ldc = N; p.sA = p.sB = p.sC = 0; p.gm = 0;
  // a recurrent step is a short, wide GEMM (M = minibatch): the smallest tile gives the most workgroups
  if (dtype == 1) return launch_gemm<Cfg<64, 64, 2, 2, 2, 16, DT_BF16>, OT_BF16>(p, 1, st);
  return launch_gemm<Cfg<64, 64, 2, 2, 2, 16, DT_F32>, OT_F32>(p, 1, st);
}

static inline size_t esz(int dtype) { return dtype == 1 ? 2 : 4; }

// Forward over T steps. xw [T, mb, 4n] (input projection + bias), rw4 [4n, n] (DL4J 'f'-order RW is exactly this memory image),
// h0 / c0 [mb, n], maskT [T, mb] fp32 or null, H / Cs [T, mb, n] outputs (activations and cell states of every step),
// SV [T, mb, 6n] saved gate activations for the backward pass (or null), gates_ws [mb, 4n] scratch.
DL4J_API int dl4j_lstm_seq_fwd(const void* xw, const void* rw4, int64_t ld_rw, const void* h0, const void* c0, const float* pf,
                               const float* po, const float* pg, const float* maskT, void* H, void* Cs, void* SV, void* gates_ws,
                               int T, int mb, int n, int dtype, int reverse, hipStream_t st) {
  if (T <= 0 || mb <= 0 || n <= 0) return 0;
  if (dtype != 0 && dtype != 1) return -2;
  const int ch = dtype == 1 ? 8 : 4;
  if (n % ch != 0 || ld_rw % ch != 0) return -2;                 // 16-byte operand chunks of the recurrent GEMM
  const size_t e = esz(dtype);
  const char* hp = (const char*)h0;
  const char* cp = (const char*)c0;
  for (int s = 0; s < T; ++s) {
    const int t = reverse ? T - 1 - s : s;
    const char* xwt = (const char*)xw + (size_t)t * mb * 4 * n * e;
    int rc = rnn_gemm(hp, rw4, gates_ws, xwt, mb, 4 * n, n, ld_rw, dtype, st);
    if (rc) return rc;
    char* ht = (char*)H + (size_t)t * mb * n * e;
    char* ct = (char*)Cs + (size_t)t * mb * n * e;
    rc = dl4j_lstm_cell_fwd(gates_ws, cp, pf, po, pg, maskT ? maskT + (size_t)t * mb : nullptr, ht, ct,
                            SV ? (char*)SV + (size_t)t * mb * 6 * n * e : nullptr, mb, n, dtype, st);
    if (rc) return rc;
    hp = ht; cp = ct;
  }
  return 0;
}

// Backward over T steps. E [T, mb, n] = dL/dH, rw4t [n, 4n] = RW4 transposed (row j = the 4n gate weights leaving hidden unit j),
// SV / Cs from the forward, c0 initial cell state. Outputs: dG [T, mb, 4n] gate pre-activation gradients, dpeep [3, n] fp32
// (accumulated; zeroed by the caller), dh0 / dc0 [mb, n] gradients of the initial state (scratch the caller may ignore).
// dh_ws [mb, n]: running dL/dh_t; dc_ws [2][mb, n] ping-pong for dL/dc_t.
DL4J_API int dl4j_lstm_seq_bwd(const void* E, const void* rw4t, int64_t ld_rwt, const void* SV, const void* Cs, const void* c0,
                               const float* pf, const float* po, const float* pg, const float* maskT, void* dG, float* dpeep,
                               void* dh_ws, void* dc_ws, int T, int mb, int n, int dtype, int reverse, hipStream_t st) {
  if (T <= 0 || mb <= 0 || n <= 0) return 0;
  if (dtype != 0 && dtype != 1) return -2;
  const int ch = dtype == 1 ? 8 : 4;
  if ((4 * n) % ch != 0 || ld_rwt % ch != 0 || n % ch != 0) return -2;
  const size_t e = esz(dtype);
  const size_t hn = (size_t)mb * n * e;
  // dh of the LAST executed step is E alone; dc_next = 0
  const int t_last = reverse ? 0 : T - 1;
  hipError_t herr = hipMemcpyAsync(dh_ws, (const char*)E + (size_t)t_last * hn, hn, hipMemcpyDeviceToDevice, st);
  if (herr != hipSuccess) return (int)herr;
  // zero through the copy engine is avoided inside captured graphs (memset nodes misbehaved on replay, softmax_xent.hip): the
  // first dc buffer is produced by a GEMM of zero rows instead — simplest: scale E[t_last] by 0 through the cell kernel's contract
  // is not available, so a tiny fill kernel is used
  extern int dl4j_fill_zero_bytes(void* p, size_t bytes, hipStream_t st);
  int rc = dl4j_fill_zero_bytes(dc_ws, hn, st);
  if (rc) return rc;
  char* dc_cur = (char*)dc_ws;
  char* dc_nxt = (char*)dc_ws + hn;
  for (int s = T - 1; s >= 0; --s) {
    const int t = reverse ? T - 1 - s : s;                 // step s of the forward order processed time index t
    const int t_prev = reverse ? t + 1 : t - 1;            // time index whose output fed step s (none for s == 0)
    const char* cprev = s == 0 ? (const char*)c0 : (const char*)Cs + (size_t)t_prev * hn;
    char* dgt = (char*)dG + (size_t)t * mb * 4 * n * e;
    rc = dl4j_lstm_cell_bwd(dh_ws, dc_cur, cprev, (const char*)SV + (size_t)t * mb * 6 * n * e, pf, po, pg,
                            maskT ? maskT + (size_t)t * mb : nullptr, dgt, dc_nxt, dpeep, mb, n, dtype, st);
    if (rc) return rc;
    if (s > 0) {
      // dL/dh of the previous step = its own eps + dgates . RW4^T
      rc = rnn_gemm(dgt, rw4t, dh_ws, (const char*)E + (size_t)t_prev * hn, mb, n, 4 * n, ld_rwt, dtype, st);
      if (rc) return rc;
    }
    char* tmp = dc_cur; dc_cur = dc_nxt; dc_nxt = tmp;
  }
  return 0;
}

__global__ void __launch_bounds__(256) fill_zero_u32_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
int dl4j_fill_zero_bytes(void* p, size_t bytes, hipStream_t st) {
  hipLaunchKernelGGL(fill_zero_u32_kernel, dim3(grid_for((int64_t)(bytes / 4), 1024)), dim3(256), 0, st, (uint32_t*)p, bytes / 4);
  return (int)hipGetLastError();
}

// ===================================================================================================================== GRU
// libnd4j gruCell / gru math (helpers/impl/gru.cpp): gate order [r | u | c],
//   r = sigmoid(xz_r + hz_r), u = sigmoid(xz_u + hz_u), c = tanh(xz_c + r * hz_c), h = u * h_prev + (1 - u) * c
// with xz = x . Wx + b and hz = h_prev . Wh (the `gru` layer form). One thread per (row, unit).
template <typename T>
__global__ void __launch_bounds__(256) gru_cell_fwd_kernel(const T* __restrict__ xz, const T* __restrict__ hz,
                                                           const T* __restrict__ hprev, T* __restrict__ h, T* __restrict__ sv,
                                                           int mb, int n) {
  const int64_t total = (int64_t)mb * n;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / n; const int j = (int)(idx - row * n);
    const int64_t g = row * 3 * n + j;
    const float r = 1.f / (1.f + __expf(-(Cvt<T>::ld(xz, g) + Cvt<T>::ld(hz, g))));
    const float u = 1.f / (1.f + __expf(-(Cvt<T>::ld(xz, g + n) + Cvt<T>::ld(hz, g + n))));
    const float hc = Cvt<T>::ld(hz, g + 2 * n);
    const float c = tanhf(Cvt<T>::ld(xz, g + 2 * n) + r * hc);
    const float hp = Cvt<T>::ld(hprev, idx);
    Cvt<T>::st(h, idx, u * hp + (1.f - u) * c);
    if (sv) { Cvt<T>::st(sv, g, r); Cvt<T>::st(sv, g + n, u); Cvt<T>::st(sv, g + 2 * n, c); }
  }
}

// dxz [mb, 3n] (gradient of the pre-activations = of xz), dhz [mb, 3n] (gradient of hz: equal to dxz for r / u, r * dc_pre for c),
// dhp [mb, n] = direct part of dL/dh_prev (u * dh); the caller adds dhz . Wh^T.
template <typename T>
__global__ void __launch_bounds__(256) gru_cell_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ sv,
                                                           const T* __restrict__ hz, const T* __restrict__ hprev,
                                                           T* __restrict__ dxz, T* __restrict__ dhz, T* __restrict__ dhp, int mb, int n) {
  const int64_t total = (int64_t)mb * n;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / n; const int j = (int)(idx - row * n);
    const int64_t g = row * 3 * n + j;
    const float r = Cvt<T>::ld(sv, g), u = Cvt<T>::ld(sv, g + n), c = Cvt<T>::ld(sv, g + 2 * n);
    const float d = Cvt<T>::ld(dh, idx), hp = Cvt<T>::ld(hprev, idx), hc = Cvt<T>::ld(hz, g + 2 * n);
    const float dc_pre = d * (1.f - u) * (1.f - c * c);
    const float du_pre = d * (hp - c) * u * (1.f - u);
    const float dr_pre = dc_pre * hc * r * (1.f - r);
    Cvt<T>::st(dxz, g, dr_pre); Cvt<T>::st(dxz, g + n, du_pre); Cvt<T>::st(dxz, g + 2 * n, dc_pre);
    Cvt<T>::st(dhz, g, dr_pre); Cvt<T>::st(dhz, g + n, du_pre); Cvt<T>::st(dhz, g + 2 * n, dc_pre * r);
    Cvt<T>::st(dhp, idx, d * u);
  }
}

DL4J_API int dl4j_gru_cell_fwd(const void* xz, const void* hz, const void* hprev, void* h, void* sv, int mb, int n, int dtype,
                               hipStream_t st) {
  if (mb <= 0 || n <= 0) return 0;
  const int grid = grid_for((int64_t)mb * n, 256);
  if (dtype == 1)
    hipLaunchKernelGGL(gru_cell_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)xz, (const bf16_t*)hz,
                       (const bf16_t*)hprev, (bf16_t*)h, (bf16_t*)sv, mb, n);
  else if (dtype == 0)
    hipLaunchKernelGGL(gru_cell_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)xz, (const float*)hz,
                       (const float*)hprev, (float*)h, (float*)sv, mb, n);
  else return -2;
  DL4J_RET_LAST();
}

DL4J_API int dl4j_gru_cell_bwd(const void* dh, const void* sv, const void* hz, const void* hprev, void* dxz, void* dhz, void* dhp,
                               int mb, int n, int dtype, hipStream_t st) {
  if (mb <= 0 || n <= 0) return 0;
  const int grid = grid_for((int64_t)mb * n, 256);
  if (dtype == 1)
    hipLaunchKernelGGL(gru_cell_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dh, (const bf16_t*)sv,
                       (const bf16_t*)hz, (const bf16_t*)hprev, (bf16_t*)dxz, (bf16_t*)dhz, (bf16_t*)dhp, mb, n);
  else if (dtype == 0)
    hipLaunchKernelGGL(gru_cell_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dh, (const float*)sv,
                       (const float*)hz, (const float*)hprev, (float*)dxz, (float*)dhz, (float*)dhp, mb, n);
  else return -2;
  DL4J_RET_LAST();
}

// gru layer forward as a native loop: xz [T, mb, 3n] = x . Wx + b, wh_t [3n, n] = Wh transposed (NT operand), h0 [mb, n];
// outputs H [T, mb, n], HZ [T, mb, 3n] and SV [T, mb, 3n] (kept for the backward pass; may be null for inference: then hz_ws
// [mb, 3n] is reused every step)
DL4J_API int dl4j_gru_seq_fwd(const void* xz, const void* wh_t, int64_t ld_wh, const void* h0, void* H, void* HZ, void* SV,
                              void* hz_ws, int T, int mb, int n, int dtype, hipStream_t st) {
  if (T <= 0 || mb <= 0 || n <= 0) return 0;
  if (dtype != 0 && dtype != 1) return -2;
  const int ch = dtype == 1 ? 8 : 4;
  if (n % ch != 0 || ld_wh % ch != 0) return -2;
  const size_t e = esz(dtype);
  const char* hp = (const char*)h0;
  for (int t = 0; t < T; ++t) {
    char* hzt = HZ ? (char*)HZ + (size_t)t * mb * 3 * n * e : (char*)hz_ws;
    int rc = rnn_gemm(hp, wh_t, hzt, nullptr, mb, 3 * n, n, ld_wh, dtype, st);
    if (rc) return rc;
    char* ht = (char*)H + (size_t)t * mb * n * e;
    rc = dl4j_gru_cell_fwd((const char*)xz + (size_t)t * mb * 3 * n * e, hzt, hp, ht,
                           SV ? (char*)SV + (size_t)t * mb * 3 * n * e : nullptr, mb, n, dtype, st);
    if (rc) return rc;
    hp = ht;
  }
  return 0;
}
