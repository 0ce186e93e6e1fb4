// Fused LSTM cell pointwise math for gfx950, DL4J gate order [a | f | o | g] with optional Graves peepholes.
//
// Reference: the reference has NO fused cell (helpers/cuda/lstm.cu composes mmul + ~20 elementwise NDArray ops per
// step; helpers/impl/lstmLayer.cpp time loop; Java LSTMHelpers.java:227-369 does the same with ND4J ops) — the
// cuDNN platform helper is its only fused path. Here one launch per timestep does all gate math after the two GEMMs:
//   a = tanh(za)  f = s(zf + c'·pf)  g = s(zg + c'·pg)  c = f·c' + g·a  o = s(zo + c·po)  h = o·tanh(c)
// and the backward kernel produces d(gates), dc' and the peephole gradients in one pass.
// Memory-bound: each thread handles one (row, unit) pair; consecutive lanes walk the unit dimension, so all eight
// gate/state streams are coalesced 64-lane accesses.
#include "common.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const T* __restrict__ gates, const T* __restrict__ c_prev,
                                                            const float* __restrict__ pf, const float* __restrict__ po,
                                                            const float* __restrict__ pg, const float* __restrict__ mask,
                                                            T* __restrict__ h, T* __restrict__ c, T* __restrict__ sv, int mb,
                                                            int n) {
  const int64_t total = (int64_t)mb * n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(idx / n), j = (int)(idx - (int64_t)r * n);
    const int64_t gb = (int64_t)r * 4 * n;
    const float cp = Cvt<T>::ld(c_prev, idx);
    float za = Cvt<T>::ld(gates, gb + j), zf = Cvt<T>::ld(gates, gb + n + j);
    float zo = Cvt<T>::ld(gates, gb + 2 * n + j), zg = Cvt<T>::ld(gates, gb + 3 * n + j);
    if (pf) { zf += cp * pf[j]; zg += cp * pg[j]; }
    const float a = tanhf(za), f = sigm(zf), g = sigm(zg);
    const float cn = f * cp + g * a;
    if (po) zo += cn * po[j];
    const float o = sigm(zo), ac = tanhf(cn);
    float hn = o * ac, cout = cn;
    if (sv) {
      const int64_t sb = (int64_t)r * 6 * n;
      Cvt<T>::st(sv, sb + j, a); Cvt<T>::st(sv, sb + n + j, f); Cvt<T>::st(sv, sb + 2 * n + j, o);
      Cvt<T>::st(sv, sb + 3 * n + j, g); Cvt<T>::st(sv, sb + 4 * n + j, cn); Cvt<T>::st(sv, sb + 5 * n + j, ac);
    }
    if (mask) { const float m = mask[r]; hn *= m; cout *= m; }
    Cvt<T>::st(h, idx, hn);
    Cvt<T>::st(c, idx, cout);
  }
}

// peephole gradients: per-thread partials are reduced over the rows a block owns before the atomics, so the
// number of atomics per unit is gridDim.y instead of mb.
template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ dc_next,
                                                            const T* __restrict__ c_prev, const T* __restrict__ sv,
                                                            const float* __restrict__ pf, const float* __restrict__ po,
                                                            const float* __restrict__ pg, const float* __restrict__ mask,
                                                            T* __restrict__ dgates, T* __restrict__ dc_prev,
                                                            float* __restrict__ dpeep, int mb, int n, int rows_per_block) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(mb, r0 + rows_per_block);
  float apf = 0.f, apo = 0.f, apg = 0.f;
  const float wf = pf ? pf[j] : 0.f, wo = po ? po[j] : 0.f, wg = pg ? pg[j] : 0.f;
  for (int r = r0; r < r1; ++r) {
    const int64_t idx = (int64_t)r * n + j, sb = (int64_t)r * 6 * n, gb = (int64_t)r * 4 * n;
    const float a = Cvt<T>::ld(sv, sb + j), f = Cvt<T>::ld(sv, sb + n + j), o = Cvt<T>::ld(sv, sb + 2 * n + j);
    const float g = Cvt<T>::ld(sv, sb + 3 * n + j), cn = Cvt<T>::ld(sv, sb + 4 * n + j), ac = Cvt<T>::ld(sv, sb + 5 * n + j);
    const float cp = Cvt<T>::ld(c_prev, idx);
    float d_h = Cvt<T>::ld(dh, idx), d_cn = Cvt<T>::ld(dc_next, idx);
    if (mask) { const float m = mask[r]; d_h *= m; d_cn *= m; }
    const float dzo = d_h * ac * o * (1.f - o);
    float dc = d_cn + d_h * o * (1.f - ac * ac);
    if (po) dc += dzo * wo;
    const float dza = dc * g * (1.f - a * a);
    const float dzf = dc * cp * f * (1.f - f);
    const float dzg = dc * a * g * (1.f - g);
    float dcp = dc * f;
    if (pf) { dcp += dzf * wf + dzg * wg; apf += dzf * cp; apo += dzo * cn; apg += dzg * cp; }
    Cvt<T>::st(dgates, gb + j, dza); Cvt<T>::st(dgates, gb + n + j, dzf);
    Cvt<T>::st(dgates, gb + 2 * n + j, dzo); Cvt<T>::st(dgates, gb + 3 * n + j, dzg);
    Cvt<T>::st(dc_prev, idx, dcp);
  }
  if (pf && dpeep) {
    atomicAdd(dpeep + j, apf);
    atomicAdd(dpeep + n + j, apo);
    atomicAdd(dpeep + 2 * n + j, apg);
  }
}

}  // namespace

DL4J_API int dl4j_lstm_cell_fwd(const void* gates, const void* c_prev, const float* pf, const float* po, const float* pg,
                                const float* mask, void* h, void* c, void* sv, int mb, int n, int dtype, hipStream_t st) {
  if (mb <= 0 || n <= 0) return 0;
  const int grid = grid_for((int64_t)mb * n, 256);
  if (dtype == 1)
    hipLaunchKernelGGL(lstm_cell_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)gates, (const bf16_t*)c_prev,
                       pf, po, pg, mask, (bf16_t*)h, (bf16_t*)c, (bf16_t*)sv, mb, n);
  else
    hipLaunchKernelGGL(lstm_cell_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)gates, (const float*)c_prev, pf,
                       po, pg, mask, (float*)h, (float*)c, (float*)sv, mb, n);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_lstm_cell_bwd(const void* dh, const void* dc_next, const void* c_prev, const void* sv, const float* pf,
                                const float* po, const float* pg, const float* mask, void* dgates, void* dc_prev, float* dpeep,
                                int mb, int n, int dtype, hipStream_t st) {
  if (mb <= 0 || n <= 0) return 0;
  const int bx = (n + 255) / 256;
  // enough row groups to fill the chip, but at most 64 atomics per unit
  int gy = (2 * DL4J_NUM_CU + bx - 1) / bx;
  if (gy > 64) gy = 64;
  if (gy > mb) gy = mb;
  if (gy < 1) gy = 1;
  const int rpb = (mb + gy - 1) / gy;
  gy = (mb + rpb - 1) / rpb;
  if (dtype == 1)
    hipLaunchKernelGGL(lstm_cell_bwd_kernel<bf16_t>, dim3(bx, gy), dim3(256), 0, st, (const bf16_t*)dh, (const bf16_t*)dc_next,
                       (const bf16_t*)c_prev, (const bf16_t*)sv, pf, po, pg, mask, (bf16_t*)dgates, (bf16_t*)dc_prev, dpeep, mb, n,
                       rpb);
  else
    hipLaunchKernelGGL(lstm_cell_bwd_kernel<float>, dim3(bx, gy), dim3(256), 0, st, (const float*)dh, (const float*)dc_next,
                       (const float*)c_prev, (const float*)sv, pf, po, pg, mask, (float*)dgates, (float*)dc_prev, dpeep, mb, n, rpb);
  DL4J_RET_LAST();
}
