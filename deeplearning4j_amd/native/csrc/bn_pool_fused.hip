// BatchNorm -> ReLU -> MaxPool as ONE forward pass and a two-pass backward that never materialises the full-resolution activation
// or its gradient (bf16, channels-last). This is the stem of the ImageNet networks (conv 7x7/2 -> BN -> ReLU -> maxpool 3x3/2): at
// batch 256 the tensor between BatchNorm and the pool is 411 MB, written once and read once forward, and its gradient is written by
// the pool backward and read twice by the BatchNorm backward.
//
//   forward:  y[n,oh,ow,c] = max(0, max_taps(x * scale[c] + shift[c])), idx = window-local tap of the maximum, 255 when no tap is
//             positive (the ReLU is closed for the whole window: no gradient flows).
//   backward: dz at a full-resolution position = sum of dy over the windows whose recorded maximum it is.
//     pass 1 (per POOLED element): sum_dz[c] += dy, sum_dzx[c] += dy * xhat where xhat = (y - beta) / gamma is recovered from the
//             pooled output itself (y = gamma * xhat + beta at the maximum), so the pass reads two pooled tensors and no
//             full-resolution one;
//     coefficients (one tiny launch): dgamma, dbeta, and k0 / k1 / gs with dx = gs * dz + x * k1 + k0
//             (= gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)));
//     pass 2 (per full-resolution element): gathers dz from the <= 4 covering windows (16-byte dy + 8-byte idx loads), reads x
//             once, writes dx once.
//
// Reference: the reference runs these as three layers (BatchNormalization.java, ActivationLayer, SubsamplingLayer.java -> libnd4j
// batchnorm / maxpool2d (+_bp) ops): three forward passes and four backward passes over the full-resolution tensor.
#include "common.h"

namespace {

struct BnPoolP {
  int N, H, W, C, OH, OW;
  int kH, kW, sH, sW, pH, pW;
};

__device__ __forceinline__ void ld8(const bf16_t* p, float (&o)[8]) {
  const s16x8 v = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = bf2f((bf16_t)v[j]);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&o)[8]) {
  s16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(o[j]);
  *reinterpret_cast<s16x8*>(p) = v;
}
__device__ __forceinline__ void ldf8(const float* p, float (&o)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
}

__global__ void __launch_bounds__(256) bn_relu_pool_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, bf16_t* __restrict__ y,
                                                               uint8_t* __restrict__ idx, BnPoolP p) {
  const unsigned cv = p.C / 8;
  const int64_t total = (int64_t)p.N * p.OH * p.OW * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    unsigned r = (unsigned)i;                       // host guarantees total < 2^32
    const int c0 = (int)(r % cv) * 8; r /= cv;
    const int ow = (int)(r % (unsigned)p.OW); r /= (unsigned)p.OW;
    const int oh = (int)(r % (unsigned)p.OH);
    const int n = (int)(r / (unsigned)p.OH);
    float sc[8], sh[8], best[8];
    ldf8(scale + c0, sc); ldf8(shift + c0, sh);
    unsigned arg_lo = 0xFFFFFFFFu, arg_hi = 0xFFFFFFFFu;       // eight tap bytes, 255 = none
#pragma unroll
    for (int j = 0; j < 8; ++j) best[j] = 0.f;
    for (int kh = 0; kh < p.kH; ++kh) {
      const int h = oh * p.sH - p.pH + kh;
      if (h < 0 || h >= p.H) continue;
      for (int kw = 0; kw < p.kW; ++kw) {
        const int w = ow * p.sW - p.pW + kw;
        if (w < 0 || w >= p.W) continue;
        float v[8];
        ld8(x + (((int64_t)n * p.H + h) * p.W + w) * p.C + c0, v);
        const unsigned code = (unsigned)(kh * p.kW + kw);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = v[j] * sc[j] + sh[j], b = v[4 + j] * sc[4 + j] + sh[4 + j];
          if (a > best[j]) { best[j] = a; arg_lo = (arg_lo & ~(0xFFu << (8 * j))) | (code << (8 * j)); }
          if (b > best[4 + j]) { best[4 + j] = b; arg_hi = (arg_hi & ~(0xFFu << (8 * j))) | (code << (8 * j)); }
        }
      }
    }
    const int64_t o = (((int64_t)n * p.OH + oh) * p.OW + ow) * p.C + c0;
    st8(y + o, best);
    if (idx != nullptr) *reinterpret_cast<uint2*>(idx + o) = make_uint2(arg_lo, arg_hi);
  }
}

// pass 1: rows = pooled positions. Block = 256 threads = (256 / cv) row lanes x cv channel vectors; per-block LDS reduction, then one
// atomicAdd per channel and block into sums[2][C] (zeroed by the caller). igamma[c] = 1 / gamma[c] (0 where gamma is 0).
__global__ void __launch_bounds__(256) bn_pool_bwd_reduce_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                                 const uint8_t* __restrict__ idx, const float* __restrict__ beta,
                                                                 const float* __restrict__ igamma, float* __restrict__ sums, int64_t rows,
                                                                 int C, int rows_per_block) {
  __shared__ float red[2 * 256 * 8 / 8 * 8];      // [2][256][8] floats = 16 KB
  const int cv = C / 8;
  const int lanes = 256 / cv;
  const int cvi = threadIdx.x % cv, rl = threadIdx.x / cv;
  const int c0 = cvi * 8;
  float bt[8], ig[8], s0[8], s1[8];
  ldf8(beta + c0, bt); ldf8(igamma + c0, ig);
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  if (rl < lanes) {
    for (int64_t r = r_begin + rl; r < r_end; r += lanes) {
      const int64_t o = r * C + c0;
      float d[8], v[8];
      ld8(dy + o, d); ld8(y + o, v);
      const uint2 am = *reinterpret_cast<const uint2*>(idx + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float da = ((am.x >> (8 * j)) & 0xFFu) != 0xFFu ? d[j] : 0.f;
        const float db = ((am.y >> (8 * j)) & 0xFFu) != 0xFFu ? d[4 + j] : 0.f;
        s0[j] += da; s1[j] += da * (v[j] - bt[j]) * ig[j];
        s0[4 + j] += db; s1[4 + j] += db * (v[4 + j] - bt[4 + j]) * ig[4 + j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[threadIdx.x * 8 + j] = s0[j]; red[2048 + threadIdx.x * 8 + j] = s1[j]; }
  __syncthreads();
  // thread t < 2 * C: which = t / C, channel = t % C; sums over the row lanes
  for (int t = threadIdx.x; t < 2 * C; t += 256) {
    const int which = t / C, ch = t - which * C;
    const int v_ = ch / 8, j = ch % 8;
    float acc = 0.f;
    for (int l = 0; l < lanes; ++l) acc += red[which * 2048 + (l * cv + v_) * 8 + j];
    atomicAdd(sums + t, acc);
  }
}

// coefficients: sums [2][C] -> dbeta, dgamma (may be null), gs / k1 / k0 [3][C] for pass 2
__global__ void __launch_bounds__(256) bn_pool_bwd_coeffs_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 float* __restrict__ dbeta, float* __restrict__ dgamma, float* __restrict__ coef,
                                                                 int C, float inv_m) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    const float sdz = sums[c], sdx = sums[C + c];
    if (dbeta) dbeta[c] = sdz;
    if (dgamma) dgamma[c] = sdx;
    const float g = gamma ? gamma[c] : 1.f, is = invstd[c], mu = mean[c];
    const float gs = g * is, a = sdz * inv_m, b = sdx * inv_m;
    coef[c] = gs;
    coef[C + c] = -gs * b * is;                       // k1: multiplies x
    coef[2 * C + c] = -gs * a + gs * b * mu * is;     // k0
  }
}

// pass 2: one thread per full-resolution (n, h, w, 8 channels)
// KS = kernel * 16 + stride when both windows are square and known at compile time (3x3 / 2: the ImageNet stem), 0 = runtime geometry:
// the covering-window arithmetic then divides by constants
template <int KS>
__global__ void __launch_bounds__(256) bn_pool_bwd_dx_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                             const uint8_t* __restrict__ idx, const float* __restrict__ coef,
                                                             bf16_t* __restrict__ dx, BnPoolP pr) {
  BnPoolP p = pr;
  if (KS != 0) { p.kH = p.kW = KS >> 4; p.sH = p.sW = KS & 15; }
  const unsigned cv = p.C / 8;
  const int64_t total = (int64_t)p.N * p.H * p.W * cv;
  const int spanH = p.kH - 1, spanW = p.kW - 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    unsigned r = (unsigned)i;
    const int c0 = (int)(r % cv) * 8; r /= cv;
    const int w = (int)(r % (unsigned)p.W); r /= (unsigned)p.W;
    const int h = (int)(r % (unsigned)p.H);
    const int n = (int)(r / (unsigned)p.H);
    float dz[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dz[j] = 0.f;
    const int hp = h + p.pH, wp = w + p.pW;
    int oh_hi = hp / p.sH; if (oh_hi > p.OH - 1) oh_hi = p.OH - 1;
    int ow_hi = wp / p.sW; if (ow_hi > p.OW - 1) ow_hi = p.OW - 1;
    int oh_lo = hp - spanH; oh_lo = oh_lo <= 0 ? 0 : (oh_lo + p.sH - 1) / p.sH;
    int ow_lo = wp - spanW; ow_lo = ow_lo <= 0 ? 0 : (ow_lo + p.sW - 1) / p.sW;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int kh = hp - oh * p.sH;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int kw = wp - ow * p.sW;
        const unsigned code = (unsigned)(kh * p.kW + kw);
        const int64_t o = (((int64_t)n * p.OH + oh) * p.OW + ow) * p.C + c0;
        float d[8];
        ld8(dy + o, d);
        const uint2 am = *reinterpret_cast<const uint2*>(idx + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dz[j] += ((am.x >> (8 * j)) & 0xFFu) == code ? d[j] : 0.f;
          dz[4 + j] += ((am.y >> (8 * j)) & 0xFFu) == code ? d[4 + j] : 0.f;
        }
      }
    }
    const int64_t xo = (((int64_t)n * p.H + h) * p.W + w) * p.C + c0;
    float xv[8], gs[8], k1[8], k0[8], out[8];
    ld8(x + xo, xv);
    ldf8(coef + c0, gs); ldf8(coef + p.C + c0, k1); ldf8(coef + 2 * p.C + c0, k0);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = gs[j] * dz[j] + xv[j] * k1[j] + k0[j];
    st8(dx + xo, out);
  }
}

bool geom_ok(const BnPoolP& p) {
  if (p.N <= 0 || p.H <= 0 || p.W <= 0 || p.C <= 0 || p.OH <= 0 || p.OW <= 0) return false;
  if (p.C % 8 != 0 || p.C > 256 || 256 % (p.C / 8) != 0) return false;         // 256 threads = row lanes x channel vectors
  if (p.kH < 1 || p.kW < 1 || p.kH * p.kW > 254 || p.sH < 1 || p.sW < 1 || p.pH < 0 || p.pW < 0) return false;
  if ((int64_t)p.N * p.H * p.W * (p.C / 8) >= 0xFFFFFFFFll) return false;
  // every window has at least its origin row / column inside the input or reaches it: OH / OW as the pooling layer computes them
  if ((p.OH - 1) * p.sH - p.pH >= p.H || (p.OW - 1) * p.sW - p.pW >= p.W) return false;
  return true;
}
BnPoolP mk(const int* g) { return BnPoolP{g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11]}; }
inline bool a16(const void* q) { return ((uintptr_t)q & 15) == 0; }

}  // namespace

// geom = [N, H, W, C, OH, OW, kH, kW, sH, sW, pH, pW]
DL4J_API int dl4j_bn_pool_supported(const int* geom) { return geom_ok(mk(geom)) ? 1 : 0; }

// x [N,H,W,C] bf16, scale / shift [C] fp32 -> y [N,OH,OW,C] bf16, idx [N,OH,OW,C] bytes (null for inference)
DL4J_API int dl4j_bn_relu_pool_fwd(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx, const int* geom, hipStream_t st) {
  const BnPoolP p = mk(geom);
  if (!geom_ok(p) || !x || !scale || !shift || !y || !a16(x) || !a16(y) || !a16(scale) || !a16(shift) || (idx && ((uintptr_t)idx & 7))) return -2;
  const int64_t total = (int64_t)p.N * p.OH * p.OW * (p.C / 8);
  hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16_t*)x, scale, shift, (bf16_t*)y, idx, p);
  DL4J_RET_LAST();
}

// dy, y [N,OH,OW,C] bf16, idx bytes, x [N,H,W,C] bf16; gamma (null = ones), beta (null = zeros), mean, invstd [C] fp32; ws: 7*C floats
// of 16-byte aligned scratch; dbeta / dgamma [C] fp32 or null -> dx [N,H,W,C] bf16.
// ws layout: [0, 2C) sums, [2C, 3C) 1 / gamma, [3C, 6C) coefficients of pass 2, [6C, 7C) beta.
DL4J_API int dl4j_bn_pool_bwd(const void* x, const void* dy, const void* y, const uint8_t* idx, const float* gamma, const float* beta,
                              const float* mean, const float* invstd, float* ws, float* dbeta, float* dgamma, void* dx, const int* geom,
                              hipStream_t st);

namespace {
__global__ void __launch_bounds__(256) bn_pool_prep_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ ws,
                                                           int C) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    ws[c] = 0.f; ws[C + c] = 0.f;
    const float g = gamma ? gamma[c] : 1.f;
    ws[2 * C + c] = g != 0.f ? 1.f / g : 0.f;
    ws[6 * C + c] = beta ? beta[c] : 0.f;          // an aligned copy: the parameter views sit at arbitrary offsets of the flat vector
  }
}
}  // namespace

// passes 1 + coefficients only (dgamma / dbeta and ws[3C, 6C) = gs / k1 / k0): pass 2 is then either dl4j_bn_pool_bwd_dx or runs inside
// the consumer of dx (dl4j_conv_c3_wgrad_bnpool recomputes dx in its loader and the 411 MB tensor is never written)
DL4J_API int dl4j_bn_pool_bwd_reduce(const void* dy, const void* y, const uint8_t* idx, const float* gamma, const float* beta,
                                     const float* mean, const float* invstd, float* ws, float* dbeta, float* dgamma, const int* geom,
                                     hipStream_t st) {
  const BnPoolP p = mk(geom);
  if (!geom_ok(p) || !dy || !y || !idx || !mean || !invstd || !ws) return -2;
  if (!a16(dy) || !a16(y) || !a16(ws) || ((uintptr_t)idx & 7)) return -2;
  const int C = p.C;
  hipLaunchKernelGGL(bn_pool_prep_kernel, dim3(1), dim3(256), 0, st, gamma, beta, ws, C);
  const int64_t rows = (int64_t)p.N * p.OH * p.OW;
  int rpb = (int)((rows + 1023) / 1024);
  if (rpb < 64) rpb = 64;
  const int nblk = (int)((rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, idx, ws + 6 * C, ws + 2 * C, ws, rows, C,
                     rpb);
  const float inv_m = 1.f / (float)((int64_t)p.N * p.H * p.W);
  hipLaunchKernelGGL(bn_pool_bwd_coeffs_kernel, dim3(1), dim3(256), 0, st, ws, gamma, mean, invstd, dbeta, dgamma, ws + 3 * C, C, inv_m);
  DL4J_RET_LAST();
}

// pass 2 on its own (ws as left by dl4j_bn_pool_bwd_reduce)
DL4J_API int dl4j_bn_pool_bwd_dx(const void* x, const void* dy, const uint8_t* idx, const float* ws, void* dx, const int* geom, hipStream_t st) {
  const BnPoolP p = mk(geom);
  if (!geom_ok(p) || !x || !dy || !idx || !ws || !dx) return -2;
  if (!a16(x) || !a16(dy) || !a16(dx) || !a16(ws) || ((uintptr_t)idx & 7)) return -2;
  const int C = p.C;
  const int64_t total = (int64_t)p.N * p.H * p.W * (C / 8);
  if (p.kH == 3 && p.kW == 3 && p.sH == 2 && p.sW == 2)
    hipLaunchKernelGGL(bn_pool_bwd_dx_kernel<3 * 16 + 2>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, idx,
                       ws + 3 * C, (bf16_t*)dx, p);
  else
    hipLaunchKernelGGL(bn_pool_bwd_dx_kernel<0>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, idx, ws + 3 * C,
                       (bf16_t*)dx, p);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_bn_pool_bwd(const void* x, const void* dy, const void* y, const uint8_t* idx, const float* gamma, const float* beta,
                              const float* mean, const float* invstd, float* ws, float* dbeta, float* dgamma, void* dx, const int* geom,
                              hipStream_t st) {
  const BnPoolP p = mk(geom);
  if (!geom_ok(p) || !x || !dy || !y || !idx || !mean || !invstd || !ws || !dx) return -2;
  if (!a16(x) || !a16(dy) || !a16(y) || !a16(dx) || !a16(ws) || ((uintptr_t)idx & 7)) return -2;
  const int C = p.C;
  hipLaunchKernelGGL(bn_pool_prep_kernel, dim3(1), dim3(256), 0, st, gamma, beta, ws, C);
  const int64_t rows = (int64_t)p.N * p.OH * p.OW;
  int rpb = (int)((rows + 1023) / 1024);
  if (rpb < 64) rpb = 64;
  const int nblk = (int)((rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, idx, ws + 6 * C, ws + 2 * C, ws, rows, C,
                     rpb);
  const float inv_m = 1.f / (float)((int64_t)p.N * p.H * p.W);
  hipLaunchKernelGGL(bn_pool_bwd_coeffs_kernel, dim3(1), dim3(256), 0, st, ws, gamma, mean, invstd, dbeta, dgamma, ws + 3 * C, C, inv_m);
  const int64_t total = (int64_t)p.N * p.H * p.W * (C / 8);
  if (p.kH == 3 && p.kW == 3 && p.sH == 2 && p.sW == 2)
    hipLaunchKernelGGL(bn_pool_bwd_dx_kernel<3 * 16 + 2>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, idx,
                       ws + 3 * C, (bf16_t*)dx, p);
  else
    hipLaunchKernelGGL(bn_pool_bwd_dx_kernel<0>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, idx, ws + 3 * C,
                       (bf16_t*)dx, p);
  DL4J_RET_LAST();
}
