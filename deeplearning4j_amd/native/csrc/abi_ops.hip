// Generic kernels behind the native op-execution ABI (rt_opctx.cpp: execCustomOp2 on an OpaqueContext).
//
// A foreign binding hands the runtime arrays of ANY layout and shape (NCHW fp32 LeNet tensors, a 1-channel 5x5 stem, a
// [K,K,iC,oC] filter): the MFMA kernels of this library take their own layouts (NHWC bf16, 16-byte channel chunks) and refuse the
// rest. These kernels are the layer underneath: strided fp32, one work item per output element, written ONCE as
// __host__ __device__ element functions and run either as a grid-stride kernel on the stream of the context or as a host loop
// over the same element function - so the ABI computes the same numbers for host buffers and device ("special") buffers.
//
// Every entry point takes `dev` (1: pointers are device memory, launch on `st`; 0: host memory, run here) and returns 0 or an
// error code. Nothing here allocates except dl4j_abi_alloc / dl4j_abi_free, which the executors use for short-lived workspaces.
//
// Reference behaviour: libnd4j declarable ops (ops/declarable/generic/nn/convo/conv2d.cpp, pooling/maxpool2d.cpp, avgpool2d.cpp,
// nn/batchnorm.cpp, loss/softmaxCrossEntropy.cpp, updaters/*.cpp, blas/matmul.cpp, transforms/gather.cpp, scatter_add.cpp).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define HDL __host__ __device__

namespace {

template <class F>
__global__ void __launch_bounds__(256) abi_kernel(F f, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) f(i);
}

template <class F>
int abi_run(const F& f, int64_t n, int dev, hipStream_t st, bool serial = false) {
  if (n <= 0) return 0;
  if (dev) {
    int64_t nb = (n + 255) / 256;
    if (nb > 262144) nb = 262144;
    hipLaunchKernelGGL(abi_kernel<F>, dim3((unsigned)nb), dim3(256), 0, st, f, n);
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }
  unsigned nt = serial || n < 2048 ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (nt <= 1) {
    for (int64_t i = 0; i < n; ++i) f(i);
    return 0;
  }
  std::vector<std::thread> th;
  const int64_t per = (n + nt - 1) / nt;
  for (unsigned k = 0; k < nt; ++k) {
    const int64_t lo = k * per, hi = std::min(n, lo + per);
    if (lo >= hi) break;
    th.emplace_back([&f, lo, hi] { for (int64_t i = lo; i < hi; ++i) f(i); });
  }
  for (auto& t : th) t.join();
  return 0;
}

// ---- convolution ---------------------------------------------------------------------------------------------------------
struct ConvG {
  int64_t N, C, H, W, K, R, S, OH, OW, sH, sW, pH, pW, dH, dW;
  int64_t xn, xc, xh, xw, wk, wc, wr, ws, yn, yk, yh, yw;
};

// ---- pooling -------------------------------------------------------------------------------------------------------------
struct PoolG {
  int64_t N, C, H, W, OH, OW, kH, kW, sH, sW, pH, pW, dH, dW, kind, xn, xc, xh, xw, yn, yc, yh, yw;
};   // kind 0 max, 1 avg dividing by the full window, 2 avg dividing by the in-bounds count

// ---- reductions ----------------------------------------------------------------------------------------------------------
struct RedG {
  int64_t rank, kind, nout, nred;
  int64_t shape[8], stride[8], red[8];
};   // kind 0 sum 1 mean 2 max 3 min 4 sum of squares 5 sum |x| 6 sqrt(sum of squares) 7 product 8 argmax 9 argmin

struct CopyG {
  int64_t rank, n, sdt, ddt;
  int64_t shape[8], ss[8], ds[8];
};   // dtype codes: 0 f32 1 bf16 2 f64 3 i32 4 i64 5 f16 6 bool / u8

HDL inline float bf16_to_f(uint16_t b) {
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)b << 16;
  return c.f;
}
HDL inline uint16_t f_to_bf16(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((c.u >> 16) | 0x40);
  return (uint16_t)((c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16);
}

HDL inline double load_as_double(const void* p, int64_t off, int64_t dt) {
  switch (dt) {
    case 0: return (double)((const float*)p)[off];
    case 1: return (double)bf16_to_f(((const uint16_t*)p)[off]);
    case 2: return ((const double*)p)[off];
    case 3: return (double)((const int32_t*)p)[off];
    case 4: return (double)((const int64_t*)p)[off];
    case 5: return (double)(float)((const _Float16*)p)[off];
    default: return ((const uint8_t*)p)[off] ? 1.0 : 0.0;
  }
}
HDL inline int64_t load_as_long(const void* p, int64_t off, int64_t dt) {
  switch (dt) {
    case 3: return (int64_t)((const int32_t*)p)[off];
    case 4: return ((const int64_t*)p)[off];
    case 6: return ((const uint8_t*)p)[off] ? 1 : 0;
    default: return (int64_t)load_as_double(p, off, dt);
  }
}
HDL inline void store_from(void* p, int64_t off, int64_t dt, double v, int64_t iv, bool integral) {
  switch (dt) {
    case 0: ((float*)p)[off] = integral ? (float)iv : (float)v; break;
    case 1: ((uint16_t*)p)[off] = f_to_bf16(integral ? (float)iv : (float)v); break;
    case 2: ((double*)p)[off] = integral ? (double)iv : v; break;
    case 3: ((int32_t*)p)[off] = integral ? (int32_t)iv : (int32_t)v; break;
    case 4: ((int64_t*)p)[off] = integral ? iv : (int64_t)v; break;
    case 5: ((_Float16*)p)[off] = (_Float16)(integral ? (float)iv : (float)v); break;
    default: ((uint8_t*)p)[off] = integral ? (iv != 0) : (v != 0.0); break;
  }
}

HDL inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

}  // namespace

// =========================================================================================================== memory helpers
DL4J_API void* dl4j_abi_alloc(int64_t bytes, int dev) {
  if (bytes <= 0) bytes = 16;
  if (!dev) return std::malloc((size_t)bytes);
  void* p = nullptr;
  return hipMalloc(&p, (size_t)bytes) == hipSuccess ? p : nullptr;
}
DL4J_API void dl4j_abi_free(void* p, int dev, hipStream_t st) {
  if (!p) return;
  if (!dev) { std::free(p); return; }
  hipStreamSynchronize(st);
  hipFree(p);
}
DL4J_API int dl4j_abi_memcpy(void* dst, const void* src, int64_t bytes, int dev, hipStream_t st) {
  if (bytes <= 0) return 0;
  if (!dev) { std::memmove(dst, src, (size_t)bytes); return 0; }
  return hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, st) == hipSuccess ? 0 : 1;
}
DL4J_API int dl4j_abi_fill(float* p, float v, int64_t n, int dev, hipStream_t st) {
  return abi_run([=] HDL(int64_t i) { p[i] = v; }, n, dev, st);
}

// =========================================================================================================== convolution
// g: the 27 entries of ConvG. x / w / y are addressed through element strides, so NCHW / NHWC data and the three filter
// formats ([kH,kW,iC,oC], [oC,iC,kH,kW], [oC,kH,kW,iC]) are one code path.
DL4J_API int dl4j_abi_conv2d(const float* x, const float* w, const float* b, float* y, const int64_t* gp, int dev, hipStream_t st) {
  ConvG g;
  std::memcpy(&g, gp, sizeof(g));
  return abi_run([=] HDL(int64_t i) {
    const int64_t ow = i % g.OW; int64_t t = i / g.OW;
    const int64_t oh = t % g.OH; t /= g.OH;
    const int64_t k = t % g.K, n = t / g.K;
    float acc = b ? b[k] : 0.f;
    for (int64_t r = 0; r < g.R; ++r) {
      const int64_t ih = oh * g.sH - g.pH + r * g.dH;
      if (ih < 0 || ih >= g.H) continue;
      for (int64_t s = 0; s < g.S; ++s) {
        const int64_t iw = ow * g.sW - g.pW + s * g.dW;
        if (iw < 0 || iw >= g.W) continue;
        const float* xp = x + n * g.xn + ih * g.xh + iw * g.xw;
        const float* wp = w + k * g.wk + r * g.wr + s * g.ws;
        for (int64_t c = 0; c < g.C; ++c) acc = fmaf(xp[c * g.xc], wp[c * g.wc], acc);
      }
    }
    y[n * g.yn + k * g.yk + oh * g.yh + ow * g.yw] = acc;
  }, g.N * g.K * g.OH * g.OW, dev, st);
}

// gx[n,c,h,w] = sum_{k,r,s} gy[n,k,oh,ow] * w[k,c,r,s] over the (oh, ow) whose window tap (r, s) lands on (h, w)
DL4J_API int dl4j_abi_conv2d_dgrad(const float* gy, const float* w, float* gx, const int64_t* gp, int dev, hipStream_t st) {
  ConvG g;
  std::memcpy(&g, gp, sizeof(g));
  return abi_run([=] HDL(int64_t i) {
    const int64_t iw = i % g.W; int64_t t = i / g.W;
    const int64_t ih = t % g.H; t /= g.H;
    const int64_t c = t % g.C, n = t / g.C;
    float acc = 0.f;
    for (int64_t r = 0; r < g.R; ++r) {
      const int64_t th = ih + g.pH - r * g.dH;
      if (th < 0 || th % g.sH) continue;
      const int64_t oh = th / g.sH;
      if (oh >= g.OH) continue;
      for (int64_t s = 0; s < g.S; ++s) {
        const int64_t tw = iw + g.pW - s * g.dW;
        if (tw < 0 || tw % g.sW) continue;
        const int64_t ow = tw / g.sW;
        if (ow >= g.OW) continue;
        const float* gp2 = gy + n * g.yn + oh * g.yh + ow * g.yw;
        const float* wp = w + c * g.wc + r * g.wr + s * g.ws;
        for (int64_t k = 0; k < g.K; ++k) acc = fmaf(gp2[k * g.yk], wp[k * g.wk], acc);
      }
    }
    gx[n * g.xn + c * g.xc + ih * g.xh + iw * g.xw] = acc;
  }, g.N * g.C * g.H * g.W, dev, st);
}

// gw[k,c,r,s] = sum_{n,oh,ow} gy[n,k,oh,ow] * x[n,c,ih,iw]
DL4J_API int dl4j_abi_conv2d_wgrad(const float* x, const float* gy, float* gw, const int64_t* gp, int dev, hipStream_t st) {
  ConvG g;
  std::memcpy(&g, gp, sizeof(g));
  return abi_run([=] HDL(int64_t i) {
    const int64_t s = i % g.S; int64_t t = i / g.S;
    const int64_t r = t % g.R; t /= g.R;
    const int64_t c = t % g.C, k = t / g.C;
    double acc = 0.0;
    for (int64_t n = 0; n < g.N; ++n)
      for (int64_t oh = 0; oh < g.OH; ++oh) {
        const int64_t ih = oh * g.sH - g.pH + r * g.dH;
        if (ih < 0 || ih >= g.H) continue;
        float part = 0.f;
        for (int64_t ow = 0; ow < g.OW; ++ow) {
          const int64_t iw = ow * g.sW - g.pW + s * g.dW;
          if (iw < 0 || iw >= g.W) continue;
          part = fmaf(gy[n * g.yn + k * g.yk + oh * g.yh + ow * g.yw], x[n * g.xn + c * g.xc + ih * g.xh + iw * g.xw], part);
        }
        acc += (double)part;
      }
    gw[k * g.wk + c * g.wc + r * g.wr + s * g.ws] = (float)acc;
  }, g.K * g.C * g.R * g.S, dev, st);
}

// =========================================================================================================== pooling
DL4J_API int dl4j_abi_pool2d(const float* x, float* y, const int64_t* gp, int dev, hipStream_t st) {
  PoolG g;
  std::memcpy(&g, gp, sizeof(g));
  return abi_run([=] HDL(int64_t i) {
    const int64_t ow = i % g.OW; int64_t t = i / g.OW;
    const int64_t oh = t % g.OH; t /= g.OH;
    const int64_t c = t % g.C, n = t / g.C;
    float best = -INFINITY, sum = 0.f;
    int64_t cnt = 0;
    for (int64_t r = 0; r < g.kH; ++r) {
      const int64_t ih = oh * g.sH - g.pH + r * g.dH;
      if (ih < 0 || ih >= g.H) continue;
      for (int64_t s = 0; s < g.kW; ++s) {
        const int64_t iw = ow * g.sW - g.pW + s * g.dW;
        if (iw < 0 || iw >= g.W) continue;
        const float v = x[n * g.xn + c * g.xc + ih * g.xh + iw * g.xw];
        best = v > best ? v : best;
        sum += v;
        ++cnt;
      }
    }
    float out = best;
    if (g.kind == 1) out = sum / (float)(g.kH * g.kW);
    else if (g.kind == 2) out = cnt ? sum / (float)cnt : 0.f;
    y[n * g.yn + c * g.yc + oh * g.yh + ow * g.yw] = out;
  }, g.N * g.C * g.OH * g.OW, dev, st);
}

// one work item per INPUT element: it visits the windows that contain it, so no atomics and no zero fill are needed
DL4J_API int dl4j_abi_pool2d_bwd(const float* x, const float* gy, float* gx, const int64_t* gp, int dev, hipStream_t st) {
  PoolG g;
  std::memcpy(&g, gp, sizeof(g));
  return abi_run([=] HDL(int64_t i) {
    const int64_t iw = i % g.W; int64_t t = i / g.W;
    const int64_t ih = t % g.H; t /= g.H;
    const int64_t c = t % g.C, n = t / g.C;
    const float* xb = x + n * g.xn + c * g.xc;
    float acc = 0.f;
    for (int64_t r = 0; r < g.kH; ++r) {
      const int64_t th = ih + g.pH - r * g.dH;
      if (th < 0 || th % g.sH) continue;
      const int64_t oh = th / g.sH;
      if (oh >= g.OH) continue;
      for (int64_t s = 0; s < g.kW; ++s) {
        const int64_t tw = iw + g.pW - s * g.dW;
        if (tw < 0 || tw % g.sW) continue;
        const int64_t ow = tw / g.sW;
        if (ow >= g.OW) continue;
        const float go = gy[n * g.yn + c * g.yc + oh * g.yh + ow * g.yw];
        if (g.kind == 1) { acc += go / (float)(g.kH * g.kW); continue; }
        // window (oh, ow): the first maximum in scan order takes the gradient / the in-bounds count divides it
        float best = -INFINITY;
        int64_t bh = -1, bw = -1, cnt = 0;
        for (int64_t r2 = 0; r2 < g.kH; ++r2) {
          const int64_t h2 = oh * g.sH - g.pH + r2 * g.dH;
          if (h2 < 0 || h2 >= g.H) continue;
          for (int64_t s2 = 0; s2 < g.kW; ++s2) {
            const int64_t w2 = ow * g.sW - g.pW + s2 * g.dW;
            if (w2 < 0 || w2 >= g.W) continue;
            const float v = xb[h2 * g.xh + w2 * g.xw];
            if (v > best) { best = v; bh = h2; bw = w2; }
            ++cnt;
          }
        }
        if (g.kind == 0) { if (bh == ih && bw == iw) acc += go; }
        else acc += go / (float)(cnt ? cnt : 1);
      }
    }
    gx[n * g.xn + c * g.xc + ih * g.xh + iw * g.xw] = acc;
  }, g.N * g.C * g.H * g.W, dev, st);
}

// =========================================================================================================== reductions
// gp: RedG (4 + 24 entries). Output element o enumerates the kept dimensions in row-major order; `out` is fp32, or int64 for the
// index reductions (kind 8 / 9).
DL4J_API int dl4j_abi_reduce(const float* x, void* out, const int64_t* gp, int dev, hipStream_t st) {
  RedG g;
  std::memcpy(&g, gp, sizeof(g));
  return abi_run([=] HDL(int64_t o) {
    int64_t base = 0, r = o;
    for (int64_t k = g.rank - 1; k >= 0; --k) {
      if (g.red[k]) continue;
      base += (r % g.shape[k]) * g.stride[k];
      r /= g.shape[k];
    }
    double acc = g.kind == 7 ? 1.0 : 0.0;
    float best = 0.f;
    int64_t bi = 0;
    for (int64_t j = 0; j < g.nred; ++j) {
      int64_t off = base, q = j;
      for (int64_t k = g.rank - 1; k >= 0; --k) {
        if (!g.red[k]) continue;
        off += (q % g.shape[k]) * g.stride[k];
        q /= g.shape[k];
      }
      const float v = x[off];
      switch (g.kind) {
        case 0: case 1: acc += (double)v; break;
        case 2: case 8: if (j == 0 || v > best) { best = v; bi = j; } break;
        case 3: case 9: if (j == 0 || v < best) { best = v; bi = j; } break;
        case 4: case 6: acc += (double)v * (double)v; break;
        case 5: acc += fabs((double)v); break;
        default: acc *= (double)v; break;
      }
    }
    if (g.kind == 8 || g.kind == 9) { ((int64_t*)out)[o] = bi; return; }
    float res;
    if (g.kind == 1) res = (float)(acc / (double)(g.nred ? g.nred : 1));
    else if (g.kind == 2 || g.kind == 3) res = best;
    else if (g.kind == 6) res = (float)sqrt(acc);
    else res = (float)acc;
    ((float*)out)[o] = res;
  }, g.nout, dev, st);
}

// =========================================================================================================== batch norm
// x viewed as [outer, C, inner]; mean / var / gamma / beta are [C] (gamma / beta may be null)
DL4J_API int dl4j_abi_batchnorm(const float* x, const float* mean, const float* var, const float* gamma, const float* beta, float* y,
                                int64_t n, int64_t C, int64_t inner, float eps, int dev, hipStream_t st) {
  return abi_run([=] HDL(int64_t i) {
    const int64_t c = (i / inner) % C;
    float v = (x[i] - mean[c]) / sqrtf(var[c] + eps);
    if (gamma) v *= gamma[c];
    if (beta) v += beta[c];
    y[i] = v;
  }, n, dev, st);
}

// gradients of y = (x - mean) * rsqrt(var + eps) * gamma + beta with respect to EVERY input (mean and var are inputs of the op,
// not functions of x - the declarable op's contract): dx elementwise, the four [C] vectors one work item per channel.
DL4J_API int dl4j_abi_batchnorm_bp(const float* x, const float* mean, const float* var, const float* gamma, const float* gy, float* gx,
                                   float* gmean, float* gvar, float* ggamma, float* gbeta, int64_t outer, int64_t C, int64_t inner,
                                   float eps, int dev, hipStream_t st) {
  int rc = abi_run([=] HDL(int64_t i) {
    const int64_t c = (i / inner) % C;
    gx[i] = gy[i] * (gamma ? gamma[c] : 1.f) / sqrtf(var[c] + eps);
  }, outer * C * inner, dev, st);
  if (rc) return rc;
  return abi_run([=] HDL(int64_t c) {
    const double m = mean[c], ve = (double)var[c] + (double)eps, inv = 1.0 / sqrt(ve), gm = gamma ? (double)gamma[c] : 1.0;
    double s_g = 0.0, s_gx = 0.0;
    for (int64_t o = 0; o < outer; ++o) {
      const float* xp = x + (o * C + c) * inner;
      const float* gp2 = gy + (o * C + c) * inner;
      for (int64_t k = 0; k < inner; ++k) { s_g += (double)gp2[k]; s_gx += (double)gp2[k] * ((double)xp[k] - m); }
    }
    if (gmean) gmean[c] = (float)(-s_g * gm * inv);
    if (gvar) gvar[c] = (float)(-0.5 * s_gx * gm * inv / ve);
    if (ggamma) ggamma[c] = (float)(s_gx * inv);
    if (gbeta) gbeta[c] = (float)s_g;
  }, C, dev, st);
}

// =========================================================================================================== layer norm
// rows of length D normalised over the last dimension, then * gain + bias ([D]; bias may be null)
DL4J_API int dl4j_abi_layer_norm(const float* x, const float* gain, const float* bias, float* y, int64_t M, int64_t D, float eps, int dev,
                                 hipStream_t st) {
  return abi_run([=] HDL(int64_t m) {
    const float* xp = x + m * D;
    double s = 0.0, sq = 0.0;
    for (int64_t k = 0; k < D; ++k) s += (double)xp[k];
    const double mu = s / (double)D;
    for (int64_t k = 0; k < D; ++k) { const double dlt = (double)xp[k] - mu; sq += dlt * dlt; }
    const float inv = (float)(1.0 / sqrt(sq / (double)D + (double)eps));
    for (int64_t k = 0; k < D; ++k) {
      float v = (xp[k] - (float)mu) * inv * gain[k];
      if (bias) v += bias[k];
      y[m * D + k] = v;
    }
  }, M, dev, st);
}

DL4J_API int dl4j_abi_layer_norm_bp(const float* x, const float* gain, const float* gy, float* gx, float* ggain, float* gbias, int64_t M,
                                    int64_t D, float eps, int dev, hipStream_t st) {
  int rc = abi_run([=] HDL(int64_t m) {
    const float* xp = x + m * D;
    const float* gp2 = gy + m * D;
    double s = 0.0, sq = 0.0;
    for (int64_t k = 0; k < D; ++k) s += (double)xp[k];
    const double mu = s / (double)D;
    for (int64_t k = 0; k < D; ++k) { const double dlt = (double)xp[k] - mu; sq += dlt * dlt; }
    const double inv = 1.0 / sqrt(sq / (double)D + (double)eps);
    double a = 0.0, b = 0.0;      // mean of dxhat, mean of dxhat * xhat
    for (int64_t k = 0; k < D; ++k) {
      const double dxh = (double)gp2[k] * (double)gain[k], xh = ((double)xp[k] - mu) * inv;
      a += dxh; b += dxh * xh;
    }
    a /= (double)D; b /= (double)D;
    for (int64_t k = 0; k < D; ++k) {
      const double dxh = (double)gp2[k] * (double)gain[k], xh = ((double)xp[k] - mu) * inv;
      gx[m * D + k] = (float)(inv * (dxh - a - xh * b));
    }
  }, M, dev, st);
  if (rc) return rc;
  return abi_run([=] HDL(int64_t k) {
    double sg = 0.0, sb = 0.0;
    for (int64_t m = 0; m < M; ++m) {
      const float* xp = x + m * D;
      double s = 0.0, sq = 0.0;
      for (int64_t j = 0; j < D; ++j) s += (double)xp[j];
      const double mu = s / (double)D;
      for (int64_t j = 0; j < D; ++j) { const double dlt = (double)xp[j] - mu; sq += dlt * dlt; }
      const double inv = 1.0 / sqrt(sq / (double)D + (double)eps);
      sg += (double)gy[m * D + k] * ((double)xp[k] - mu) * inv;
      sb += (double)gy[m * D + k];
    }
    if (ggain) ggain[k] = (float)sg;
    if (gbias) gbias[k] = (float)sb;
  }, D, dev, st);
}

// =========================================================================================================== softmax cross entropy
// logits / labels [M, C]; weights: nw == 1 (scalar) or nw == M. mode: 0 none, 1 sum, 2 mean by weight, 3 mean by non-zero weights.
// per (workspace, [M]) receives the unweighted row losses; out is [M] for mode 0 and [1] otherwise.
DL4J_API int dl4j_abi_softmax_xent(const float* z, const float* w, int64_t nw, const float* lab, float* per, float* out, int64_t M,
                                   int64_t C, int mode, float smooth, int dev, hipStream_t st) {
  int rc = abi_run([=] HDL(int64_t m) {
    const float* zp = z + m * C;
    const float* lp = lab + m * C;
    float mx = -INFINITY;
    for (int64_t k = 0; k < C; ++k) mx = zp[k] > mx ? zp[k] : mx;
    double s = 0.0;
    for (int64_t k = 0; k < C; ++k) s += exp((double)(zp[k] - mx));
    const double lse = (double)mx + log(s);
    double loss = 0.0;
    for (int64_t k = 0; k < C; ++k) {
      const double l = smooth != 0.f ? (double)lp[k] * (1.0 - (double)smooth) + (double)smooth / (double)C : (double)lp[k];
      loss -= l * ((double)zp[k] - lse);
    }
    per[m] = (float)loss;
    if (mode == 0) out[m] = (float)(loss * (double)(nw == 1 ? w[0] : w[m]));
  }, M, dev, st);
  if (rc || mode == 0) return rc;
  return abi_run([=] HDL(int64_t) {
    double s = 0.0, ws = 0.0, nz = 0.0;
    for (int64_t m = 0; m < M; ++m) {
      const double wm = nw == 1 ? w[0] : w[m];
      s += (double)per[m] * wm; ws += wm; nz += wm != 0.0 ? 1.0 : 0.0;
    }
    out[0] = (float)(mode == 1 ? s : mode == 2 ? (ws == 0.0 ? 0.0 : s / ws) : (nz == 0.0 ? 0.0 : s / nz));
  }, 1, dev, st);
}

// gradients with respect to logits [M,C], weights [nw] and labels [M,C]. tmp: workspace of 4 floats + [M] row losses.
DL4J_API int dl4j_abi_softmax_xent_grad(const float* z, const float* w, int64_t nw, const float* lab, float* tmp, float* gz, float* gw,
                                        float* gl, int64_t M, int64_t C, int mode, float smooth, int dev, hipStream_t st) {
  float* per = tmp + 4;
  int rc = abi_run([=] HDL(int64_t m) {
    const float* zp = z + m * C;
    const float* lp = lab + m * C;
    float mx = -INFINITY;
    for (int64_t k = 0; k < C; ++k) mx = zp[k] > mx ? zp[k] : mx;
    double s = 0.0;
    for (int64_t k = 0; k < C; ++k) s += exp((double)(zp[k] - mx));
    const double lse = (double)mx + log(s);
    double loss = 0.0;
    for (int64_t k = 0; k < C; ++k) {
      const double l = smooth != 0.f ? (double)lp[k] * (1.0 - (double)smooth) + (double)smooth / (double)C : (double)lp[k];
      loss -= l * ((double)zp[k] - lse);
    }
    per[m] = (float)loss;
  }, M, dev, st);
  if (rc) return rc;
  rc = abi_run([=] HDL(int64_t) {
    double s = 0.0, ws = 0.0, nz = 0.0;
    for (int64_t m = 0; m < M; ++m) {
      const double wm = nw == 1 ? w[0] : w[m];
      s += (double)per[m] * wm; ws += wm; nz += wm != 0.0 ? 1.0 : 0.0;
    }
    tmp[0] = (float)s; tmp[1] = (float)ws; tmp[2] = (float)nz;
    double us = 0.0;
    for (int64_t m = 0; m < M; ++m) us += (double)per[m];
    tmp[3] = (float)us;
  }, 1, dev, st);
  if (rc) return rc;
  rc = abi_run([=] HDL(int64_t m) {
    const double ws = tmp[1], nz = tmp[2];
    const double scale = mode <= 1 ? 1.0 : mode == 2 ? (ws == 0.0 ? 0.0 : 1.0 / ws) : (nz == 0.0 ? 0.0 : 1.0 / nz);
    const double coef = (double)(nw == 1 ? w[0] : w[m]) * scale;
    const float* zp = z + m * C;
    const float* lp = lab + m * C;
    float mx = -INFINITY;
    for (int64_t k = 0; k < C; ++k) mx = zp[k] > mx ? zp[k] : mx;
    double s = 0.0, lsum = 0.0;
    for (int64_t k = 0; k < C; ++k) {
      s += exp((double)(zp[k] - mx));
      lsum += smooth != 0.f ? (double)lp[k] * (1.0 - (double)smooth) + (double)smooth / (double)C : (double)lp[k];
    }
    const double lse = (double)mx + log(s);
    for (int64_t k = 0; k < C; ++k) {
      const double l = smooth != 0.f ? (double)lp[k] * (1.0 - (double)smooth) + (double)smooth / (double)C : (double)lp[k];
      const double p = exp((double)zp[k] - lse);
      gz[m * C + k] = (float)(coef * (p * lsum - l));
      if (gl) gl[m * C + k] = (float)(-coef * (1.0 - (double)smooth) * ((double)zp[k] - lse));
    }
    if (gw && nw != 1) {
      // d/dw_m of sum(per * w) * scale: mode 2 also differentiates the 1 / sum(w) normaliser
      double v = (double)per[m] * scale;
      if (mode == 2 && ws != 0.0) v -= (double)tmp[0] / (ws * ws);
      gw[m] = (float)v;
    }
  }, M, dev, st);
  if (rc || !gw || nw != 1) return rc;
  return abi_run([=] HDL(int64_t) {
    // scalar weight: sum(per) * w / (M w) is constant in w for mode 2; modes 0 / 1 give sum(per); mode 3 sum(per) / count
    const double us = tmp[3], nz = tmp[2];
    gw[0] = (float)(mode <= 1 ? us : mode == 2 ? 0.0 : (nz == 0.0 ? 0.0 : us / nz));
  }, 1, dev, st);
}

// =========================================================================================================== updaters
// kind: ops/updater.py KIND (1 Sgd 2 Nesterovs 3 Adam 4 RmsProp 5 AdaGrad 6 AdaDelta 7 AdaMax 8 Nadam 9 AMSGrad 10 AdaBelief).
// s0 / s1 / s2 are the state inputs in the kernel's own order (m, v, h); o0 / o1 / o2 the matching outputs; t = iteration + 1.
DL4J_API int dl4j_abi_updater(int kind, const float* grad, const float* s0, const float* s1, const float* s2, float* upd, float* o0,
                              float* o1, float* o2, int64_t n, double lr, double a, double b, double eps_in, double t, int dev,
                              hipStream_t st) {
  const double eps = (eps_in == 0.0 && kind >= 3) ? 1e-7 : eps_in;
  double alpha = lr;
  if (kind == 3 || kind == 9 || kind == 10) {
    const double den = 1.0 - std::pow(a, t);
    alpha = den != 0.0 ? lr * std::sqrt(1.0 - std::pow(b, t)) / den : eps;
    if (!(alpha == alpha) || alpha == 0.0 || std::isinf(alpha)) alpha = eps;
  } else if (kind == 7) {
    const double den = 1.0 - std::pow(a, t);
    alpha = den != 0.0 ? lr / den : eps;
    if (!(alpha == alpha) || alpha == 0.0 || std::isinf(alpha)) alpha = eps;
  }
  const float fa = (float)a, fb = (float)b, flr = (float)lr, fe = (float)eps, fal = (float)alpha;
  const float nad = (float)(1.0 - std::pow(a, t));
  return abi_run([=] HDL(int64_t i) {
    const float g = grad[i];
    float u = g;
    switch (kind) {
      case 1: u = flr * g; break;
      case 2: { const float vp = fa * s0[i], v = vp - flr * g; o0[i] = v; u = vp - (fa + 1.f) * v; break; }
      case 3: { const float m = fa * s0[i] + (1.f - fa) * g, v = fb * s1[i] + (1.f - fb) * g * g; o0[i] = m; o1[i] = v;
                u = m * fal / (sqrtf(v) + fe); break; }
      case 9: { const float m = fa * s0[i] + (1.f - fa) * g, v = fb * s1[i] + (1.f - fb) * g * g, h = fmaxf(s2[i], v);
                o0[i] = m; o1[i] = v; o2[i] = h; u = fal * m / (sqrtf(h) + fe); break; }
      case 10: { const float m = fa * s0[i] + (1.f - fa) * g, dl = g - m, sv = fb * s1[i] + (1.f - fb) * dl * dl + fe;
                 o0[i] = m; o1[i] = sv; u = fal * m / (sqrtf(sv) + fe); break; }
      case 4: { const float r = fa * s0[i] + (1.f - fa) * g * g; o0[i] = r; u = flr * g / (sqrtf(r) + fe); break; }
      case 5: { const float h = s0[i] + g * g; o0[i] = h; u = flr * g / (sqrtf(h) + fe); break; }
      case 6: { const float msg = fa * s0[i] + (1.f - fa) * g * g, d0 = s1[i]; u = g * sqrtf(d0 + fe) / sqrtf(msg + fe);
                o0[i] = msg; o1[i] = fa * d0 + (1.f - fa) * u * u; break; }
      case 7: { const float m = fa * s0[i] + (1.f - fa) * g, ui = fmaxf(fb * s1[i], fabsf(g)) + 1e-32f; o0[i] = m; o1[i] = ui;
                u = m * fal / ui; break; }
      case 8: { const float m = fa * s0[i] + (1.f - fa) * g, v = fb * s1[i] + (1.f - fb) * g * g; o0[i] = m; o1[i] = v;
                u = flr * ((fa * m + (1.f - fa) * g) / nad) / (sqrtf(v) + fe); break; }
      default: break;
    }
    upd[i] = u;
  }, n, dev, st);
}

// =========================================================================================================== strided GEMM
// C[m,n] = alpha * sum_k A(m,k) * B(k,n) + beta * C[m,n] with element strides for every operand (transposes are strides)
DL4J_API int dl4j_abi_gemm(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, int64_t am, int64_t ak, int64_t bk,
                           int64_t bn, int64_t cm, int64_t cn, float alpha, float beta, int dev, hipStream_t st) {
  return abi_run([=] HDL(int64_t i) {
    const int64_t n = i % N, m = i / N;
    const float* ap = A + m * am;
    const float* bp = B + n * bn;
    float acc = 0.f;
    for (int64_t k = 0; k < K; ++k) acc = fmaf(ap[k * ak], bp[k * bk], acc);
    float* cp = C + m * cm + n * cn;
    *cp = beta != 0.f ? alpha * acc + beta * *cp : alpha * acc;
  }, M * N, dev, st);
}

// =========================================================================================================== copies / casts
// dst[ds . idx] = convert(src[ss . idx]) over the logical shape (permute, transpose, concat slices, cast)
DL4J_API int dl4j_abi_copy(const void* src, void* dst, const int64_t* gp, int dev, hipStream_t st) {
  CopyG g;
  std::memcpy(&g, gp, sizeof(g));
  const bool integral = (g.sdt == 3 || g.sdt == 4 || g.sdt == 6) && (g.ddt == 3 || g.ddt == 4 || g.ddt == 6);
  return abi_run([=] HDL(int64_t i) {
    int64_t so = 0, dof = 0, r = i;
    for (int64_t k = g.rank - 1; k >= 0; --k) {
      const int64_t c = r % g.shape[k];
      r /= g.shape[k];
      so += c * g.ss[k]; dof += c * g.ds[k];
    }
    if (integral) store_from(dst, dof, g.ddt, 0.0, load_as_long(src, so, g.sdt), true);
    else store_from(dst, dof, g.ddt, load_as_double(src, so, g.sdt), 0, false);
  }, g.n, dev, st);
}

// =========================================================================================================== gather / scatter
// x [outer, A, inner], idx [J] (idt 3 int32 / 4 int64) -> out [outer, J, inner]; esz = element size in bytes (4 or 8 or 2)
DL4J_API int dl4j_abi_gather(const void* x, const void* idx, int idt, void* out, int64_t outer, int64_t A, int64_t J, int64_t inner, int esz,
                             int dev, hipStream_t st) {
  return abi_run([=] HDL(int64_t i) {
    const int64_t in = i % inner; int64_t t = i / inner;
    const int64_t j = t % J, o = t / J;
    int64_t a = load_as_long(idx, j, idt);
    if (a < 0) a += A;
    if (a < 0 || a >= A) return;
    const int64_t so = (o * A + a) * inner + in;
    if (esz == 4) ((uint32_t*)out)[i] = ((const uint32_t*)x)[so];
    else if (esz == 8) ((uint64_t*)out)[i] = ((const uint64_t*)x)[so];
    else ((uint16_t*)out)[i] = ((const uint16_t*)x)[so];
  }, outer * J * inner, dev, st);
}

// out[idx[j], :] += upd[j, :]   (out already holds the reference array); rows may repeat -> atomics on the device, serial on the host
DL4J_API int dl4j_abi_scatter_add(float* out, const void* idx, int idt, const float* upd, int64_t A, int64_t J, int64_t inner, int dev,
                                  hipStream_t st) {
  return abi_run([=] HDL(int64_t i) {
    const int64_t in = i % inner, j = i / inner;
    int64_t a = load_as_long(idx, j, idt);
    if (a < 0) a += A;
    if (a < 0 || a >= A) return;
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(out + a * inner + in, upd[i]);
#else
    out[a * inner + in] += upd[i];
#endif
  }, J * inner, dev, st, /*serial=*/true);
}

DL4J_API int dl4j_abi_onehot(const void* idx, int idt, float* out, int64_t n, int64_t depth, float on, float off, int dev, hipStream_t st) {
  return abi_run([=] HDL(int64_t i) {
    const int64_t d = i % depth, j = i / depth;
    out[i] = load_as_long(idx, j, idt) == d ? on : off;
  }, n * depth, dev, st);
}

// =========================================================================================================== activation gradients
// gx = gy * f'(x). kind: 0 relu (cut t0) 1 sigmoid 2 tanh 3 relu6 (cut t0) 4 lrelu (alpha t0) 5 softplus 6 softsign 7 cube 8 hardtanh
// 9 hardsigmoid 10 elu (alpha t0) 11 selu
DL4J_API int dl4j_abi_act_bp(int kind, const float* x, const float* gy, float* gx, int64_t n, float t0, int dev, hipStream_t st) {
  return abi_run([=] HDL(int64_t i) {
    const float v = x[i], g = gy[i];
    float d = 1.f;
    switch (kind) {
      case 0: d = v > t0 ? 1.f : 0.f; break;
      case 1: { const float s = sigmoidf_(v); d = s * (1.f - s); break; }
      case 2: { const float th = tanhf(v); d = 1.f - th * th; break; }
      case 3: d = (v > t0 && v < 6.f) ? 1.f : 0.f; break;
      case 4: d = v >= 0.f ? 1.f : t0; break;
      case 5: d = sigmoidf_(v); break;
      case 6: { const float q = 1.f + fabsf(v); d = 1.f / (q * q); break; }
      case 7: d = 3.f * v * v; break;
      case 8: d = (v >= -1.f && v <= 1.f) ? 1.f : 0.f; break;
      case 9: d = (v > -2.5f && v < 2.5f) ? 0.2f : 0.f; break;
      case 10: d = v > 0.f ? 1.f : t0 * expf(v); break;
      default: d = v > 0.f ? 1.0507009873554805f : 1.0507009873554805f * 1.6732632423543772f * expf(v); break;
    }
    gx[i] = g * d;
  }, n, dev, st);
}

// softmax backward over the last dimension from the INPUT of the softmax: gx = y * (gy - sum(gy * y)), y = softmax(x)
DL4J_API int dl4j_abi_softmax_bp(const float* x, const float* gy, float* gx, int64_t M, int64_t C, int is_log, int dev, hipStream_t st) {
  return abi_run([=] HDL(int64_t m) {
    const float* xp = x + m * C;
    const float* gp2 = gy + m * C;
    float mx = -INFINITY;
    for (int64_t k = 0; k < C; ++k) mx = xp[k] > mx ? xp[k] : mx;
    double s = 0.0;
    for (int64_t k = 0; k < C; ++k) s += exp((double)(xp[k] - mx));
    double dot = 0.0, gs = 0.0;
    for (int64_t k = 0; k < C; ++k) { const double y = exp((double)(xp[k] - mx)) / s; dot += y * (double)gp2[k]; gs += (double)gp2[k]; }
    for (int64_t k = 0; k < C; ++k) {
      const double y = exp((double)(xp[k] - mx)) / s;
      gx[m * C + k] = (float)(is_log ? (double)gp2[k] - y * gs : y * ((double)gp2[k] - dot));
    }
  }, M, dev, st);
}
