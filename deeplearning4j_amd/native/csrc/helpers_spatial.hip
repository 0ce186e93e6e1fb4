// Declarable-op helper kernels, spatial family (gfx950): depthwise conv2d (fwd / input grad / weight grad), 3-D
// pooling (max / avg / pnorm, fwd + bwd), nearest 3-D upsampling fwd + bwd, grayscale dilation2d and
// extract_image_patches.
//
// Reference counterparts (libnd4j/include/ops/declarable/helpers/cuda): convolutions_depthwiseConv2d(BP).cu (im2col +
// batched GEMM there), convolutions_pooling3d(BP).cu, convolutions_upsampling3d(BP).cu, dilation2d.cu,
// extract_patches.cu.
// Own design notes: depthwise convolution has one input channel per group, so there is no GEMM to feed the matrix
// cores with — it is a memory-bound stencil. The kernels are direct NCHW stencils with W innermost across the 64
// lanes (coalesced rows, the k x k taps hit L1/L2), the weight gradient is a per-(channel, tap) block reduction.
// Pooling backward is the gather formulation (every input voxel visits the windows that cover it) so it is
// deterministic and needs no atomics and no pre-zeroed output.
#include "common.h"

namespace {

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ depthwise conv2d, x [N, C, H, W], w [C*mC, kH, kW], y [N, C*mC, oH, oW]
struct DwP {
  int N, C, mC, H, W, kH, kW, oH, oW, sH, sW, pH, pW, dH, dW;
};

template <typename T>
__global__ void dw_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, DwP p) {
  const int OC = p.C * p.mC;
  const int64_t total = (int64_t)p.N * OC * p.oH * p.oW;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int ow = (int)(t % p.oW); t /= p.oW;
    const int oh = (int)(t % p.oH); t /= p.oH;
    const int oc = (int)(t % OC);
    const int n = (int)(t / OC);
    const int c = oc / p.mC;
    const T* xp = x + ((int64_t)n * p.C + c) * p.H * p.W;
    const T* wp = w + (int64_t)oc * p.kH * p.kW;
    float s = bias ? bias[oc] : 0.f;
    for (int kh = 0; kh < p.kH; ++kh) {
      const int ih = oh * p.sH - p.pH + kh * p.dH;
      if (ih < 0 || ih >= p.H) continue;
      for (int kw = 0; kw < p.kW; ++kw) {
        const int iw = ow * p.sW - p.pW + kw * p.dW;
        if (iw < 0 || iw >= p.W) continue;
        s += Cvt<T>::ld(xp, (int64_t)ih * p.W + iw) * Cvt<T>::ld(wp, kh * p.kW + kw);
      }
    }
    Cvt<T>::st(y, i, s);
  }
}

// dx[n, c, ih, iw] = sum_m sum_taps dy[n, c*mC+m, oh, ow] * w[c*mC+m, kh, kw]   (gather over the covering outputs)
template <typename T>
__global__ void dw_bwd_x_kernel(const T* __restrict__ dy, const T* __restrict__ w, T* __restrict__ dx, DwP p) {
  const int64_t total = (int64_t)p.N * p.C * p.H * p.W;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int iw = (int)(t % p.W); t /= p.W;
    const int ih = (int)(t % p.H); t /= p.H;
    const int c = (int)(t % p.C);
    const int n = (int)(t / p.C);
    float s = 0.f;
    for (int m = 0; m < p.mC; ++m) {
      const int oc = c * p.mC + m;
      const T* gp = dy + ((int64_t)n * p.C * p.mC + oc) * p.oH * p.oW;
      const T* wp = w + (int64_t)oc * p.kH * p.kW;
      for (int kh = 0; kh < p.kH; ++kh) {
        const int hh = ih + p.pH - kh * p.dH;
        if (hh < 0 || hh % p.sH) continue;
        const int oh = hh / p.sH;
        if (oh >= p.oH) continue;
        for (int kw = 0; kw < p.kW; ++kw) {
          const int ww = iw + p.pW - kw * p.dW;
          if (ww < 0 || ww % p.sW) continue;
          const int ow = ww / p.sW;
          if (ow >= p.oW) continue;
          s += Cvt<T>::ld(gp, (int64_t)oh * p.oW + ow) * Cvt<T>::ld(wp, kh * p.kW + kw);
        }
      }
    }
    Cvt<T>::st(dx, i, s);
  }
}

// dw[oc, kh, kw] = sum_{n, oh, ow} dy[n, oc, oh, ow] * x[n, c, ih, iw]; one block per (oc, tap); db[oc] by the tap-0 block
template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_w_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw,
                                                       float* __restrict__ db, DwP p) {
  __shared__ float red[16];
  const int taps = p.kH * p.kW;
  const int oc = blockIdx.x / taps, tap = blockIdx.x - oc * taps;
  const int kh = tap / p.kW, kw = tap - kh * p.kW;
  const int c = oc / p.mC;
  const int64_t per = (int64_t)p.oH * p.oW, total = (int64_t)p.N * per;
  float s = 0.f, sb = 0.f;
  for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
    const int n = (int)(i / per);
    const int r = (int)(i - (int64_t)n * per);
    const int oh = r / p.oW, ow = r - oh * p.oW;
    const float g = Cvt<T>::ld(dy, ((int64_t)n * p.C * p.mC + oc) * per + r);
    sb += g;
    const int ih = oh * p.sH - p.pH + kh * p.dH, iw = ow * p.sW - p.pW + kw * p.dW;
    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) s += g * Cvt<T>::ld(x, (((int64_t)n * p.C + c) * p.H + ih) * p.W + iw);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) dw[blockIdx.x] = s;
  if (db && tap == 0) {
    sb = block_sum(sb, red);
    if (threadIdx.x == 0) db[oc] = sb;
  }
}

// ------------------------------------------------------------------ pooling3d, x [N*C, D, H, W] -> y [N*C, oD, oH, oW]
struct P3 {
  int64_t planes;
  int D, H, W, kD, kH, kW, oD, oH, oW, sD, sH, sW, pD, pH, pW, dD, dH, dW;
  int mode;               // 0 max, 1 avg, 2 pnorm
  int include_pad;        // avg: divide by the full window
  float pnorm;
};

__device__ __forceinline__ int win_count(const P3& p, int od, int oh, int ow) {
  int cd = 0, ch = 0, cw = 0;
  for (int k = 0; k < p.kD; ++k) { const int v = od * p.sD - p.pD + k * p.dD; cd += (v >= 0 && v < p.D); }
  for (int k = 0; k < p.kH; ++k) { const int v = oh * p.sH - p.pH + k * p.dH; ch += (v >= 0 && v < p.H); }
  for (int k = 0; k < p.kW; ++k) { const int v = ow * p.sW - p.pW + k * p.dW; cw += (v >= 0 && v < p.W); }
  return cd * ch * cw;
}

template <typename T>
__global__ void pool3d_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, P3 p) {
  const int64_t total = p.planes * p.oD * p.oH * p.oW;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int ow = (int)(t % p.oW); t /= p.oW;
    const int oh = (int)(t % p.oH); t /= p.oH;
    const int od = (int)(t % p.oD);
    const int64_t pl = t / p.oD;
    const T* xp = x + pl * p.D * p.H * p.W;
    float acc = p.mode == 0 ? -INFINITY : 0.f;
    int cnt = 0;
    for (int kd = 0; kd < p.kD; ++kd) {
      const int id = od * p.sD - p.pD + kd * p.dD;
      if (id < 0 || id >= p.D) continue;
      for (int kh = 0; kh < p.kH; ++kh) {
        const int ih = oh * p.sH - p.pH + kh * p.dH;
        if (ih < 0 || ih >= p.H) continue;
        for (int kw = 0; kw < p.kW; ++kw) {
          const int iw = ow * p.sW - p.pW + kw * p.dW;
          if (iw < 0 || iw >= p.W) continue;
          const float v = Cvt<T>::ld(xp, ((int64_t)id * p.H + ih) * p.W + iw);
          if (p.mode == 0) acc = fmaxf(acc, v);
          else if (p.mode == 1) acc += v;
          else acc += __powf(fabsf(v), p.pnorm);
          ++cnt;
        }
      }
    }
    if (p.mode == 1) acc /= (float)(p.include_pad ? p.kD * p.kH * p.kW : (cnt > 0 ? cnt : 1));
    else if (p.mode == 2) acc = __powf(acc, 1.f / p.pnorm);
    Cvt<T>::st(y, i, acc);
  }
}

// gather backward. max: the first voxel (scan order) equal to the window maximum receives the gradient, as in ATen.
template <typename T>
__global__ void pool3d_bwd_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ g, T* __restrict__ dx, P3 p) {
  const int64_t total = p.planes * p.D * p.H * p.W;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int iw = (int)(t % p.W); t /= p.W;
    const int ih = (int)(t % p.H); t /= p.H;
    const int id = (int)(t % p.D);
    const int64_t pl = t / p.D;
    const T* xp = x + pl * p.D * p.H * p.W;
    const int64_t obase = pl * p.oD * p.oH * p.oW;
    const float xv = Cvt<T>::ld(x, i);
    float s = 0.f;
    for (int kd = 0; kd < p.kD; ++kd) {
      const int dd = id + p.pD - kd * p.dD;
      if (dd < 0 || dd % p.sD) continue;
      const int od = dd / p.sD;
      if (od >= p.oD) continue;
      for (int kh = 0; kh < p.kH; ++kh) {
        const int hh = ih + p.pH - kh * p.dH;
        if (hh < 0 || hh % p.sH) continue;
        const int oh = hh / p.sH;
        if (oh >= p.oH) continue;
        for (int kw = 0; kw < p.kW; ++kw) {
          const int ww = iw + p.pW - kw * p.dW;
          if (ww < 0 || ww % p.sW) continue;
          const int ow = ww / p.sW;
          if (ow >= p.oW) continue;
          const int64_t o = obase + ((int64_t)od * p.oH + oh) * p.oW + ow;
          const float gv = Cvt<T>::ld(g, o);
          if (p.mode == 1) {
            s += gv / (float)(p.include_pad ? p.kD * p.kH * p.kW : win_count(p, od, oh, ow));
          } else if (p.mode == 2) {
            const float yv = Cvt<T>::ld(y, o);
            if (yv > 0.f) s += gv * __powf(fabsf(xv), p.pnorm - 1.f) * (xv >= 0.f ? 1.f : -1.f) * __powf(yv, 1.f - p.pnorm);
          } else {
            const float yv = Cvt<T>::ld(y, o);
            if (xv != yv) continue;
            // am I the first voxel of this window (scan order) that equals the maximum?
            bool first = true, done = false;
            for (int a = 0; a < p.kD && !done; ++a) {
              const int jd = od * p.sD - p.pD + a * p.dD;
              if (jd < 0 || jd >= p.D) continue;
              for (int b = 0; b < p.kH && !done; ++b) {
                const int jh = oh * p.sH - p.pH + b * p.dH;
                if (jh < 0 || jh >= p.H) continue;
                for (int c = 0; c < p.kW && !done; ++c) {
                  const int jw = ow * p.sW - p.pW + c * p.dW;
                  if (jw < 0 || jw >= p.W) continue;
                  if (jd == id && jh == ih && jw == iw) done = true;                                   // reached myself
                  else if (Cvt<T>::ld(xp, ((int64_t)jd * p.H + jh) * p.W + jw) == yv) { first = false; done = true; }
                }
              }
            }
            if (first) s += gv;
          }
        }
      }
    }
    Cvt<T>::st(dx, i, s);
  }
}

// ------------------------------------------------------------------ nearest upsampling 3d, [planes, D, H, W] <-> [planes, D*fD, H*fH, W*fW]
template <typename T>
__global__ void upsample3d_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t planes, int D, int H, int W, int fD, int fH,
                                      int fW) {
  const int oD = D * fD, oH = H * fH, oW = W * fW;
  const int64_t total = planes * oD * oH * oW;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int ow = (int)(t % oW); t /= oW;
    const int oh = (int)(t % oH); t /= oH;
    const int od = (int)(t % oD);
    const int64_t pl = t / oD;
    y[i] = x[((pl * D + od / fD) * H + oh / fH) * W + ow / fW];
  }
}

template <typename T>
__global__ void upsample3d_bwd_kernel(const T* __restrict__ g, T* __restrict__ dx, int64_t planes, int D, int H, int W, int fD, int fH,
                                      int fW) {
  const int oH = H * fH, oW = W * fW;
  const int64_t total = planes * D * H * W;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int d = (int)(t % D);
    const int64_t pl = t / D;
    float s = 0.f;
    for (int a = 0; a < fD; ++a)
      for (int b = 0; b < fH; ++b)
        for (int c = 0; c < fW; ++c) s += Cvt<T>::ld(g, ((pl * D * fD + d * fD + a) * oH + h * fH + b) * (int64_t)oW + w * fW + c);
    Cvt<T>::st(dx, i, s);
  }
}

// ------------------------------------------------------------------ dilation2d (NHWC, w [kH, kW, C]): y = max_taps(x + w)
template <typename T>
__global__ void dilation2d_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int N, int H, int W, int C, int kH,
                                  int kW, int oH, int oW, int sH, int sW, int pH, int pW, int rH, int rW) {
  const int64_t total = (int64_t)N * oH * oW * C;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int c = (int)(t % C); t /= C;
    const int ow = (int)(t % oW); t /= oW;
    const int oh = (int)(t % oH);
    const int n = (int)(t / oH);
    float m = -INFINITY;
    for (int kh = 0; kh < kH; ++kh) {
      const int ih = oh * sH - pH + kh * rH;
      if (ih < 0 || ih >= H) continue;
      for (int kw = 0; kw < kW; ++kw) {
        const int iw = ow * sW - pW + kw * rW;
        if (iw < 0 || iw >= W) continue;
        m = fmaxf(m, Cvt<T>::ld(x, (((int64_t)n * H + ih) * W + iw) * C + c) + Cvt<T>::ld(w, ((int64_t)kh * kW + kw) * C + c));
      }
    }
    Cvt<T>::st(y, i, m);
  }
}

// ------------------------------------------------------------------ extract_image_patches (NHWC -> [N, oH, oW, kH*kW*C]), zero padding
template <typename T>
__global__ void extract_patches_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int kH, int kW, int oH,
                                       int oW, int sH, int sW, int pH, int pW, int rH, int rW) {
  const int64_t total = (int64_t)N * oH * oW * kH * kW * C;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int c = (int)(t % C); t /= C;
    const int kw = (int)(t % kW); t /= kW;
    const int kh = (int)(t % kH); t /= kH;
    const int ow = (int)(t % oW); t /= oW;
    const int oh = (int)(t % oH);
    const int n = (int)(t / oH);
    const int ih = oh * sH - pH + kh * rH, iw = ow * sW - pW + kw * rW;
    y[i] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[(((int64_t)n * H + ih) * W + iw) * C + c] : T(0);
  }
}

}  // namespace

#define DISPATCH_T(...)                             \
  if (dtype == 1) { typedef bf16_t T; __VA_ARGS__ } \
  else { typedef float T; __VA_ARGS__ }
#define DISPATCH_BYTES(esize, ...)                                  \
  switch (esize) {                                                  \
    case 1: { typedef uint8_t T; __VA_ARGS__ } break;               \
    case 2: { typedef uint16_t T; __VA_ARGS__ } break;              \
    case 4: { typedef uint32_t T; __VA_ARGS__ } break;              \
    case 8: { typedef uint64_t T; __VA_ARGS__ } break;              \
    default: return 1;                                              \
  }

// geom = {N, C, mC, H, W, kH, kW, oH, oW, sH, sW, pH, pW, dH, dW}
static DwP mkdw(const int* g) { return DwP{g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[13], g[14]}; }
static bool dw_valid(const DwP& p) {
  // every load in the kernels is bounds-checked against H / W; the outputs are sized by the caller from oH / oW
  return p.N > 0 && p.C > 0 && p.mC > 0 && p.H > 0 && p.W > 0 && p.kH > 0 && p.kW > 0 && p.oH > 0 && p.oW > 0 && p.sH > 0 && p.sW > 0 &&
         p.dH > 0 && p.dW > 0 && p.pH >= 0 && p.pW >= 0;
}

DL4J_API int dl4j_depthwise_fwd(const void* x, const void* w, const float* bias, void* y, const int* geom, int dtype, hipStream_t st) {
  const DwP p = mkdw(geom);
  if (!dw_valid(p)) return 1;
  const int64_t total = (int64_t)p.N * p.C * p.mC * p.oH * p.oW;
  DISPATCH_T(hipLaunchKernelGGL(dw_fwd_kernel<T>, dim3(grid_for(total, 512)), dim3(256), 0, st, (const T*)x, (const T*)w, bias, (T*)y, p);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_depthwise_bwd_x(const void* dy, const void* w, void* dx, const int* geom, int dtype, hipStream_t st) {
  const DwP p = mkdw(geom);
  if (!dw_valid(p)) return 1;
  const int64_t total = (int64_t)p.N * p.C * p.H * p.W;
  DISPATCH_T(hipLaunchKernelGGL(dw_bwd_x_kernel<T>, dim3(grid_for(total, 512)), dim3(256), 0, st, (const T*)dy, (const T*)w, (T*)dx, p);)
  DL4J_RET_LAST();
}

// dw: fp32 [C*mC, kH, kW]; db: fp32 [C*mC] or null
DL4J_API int dl4j_depthwise_bwd_w(const void* dy, const void* x, float* dw, float* db, const int* geom, int dtype, hipStream_t st) {
  const DwP p = mkdw(geom);
  if (!dw_valid(p)) return 1;
  const int64_t blocks = (int64_t)p.C * p.mC * p.kH * p.kW;
  if (blocks > 0x7fffffff) return 1;
  DISPATCH_T(hipLaunchKernelGGL(dw_bwd_w_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)dy, (const T*)x, dw, db, p);)
  DL4J_RET_LAST();
}

// geom = {D,H,W, kD,kH,kW, oD,oH,oW, sD,sH,sW, pD,pH,pW, dD,dH,dW, mode, include_pad}
static bool mkp3(int64_t planes, const int* g, float pnorm, P3* out) {
  P3 p;
  p.planes = planes;
  p.D = g[0]; p.H = g[1]; p.W = g[2]; p.kD = g[3]; p.kH = g[4]; p.kW = g[5]; p.oD = g[6]; p.oH = g[7]; p.oW = g[8];
  p.sD = g[9]; p.sH = g[10]; p.sW = g[11]; p.pD = g[12]; p.pH = g[13]; p.pW = g[14]; p.dD = g[15]; p.dH = g[16]; p.dW = g[17];
  p.mode = g[18]; p.include_pad = g[19]; p.pnorm = pnorm;
  *out = p;
  for (int j = 0; j < 12; ++j) if (g[j] <= 0) return false;
  for (int j = 12; j < 15; ++j) if (g[j] < 0) return false;
  for (int j = 15; j < 18; ++j) if (g[j] <= 0) return false;
  return planes > 0 && p.mode >= 0 && p.mode <= 2 && (p.mode != 2 || pnorm > 0.f);
}

DL4J_API int dl4j_pool3d_fwd(const void* x, void* y, int64_t planes, const int* geom, float pnorm, int dtype, hipStream_t st) {
  P3 p;
  if (!mkp3(planes, geom, pnorm, &p)) return 1;
  const int64_t total = planes * p.oD * p.oH * p.oW;
  DISPATCH_T(hipLaunchKernelGGL(pool3d_fwd_kernel<T>, dim3(grid_for(total, 512)), dim3(256), 0, st, (const T*)x, (T*)y, p);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_pool3d_bwd(const void* x, const void* y, const void* g, void* dx, int64_t planes, const int* geom, float pnorm, int dtype,
                             hipStream_t st) {
  P3 p;
  if (!mkp3(planes, geom, pnorm, &p)) return 1;
  const int64_t total = planes * p.D * p.H * p.W;
  DISPATCH_T(hipLaunchKernelGGL(pool3d_bwd_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)x, (const T*)y, (const T*)g,
                                (T*)dx, p);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_upsample3d_fwd(const void* x, void* y, int64_t planes, int D, int H, int W, int fD, int fH, int fW, int esize,
                                 hipStream_t st) {
  const int64_t total = planes * D * fD * H * fH * W * fW;
  if (total <= 0) return 0;
  DISPATCH_BYTES(esize, hipLaunchKernelGGL(upsample3d_fwd_kernel<T>, dim3(grid_for(total, 1024)), dim3(256), 0, st, (const T*)x, (T*)y,
                                           planes, D, H, W, fD, fH, fW);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_upsample3d_bwd(const void* g, void* dx, int64_t planes, int D, int H, int W, int fD, int fH, int fW, int dtype,
                                 hipStream_t st) {
  const int64_t total = planes * D * H * W;
  if (total <= 0) return 0;
  DISPATCH_T(hipLaunchKernelGGL(upsample3d_bwd_kernel<T>, dim3(grid_for(total, 512)), dim3(256), 0, st, (const T*)g, (T*)dx, planes, D, H, W,
                                fD, fH, fW);)
  DL4J_RET_LAST();
}

// geom = {N, H, W, C, kH, kW, oH, oW, sH, sW, pH, pW, rH, rW}
DL4J_API int dl4j_dilation2d(const void* x, const void* w, void* y, const int* g, int dtype, hipStream_t st) {
  for (int j = 0; j < 10; ++j) if (g[j] <= 0) return 1;
  if (g[12] <= 0 || g[13] <= 0) return 1;
  const int64_t total = (int64_t)g[0] * g[6] * g[7] * g[3];
  DISPATCH_T(hipLaunchKernelGGL(dilation2d_kernel<T>, dim3(grid_for(total, 512)), dim3(256), 0, st, (const T*)x, (const T*)w, (T*)y, g[0],
                                g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[13]);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_extract_patches(const void* x, void* y, const int* g, int esize, hipStream_t st) {
  for (int j = 0; j < 10; ++j) if (g[j] <= 0) return 1;
  if (g[12] <= 0 || g[13] <= 0) return 1;
  const int64_t total = (int64_t)g[0] * g[6] * g[7] * g[4] * g[5] * g[3];
  DISPATCH_BYTES(esize, hipLaunchKernelGGL(extract_patches_kernel<T>, dim3(grid_for(total, 1024)), dim3(256), 0, st, (const T*)x, (T*)y,
                                           g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[13]);)
  DL4J_RET_LAST();
}
