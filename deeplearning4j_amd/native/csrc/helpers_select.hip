// Declarable-op helper kernels, selection / indexed-reduction / recurrent-elementwise family (gfx950):
// segment max/min/prod, N-d index flattening (gather_nd / scatter_nd*), row scatter-set, top-k, fixed-width
// histogram, LRN backward, the SRU recurrence forward + backward, colour-space conversions and greedy NMS.
//
// Reference counterparts (libnd4j/include/ops/declarable/helpers/cuda): segment_max.cu / segment_min.cu /
// segment_prod.cu, scatter.cu (scatterND*), gather_nd.cu, top_k.cu, histogramFixedWidth.cu, lrn.cu (lrnBP),
// sru.cu, imagesHelpers.cu (rgb<->hsv/yuv/yiq), image_suppression.cu.
// Own design notes: the nd variants of gather/scatter first flatten the index tuples to row numbers and then reuse
// the coalesced row kernels; top-k is a per-wave k-pass selection with a (value desc, index asc) total order so the
// result is deterministic; the histogram is privatised in LDS per block; LRN backward stages one row in LDS; the SRU
// kernels give each (batch, unit) pair one lane that walks time (the recurrence is elementwise, only the input
// projection is a GEMM and that stays in the BLAS library); NMS is one block walking the score order with the
// IoU tests against the kept set spread over the lanes.
#include "common.h"

namespace {

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ fp32 atomics built on compare-and-swap
__device__ __forceinline__ void atomic_max_f32(float* a, float v) {
  unsigned* ua = (unsigned*)a;
  unsigned old = *ua;
  while (__uint_as_float(old) < v) {
    const unsigned assumed = old;
    old = atomicCAS(ua, assumed, __float_as_uint(v));
    if (old == assumed) break;
  }
}
__device__ __forceinline__ void atomic_min_f32(float* a, float v) {
  unsigned* ua = (unsigned*)a;
  unsigned old = *ua;
  while (__uint_as_float(old) > v) {
    const unsigned assumed = old;
    old = atomicCAS(ua, assumed, __float_as_uint(v));
    if (old == assumed) break;
  }
}
__device__ __forceinline__ void atomic_mul_f32(float* a, float v) {
  unsigned* ua = (unsigned*)a;
  unsigned old = *ua, assumed;
  do {
    assumed = old;
    old = atomicCAS(ua, assumed, __float_as_uint(__uint_as_float(assumed) * v));
  } while (old != assumed);
}

// out[seg[r], :] = op(out[seg[r], :], x[r, :]); out pre-filled with the identity by the caller. mode 1 max, 2 min, 3 prod
template <typename T, typename I>
__global__ void segment_reduce_kernel(const T* __restrict__ x, const I* __restrict__ seg, float* __restrict__ out, int64_t rows, int D,
                                      int64_t nseg, int mode) {
  const int64_t total = rows * D;
  GRID_STRIDE(i, total) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    const int64_t s = (int64_t)seg[r];
    if (s < 0 || s >= nseg) continue;
    const float v = Cvt<T>::ld(x, i);
    float* o = out + s * D + c;
    if (mode == 1) atomic_max_f32(o, v);
    else if (mode == 2) atomic_min_f32(o, v);
    else atomic_mul_f32(o, v);
  }
}

// ------------------------------------------------------------------ [n, k] index tuples -> flat row numbers (-1 when out of range)
struct NdShape {
  int64_t dim[8];
};
template <typename I>
__global__ void nd_flatten_kernel(const I* __restrict__ idx, int64_t* __restrict__ flat, int64_t n, int k, NdShape shp) {
  GRID_STRIDE(r, n) {
    int64_t f = 0;
    bool okr = true;
    for (int j = 0; j < k; ++j) {
      int64_t v = (int64_t)idx[r * k + j];
      if (v < 0) v += shp.dim[j];
      okr = okr && v >= 0 && v < shp.dim[j];
      f = f * shp.dim[j] + v;
    }
    flat[r] = okr ? f : -1;
  }
}

// table[idx[r], :] = src[r, :]  (raw element copy; with duplicate indices one of the writers wins, as in the reference)
template <typename T>
__global__ void scatter_set_rows_kernel(const T* __restrict__ src, const int64_t* __restrict__ idx, T* __restrict__ table, int64_t n_idx,
                                        int D, int64_t V) {
  const int64_t total = n_idx * D;
  GRID_STRIDE(i, total) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    const int64_t v = idx[r];
    if (v >= 0 && v < V) table[v * D + c] = src[i];
  }
}

// ------------------------------------------------------------------ top-k per row, one wave per row, k selection passes
// order: larger value first, lower index first among equal values
__device__ __forceinline__ bool tk_before(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

template <typename T>
__global__ __launch_bounds__(256) void top_k_kernel(const T* __restrict__ x, T* __restrict__ vals, int64_t* __restrict__ inds,
                                                    int64_t rows, int cols, int k) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const T* xr = x + r * cols;
  float pv = INFINITY;
  int pi = -1;
  for (int s = 0; s < k; ++s) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < cols; c += 64) {
      const float v = Cvt<T>::ld(xr, c);
      if (tk_before(pv, pi, v, c) && tk_before(v, c, bv, bi)) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (tk_before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
      Cvt<T>::st(vals, r * k + s, bv);
      inds[r * k + s] = bi == 0x7fffffff ? 0 : bi;
    }
    pv = bv;
    pi = bi;
  }
}

// ------------------------------------------------------------------ fixed-width histogram, LDS-privatised
#define HIST_LDS_BINS 4096
template <typename T>
__global__ __launch_bounds__(256) void hist_kernel(const T* __restrict__ x, long long* __restrict__ counts, int64_t n, int bins, float lo,
                                                   float hi) {
  __shared__ unsigned local[HIST_LDS_BINS];
  const bool priv = bins <= HIST_LDS_BINS;
  if (priv) {
    for (int b = threadIdx.x; b < bins; b += blockDim.x) local[b] = 0u;
    __syncthreads();
  }
  const float scale = (float)bins / (hi - lo);
  GRID_STRIDE(i, n) {
    const float v = Cvt<T>::ld(x, i);
    int b = (int)floorf((v - lo) * scale);
    b = b < 0 ? 0 : (b >= bins ? bins - 1 : b);
    if (priv) atomicAdd(&local[b], 1u);
    else atomicAdd((unsigned long long*)counts + b, 1ull);
  }
  if (priv) {
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += blockDim.x)
      if (local[b]) atomicAdd((unsigned long long*)counts + b, (unsigned long long)local[b]);
  }
}

// ------------------------------------------------------------------ LRN backward over the last axis (one block per row, row in LDS)
//   s_i = bias + alpha * sum_{j in W(i)} x_j^2 ;  y_i = x_i * s_i^-beta
//   dx_k = g_k * s_k^-beta - 2*alpha*beta * x_k * sum_{i in W(k)} g_i * x_i * s_i^(-beta-1)
#define LRN_MAX_C 4096
template <typename T>
__global__ __launch_bounds__(256) void lrn_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ dx, int64_t M, int C,
                                                      int depth, float bias, float alpha, float beta) {
  __shared__ float xs[LRN_MAX_C];
  __shared__ float ts[LRN_MAX_C];
  for (int64_t r = blockIdx.x; r < M; r += gridDim.x) {
    const int64_t base = r * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) xs[c] = Cvt<T>::ld(x, base + c);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const int lo = c - depth < 0 ? 0 : c - depth, hi = c + depth >= C ? C - 1 : c + depth;
      float s = 0.f;
      for (int k = lo; k <= hi; ++k) s += xs[k] * xs[k];
      const float sc = bias + alpha * s;
      const float gv = Cvt<T>::ld(g, base + c);
      const float p = __powf(sc, -beta);
      ts[c] = gv * xs[c] * p / sc;
      Cvt<T>::st(dx, base + c, gv * p);          // first term; the window term is added below by the same lane
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const int lo = c - depth < 0 ? 0 : c - depth, hi = c + depth >= C ? C - 1 : c + depth;
      float s = 0.f;
      for (int k = lo; k <= hi; ++k) s += ts[k];
      Cvt<T>::st(dx, base + c, Cvt<T>::ld(dx, base + c) - 2.f * alpha * beta * xs[c] * s);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ SRU recurrence, x [B, K, T], z = W x [B, 3K, T]
//   f = sigmoid(z1 + bf), r = sigmoid(z2 + br);  c_t = f c_{t-1} + (1 - f) z0;  h_t = r tanh(c_t) + (1 - r) x_t
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ void sru_fwd_kernel(const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ bias,
                               const float* __restrict__ c0, float* __restrict__ h, float* __restrict__ cs, int B, int K, int T) {
  GRID_STRIDE(i, (int64_t)B * K) {
    const int b = (int)(i / K), k = (int)(i - (int64_t)b * K);
    const float* xr = x + ((int64_t)b * K + k) * T;
    const float* z0 = z + ((int64_t)b * 3 * K + k) * T;
    const float* z1 = z0 + (int64_t)K * T;
    const float* z2 = z1 + (int64_t)K * T;
    float* hr = h + ((int64_t)b * K + k) * T;
    float* cr = cs + ((int64_t)b * K + k) * T;
    const float bf = bias[k], br = bias[K + k];
    float c = c0[i];
    for (int t = 0; t < T; ++t) {
      const float f = sigm(z1[t] + bf), r = sigm(z2[t] + br);
      c = f * c + (1.f - f) * z0[t];
      cr[t] = c;
      hr[t] = r * tanhf(c) + (1.f - r) * xr[t];
    }
  }
}

// reverse-time pass: dz [B, 3K, T], dxd [B, K, T] (highway term), dc0 [B, K]; gc_last optional [B, K]
__global__ void sru_bwd_kernel(const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ bias,
                               const float* __restrict__ c0, const float* __restrict__ cs, const float* __restrict__ gh,
                               const float* __restrict__ gc_last, float* __restrict__ dz, float* __restrict__ dxd,
                               float* __restrict__ dc0, int B, int K, int T) {
  GRID_STRIDE(i, (int64_t)B * K) {
    const int b = (int)(i / K), k = (int)(i - (int64_t)b * K);
    const int64_t row = ((int64_t)b * K + k) * T;
    const int64_t zrow = ((int64_t)b * 3 * K + k) * T;
    const int64_t KT = (int64_t)K * T;
    const float bf = bias[k], br = bias[K + k];
    float gc = gc_last ? gc_last[i] : 0.f;
    for (int t = T - 1; t >= 0; --t) {
      const float f = sigm(z[zrow + KT + t] + bf), r = sigm(z[zrow + 2 * KT + t] + br);
      const float c = cs[row + t], cp = t > 0 ? cs[row + t - 1] : c0[i];
      const float tc = tanhf(c), g = gh[row + t], xv = x[row + t], z0 = z[zrow + t];
      gc += g * r * (1.f - tc * tc);
      dz[zrow + 2 * KT + t] = g * (tc - xv) * r * (1.f - r);
      dxd[row + t] = g * (1.f - r);
      dz[zrow + KT + t] = gc * (cp - z0) * f * (1.f - f);
      dz[zrow + t] = gc * (1.f - f);
      gc *= f;
    }
    dc0[i] = gc;
  }
}

// ------------------------------------------------------------------ colour conversions on channel-last pixels [n, 3]
// mode 0 rgb->hsv, 1 hsv->rgb, 2 3x3 matrix (rgb<->yuv / yiq; m row-major), 3 adjust_hue (delta = m[0]),
// 4 adjust_saturation (factor = m[0])
struct Mat3 {
  float m[9];
};
__device__ __forceinline__ void rgb2hsv(float a, float b, float c, float& h, float& s, float& v) {
  const float mx = fmaxf(a, fmaxf(b, c)), mn = fminf(a, fminf(b, c)), dl = mx - mn;
  s = mx > 0.f ? dl / fmaxf(mx, 1e-30f) : 0.f;
  const float dz = fmaxf(dl, 1e-30f);
  float hh = mx == a ? (b - c) / dz : (mx == b ? 2.f + (c - a) / dz : 4.f + (a - b) / dz);
  hh = hh / 6.f;
  hh = hh - floorf(hh);
  h = dl == 0.f ? 0.f : hh;
  v = mx;
}
__device__ __forceinline__ float hsv_chan(float n, float h6, float s, float v) {
  float k = fmodf(n + h6, 6.f);
  if (k < 0.f) k += 6.f;
  return v - v * s * fminf(fmaxf(fminf(k, 4.f - k), 0.f), 1.f);
}
template <typename T>
__global__ void color_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int mode, Mat3 M) {
  GRID_STRIDE(p, n) {
    const float a = Cvt<T>::ld(x, p * 3), b = Cvt<T>::ld(x, p * 3 + 1), c = Cvt<T>::ld(x, p * 3 + 2);
    float o0, o1, o2;
    if (mode == 0) {
      rgb2hsv(a, b, c, o0, o1, o2);
    } else if (mode == 1) {
      o0 = hsv_chan(5.f, a * 6.f, b, c);
      o1 = hsv_chan(3.f, a * 6.f, b, c);
      o2 = hsv_chan(1.f, a * 6.f, b, c);
    } else if (mode == 2) {
      o0 = M.m[0] * a + M.m[1] * b + M.m[2] * c;
      o1 = M.m[3] * a + M.m[4] * b + M.m[5] * c;
      o2 = M.m[6] * a + M.m[7] * b + M.m[8] * c;
    } else {
      float h, s, v;
      rgb2hsv(a, b, c, h, s, v);
      if (mode == 3) { h += M.m[0]; h -= floorf(h); }
      else s = fminf(fmaxf(s * M.m[0], 0.f), 1.f);
      o0 = hsv_chan(5.f, h * 6.f, s, v);
      o1 = hsv_chan(3.f, h * 6.f, s, v);
      o2 = hsv_chan(1.f, h * 6.f, s, v);
    }
    Cvt<T>::st(y, p * 3, o0);
    Cvt<T>::st(y, p * 3 + 1, o1);
    Cvt<T>::st(y, p * 3 + 2, o2);
  }
}

// ------------------------------------------------------------------ Barnes-Hut t-SNE attractive (edge) forces over a CSR graph
// out[n, :] = sum_{k in row n} val[k] / (1 + |y_n - y_col[k]|^2) * (y_n - y_col[k]); one thread per row, D <= 8 in registers
template <typename I>
__global__ void barnes_edge_forces_kernel(const I* __restrict__ rowP, const I* __restrict__ colP, const float* __restrict__ val,
                                          const float* __restrict__ y, float* __restrict__ out, int N, int D, int64_t nnz) {
  GRID_STRIDE(n, (int64_t)N) {
    float yn[8], acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { yn[j] = j < D ? y[n * D + j] : 0.f; acc[j] = 0.f; }
    int64_t k0 = (int64_t)rowP[n], k1 = (int64_t)rowP[n + 1];
    if (k1 > nnz) k1 = nnz;
    for (int64_t k = k0; k < k1; ++k) {
      const int64_t c = (int64_t)colP[k];
      if (c < 0 || c >= N) continue;
      float diff[8], d2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { diff[j] = j < D ? yn[j] - y[c * D + j] : 0.f; d2 += diff[j] * diff[j]; }
      const float w = val[k] / (1.f + d2);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += w * diff[j];
    }
    for (int j = 0; j < D; ++j) out[n * D + j] = acc[j];
  }
}

// ------------------------------------------------------------------ greedy non-max suppression, one block
// order: candidate indices by descending score; overlaps (optional) [N, N] replaces the box IoU
__device__ __forceinline__ float box_iou(const float* a, const float* b) {
  const float ay1 = fminf(a[0], a[2]), ax1 = fminf(a[1], a[3]), ay2 = fmaxf(a[0], a[2]), ax2 = fmaxf(a[1], a[3]);
  const float by1 = fminf(b[0], b[2]), bx1 = fminf(b[1], b[3]), by2 = fmaxf(b[0], b[2]), bx2 = fmaxf(b[1], b[3]);
  const float iy = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f), ix = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f);
  const float inter = iy * ix, uni = (ay2 - ay1) * (ax2 - ax1) + (by2 - by1) * (bx2 - bx1) - inter;
  return uni > 0.f ? inter / fmaxf(uni, 1e-30f) : 0.f;
}

__global__ __launch_bounds__(256) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ overlaps,
                                                  const float* __restrict__ scores, const int64_t* __restrict__ order,
                                                  int* __restrict__ keep, int* __restrict__ count, int N, int max_out, float thr,
                                                  float score_thr) {
  __shared__ int nkeep, hit;
  if (threadIdx.x == 0) nkeep = 0;
  __syncthreads();
  for (int c = 0; c < N; ++c) {
    const int k = (int)order[c];
    if (nkeep >= max_out || !(scores[k] > score_thr)) break;       // block-uniform (scores are sorted descending)
    if (threadIdx.x == 0) hit = 0;
    __syncthreads();
    const int nk = nkeep;
    for (int j = threadIdx.x; j < nk; j += blockDim.x) {
      const int q = keep[j];
      const float ov = overlaps ? overlaps[(int64_t)k * N + q] : box_iou(boxes + (int64_t)k * 4, boxes + (int64_t)q * 4);
      if (ov > thr) hit = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && !hit) { keep[nk] = k; nkeep = nk + 1; }
    __syncthreads();
  }
  if (threadIdx.x == 0) count[0] = nkeep;
}

}  // namespace

#define DISPATCH_T(...)                             \
  if (dtype == 1) { typedef bf16_t T; __VA_ARGS__ } \
  else { typedef float T; __VA_ARGS__ }
#define DISPATCH_I(...)                                 \
  if (idx64) { typedef int64_t I; __VA_ARGS__ }         \
  else { typedef int32_t I; __VA_ARGS__ }
#define DISPATCH_BYTES(esize, ...)                                  \
  switch (esize) {                                                  \
    case 1: { typedef uint8_t T; __VA_ARGS__ } break;               \
    case 2: { typedef uint16_t T; __VA_ARGS__ } break;              \
    case 4: { typedef uint32_t T; __VA_ARGS__ } break;              \
    case 8: { typedef uint64_t T; __VA_ARGS__ } break;              \
    default: return 1;                                              \
  }

// out: fp32 [nseg, D] pre-filled with the identity of `mode` (1 max, 2 min, 3 prod)
DL4J_API int dl4j_segment_reduce(const void* x, const void* seg, float* out, int64_t rows, int D, int64_t nseg, int mode, int dtype,
                                 int idx64, hipStream_t st) {
  if (rows <= 0 || D <= 0) return 0;
  if (mode < 1 || mode > 3) return 1;
  const int g = grid_for(rows * D, 1024);
  DISPATCH_T(DISPATCH_I(hipLaunchKernelGGL((segment_reduce_kernel<T, I>), dim3(g), dim3(256), 0, st, (const T*)x, (const I*)seg, out, rows,
                                           D, nseg, mode);))
  DL4J_RET_LAST();
}

DL4J_API int dl4j_nd_flatten(const void* idx, int64_t* flat, int64_t n, int k, const int64_t* shape, int idx64, hipStream_t st) {
  if (n <= 0) return 0;
  if (k < 1 || k > 8) return 1;
  NdShape s;
  for (int j = 0; j < 8; ++j) s.dim[j] = j < k ? shape[j] : 1;
  DISPATCH_I(hipLaunchKernelGGL(nd_flatten_kernel<I>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const I*)idx, flat, n, k, s);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_scatter_set_rows(const void* src, const int64_t* idx, void* table, int64_t n_idx, int D, int64_t V, int esize,
                                   hipStream_t st) {
  if (n_idx <= 0 || D <= 0) return 0;
  const int g = grid_for(n_idx * D, 1024);
  DISPATCH_BYTES(esize, hipLaunchKernelGGL(scatter_set_rows_kernel<T>, dim3(g), dim3(256), 0, st, (const T*)src, idx, (T*)table, n_idx, D,
                                           V);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_top_k(const void* x, void* vals, int64_t* inds, int64_t rows, int cols, int k, int dtype, hipStream_t st) {
  if (rows <= 0 || cols <= 0 || k <= 0) return 0;
  if (k > cols) return 1;
  const int64_t g = (rows + 3) / 4;
  if (g > 0x7fffffff) return 1;
  DISPATCH_T(hipLaunchKernelGGL(top_k_kernel<T>, dim3((unsigned)g), dim3(256), 0, st, (const T*)x, (T*)vals, inds, rows, cols, k);)
  DL4J_RET_LAST();
}

// counts: int64 [bins], zeroed by the caller
DL4J_API int dl4j_histogram(const void* x, long long* counts, int64_t n, int bins, float lo, float hi, int dtype, hipStream_t st) {
  if (n <= 0 || bins <= 0) return 0;
  if (!(hi > lo)) return 1;
  DISPATCH_T(hipLaunchKernelGGL(hist_kernel<T>, dim3(grid_for(n, 4096)), dim3(256), 0, st, (const T*)x, counts, n, bins, lo, hi);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_lrn_bwd(const void* x, const void* g, void* dx, int64_t M, int C, int depth, float bias, float alpha, float beta,
                          int dtype, hipStream_t st) {
  if (M <= 0 || C <= 0) return 0;
  if (C > LRN_MAX_C) return 1;
  const int grid = (int)(M < DL4J_MAX_GRID ? M : DL4J_MAX_GRID);
  DISPATCH_T(hipLaunchKernelGGL(lrn_bwd_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)x, (const T*)g, (T*)dx, M, C, depth, bias, alpha,
                                beta);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_sru_fwd(const float* x, const float* z, const float* bias, const float* c0, float* h, float* cs, int B, int K, int T,
                          hipStream_t st) {
  if (B <= 0 || K <= 0 || T <= 0) return 0;
  hipLaunchKernelGGL(sru_fwd_kernel, dim3(grid_for((int64_t)B * K, 64)), dim3(64), 0, st, x, z, bias, c0, h, cs, B, K, T);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_sru_bwd(const float* x, const float* z, const float* bias, const float* c0, const float* cs, const float* gh,
                          const float* gc_last, float* dz, float* dxd, float* dc0, int B, int K, int T, hipStream_t st) {
  if (B <= 0 || K <= 0 || T <= 0) return 0;
  hipLaunchKernelGGL(sru_bwd_kernel, dim3(grid_for((int64_t)B * K, 64)), dim3(64), 0, st, x, z, bias, c0, cs, gh, gc_last, dz, dxd, dc0, B,
                     K, T);
  DL4J_RET_LAST();
}

// x, y: [n, 3] channel-last pixels; mode 0 rgb->hsv, 1 hsv->rgb, 2 matrix (m: 9 host floats, row-major), 3 adjust_hue and
// 4 adjust_saturation (m[0] = delta / factor)
DL4J_API int dl4j_color_convert(const void* x, void* y, int64_t n, int mode, const float* m, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  if (mode < 0 || mode > 4 || (mode >= 2 && !m)) return 1;
  Mat3 M;
  for (int j = 0; j < 9; ++j) M.m[j] = (mode == 2 || (mode > 2 && j == 0)) ? m[j] : 0.f;
  DISPATCH_T(hipLaunchKernelGGL(color_kernel<T>, dim3(grid_for(n, 512)), dim3(256), 0, st, (const T*)x, (T*)y, n, mode, M);)
  DL4J_RET_LAST();
}

// keep: int32 [max_out]; count: int32 [1]
DL4J_API int dl4j_nms(const float* boxes, const float* overlaps, const float* scores, const int64_t* order, int* keep, int* count, int N,
                      int max_out, float thr, float score_thr, hipStream_t st) {
  if (N <= 0 || max_out <= 0) return 1;
  if (!boxes && !overlaps) return 1;
  hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(256), 0, st, boxes, overlaps, scores, order, keep, count, N, max_out, thr, score_thr);
  DL4J_RET_LAST();
}

// rowP [N+1], colP / val [nnz] (CSR), y / out fp32 [N, D], D <= 8
DL4J_API int dl4j_barnes_edge_forces(const void* rowP, const void* colP, const float* val, const float* y, float* out, int N, int D,
                                     int64_t nnz, int idx64, hipStream_t st) {
  if (N <= 0) return 0;
  if (D < 1 || D > 8) return 1;
  DISPATCH_I(hipLaunchKernelGGL(barnes_edge_forces_kernel<I>, dim3(grid_for(N, 256)), dim3(256), 0, st, (const I*)rowP, (const I*)colP,
                                val, y, out, N, D, nnz);)
  DL4J_RET_LAST();
}
