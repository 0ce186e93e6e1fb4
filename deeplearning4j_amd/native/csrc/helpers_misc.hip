// Declarable-op helper kernels, third family (gfx950): order statistics, row sort, vol2col / col2vol, generators
// (range / lin_space / diag / meshgrid), TF-exact bicubic and area image resize, adjust_contrast, draw_bounding_boxes,
// the t-SNE gains and CSR symmetrisation steps, and the batched skip-gram / CBOW training step with hierarchical
// softmax and negative sampling.
//
// Reference counterparts (libnd4j/include/ops/declarable/helpers/cuda): nth_element.cu, percentile.cu, sort (specials_cuda
// bitonic / oes sort), convolutions_vol2col.cu / convolutions_col2vol.cu, range.cu, diag.cu, meshgrid.cu, image_resize.cu
// (bicubic / area), adjust_contrast (imagesHelpers), image_draw_bounding_boxes.cu, BarnesHutTsne.cu (gains, symmetrize),
// sg_cb.cu (skipgramBatchExec / cbowBatchExec).
// Own design notes:
//  * nth_element is a 4-pass byte radix SELECT on order-preserving keys (one workgroup per row, LDS digit histogram):
//    O(4 N) reads and no sort, where the reference sorts every TAD and picks element n.
//  * row sort is a bitonic network held entirely in LDS (keys + original positions, up to 8192 per row), stable through
//    the (key, position) total order.
//  * vol2col is a pure gather, col2vol the matching gather over the contributing kernel taps (deterministic, no atomics).
//  * bicubic follows the TensorFlow kernel bit for bit in structure: 1024-entry quantised Keys coefficients (A = -0.75, or
//    -0.5 with half-pixel centres and weight renormalisation at the borders); area averages the exact fractional footprint.
//  * symmetrisation = count kernel + host-side exclusive scan + cursor fill + per-row insertion sort, which makes the
//    column order deterministic (ascending) although the fill uses atomics.
//  * skip-gram / CBOW: one 64-lane wave per batch item keeps the hidden vector and its error in registers, walks the Huffman
//    code points, then the negative samples (same LCG as the reference so the sample sequence is reproducible), and writes
//    syn1 / syn1neg / syn0 with fp32 atomics (hogwild across items, as in the reference's parallel batch).
#include "common.h"
#include <string.h>

namespace {

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ order-preserving float keys
__device__ __forceinline__ uint32_t f2key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// n-th smallest (0-based) element of every row; reverse handled by the caller (n -> cols-1-n)
template <typename T>
__global__ __launch_bounds__(256) void nth_element_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t rows, int cols, int n) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_rank;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const T* xr = x + row * cols;
    uint32_t prefix = 0u, mask = 0u, rank = (uint32_t)n;
    for (int shift = 24; shift >= 0; shift -= 8) {
      hist[threadIdx.x] = 0u;
      __syncthreads();
      for (int c = threadIdx.x; c < cols; c += 256) {
        const uint32_t k = f2key(Cvt<T>::ld(xr, c));
        if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned acc = 0u;
        int dsel = 255;
        for (int dgt = 0; dgt < 256; ++dgt) {
          const unsigned h = hist[dgt];
          if (acc + h > rank) { dsel = dgt; break; }
          acc += h;
        }
        s_prefix = prefix | ((uint32_t)dsel << shift);
        s_rank = rank - acc;
      }
      __syncthreads();
      prefix = s_prefix;
      rank = s_rank;
      mask |= 255u << shift;
      __syncthreads();
    }
    if (threadIdx.x == 0) Cvt<T>::st(out, row, key2f(prefix));
  }
}

// ------------------------------------------------------------------ per-row bitonic sort in LDS (values + original index)
// dynamic LDS: P keys (uint32) + P positions (int32), P = next power of two >= cols
template <typename T>
__global__ __launch_bounds__(256) void sort_rows_kernel(const T* __restrict__ x, T* __restrict__ vals, int64_t* __restrict__ inds, int64_t rows,
                                                        int cols, int P, int descending) {
  extern __shared__ unsigned char smem_raw[];
  uint32_t* key = (uint32_t*)smem_raw;
  int* pos = (int*)(key + P);
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const T* xr = x + row * cols;
    for (int c = threadIdx.x; c < P; c += 256) {
      if (c < cols) {
        const uint32_t k = f2key(Cvt<T>::ld(xr, c));
        key[c] = descending ? ~k : k;
        pos[c] = c;
      } else {
        key[c] = 0xffffffffu;        // padding sorts last in both directions
        pos[c] = 0x7fffffff;
      }
    }
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < P; t += 256) {
          const int p = t ^ j;
          if (p > t) {
            const bool up = (t & k2) == 0;
            const uint32_t ka = key[t], kb = key[p];
            const int ia = pos[t], ib = pos[p];
            const bool a_after_b = ka > kb || (ka == kb && ia > ib);
            if (a_after_b == up) { key[t] = kb; key[p] = ka; pos[t] = ib; pos[p] = ia; }
          }
        }
        __syncthreads();
      }
    }
    for (int c = threadIdx.x; c < cols; c += 256) {
      const uint32_t k = descending ? ~key[c] : key[c];
      if (vals) Cvt<T>::st(vals, row * cols + c, key2f(k));
      if (inds) inds[row * cols + c] = pos[c];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ vol2col / col2vol (NCDHW)
struct Vol {
  int N, C, D, H, W, kD, kH, kW, oD, oH, oW, sD, sH, sW, pD, pH, pW, dD, dH, dW;
};

template <typename T>
__global__ __launch_bounds__(256) void vol2col_kernel(const T* __restrict__ x, T* __restrict__ col, Vol v, int64_t total) {
  GRID_STRIDE(i, total) {
    int64_t r = i;
    const int ow = (int)(r % v.oW); r /= v.oW;
    const int oh = (int)(r % v.oH); r /= v.oH;
    const int od = (int)(r % v.oD); r /= v.oD;
    const int kw = (int)(r % v.kW); r /= v.kW;
    const int kh = (int)(r % v.kH); r /= v.kH;
    const int kd = (int)(r % v.kD); r /= v.kD;
    const int c = (int)(r % v.C);
    const int n = (int)(r / v.C);
    const int d = od * v.sD - v.pD + kd * v.dD, h = oh * v.sH - v.pH + kh * v.dH, w = ow * v.sW - v.pW + kw * v.dW;
    T val = T(0);
    if (d >= 0 && d < v.D && h >= 0 && h < v.H && w >= 0 && w < v.W) val = x[((((int64_t)n * v.C + c) * v.D + d) * v.H + h) * v.W + w];
    col[i] = val;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void col2vol_kernel(const T* __restrict__ col, T* __restrict__ x, Vol v, int64_t total) {
  GRID_STRIDE(i, total) {
    int64_t r = i;
    const int w = (int)(r % v.W); r /= v.W;
    const int h = (int)(r % v.H); r /= v.H;
    const int d = (int)(r % v.D); r /= v.D;
    const int c = (int)(r % v.C);
    const int n = (int)(r / v.C);
    float acc = 0.f;
    for (int kd = 0; kd < v.kD; ++kd) {
      const int td = d + v.pD - kd * v.dD;
      if (td < 0 || td % v.sD) continue;
      const int od = td / v.sD;
      if (od >= v.oD) continue;
      for (int kh = 0; kh < v.kH; ++kh) {
        const int th = h + v.pH - kh * v.dH;
        if (th < 0 || th % v.sH) continue;
        const int oh = th / v.sH;
        if (oh >= v.oH) continue;
        for (int kw = 0; kw < v.kW; ++kw) {
          const int tw = w + v.pW - kw * v.dW;
          if (tw < 0 || tw % v.sW) continue;
          const int ow = tw / v.sW;
          if (ow >= v.oW) continue;
          const int64_t ci = (((((((int64_t)n * v.C + c) * v.kD + kd) * v.kH + kh) * v.kW + kw) * v.oD + od) * v.oH + oh) * v.oW + ow;
          acc += Cvt<T>::ld(col, ci);
        }
      }
    }
    Cvt<T>::st(x, i, acc);
  }
}

// ------------------------------------------------------------------ generators
// dtype codes of the generic kernels: 0 f32, 1 bf16, 2 f64, 3 i32, 4 i64
__global__ __launch_bounds__(256) void fill_range_kernel(void* __restrict__ out, int64_t n, double start, double step, int dtype) {
  GRID_STRIDE(i, n) {
    const double v = start + step * (double)i;
    switch (dtype) {
      case 0: ((float*)out)[i] = (float)v; break;
      case 1: ((bf16_t*)out)[i] = f2bf((float)v); break;
      case 2: ((double*)out)[i] = v; break;
      case 3: ((int32_t*)out)[i] = (int32_t)v; break;
      default: ((int64_t*)out)[i] = (int64_t)v; break;
    }
  }
}
// integer ranges keep exact 64-bit arithmetic
__global__ __launch_bounds__(256) void fill_range_i64_kernel(void* __restrict__ out, int64_t n, int64_t start, int64_t step, int dtype) {
  GRID_STRIDE(i, n) {
    const int64_t v = start + step * i;
    if (dtype == 3) ((int32_t*)out)[i] = (int32_t)v;
    else ((int64_t*)out)[i] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void diag_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t n) {
  const int64_t total = n * n;
  GRID_STRIDE(i, total) {
    const int64_t r = i / n, c = i - r * n;
    out[i] = r == c ? x[r] : T(0);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void diag_part_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t n) {
  GRID_STRIDE(i, n) out[i] = x[i * (n + 1)];
}
// out[idx] = x[(idx / inner) % len]
template <typename T>
__global__ __launch_bounds__(256) void meshgrid_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t total, int64_t inner, int64_t len) {
  GRID_STRIDE(i, total) out[i] = x[(i / inner) % len];
}

// ------------------------------------------------------------------ TF-exact bicubic resize (NHWC in, fp32 out)
#define CUBIC_TABLE 1024
__device__ __forceinline__ float cubic_w_near(int off, float A) {   // |x| <= 1
  const float x = (float)off / CUBIC_TABLE;
  return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
}
__device__ __forceinline__ float cubic_w_far(int off, float A) {    // 1 < |x| < 2 (x = off/1024 + 1)
  const float x = (float)off / CUBIC_TABLE + 1.f;
  return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
__device__ __forceinline__ int bound_i(int64_t v, int lim) { return (int)(v < 0 ? 0 : (v > lim - 1 ? lim - 1 : v)); }

__device__ __forceinline__ void cubic_taps(int out_loc, float scale, int limit, bool half_pixel, int idx[4], float w[4]) {
  if (half_pixel) {
    const float A = -0.5f;
    const float loc = ((float)out_loc + 0.5f) * scale - 0.5f;
    const int64_t in_loc = (int64_t)floorf(loc);
    const float delta = loc - (float)in_loc;
    const int off = (int)lrintf(delta * CUBIC_TABLE);
    idx[0] = bound_i(in_loc - 1, limit); w[0] = idx[0] == in_loc - 1 ? cubic_w_far(off, A) : 0.f;
    idx[1] = bound_i(in_loc, limit);     w[1] = idx[1] == in_loc ? cubic_w_near(off, A) : 0.f;
    idx[2] = bound_i(in_loc + 1, limit); w[2] = idx[2] == in_loc + 1 ? cubic_w_near(CUBIC_TABLE - off, A) : 0.f;
    idx[3] = bound_i(in_loc + 2, limit); w[3] = idx[3] == in_loc + 2 ? cubic_w_far(CUBIC_TABLE - off, A) : 0.f;
    const float s = w[0] + w[1] + w[2] + w[3];
    if (fabsf(s) >= 1000.f * 1.17549435e-38f) {
      const float inv = 1.f / s;
      w[0] *= inv; w[1] *= inv; w[2] *= inv; w[3] *= inv;
    }
  } else {
    const float A = -0.75f;
    const float loc = (float)out_loc * scale;
    const int64_t in_loc = (int64_t)floorf(loc);
    const float delta = loc - (float)in_loc;
    const int off = (int)lrintf(delta * CUBIC_TABLE);
    idx[0] = bound_i(in_loc - 1, limit); w[0] = cubic_w_far(off, A);
    idx[1] = bound_i(in_loc, limit);     w[1] = cubic_w_near(off, A);
    idx[2] = bound_i(in_loc + 1, limit); w[2] = cubic_w_near(CUBIC_TABLE - off, A);
    idx[3] = bound_i(in_loc + 2, limit); w[3] = cubic_w_far(CUBIC_TABLE - off, A);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void resize_bicubic_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int oH,
                                                             int oW, float sh, float sw, int half_pixel) {
  const int64_t total = (int64_t)N * oH * oW * C;
  GRID_STRIDE(i, total) {
    int64_t r = i;
    const int c = (int)(r % C); r /= C;
    const int ox = (int)(r % oW); r /= oW;
    const int oy = (int)(r % oH);
    const int n = (int)(r / oH);
    int iy[4], ix[4];
    float wy[4], wx[4];
    cubic_taps(oy, sh, H, half_pixel != 0, iy, wy);
    cubic_taps(ox, sw, W, half_pixel != 0, ix, wx);
    const T* xn = x + (int64_t)n * H * W * C;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) row += wx[b] * Cvt<T>::ld(xn, ((int64_t)iy[a] * W + ix[b]) * C + c);
      acc += wy[a] * row;
    }
    y[i] = acc;
  }
}

// ------------------------------------------------------------------ TF-exact area resize (NHWC in, fp32 out)
template <typename T>
__global__ __launch_bounds__(256) void resize_area_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int oH,
                                                          int oW, float sh, float sw) {
  const int64_t total = (int64_t)N * oH * oW * C;
  const float inv = 1.f / (sh * sw);
  GRID_STRIDE(i, total) {
    int64_t r = i;
    const int c = (int)(r % C); r /= C;
    const int ox = (int)(r % oW); r /= oW;
    const int oy = (int)(r % oH);
    const int n = (int)(r / oH);
    const float y0 = oy * sh, y1 = (oy + 1) * sh, x0 = ox * sw, x1 = (ox + 1) * sw;
    const int64_t ys = (int64_t)floorf(y0), ye = (int64_t)ceilf(y1), xs = (int64_t)floorf(x0), xe = (int64_t)ceilf(x1);
    const T* xn = x + (int64_t)n * H * W * C;
    float acc = 0.f;
    for (int64_t yy = ys; yy < ye; ++yy) {
      const float wy = yy < y0 ? (yy + 1 > y1 ? sh : yy + 1 - y0) : (yy + 1 > y1 ? y1 - yy : 1.f);
      const int by = bound_i(yy, H);
      float row = 0.f;
      for (int64_t xx = xs; xx < xe; ++xx) {
        const float wx = xx < x0 ? (xx + 1 > x1 ? sw : xx + 1 - x0) : (xx + 1 > x1 ? x1 - xx : 1.f);
        row += wx * Cvt<T>::ld(xn, ((int64_t)by * W + bound_i(xx, W)) * C + c);
      }
      acc += wy * row;
    }
    y[i] = acc * inv;
  }
}

// ------------------------------------------------------------------ adjust_contrast: per (image, channel) mean, then (x - m) * f + m
// sums: [N, C] fp32, zeroed by the caller. grid (splits, N); only the first `act` threads (a multiple of C) work so that
// each thread keeps one channel.
template <typename T>
__global__ __launch_bounds__(256) void chan_sum_kernel(const T* __restrict__ x, float* __restrict__ sums, int64_t per_img, int C) {
  __shared__ float red[256];
  const int act = 256 - (256 % C);
  const int n = blockIdx.y;
  const T* xn = x + (int64_t)n * per_img;
  float a = 0.f;
  if ((int)threadIdx.x < act)
    for (int64_t i = (int64_t)blockIdx.x * act + threadIdx.x; i < per_img; i += (int64_t)gridDim.x * act) a += Cvt<T>::ld(xn, i);
  red[threadIdx.x] = a;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    float s = 0.f;
    for (int t = threadIdx.x; t < act; t += C) s += red[t];
    atomicAdd(sums + (int64_t)n * C + threadIdx.x, s);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void contrast_apply_kernel(const T* __restrict__ x, const float* __restrict__ sums, T* __restrict__ y,
                                                             int64_t total, int64_t per_img, int C, float inv_hw, float factor) {
  GRID_STRIDE(i, total) {
    const int64_t n = i / per_img;
    const int c = (int)(i % C);
    const float m = sums[n * C + c] * inv_hw;
    Cvt<T>::st(y, i, (Cvt<T>::ld(x, i) - m) * factor + m);
  }
}

// ------------------------------------------------------------------ draw_bounding_boxes (NHWC), later boxes overwrite earlier ones
template <typename T>
__global__ __launch_bounds__(256) void draw_boxes_kernel(const T* __restrict__ img, const float* __restrict__ boxes, const float* __restrict__ colors,
                                                         T* __restrict__ out, int N, int H, int W, int C, int B, int ncol) {
  const int64_t total = (int64_t)N * H * W;
  GRID_STRIDE(p, total) {
    int64_t r = p;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    int hit = -1;
    for (int k = 0; k < B; ++k) {
      const float* bx = boxes + ((int64_t)n * B + k) * 4;
      const int r1 = (int)(bx[0] * (H - 1)), c1 = (int)(bx[1] * (W - 1)), r2 = (int)(bx[2] * (H - 1)), c2 = (int)(bx[3] * (W - 1));
      if (r1 > r2 || c1 > c2 || r1 >= H || c1 >= W || r2 < 0 || c2 < 0) continue;
      const bool in_rows = h >= r1 && h <= r2, in_cols = w >= c1 && w <= c2;
      if ((in_cols && (h == r1 || h == r2)) || (in_rows && (w == c1 || w == c2))) hit = k;
    }
    for (int c = 0; c < C; ++c) {
      const int64_t o = p * C + c;
      if (hit < 0) out[o] = img[o];
      else Cvt<T>::st(out, o, colors[(int64_t)(hit % ncol) * C + c]);
    }
  }
}

// ------------------------------------------------------------------ t-SNE: gains update, CSR symmetrisation
__global__ __launch_bounds__(256) void barnes_gains_kernel(const float* __restrict__ gains, const float* __restrict__ gradx,
                                                           const float* __restrict__ eps, float* __restrict__ out, int64_t n) {
  GRID_STRIDE(i, n) {
    const float g = gradx[i], e = eps[i];
    const int sg = (g > 0.f) - (g < 0.f), se = (e > 0.f) - (e < 0.f);
    const float v = sg == se ? gains[i] * 0.8f : gains[i] + 0.2f;
    out[i] = v < 0.01f ? 0.01f : v;
  }
}

__device__ __forceinline__ int csr_find(const int* __restrict__ rowP, const int* __restrict__ colP, int row, int col) {
  for (int k = rowP[row]; k < rowP[row + 1]; ++k)
    if (colP[k] == col) return k;
  return -1;
}
// counts[n] = entries of row n of the symmetrised matrix (pattern of P + P^T)
__global__ __launch_bounds__(256) void sym_count_kernel(const int* __restrict__ rowP, const int* __restrict__ colP, int* __restrict__ counts, int N) {
  GRID_STRIDE(n, N) {
    for (int k = rowP[n]; k < rowP[n + 1]; ++k) {
      const int j = colP[k];
      if (j < 0 || j >= N) continue;
      atomicAdd(counts + n, 1);
      if (j != n && csr_find(rowP, colP, j, (int)n) < 0) atomicAdd(counts + j, 1);
    }
  }
}
__global__ __launch_bounds__(256) void sym_fill_kernel(const int* __restrict__ rowP, const int* __restrict__ colP, const float* __restrict__ val,
                                                       const int* __restrict__ symRowP, int* __restrict__ cursor, int* __restrict__ symCol,
                                                       float* __restrict__ symVal, int N) {
  GRID_STRIDE(n, N) {
    for (int k = rowP[n]; k < rowP[n + 1]; ++k) {
      const int j = colP[k];
      if (j < 0 || j >= N) continue;
      const int m = j == n ? k : csr_find(rowP, colP, j, (int)n);
      const float v = j == n ? val[k] : 0.5f * (val[k] + (m >= 0 ? val[m] : 0.f));
      const int s = symRowP[n] + atomicAdd(cursor + n, 1);
      symCol[s] = j;
      symVal[s] = v;
      if (j != n && m < 0) {                 // the transposed entry exists only through this one
        const int s2 = symRowP[j] + atomicAdd(cursor + j, 1);
        symCol[s2] = (int)n;
        symVal[s2] = v;
      }
    }
  }
}
__global__ __launch_bounds__(256) void sym_sort_kernel(const int* __restrict__ symRowP, int* __restrict__ symCol, float* __restrict__ symVal, int N) {
  GRID_STRIDE(n, N) {
    const int b = symRowP[n], e = symRowP[n + 1];
    for (int i = b + 1; i < e; ++i) {
      const int c = symCol[i];
      const float v = symVal[i];
      int k = i - 1;
      while (k >= b && symCol[k] > c) { symCol[k + 1] = symCol[k]; symVal[k + 1] = symVal[k]; --k; }
      symCol[k + 1] = c;
      symVal[k + 1] = v;
    }
  }
}

// ------------------------------------------------------------------ batched skip-gram / CBOW step (hs + ns)
#define SG_MAX_EXP 6.0f
#define SG_MAXP 16          // per-lane slice of the vector: D <= 64 * 16
// fully unrolled so h / e / wv stay in registers (a runtime trip count would index them dynamically -> scratch)
#define FOR_J _Pragma("unroll") for (int j = 0; j < SG_MAXP; ++j) if (j < P)
struct SgArgs {
  float* syn0; float* syn1; float* syn1neg; const int* negTable;
  const int* target; const int* ngStarter; const int* context; const int* indices; const signed char* codes;
  const float* alpha; const int64_t* rnd; float* infer;
  int B, D, V, codeLen, ctxW, nsRounds, negLen, trainWords, cbow;
};

__global__ __launch_bounds__(256) void sg_cb_batch_kernel(SgArgs a) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= a.B) return;
  const int D = a.D, P = (D + 63) >> 6;
  float h[SG_MAXP], e[SG_MAXP];
#pragma unroll
  for (int j = 0; j < SG_MAXP; ++j) { h[j] = 0.f; e[j] = 0.f; }
  const int tgt = a.target[item];
  int nctx = 0;
  if (a.infer) {
    FOR_J { const int c = lane + 64 * j; if (c < D) h[j] = a.infer[(int64_t)item * D + c]; }
  } else if (a.cbow) {
    for (int k = 0; k < a.ctxW; ++k) {
      const int w = a.context[(int64_t)item * a.ctxW + k];
      if (w < 0 || w >= a.V) continue;
      ++nctx;
      FOR_J { const int c = lane + 64 * j; if (c < D) h[j] += a.syn0[(int64_t)w * D + c]; }
    }
    if (nctx == 0) return;
    const float inv = 1.f / (float)nctx;
    FOR_J h[j] *= inv;
  } else {
    if (tgt < 0 || tgt >= a.V) return;
    FOR_J { const int c = lane + 64 * j; if (c < D) h[j] = a.syn0[(int64_t)tgt * D + c]; }
  }
  const float lr = a.alpha[item];
  // hierarchical softmax over the Huffman path
  if (a.syn1 && a.indices) {
    for (int k = 0; k < a.codeLen; ++k) {
      const int row = a.indices[(int64_t)item * a.codeLen + k];
      if (row < 0 || row >= a.V) continue;
      float* w = a.syn1 + (int64_t)row * D;
      float wv[SG_MAXP];
      float dot = 0.f;
      FOR_J { const int c = lane + 64 * j; wv[j] = c < D ? w[c] : 0.f; dot += h[j] * wv[j]; }
      dot = wave_sum(dot);
      if (dot < -SG_MAX_EXP || dot >= SG_MAX_EXP) continue;
      const float f = 1.f / (1.f + __expf(-dot));
      const float g = (1.f - (float)a.codes[(int64_t)item * a.codeLen + k] - f) * lr;
      FOR_J {
        const int c = lane + 64 * j;
        e[j] += g * wv[j];
        if (c < D && !a.infer) atomicAdd(w + c, g * h[j]);
      }
    }
  }
  // negative sampling: round 0 is the positive sample, the others follow the reference LCG
  if (a.syn1neg && a.nsRounds > 0) {
    const int starter = a.ngStarter[item];
    uint64_t r = (uint64_t)a.rnd[item];
    for (int s = 0; s <= a.nsRounds; ++s) {
      int row = starter;
      if (s > 0) {
        r = r * 25214903917ull + 11ull;
        row = a.negLen > 0 ? a.negTable[(r >> 16) % (uint64_t)a.negLen] : (int)((r >> 16) % (uint64_t)a.V);
        if (row == starter) continue;
      }
      if (row < 0 || row >= a.V) continue;
      float* w = a.syn1neg + (int64_t)row * D;
      float wv[SG_MAXP];
      float dot = 0.f;
      FOR_J { const int c = lane + 64 * j; wv[j] = c < D ? w[c] : 0.f; dot += h[j] * wv[j]; }
      dot = wave_sum(dot);
      const float label = s == 0 ? 1.f : 0.f;
      float g;
      if (dot > SG_MAX_EXP) g = (label - 1.f) * lr;
      else if (dot < -SG_MAX_EXP) g = label * lr;
      else g = (label - 1.f / (1.f + __expf(-dot))) * lr;
      FOR_J {
        const int c = lane + 64 * j;
        e[j] += g * wv[j];
        if (c < D && !a.infer) atomicAdd(w + c, g * h[j]);
      }
    }
  }
  // input side
  if (a.infer) {
    FOR_J { const int c = lane + 64 * j; if (c < D) a.infer[(int64_t)item * D + c] = h[j] + e[j]; }
  } else if (a.cbow) {
    if (a.trainWords)
      for (int k = 0; k < a.ctxW; ++k) {
        const int w = a.context[(int64_t)item * a.ctxW + k];
        if (w < 0 || w >= a.V) continue;
        FOR_J { const int c = lane + 64 * j; if (c < D) atomicAdd(a.syn0 + (int64_t)w * D + c, e[j]); }
      }
  } else {
    FOR_J { const int c = lane + 64 * j; if (c < D) atomicAdd(a.syn0 + (int64_t)tgt * D + c, e[j]); }
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

DL4J_API int dl4j_nth_element(const void* x, void* out, int64_t rows, int cols, int n, int dtype, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  if (n < 0 || n >= cols) return 1;
  const int g = (int)(rows < DL4J_MAX_GRID ? rows : DL4J_MAX_GRID);
  DISPATCH_T(hipLaunchKernelGGL(nth_element_kernel<T>, dim3(g), dim3(256), 0, st, (const T*)x, (T*)out, rows, cols, n);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_sort_rows_max_cols() { return 8192; }

// vals and/or inds may be null; rows of up to dl4j_sort_rows_max_cols() elements
DL4J_API int dl4j_sort_rows(const void* x, void* vals, int64_t* inds, int64_t rows, int cols, int descending, int dtype, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  if (cols > 8192) return 1;
  int P = 2;
  while (P < cols) P <<= 1;
  const int lds = P * 8;
  const int g = (int)(rows < DL4J_MAX_GRID ? rows : DL4J_MAX_GRID);
  DISPATCH_T(hipLaunchKernelGGL(sort_rows_kernel<T>, dim3(g), dim3(256), lds, st, (const T*)x, (T*)vals, inds, rows, cols, P, descending);)
  DL4J_RET_LAST();
}

// geom: 20 ints in the order of struct Vol
DL4J_API int dl4j_vol2col(const void* x, void* col, const int* geom, int esize, hipStream_t st) {
  Vol v;
  memcpy(&v, geom, sizeof(Vol));
  const int64_t total = (int64_t)v.N * v.C * v.kD * v.kH * v.kW * v.oD * v.oH * v.oW;
  if (total <= 0) return 0;
  if (v.sD <= 0 || v.sH <= 0 || v.sW <= 0) return 1;
  DISPATCH_BYTES(esize, hipLaunchKernelGGL(vol2col_kernel<T>, dim3(grid_for(total, 1024)), dim3(256), 0, st, (const T*)x, (T*)col, v, total);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_col2vol(const void* col, void* x, const int* geom, int dtype, hipStream_t st) {
  Vol v;
  memcpy(&v, geom, sizeof(Vol));
  const int64_t total = (int64_t)v.N * v.C * v.D * v.H * v.W;
  if (total <= 0) return 0;
  if (v.sD <= 0 || v.sH <= 0 || v.sW <= 0) return 1;
  DISPATCH_T(hipLaunchKernelGGL(col2vol_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)col, (T*)x, v, total);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_fill_range(void* out, int64_t n, double start, double step, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  if (dtype < 0 || dtype > 4) return 1;
  hipLaunchKernelGGL(fill_range_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, out, n, start, step, dtype);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_fill_range_i64(void* out, int64_t n, int64_t start, int64_t step, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  if (dtype != 3 && dtype != 4) return 1;
  hipLaunchKernelGGL(fill_range_i64_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, out, n, start, step, dtype);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_diag(const void* x, void* out, int64_t n, int esize, hipStream_t st) {
  if (n <= 0) return 0;
  DISPATCH_BYTES(esize, hipLaunchKernelGGL(diag_kernel<T>, dim3(grid_for(n * n, 1024)), dim3(256), 0, st, (const T*)x, (T*)out, n);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_diag_part(const void* x, void* out, int64_t n, int esize, hipStream_t st) {
  if (n <= 0) return 0;
  DISPATCH_BYTES(esize, hipLaunchKernelGGL(diag_part_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const T*)x, (T*)out, n);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_meshgrid_axis(const void* x, void* out, int64_t total, int64_t inner, int64_t len, int esize, hipStream_t st) {
  if (total <= 0) return 0;
  if (inner <= 0 || len <= 0) return 1;
  DISPATCH_BYTES(esize, hipLaunchKernelGGL(meshgrid_kernel<T>, dim3(grid_for(total, 1024)), dim3(256), 0, st, (const T*)x, (T*)out, total, inner,
                                           len);)
  DL4J_RET_LAST();
}

static inline float resize_scale(int in, int out, int align) { return (align && out > 1) ? (float)(in - 1) / (float)(out - 1) : (float)in / (float)out; }

// mode 0 = bicubic, 1 = area; y is fp32 [N, oH, oW, C]
DL4J_API int dl4j_resize_tf(const void* x, float* y, int N, int H, int W, int C, int oH, int oW, int mode, int align_corners, int half_pixel,
                            int dtype, hipStream_t st) {
  const int64_t total = (int64_t)N * oH * oW * C;
  if (total <= 0) return 0;
  if (H <= 0 || W <= 0) return 1;
  const float sh = resize_scale(H, oH, align_corners), sw = resize_scale(W, oW, align_corners);
  const int g = grid_for(total, 512);
  if (mode == 0) {
    DISPATCH_T(hipLaunchKernelGGL(resize_bicubic_kernel<T>, dim3(g), dim3(256), 0, st, (const T*)x, y, N, H, W, C, oH, oW, sh, sw, half_pixel);)
  } else {
    DISPATCH_T(hipLaunchKernelGGL(resize_area_kernel<T>, dim3(g), dim3(256), 0, st, (const T*)x, y, N, H, W, C, oH, oW, sh, sw);)
  }
  DL4J_RET_LAST();
}

// sums: [N, C] fp32 scratch, zeroed by the caller
DL4J_API int dl4j_adjust_contrast(const void* x, void* y, float* sums, int N, int64_t HW, int C, float factor, int dtype, hipStream_t st) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  if (C > 256) return 1;
  const int64_t per_img = HW * C;
  int splits = (int)((per_img + 16383) / 16384);
  splits = splits < 1 ? 1 : (splits > 64 ? 64 : splits);
  DISPATCH_T(hipLaunchKernelGGL(chan_sum_kernel<T>, dim3(splits, N), dim3(256), 0, st, (const T*)x, sums, per_img, C);
             hipLaunchKernelGGL(contrast_apply_kernel<T>, dim3(grid_for((int64_t)N * per_img, 1024)), dim3(256), 0, st, (const T*)x, sums, (T*)y,
                                (int64_t)N * per_img, per_img, C, 1.f / (float)HW, factor);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_draw_boxes(const void* img, const float* boxes, const float* colors, void* out, int N, int H, int W, int C, int B, int ncol,
                             int dtype, hipStream_t st) {
  const int64_t total = (int64_t)N * H * W;
  if (total <= 0) return 0;
  if (ncol <= 0 && B > 0) return 1;
  DISPATCH_T(hipLaunchKernelGGL(draw_boxes_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)img, boxes, colors, (T*)out, N, H, W,
                                C, B, ncol);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_barnes_gains(const float* gains, const float* gradx, const float* eps, float* out, int64_t n, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(barnes_gains_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, gains, gradx, eps, out, n);
  DL4J_RET_LAST();
}

// counts: int32 [N], zeroed by the caller
DL4J_API int dl4j_barnes_sym_count(const int* rowP, const int* colP, int* counts, int N, hipStream_t st) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(sym_count_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, rowP, colP, counts, N);
  DL4J_RET_LAST();
}

// symRowP: int32 [N+1] exclusive scan of the counts; cursor: int32 [N] zeroed by the caller
DL4J_API int dl4j_barnes_sym_fill(const int* rowP, const int* colP, const float* val, const int* symRowP, int* cursor, int* symCol,
                                  float* symVal, int N, hipStream_t st) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(sym_fill_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, rowP, colP, val, symRowP, cursor, symCol, symVal, N);
  hipLaunchKernelGGL(sym_sort_kernel, dim3(grid_for(N, 256)), dim3(256), 0, st, symRowP, symCol, symVal, N);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_sg_cb_max_dim() { return 64 * SG_MAXP; }

// One training step for B (target, context) items. Any of syn1 / syn1neg / negTable / context / indices / codes / infer may be null.
DL4J_API int dl4j_sg_cb_batch(float* syn0, float* syn1, float* syn1neg, const int* negTable, const int* target, const int* ngStarter,
                              const int* context, const int* indices, const signed char* codes, const float* alpha, const int64_t* rnd,
                              float* infer, int B, int D, int V, int codeLen, int ctxW, int nsRounds, int negLen, int trainWords, int cbow,
                              hipStream_t st) {
  if (B <= 0 || D <= 0) return 0;
  if (D > 64 * SG_MAXP || V <= 0) return 1;
  if (cbow && !context && !infer) return 1;
  SgArgs a{syn0, syn1, syn1neg, negTable, target, ngStarter, context, indices, codes, alpha, rnd, infer,
           B, D, V, codeLen, ctxW, nsRounds, negLen, trainWords, cbow};
  hipLaunchKernelGGL(sg_cb_batch_kernel, dim3((B + 3) / 4), dim3(256), 0, st, a);
  DL4J_RET_LAST();
}
