// Declarable-op helper kernels, data-movement / indexed / scan family (gfx950): gather, scatter-add, one-hot,
// cumsum, im2col / col2im, nearest upsampling fwd+bwd, N-ary merge, segment sums, image resize, and the batched
// skip-gram / CBOW negative-sampling step.
//
// Reference counterparts (libnd4j/include/ops/declarable/helpers/cuda): gather.cu, scatter.cu (lock-based duplicate
// handling -> here fp32 atomics), one_hot.cu, prefix.cu (prefixPerBlockCuda), im2col.cu, col2im.cu,
// convolutions_upsampling2d(BP).cu, merge.cu (pointer-array launchers), segment_sum.cu, image_resize.cu, sg_cb.cu.
// Own design notes: row-oriented kernels give one wave (64 lanes) a contiguous row so every access is a full
// coalesced line; scans use 64-lane shuffle prefix sums with a per-wave carry instead of a shared-memory
// Blelloch tree; col2im is a gather over the covering windows (deterministic, no atomics).
#include "common.h"

namespace {

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ gather / scatter-add rows of a [V, D] table
template <typename T, typename I>
__global__ void gather_rows_kernel(const T* __restrict__ table, const I* __restrict__ idx, T* __restrict__ out, int64_t n_idx, int D,
                                   int64_t V) {
  const int64_t total = n_idx * D;
  GRID_STRIDE(i, total) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    int64_t v = (int64_t)idx[r];
    if (v < 0) v += V;
    out[i] = (v >= 0 && v < V) ? table[v * D + c] : T(0);
  }
}

// rows that are whole 16-byte pieces: one thread moves one piece (one 32-bit division per 16 bytes; the index is read once per piece and
// hits the cache for the other pieces of the row)
template <typename I>
__global__ void __launch_bounds__(256) gather_rows_vec_kernel(const uint4* __restrict__ table, const I* __restrict__ idx, uint4* __restrict__ out,
                                                              uint32_t total_pieces, uint32_t ppr, int64_t V) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total_pieces; i += gridDim.x * 256u) {
    const uint32_t r = i / ppr, c = i - r * ppr;
    int64_t v = (int64_t)idx[r];
    if (v < 0) v += V;
    out[i] = (v >= 0 && v < V) ? table[v * ppr + c] : make_uint4(0u, 0u, 0u, 0u);
  }
}

// table[idx[r], :] += alpha * src[r, :]   (fp32 accumulate; duplicates are summed)
template <typename T, typename I>
__global__ void scatter_add_rows_kernel(const T* __restrict__ src, const I* __restrict__ idx, float* __restrict__ table, int64_t n_idx,
                                        int D, int64_t V, float alpha) {
  const int64_t total = n_idx * D;
  GRID_STRIDE(i, total) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    int64_t v = (int64_t)idx[r];
    if (v < 0) v += V;
    if (v >= 0 && v < V) atomicAdd(table + v * D + c, alpha * Cvt<T>::ld(src, i));
  }
}

template <typename T, typename I>
__global__ void one_hot_kernel(const I* __restrict__ idx, T* __restrict__ out, int64_t n, int depth, float on, float off) {
  const int64_t total = n * depth;
  GRID_STRIDE(i, total) {
    const int64_t r = i / depth;
    const int c = (int)(i - r * depth);
    Cvt<T>::st(out, i, (int64_t)idx[r] == c ? on : off);
  }
}

// rows that are whole 16-byte pieces: one thread writes one piece (V = 4 fp32 / 8 bf16 elements), one 32-bit division per piece
template <typename T, typename I>
__global__ void __launch_bounds__(256) one_hot_vec_kernel(const I* __restrict__ idx, T* __restrict__ out, uint32_t total_pieces, uint32_t ppr,
                                                          float on, float off) {
  constexpr int V = 16 / (int)sizeof(T);
  struct alignas(16) Vec { T v[V]; };
  T t_on, t_off;
  Cvt<T>::st(&t_on, 0, on);
  Cvt<T>::st(&t_off, 0, off);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total_pieces; i += gridDim.x * 256u) {
    const uint32_t r = i / ppr, c0 = (i - r * ppr) * V;
    const int64_t hot = (int64_t)idx[r] - (int64_t)c0;          // position of the hot element inside this piece, if any
    Vec o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = hot == j ? t_on : t_off;
    *reinterpret_cast<Vec*>(out + (int64_t)i * V) = o;
  }
}

// ------------------------------------------------------------------ cumsum along the innermost dim, one block (4 waves) per row
__device__ __forceinline__ float wave_incl_scan(float v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void cumsum_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int cols,
                                                          int exclusive, int reverse) {
  __shared__ float wsum[4];
  __shared__ float carry_s;
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  const T* xr = x + r * cols;
  T* yr = y + r * cols;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0.f;
  __syncthreads();
  for (int base = 0; base < cols; base += 256) {
    const int p = base + threadIdx.x;                 // position in scan order
    const int c = reverse ? cols - 1 - p : p;
    const float v = p < cols ? Cvt<T>::ld(xr, c) : 0.f;
    const float incl = wave_incl_scan(v);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    float off = carry_s;
    for (int w = 0; w < wid; ++w) off += wsum[w];
    if (p < cols) Cvt<T>::st(yr, c, (exclusive ? incl - v : incl) + off);
    __syncthreads();
    if (threadIdx.x == 0) carry_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

// ------------------------------------------------------------------ im2col / col2im, NCHW <-> [N, C, kH, kW, oH, oW]
struct ColP {
  int N, C, H, W, kH, kW, oH, oW, sH, sW, pH, pW, dH, dW;
};

template <typename T>
__global__ void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, ColP p, float pad_value) {
  const int64_t total = (int64_t)p.N * p.C * p.kH * p.kW * p.oH * p.oW;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int ow = (int)(t % p.oW); t /= p.oW;
    const int oh = (int)(t % p.oH); t /= p.oH;
    const int kw = (int)(t % p.kW); t /= p.kW;
    const int kh = (int)(t % p.kH); t /= p.kH;
    const int c = (int)(t % p.C);
    const int n = (int)(t / p.C);
    const int ih = oh * p.sH - p.pH + kh * p.dH, iw = ow * p.sW - p.pW + kw * p.dW;
    float v = pad_value;
    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) v = Cvt<T>::ld(x, (((int64_t)n * p.C + c) * p.H + ih) * p.W + iw);
    Cvt<T>::st(col, i, v);
  }
}

// gather formulation: each input pixel sums the (kh, kw, oh, ow) entries that cover it
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ col, T* __restrict__ x, ColP p) {
  const int64_t total = (int64_t)p.N * p.C * p.H * p.W;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int iw = (int)(t % p.W); t /= p.W;
    const int ih = (int)(t % p.H); t /= p.H;
    const int c = (int)(t % p.C);
    const int n = (int)(t / p.C);
    float s = 0.f;
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
        s += Cvt<T>::ld(col, (((((int64_t)n * p.C + c) * p.kH + kh) * p.kW + kw) * p.oH + oh) * p.oW + ow);
      }
    }
    Cvt<T>::st(x, i, s);
  }
}

// ------------------------------------------------------------------ nearest upsampling NCHW (planes = N*C)
template <typename T>
__global__ void upsample2d_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t planes, int H, int W, int fH, int fW) {
  const int oH = H * fH, oW = W * fW;
  const int64_t total = planes * oH * oW;
  GRID_STRIDE(i, total) {
    const int ow = (int)(i % oW);
    const int oh = (int)((i / oW) % oH);
    const int64_t pl = i / ((int64_t)oW * oH);
    y[i] = x[(pl * H + oh / fH) * W + ow / fW];
  }
}

// factor-2 width, rows that are whole 16-byte pieces: one thread reads one 16-byte piece of an input row and writes the two 16-byte
// pieces it expands to into each of the fH output rows (one 32-bit division per 16 bytes read instead of five 64-bit ones per element)
template <typename T>
__global__ void __launch_bounds__(256) upsample2d_fwd_w2_kernel(const T* __restrict__ x, T* __restrict__ y, uint32_t total_vec, int W, int fH) {
  constexpr int V = 16 / (int)sizeof(T);
  struct alignas(16) Vec { T v[V]; };
  const uint32_t vpr = (uint32_t)(W / V);
  const int64_t oW = (int64_t)W * 2;
  for (uint32_t iv = blockIdx.x * 256u + threadIdx.x; iv < total_vec; iv += gridDim.x * 256u) {
    const uint32_t row = iv / vpr, wv = iv - row * vpr;
    const Vec in = *reinterpret_cast<const Vec*>(x + (int64_t)iv * V);
    Vec lo, hi;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      lo.v[j] = in.v[j / 2];
      hi.v[j] = in.v[V / 2 + j / 2];
    }
    T* o = y + ((int64_t)row * fH) * oW + (int64_t)wv * V * 2;
    for (int a = 0; a < fH; ++a) {
      *reinterpret_cast<Vec*>(o + a * oW) = lo;
      *reinterpret_cast<Vec*>(o + a * oW + V) = hi;
    }
  }
}

template <typename T>
__global__ void upsample2d_bwd_kernel(const T* __restrict__ g, T* __restrict__ dx, int64_t planes, int H, int W, int fH, int fW) {
  const int oW = W * fW;
  const int64_t total = planes * H * W;
  GRID_STRIDE(i, total) {
    const int w = (int)(i % W);
    const int h = (int)((i / W) % H);
    const int64_t pl = i / ((int64_t)W * H);
    float s = 0.f;
    for (int a = 0; a < fH; ++a)
      for (int b = 0; b < fW; ++b) s += Cvt<T>::ld(g, (pl * H * fH + h * fH + a) * oW + w * fW + b);
    Cvt<T>::st(dx, i, s);
  }
}

// ------------------------------------------------------------------ N-ary merge over a device table of pointers: 0 add, 1 avg, 2 max
template <typename T>
__global__ void merge_kernel(const int64_t* __restrict__ ptrs, int k, T* __restrict__ out, int64_t n, int mode) {
  GRID_STRIDE(i, n) {
    float acc = Cvt<T>::ld((const T*)ptrs[0], i);
    for (int a = 1; a < k; ++a) {
      const float v = Cvt<T>::ld((const T*)ptrs[a], i);
      acc = mode == 2 ? fmaxf(acc, v) : acc + v;
    }
    Cvt<T>::st(out, i, mode == 1 ? acc / k : acc);
  }
}

// ------------------------------------------------------------------ segment sum: out[seg[r], :] += x[r, :]  (out fp32, pre-zeroed)
template <typename T, typename I>
__global__ void segment_sum_kernel(const T* __restrict__ x, const I* __restrict__ seg, float* __restrict__ out, int64_t rows, int D,
                                   int64_t nseg) {
  const int64_t total = rows * D;
  GRID_STRIDE(i, total) {
    const int64_t r = i / D;
    const int64_t s = (int64_t)seg[r];
    if (s >= 0 && s < nseg) atomicAdd(out + s * D + (i - r * D), Cvt<T>::ld(x, i));
  }
}

// ------------------------------------------------------------------ image resize NHWC: mode 0 nearest, 1 bilinear
template <typename T>
__global__ void resize_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int oH, int oW, int mode,
                                   int align_corners, int half_pixel) {
  const int64_t total = (int64_t)N * oH * oW * C;
  const float sh = align_corners && oH > 1 ? (float)(H - 1) / (oH - 1) : (float)H / oH;
  const float sw = align_corners && oW > 1 ? (float)(W - 1) / (oW - 1) : (float)W / oW;
  GRID_STRIDE(i, total) {
    int64_t t = i;
    const int c = (int)(t % C); t /= C;
    const int ow = (int)(t % oW); t /= oW;
    const int oh = (int)(t % oH);
    const int n = (int)(t / oH);
    const T* xn = x + (int64_t)n * H * W * C;
    if (mode == 0) {
      const float fh = half_pixel ? (oh + 0.5f) * sh : oh * sh, fw = half_pixel ? (ow + 0.5f) * sw : ow * sw;
      int ih = (int)(align_corners ? roundf(fh) : floorf(fh)), iw = (int)(align_corners ? roundf(fw) : floorf(fw));
      ih = ih < 0 ? 0 : (ih > H - 1 ? H - 1 : ih);
      iw = iw < 0 ? 0 : (iw > W - 1 ? W - 1 : iw);
      y[i] = xn[((int64_t)ih * W + iw) * C + c];
    } else {
      float fh = half_pixel ? (oh + 0.5f) * sh - 0.5f : oh * sh, fw = half_pixel ? (ow + 0.5f) * sw - 0.5f : ow * sw;
      fh = fh < 0.f ? 0.f : fh; fw = fw < 0.f ? 0.f : fw;
      int h0 = (int)floorf(fh), w0 = (int)floorf(fw);
      h0 = h0 > H - 1 ? H - 1 : h0; w0 = w0 > W - 1 ? W - 1 : w0;
      const int h1 = h0 + 1 > H - 1 ? H - 1 : h0 + 1, w1 = w0 + 1 > W - 1 ? W - 1 : w0 + 1;
      const float ah = fh - h0, aw = fw - w0;
      const float v00 = Cvt<T>::ld(xn, ((int64_t)h0 * W + w0) * C + c), v01 = Cvt<T>::ld(xn, ((int64_t)h0 * W + w1) * C + c);
      const float v10 = Cvt<T>::ld(xn, ((int64_t)h1 * W + w0) * C + c), v11 = Cvt<T>::ld(xn, ((int64_t)h1 * W + w1) * C + c);
      Cvt<T>::st(y, i, (v00 * (1 - aw) + v01 * aw) * (1 - ah) + (v10 * (1 - aw) + v11 * aw) * ah);
    }
  }
}

// ------------------------------------------------------------------ skip-gram / CBOW negative-sampling (or HS) batched step
// One wave per (input row, output row, label) triple: dot over D with a 64-lane reduction, then both rows are
// updated with fp32 atomics (duplicates inside a batch accumulate, which is what the sequential reference does).
//   f = sigmoid(<h, syn1[t]>);  g = (label - f) * lr;  neu1e += g * syn1[t];  syn1[t] += g * h
// `h` is syn0[in] (skip-gram) or a precomputed context mean (CBOW, `hidden` != null); input-side gradients are
// written to dh[pair, :] and scatter-added into syn0 by the caller (dl4j_scatter_add_rows).
__global__ __launch_bounds__(256) void sg_pairs_kernel(const float* __restrict__ syn0, float* __restrict__ syn1,
                                                       const float* __restrict__ hidden, const int64_t* __restrict__ in_idx,
                                                       const int64_t* __restrict__ tgt_idx, const float* __restrict__ label,
                                                       float* __restrict__ dh, int64_t n_pairs, int D, float lr, float scale,
                                                       int64_t V0, int64_t V1, int sat_mode) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t p = wave; p < n_pairs; p += nwaves) {
    const int64_t t = tgt_idx[p];
    const int64_t iv = hidden ? 0 : in_idx[p];
    if (t < 0 || t >= V1 || iv < 0 || iv >= V0) {          // padded / invalid slot: no update, zero input gradient
      for (int c = lane; c < D; c += 64) dh[p * D + c] = 0.f;
      continue;
    }
    const float* h = hidden ? hidden + p * D : syn0 + iv * D;
    float* w = syn1 + t * D;
    float dot = 0.f;
    for (int c = lane; c < D; c += 64) dot += h[c] * w[c];
    dot = wave_sum(dot);
    // the reference's exp-table bounds (MAX_EXP = 6, sg_cb.cu): a saturated hierarchic-softmax node is skipped, a saturated negative
    // sample is treated as exactly 0 / 1. Without the skip the batched hogwild update can run away (NaN after ~40 epochs on a toy corpus)
    float f = 1.f / (1.f + __expf(-dot));
    if (sat_mode == 1 && !(dot > -6.f && dot < 6.f)) {
      for (int c = lane; c < D; c += 64) dh[p * D + c] = 0.f;
      continue;
    }
    if (sat_mode == 2) f = dot > 6.f ? 1.f : (dot < -6.f ? 0.f : f);
    const float g = (label[p] - f) * lr * scale;
    for (int c = lane; c < D; c += 64) {
      const float wv = w[c], hv = h[c];
      dh[p * D + c] = g * wv;                              // this wave owns row p
      atomicAdd(w + c, g * hv);
    }
  }
}

}  // namespace

#define DISPATCH_T(...)                             \
  if (dtype == 1) { typedef bf16_t T; __VA_ARGS__ } \
  else { typedef float T; __VA_ARGS__ }
#define DISPATCH_I(...)                                 \
  if (idx64) { typedef int64_t I; __VA_ARGS__ }         \
  else { typedef int32_t I; __VA_ARGS__ }

// element size dispatch for pure copies (1/2/4/8 bytes)
#define DISPATCH_BYTES(esize, ...)                                  \
  switch (esize) {                                                  \
    case 1: { typedef uint8_t T; __VA_ARGS__ } break;               \
    case 2: { typedef uint16_t T; __VA_ARGS__ } break;              \
    case 4: { typedef uint32_t T; __VA_ARGS__ } break;              \
    case 8: { typedef uint64_t T; __VA_ARGS__ } break;              \
    default: return 1;                                              \
  }

DL4J_API int dl4j_gather_rows(const void* table, const void* idx, void* out, int64_t n_idx, int D, int64_t V, int esize, int idx64,
                              hipStream_t st) {
  if (n_idx <= 0 || D <= 0) return 0;
  const int64_t row_bytes = (int64_t)D * esize;
  if (row_bytes % 16 == 0 && n_idx * (row_bytes / 16) < 4294967295LL && !((uintptr_t)table & 15) && !((uintptr_t)out & 15)) {
    const uint32_t ppr = (uint32_t)(row_bytes / 16), total = (uint32_t)(n_idx * ppr);
    DISPATCH_I(hipLaunchKernelGGL((gather_rows_vec_kernel<I>), dim3(grid_for(total, 512)), dim3(256), 0, st, (const uint4*)table, (const I*)idx,
                                  (uint4*)out, total, ppr, V);)
    DL4J_RET_LAST();
  }
  const int g = grid_for(n_idx * D, 1024);
  DISPATCH_BYTES(esize, DISPATCH_I(hipLaunchKernelGGL((gather_rows_kernel<T, I>), dim3(g), dim3(256), 0, st, (const T*)table,
                                                      (const I*)idx, (T*)out, n_idx, D, V);))
  DL4J_RET_LAST();
}

DL4J_API int dl4j_scatter_add_rows(const void* src, const void* idx, float* table, int64_t n_idx, int D, int64_t V, float alpha, int dtype,
                                   int idx64, hipStream_t st) {
  if (n_idx <= 0 || D <= 0) return 0;
  const int g = grid_for(n_idx * D, 1024);
  DISPATCH_T(DISPATCH_I(hipLaunchKernelGGL((scatter_add_rows_kernel<T, I>), dim3(g), dim3(256), 0, st, (const T*)src, (const I*)idx, table,
                                           n_idx, D, V, alpha);))
  DL4J_RET_LAST();
}

DL4J_API int dl4j_one_hot(const void* idx, void* out, int64_t n, int depth, float on, float off, int dtype, int idx64, hipStream_t st) {
  if (n <= 0 || depth <= 0) return 0;
  const int V = dtype == 1 ? 8 : 4;
  if (depth % V == 0 && n * (depth / V) < 4294967295LL && !((uintptr_t)out & 15)) {
    const uint32_t ppr = (uint32_t)(depth / V), total = (uint32_t)(n * ppr);
    DISPATCH_T(DISPATCH_I(hipLaunchKernelGGL((one_hot_vec_kernel<T, I>), dim3(grid_for(total, 512)), dim3(256), 0, st, (const I*)idx, (T*)out, total,
                                             ppr, on, off);))
    DL4J_RET_LAST();
  }
  const int g = grid_for(n * depth, 1024);
  DISPATCH_T(DISPATCH_I(hipLaunchKernelGGL((one_hot_kernel<T, I>), dim3(g), dim3(256), 0, st, (const I*)idx, (T*)out, n, depth, on, off);))
  DL4J_RET_LAST();
}

DL4J_API int dl4j_cumsum_rows(const void* x, void* y, int64_t rows, int cols, int exclusive, int reverse, int dtype, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  DISPATCH_T(hipLaunchKernelGGL(cumsum_rows_kernel<T>, dim3((unsigned)rows), dim3(256), 0, st, (const T*)x, (T*)y, rows, cols, exclusive,
                                reverse);)
  DL4J_RET_LAST();
}

static ColP mkcol(const int* g) { return ColP{g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[13]}; }

// geom = {N, C, H, W, kH, kW, oH, oW, sH, sW, pH, pW, dH, dW}
DL4J_API int dl4j_im2col(const void* x, void* col, const int* geom, float pad_value, int dtype, hipStream_t st) {
  const ColP p = mkcol(geom);
  const int64_t total = (int64_t)p.N * p.C * p.kH * p.kW * p.oH * p.oW;
  if (total <= 0) return 0;
  DISPATCH_T(hipLaunchKernelGGL(im2col_kernel<T>, dim3(grid_for(total, 1024)), dim3(256), 0, st, (const T*)x, (T*)col, p, pad_value);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_col2im(const void* col, void* x, const int* geom, int dtype, hipStream_t st) {
  const ColP p = mkcol(geom);
  const int64_t total = (int64_t)p.N * p.C * p.H * p.W;
  if (total <= 0) return 0;
  DISPATCH_T(hipLaunchKernelGGL(col2im_kernel<T>, dim3(grid_for(total, 512)), dim3(256), 0, st, (const T*)col, (T*)x, p);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_upsample2d_fwd(const void* x, void* y, int64_t planes, int H, int W, int fH, int fW, int esize, hipStream_t st) {
  const int64_t total = planes * H * fH * W * fW;
  if (total <= 0) return 0;
  const int64_t in_bytes = planes * H * W * esize;
  if (fW == 2 && esize >= 1 && esize <= 8 && (W * esize) % 16 == 0 && in_bytes / 16 < 2147483647LL && !((uintptr_t)x & 15) && !((uintptr_t)y & 15)) {
    const uint32_t tv = (uint32_t)(in_bytes / 16);
    DISPATCH_BYTES(esize, hipLaunchKernelGGL(upsample2d_fwd_w2_kernel<T>, dim3(grid_for(tv, 512)), dim3(256), 0, st, (const T*)x, (T*)y, tv, W, fH);)
    DL4J_RET_LAST();
  }
  DISPATCH_BYTES(esize, hipLaunchKernelGGL(upsample2d_fwd_kernel<T>, dim3(grid_for(total, 1024)), dim3(256), 0, st, (const T*)x, (T*)y,
                                           planes, H, W, fH, fW);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_upsample2d_bwd(const void* g, void* dx, int64_t planes, int H, int W, int fH, int fW, int dtype, hipStream_t st) {
  const int64_t total = planes * H * W;
  if (total <= 0) return 0;
  DISPATCH_T(hipLaunchKernelGGL(upsample2d_bwd_kernel<T>, dim3(grid_for(total, 512)), dim3(256), 0, st, (const T*)g, (T*)dx, planes, H, W,
                                fH, fW);)
  DL4J_RET_LAST();
}

// ptrs: device int64[k] of array base addresses (all arrays contiguous, same dtype and length n)
DL4J_API int dl4j_merge(const int64_t* ptrs, int k, void* out, int64_t n, int mode, int dtype, hipStream_t st) {
  if (n <= 0 || k <= 0) return 0;
  DISPATCH_T(hipLaunchKernelGGL(merge_kernel<T>, dim3(grid_for(n, 1024)), dim3(256), 0, st, ptrs, k, (T*)out, n, mode);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_segment_sum(const void* x, const void* seg, float* out, int64_t rows, int D, int64_t nseg, int dtype, int idx64,
                              hipStream_t st) {
  if (rows <= 0 || D <= 0) return 0;
  const int g = grid_for(rows * D, 1024);
  DISPATCH_T(DISPATCH_I(hipLaunchKernelGGL((segment_sum_kernel<T, I>), dim3(g), dim3(256), 0, st, (const T*)x, (const I*)seg, out, rows, D,
                                           nseg);))
  DL4J_RET_LAST();
}

DL4J_API int dl4j_resize_nhwc(const void* x, void* y, int N, int H, int W, int C, int oH, int oW, int mode, int align_corners,
                              int half_pixel, int dtype, hipStream_t st) {
  const int64_t total = (int64_t)N * oH * oW * C;
  if (total <= 0) return 0;
  DISPATCH_T(hipLaunchKernelGGL(resize_nhwc_kernel<T>, dim3(grid_for(total, 1024)), dim3(256), 0, st, (const T*)x, (T*)y, N, H, W, C, oH,
                                oW, mode, align_corners, half_pixel);)
  DL4J_RET_LAST();
}

// dh: [n_pairs, D] fp32, fully written here
DL4J_API int dl4j_sg_pairs(const float* syn0, float* syn1, const float* hidden, const int64_t* in_idx, const int64_t* tgt_idx,
                           const float* label, float* dh, int64_t n_pairs, int D, float lr, float scale, int64_t V0, int64_t V1,
                           int sat_mode, hipStream_t st) {
  if (n_pairs <= 0 || D <= 0) return 0;
  const int g = grid_for(n_pairs, 4);      // 4 waves per block, one pair per wave
  hipLaunchKernelGGL(sg_pairs_kernel, dim3(g), dim3(256), 0, st, syn0, syn1, hidden, in_idx, tgt_idx, label, dh, n_pairs, D, lr, scale,
                     V0, V1, sat_mode);
  DL4J_RET_LAST();
}
