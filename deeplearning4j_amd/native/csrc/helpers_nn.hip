// Declarable-op helper kernels, nn family (gfx950): softmax / log-softmax backward, layer norm fwd+bwd,
// channel bias add, Philox dropout, LRN, PReLU, clip-by-norm.
//
// Reference counterparts (libnd4j/include/ops/declarable/helpers/cuda): activations.cu (softMaxCuda + derivative,
// prelu), addBias.cu, dropout.cu, lrn.cu, clip.cu, and the layer_norm composition (generic/nn/layer_norm.cpp:
// standardize + mul + add as three passes). Own design: rows are reduced with 64-lane shuffles (one wave per
// row for short rows, 4 waves for long ones) instead of shared-memory trees; layer norm is one pass forward and
// a dx pass + a partial column reduction backward; dropout draws 4 uniforms per Philox block so a wave fills
// 256 elements per round without any RNG state in memory.
#include "common.h"
#include "philox.h"

namespace {

// ------------------------------------------------------------------ softmax / log-softmax backward
// dx = y * (g - sum(g*y))            (softmax, y = softmax(x) recomputed from x)
// dx = g - softmax(x) * sum(g)       (log-softmax)
template <typename T>
__global__ void softmax_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ dx, int64_t rows, int cols,
                                   int is_log) {
  __shared__ float red[16];
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  const T* xr = x + r * cols;
  const T* gr = g + r * cols;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) mx = fmaxf(mx, Cvt<T>::ld(xr, c));
  mx = block_max(mx, red);
  float se = 0.f, sg = 0.f, sgy = 0.f;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    const float e = __expf(Cvt<T>::ld(xr, c) - mx), gv = Cvt<T>::ld(gr, c);
    se += e; sg += gv; sgy += gv * e;
  }
  se = block_sum(se, red);
  sg = block_sum(sg, red);
  sgy = block_sum(sgy, red);
  const float inv = 1.f / se;
  sgy *= inv;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    const float y = __expf(Cvt<T>::ld(xr, c) - mx) * inv, gv = Cvt<T>::ld(gr, c);
    Cvt<T>::st(dx + r * cols, c, is_log ? gv - y * sg : y * (gv - sgy));
  }
}

// ------------------------------------------------------------------ layer norm
template <typename T>
__global__ void layer_norm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int cols,
                                      float eps) {
  __shared__ float red[16];
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  const T* xr = x + r * cols;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) s += Cvt<T>::ld(xr, c);
  const float mu = block_sum(s, red) / cols;
  float v = 0.f;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) { const float d = Cvt<T>::ld(xr, c) - mu; v += d * d; }
  const float rs = rsqrtf(block_sum(v, red) / cols + eps);
  if (threadIdx.x == 0) { if (mean) mean[r] = mu; if (rstd) rstd[r] = rs; }
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    float o = (Cvt<T>::ld(xr, c) - mu) * rs;
    if (gamma) o *= gamma[c];
    if (beta) o += beta[c];
    Cvt<T>::st(y + r * cols, c, o);
  }
}

template <typename T>
__global__ void layer_norm_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ g, const float* __restrict__ gamma,
                                         const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx,
                                         int64_t rows, int cols) {
  __shared__ float red[16];
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  const T* xr = x + r * cols;
  const T* gr = g + r * cols;
  const float mu = mean[r], rs = rstd[r];
  float s1 = 0.f, s2 = 0.f;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    const float gg = Cvt<T>::ld(gr, c) * (gamma ? gamma[c] : 1.f), xh = (Cvt<T>::ld(xr, c) - mu) * rs;
    s1 += gg; s2 += gg * xh;
  }
  s1 = block_sum(s1, red) / cols;
  s2 = block_sum(s2, red) / cols;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    const float gg = Cvt<T>::ld(gr, c) * (gamma ? gamma[c] : 1.f), xh = (Cvt<T>::ld(xr, c) - mu) * rs;
    Cvt<T>::st(dx + r * cols, c, rs * (gg - s1 - xh * s2));
  }
}

// ---- register-resident forms (rows of whole 16-byte chunks, up to MAXC chunks per lane): ONE wave owns a row, reads it once with
// 16-byte loads, keeps it in registers for the mean, the (two-pass) variance and the output, and reduces with 64-lane shuffles - no
// LDS, no workgroup barrier. The block-per-row kernels above re-read the row three times with 2-byte loads and two barriers: 17.9 us
// forward / 13.9 + 7.0 us backward at [4096, 768] bf16 (profiles/r3_bert_kernels.md), i.e. ~0.7 TB/s. The backward form also folds the
// dgamma / dbeta column partials in (registers -> LDS atomics -> one partial row per block), so x and g are read once instead of twice.
template <typename T> struct Vec16;
template <> struct Vec16<bf16_t> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&f)[8]) {
    const s16x8 v = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = bf2f((bf16_t)v[i]);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&f)[8]) {
    s16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (short)f2bf(f[i]);
    *reinterpret_cast<s16x8*>(p) = v;
  }
};
template <> struct Vec16<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void ld(const float* p, float (&f)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
  }
  static __device__ __forceinline__ void st(float* p, const float (&f)[4]) { *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]}; }
};
// parameter vectors are views of a flat parameter vector: 16-byte alignment is usual but not guaranteed (wave-uniform branch)
template <int V> __device__ __forceinline__ void ld_f32(const float* p, float (&f)[V]) {
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
#pragma unroll
    for (int i = 0; i < V; i += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + i);
      f[i] = v[0]; f[i + 1] = v[1]; f[i + 2] = v[2]; f[i + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) f[i] = p[i];
  }
}

template <typename T, int MAXC>
__global__ void __launch_bounds__(256) layer_norm_fwd_wave_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ mean,
                                                                  float* __restrict__ rstd, int64_t rows, int cols, float eps) {
  constexpr int V = Vec16<T>::V;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= rows) return;
  const int nch = cols / V;
  const T* xr = x + r * cols;
  float v[MAXC][V];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      Vec16<T>::ld(xr + c * V, v[i]);
#pragma unroll
      for (int k = 0; k < V; ++k) s += v[i][k];
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) v[i][k] = 0.f;
    }
  }
  const float mu = wave_sum(s) / cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (lane + 64 * i < nch) {
#pragma unroll
      for (int k = 0; k < V; ++k) { const float d = v[i][k] - mu; q += d * d; }
    }
  const float rs = rsqrtf(wave_sum(q) / cols + eps);
  if (lane == 0) { if (mean) mean[r] = mu; if (rstd) rstd[r] = rs; }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float o[V], gm[V], bt[V];
      if (gamma) ld_f32<V>(gamma + c * V, gm);
      if (beta) ld_f32<V>(beta + c * V, bt);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        o[k] = (v[i][k] - mu) * rs;
        if (gamma) o[k] *= gm[k];
        if (beta) o[k] += bt[k];
      }
      Vec16<T>::st(y + r * cols + c * V, o);
    }
  }
}

// block = 8 waves, rows [blockIdx.x * rpb, + rpb) dealt to the waves round-robin; partial (optional) [gridDim.x][2][cols].
// The row loop is software-pipelined (the 16-byte loads of the wave's next row are in flight while the two wave reductions of the
// current row run: with two rows per wave the loop was a chain of exposed load latencies, 24 us at [4096, 768] = 0.75 TB/s), and the
// column partials leave the registers through one LDS read-modify-write turn per wave (16-byte accesses, a barrier between turns)
// instead of 32 LDS atomics per lane.
template <typename T> struct Raw16;
template <> struct Raw16<bf16_t> {
  static __device__ __forceinline__ void cv(const uint4& v, float (&f)[8]) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(w[j] << 16); f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
  }
};
template <> struct Raw16<float> {
  static __device__ __forceinline__ void cv(const uint4& v, float (&f)[4]) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
};

template <typename T, int MAXC>
__global__ void __launch_bounds__(512) layer_norm_bwd_wave_kernel(const T* __restrict__ x, const T* __restrict__ g, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  T* __restrict__ dx, float* __restrict__ partial, int64_t rows, int cols,
                                                                  int64_t rpb) {
  constexpr int V = Vec16<T>::V;
  constexpr int MAXCOLS = MAXC * 64 * V;
  __shared__ __attribute__((aligned(16))) float acc[2 * MAXCOLS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = cols / V;
  float gm[MAXC][V], dg[MAXC][V], db[MAXC][V];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
#pragma unroll
    for (int k = 0; k < V; ++k) { gm[i][k] = 1.f; dg[i][k] = 0.f; db[i][k] = 0.f; }
    if (gamma && c < nch) ld_f32<V>(gamma + c * V, gm[i]);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = r0 + rpb < rows ? r0 + rpb : rows;
  uint4 xr[MAXC], gr[MAXC], xn[MAXC], gn[MAXC];
  float mu_n = 0.f, rs_n = 0.f;
  auto issue = [&](int64_t r) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      xn[i] = make_uint4(0, 0, 0, 0); gn[i] = xn[i];
      if (c < nch) {
        xn[i] = *reinterpret_cast<const uint4*>(x + r * cols + c * V);
        gn[i] = *reinterpret_cast<const uint4*>(g + r * cols + c * V);
      }
    }
    mu_n = mean[r]; rs_n = rstd[r];
  };
  int64_t r = r0 + wave;
  if (r < r1) issue(r);
  for (; r < r1; r += 8) {
    const float mu = mu_n, rs = rs_n;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) { xr[i] = xn[i]; gr[i] = gn[i]; }
    if (r + 8 < r1) issue(r + 8);
    float xv[MAXC][V], gv[MAXC][V];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      Raw16<T>::cv(xr[i], xv[i]);
      Raw16<T>::cv(gr[i], gv[i]);
      if (lane + 64 * i < nch) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const float xh = (xv[i][k] - mu) * rs, gg = gv[i][k] * gm[i][k];
          xv[i][k] = xh;
          s1 += gg; s2 += gg * xh;
          dg[i][k] += gv[i][k] * xh;
          db[i][k] += gv[i][k];
        }
      }
    }
    s1 = wave_sum(s1) / cols;
    s2 = wave_sum(s2) / cols;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float o[V];
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = rs * (gv[i][k] * gm[i][k] - s1 - xv[i][k] * s2);
        Vec16<T>::st(dx + r * cols + c * V, o);
      }
    }
  }
  if (!partial) return;
  // wave w adds its registers into the LDS rows in turn w (wave 0 stores): plain 16-byte LDS accesses, 8 short turns
  for (int w = 0; w < 8; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
#pragma unroll
          for (int k = 0; k < V; k += 4) {
            f32x4* pg = reinterpret_cast<f32x4*>(&acc[c * V + k]);
            f32x4* pb = reinterpret_cast<f32x4*>(&acc[MAXCOLS + c * V + k]);
            f32x4 vg = {dg[i][k], dg[i][k + 1], dg[i][k + 2], dg[i][k + 3]};
            f32x4 vb = {db[i][k], db[i][k + 1], db[i][k + 2], db[i][k + 3]};
            if (w != 0) { vg += *pg; vb += *pb; }
            *pg = vg; *pb = vb;
          }
        }
      }
    }
    __syncthreads();
  }
  float* out = partial + (int64_t)blockIdx.x * 2 * cols;
  for (int c = threadIdx.x; c < cols; c += 512) {
    out[c] = acc[c];
    out[cols + c] = acc[MAXCOLS + c];
  }
}

template <typename T> int ln_maxc(int cols) {
  constexpr int V = Vec16<T>::V;
  if (cols % V) return 0;
  const int nch = cols / V;
  return nch <= 64 ? 1 : nch <= 128 ? 2 : nch <= 256 ? 4 : 0;
}
template <typename T>
bool ln_fwd_wave(const T* x, const float* gamma, const float* beta, T* y, float* mean, float* rstd, int64_t rows, int cols, float eps,
                 hipStream_t st) {
  const dim3 grid((unsigned)((rows + 3) / 4));
  switch (ln_maxc<T>(cols)) {
    case 1: hipLaunchKernelGGL((layer_norm_fwd_wave_kernel<T, 1>), grid, dim3(256), 0, st, x, gamma, beta, y, mean, rstd, rows, cols, eps); return true;
    case 2: hipLaunchKernelGGL((layer_norm_fwd_wave_kernel<T, 2>), grid, dim3(256), 0, st, x, gamma, beta, y, mean, rstd, rows, cols, eps); return true;
    case 4: hipLaunchKernelGGL((layer_norm_fwd_wave_kernel<T, 4>), grid, dim3(256), 0, st, x, gamma, beta, y, mean, rstd, rows, cols, eps); return true;
    default: return false;
  }
}
template <typename T>
bool ln_bwd_wave(const T* x, const T* g, const float* gamma, const float* mean, const float* rstd, T* dx, float* partial, int64_t rows,
                 int cols, int64_t rpb, int gy, hipStream_t st) {
  switch (ln_maxc<T>(cols)) {
    case 1: hipLaunchKernelGGL((layer_norm_bwd_wave_kernel<T, 1>), dim3(gy), dim3(512), 0, st, x, g, gamma, mean, rstd, dx, partial, rows, cols, rpb); return true;
    case 2: hipLaunchKernelGGL((layer_norm_bwd_wave_kernel<T, 2>), dim3(gy), dim3(512), 0, st, x, g, gamma, mean, rstd, dx, partial, rows, cols, rpb); return true;
    case 4: hipLaunchKernelGGL((layer_norm_bwd_wave_kernel<T, 4>), dim3(gy), dim3(512), 0, st, x, g, gamma, mean, rstd, dx, partial, rows, cols, rpb); return true;
    default: return false;
  }
}
bool ln_aligned(const void* a, const void* b, const void* c) {
  return !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15);
}

// partial[gy][2][cols]: row 0 = sum g*xhat (dgamma), row 1 = sum g (dbeta); finished by dl4j_bn_sum_partials
template <typename T>
__global__ void layer_norm_bwd_param_kernel(const T* __restrict__ x, const T* __restrict__ g, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, float* __restrict__ partial, int64_t rows, int cols,
                                            int64_t rows_per_block) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float dg = 0.f, db = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const float gv = Cvt<T>::ld(g, r * cols + c);
    dg += gv * (Cvt<T>::ld(x, r * cols + c) - mean[r]) * rstd[r];
    db += gv;
  }
  partial[((int64_t)blockIdx.y * 2) * cols + c] = dg;
  partial[((int64_t)blockIdx.y * 2 + 1) * cols + c] = db;
}

// ------------------------------------------------------------------ bias add: channel = (i / inner) % C
template <typename T>
__global__ void bias_add_kernel(const T* __restrict__ x, const float* __restrict__ bias, T* __restrict__ y, int64_t n, int C,
                                int64_t inner) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = inner == 1 ? (int)(i % C) : (int)((i / inner) % C);
    Cvt<T>::st(y, i, Cvt<T>::ld(x, i) + bias[c]);
  }
}

// 16-byte pieces: V elements per thread and one 32-bit channel computation per piece. inner % V == 0 (NCHW: a piece lies inside one
// channel) or inner == 1 with C % V == 0 (channels-last: a piece covers V consecutive channels).
template <typename T> struct BPack;
template <> struct BPack<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void ld(const float* p, float* x) { const float4 v = *(const float4*)p; x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
  static __device__ __forceinline__ void st(float* p, const float* x) { *(float4*)p = make_float4(x[0], x[1], x[2], x[3]); }
};
template <> struct BPack<bf16_t> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void ld(const bf16_t* p, float* x) {
    const uint4 v = *(const uint4*)p;
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[2 * j] = __uint_as_float(w[j] << 16); x[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float* x) {
    alignas(16) bf16_t h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = f2bf(x[j]);
    *(uint4*)p = *(const uint4*)h;
  }
};

template <typename T>
__global__ void __launch_bounds__(256) bias_add_vec_kernel(const T* __restrict__ x, const float* __restrict__ bias, T* __restrict__ y,
                                                           uint32_t nvec, uint32_t C, uint32_t inner_vec) {
  constexpr int V = BPack<T>::V;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nvec; i += gridDim.x * 256u) {
    float v[V];
    BPack<T>::ld(x + (int64_t)i * V, v);
    if (inner_vec) {                                   // channels-first: one channel per piece
      const float b = bias[(i / inner_vec) % C];
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] += b;
    } else {                                           // channels-last: V consecutive channels
      const float* bp = bias + ((uint64_t)i * V) % C;
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] += bp[j];
    }
    BPack<T>::st(y + (int64_t)i * V, v);
  }
}

// ------------------------------------------------------------------ dropout (inverted): keep with prob p, scale 1/p
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ mask, int64_t n, float p,
                               float scale, uint64_t seed, uint64_t stream) {
  const int64_t nblk = (n + 3) >> 2;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (int64_t)gridDim.x * blockDim.x) {
    const philox::u4 r = philox::block((uint64_t)b, stream, seed);
    const uint32_t bits[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = b * 4 + k;
      if (i < n) {
        const bool keep = philox::u01(bits[k]) < p;
        Cvt<T>::st(y, i, keep ? Cvt<T>::ld(x, i) * scale : 0.f);
        if (mask) mask[i] = keep ? 1 : 0;
      }
    }
  }
}

template <typename T>
__global__ void dropout_bwd_kernel(const T* __restrict__ g, const uint8_t* __restrict__ mask, T* __restrict__ dx, int64_t n,
                                   float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    Cvt<T>::st(dx, i, mask[i] ? Cvt<T>::ld(g, i) * scale : 0.f);
}

// ------------------------------------------------------------------ LRN across the innermost (channel) dim of [M, C] rows
// y = x / (bias + alpha * sum_{c-d..c+d} x^2)^beta ; also emits the denominators' base for the backward
template <typename T>
__global__ void lrn_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, float* __restrict__ scale_out, int64_t M, int C, int depth,
                               float bias, float alpha, float beta) {
  const int64_t n = M * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t base = i - c;
    const int lo = c - depth < 0 ? 0 : c - depth, hi = c + depth >= C ? C - 1 : c + depth;
    float s = 0.f;
    for (int k = lo; k <= hi; ++k) { const float v = Cvt<T>::ld(x, base + k); s += v * v; }
    const float sc = bias + alpha * s;
    if (scale_out) scale_out[i] = sc;
    Cvt<T>::st(y, i, Cvt<T>::ld(x, i) * __powf(sc, -beta));
  }
}

// ------------------------------------------------------------------ PReLU with per-channel alpha (channel = (i / inner) % C)
template <typename T>
__global__ void prelu_kernel(const T* __restrict__ x, const float* __restrict__ alpha, T* __restrict__ y, int64_t n, int C,
                             int64_t inner) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = Cvt<T>::ld(x, i);
    const int c = (int)((i / inner) % C);
    Cvt<T>::st(y, i, v >= 0.f ? v : v * alpha[c]);
  }
}

// ------------------------------------------------------------------ clip by (global) L2 norm: two kernels, no host round trip
template <typename T>
__global__ void sumsq_partial_kernel(const T* __restrict__ x, float* __restrict__ partial, int64_t n) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = Cvt<T>::ld(x, i);
    s += v * v;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

template <typename T>
__global__ void clip_scale_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ partial, int nparts, int64_t n,
                                  float clip, float* __restrict__ norm_out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int k = threadIdx.x; k < nparts; k += blockDim.x) s += partial[k];     // kernel boundary made the partials visible
  const float nrm = sqrtf(block_sum(s, red));
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = nrm;
  const float f = nrm > clip ? clip / nrm : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    Cvt<T>::st(y, i, Cvt<T>::ld(x, i) * f);
}

inline int row_threads(int cols) { return cols <= 1024 ? 64 : 256; }

}  // namespace

#define DISPATCH_T(...)                          \
  if (dtype == 1) { typedef bf16_t T; __VA_ARGS__ } \
  else { typedef float T; __VA_ARGS__ }

DL4J_API int dl4j_softmax_bwd_x(const void* x, const void* g, void* dx, int64_t rows, int cols, int is_log, int dtype, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  DISPATCH_T(hipLaunchKernelGGL(softmax_bwd_kernel<T>, dim3((unsigned)rows), dim3(row_threads(cols)), 0, st, (const T*)x, (const T*)g,
                                (T*)dx, rows, cols, is_log);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_layer_norm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
                                 int cols, float eps, int dtype, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  if (ln_aligned(x, y, nullptr)) {
    bool done = false;
    DISPATCH_T(done = ln_fwd_wave<T>((const T*)x, gamma, beta, (T*)y, mean, rstd, rows, cols, eps, st);)
    if (done) DL4J_RET_LAST();
  }
  DISPATCH_T(hipLaunchKernelGGL(layer_norm_fwd_kernel<T>, dim3((unsigned)rows), dim3(row_threads(cols)), 0, st, (const T*)x, gamma, beta,
                                (T*)y, mean, rstd, rows, cols, eps);)
  DL4J_RET_LAST();
}

// rows of the partial buffer: one per workgroup of the backward pass (16 rows each, at most 512 of them)
DL4J_API int dl4j_layer_norm_bwd_rows(int64_t rows) {
  int64_t gy = (rows + 15) / 16;
  if (gy > 512) gy = 512;
  return (int)(gy < 1 ? 1 : gy);
}

// partial must hold dl4j_layer_norm_bwd_rows(rows) * 2 * cols floats ([gy][2][cols]; finished by dl4j_bn_sum_partials)
DL4J_API int dl4j_layer_norm_bwd(const void* x, const void* g, const float* gamma, const float* mean, const float* rstd, void* dx,
                                 float* partial, int64_t rows, int cols, int dtype, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  const int gy = dl4j_layer_norm_bwd_rows(rows);
  const int64_t rpb = (rows + gy - 1) / gy;
  if (ln_aligned(x, g, dx)) {
    bool done = false;
    DISPATCH_T(done = ln_bwd_wave<T>((const T*)x, (const T*)g, gamma, mean, rstd, (T*)dx, partial, rows, cols, rpb, gy, st);)
    if (done) DL4J_RET_LAST();
  }
  const dim3 pg((cols + 255) / 256, gy);
  DISPATCH_T(
      hipLaunchKernelGGL(layer_norm_bwd_dx_kernel<T>, dim3((unsigned)rows), dim3(row_threads(cols)), 0, st, (const T*)x, (const T*)g,
                         gamma, mean, rstd, (T*)dx, rows, cols);
      if (partial) hipLaunchKernelGGL(layer_norm_bwd_param_kernel<T>, pg, dim3(256), 0, st, (const T*)x, (const T*)g, mean, rstd, partial,
                                      rows, cols, rpb);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_bias_add(const void* x, const float* bias, void* y, int64_t n, int C, int64_t inner, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  const int V = dtype == 1 ? 8 : 4;
  const bool cf = inner > 1 && inner % V == 0, cl = inner == 1 && C % V == 0;
  if ((cf || cl) && n % V == 0 && n / V < 4294967295LL && inner / V < 4294967295LL && !((uintptr_t)x & 15) && !((uintptr_t)y & 15)) {
    const uint32_t nvec = (uint32_t)(n / V);
    DISPATCH_T(hipLaunchKernelGGL(bias_add_vec_kernel<T>, dim3(grid_for(nvec, 512)), dim3(256), 0, st, (const T*)x, bias, (T*)y, nvec, (uint32_t)C,
                                  cf ? (uint32_t)(inner / V) : 0u);)
    DL4J_RET_LAST();
  }
  DISPATCH_T(hipLaunchKernelGGL(bias_add_kernel<T>, dim3(grid_for(n, 1024)), dim3(256), 0, st, (const T*)x, bias, (T*)y, n, C, inner);)
  DL4J_RET_LAST();
}

// keep-probability p (DL4J convention); mask (optional, 1 byte/element) is what the backward needs
DL4J_API int dl4j_dropout(const void* x, void* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t stream_id, int dtype,
                          hipStream_t st) {
  if (n <= 0) return 0;
  const float scale = p > 0.f ? 1.f / p : 0.f;
  DISPATCH_T(hipLaunchKernelGGL(dropout_kernel<T>, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, st, (const T*)x, (T*)y, mask, n, p,
                                scale, seed, stream_id);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_dropout_bwd(const void* g, const uint8_t* mask, void* dx, int64_t n, float p, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  const float scale = p > 0.f ? 1.f / p : 0.f;
  DISPATCH_T(hipLaunchKernelGGL(dropout_bwd_kernel<T>, dim3(grid_for(n, 1024)), dim3(256), 0, st, (const T*)g, mask, (T*)dx, n, scale);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_lrn_fwd(const void* x, void* y, float* scale_out, int64_t M, int C, int depth, float bias, float alpha, float beta,
                          int dtype, hipStream_t st) {
  if (M <= 0 || C <= 0) return 0;
  DISPATCH_T(hipLaunchKernelGGL(lrn_fwd_kernel<T>, dim3(grid_for(M * C, 512)), dim3(256), 0, st, (const T*)x, (T*)y, scale_out, M, C,
                                depth, bias, alpha, beta);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_prelu(const void* x, const float* alpha, void* y, int64_t n, int C, int64_t inner, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  DISPATCH_T(hipLaunchKernelGGL(prelu_kernel<T>, dim3(grid_for(n, 1024)), dim3(256), 0, st, (const T*)x, alpha, (T*)y, n, C, inner);)
  DL4J_RET_LAST();
}

DL4J_API int dl4j_clip_ws_floats() { return DL4J_MAX_GRID; }

// y = x * min(1, clip / ||x||_2); partial_ws: dl4j_clip_ws_floats() floats; norm_out optional (device float)
DL4J_API int dl4j_clip_by_norm(const void* x, void* y, float* partial_ws, int64_t n, float clip, float* norm_out, int dtype,
                               hipStream_t st) {
  if (n <= 0) return 0;
  const int g = grid_for(n, 2048);
  DISPATCH_T(hipLaunchKernelGGL(sumsq_partial_kernel<T>, dim3(g), dim3(256), 0, st, (const T*)x, partial_ws, n);
             hipLaunchKernelGGL(clip_scale_kernel<T>, dim3(g), dim3(256), 0, st, (const T*)x, (T*)y, partial_ws, g, n, clip, norm_out);)
  DL4J_RET_LAST();
}
