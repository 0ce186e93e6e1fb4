// Fused BatchNorm kernels for channels-last activations ([M, C], C contiguous; M = N*H*W).
// Semantics follow DL4J BatchNormalization train-time math
// (reference: deeplearning4j-nn/.../nn/layers/normalization/BatchNormalization.java:492-566 fwd,
// :280-353 bwd incl. running-stat "gradients"), but the ~25 legacy ops issued from Java per
// layer are replaced by: one stats pass, one tiny finalize, one apply pass (optionally fused
// with a residual add — itself optionally batch-normalised — and ReLU), and a 2-pass backward.
// Memory-bound: 16-byte vector accesses, fp32 accumulation, one float atomic per block/channel.
#include "common.h"
#include "ticket.h"
#include <stdlib.h>

struct f16_t { _Float16 v; };     // IEEE half activations (DL4J network dtype HALF); distinct type so the templates can dispatch
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
template <typename T> struct VecOf;
template <> struct VecOf<bf16_t> { static constexpr int N = 8; };
template <> struct VecOf<f16_t> { static constexpr int N = 8; };
template <> struct VecOf<float> { static constexpr int N = 4; };

template <typename T, int V>
__device__ __forceinline__ void load_vec(const T* p, float (&o)[V]);
template <>
__device__ __forceinline__ void load_vec<bf16_t, 8>(const bf16_t* p, float (&o)[8]) {
  s16x8 v = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = bf2f((bf16_t)v[j]);
}
template <>
__device__ __forceinline__ void load_vec<float, 4>(const float* p, float (&o)[4]) {
  f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = v[j];
}
template <>
__device__ __forceinline__ void load_vec<bf16_t, 1>(const bf16_t* p, float (&o)[1]) { o[0] = bf2f(*p); }
template <>
__device__ __forceinline__ void load_vec<f16_t, 8>(const f16_t* p, float (&o)[8]) {
  const f16x8v v = *reinterpret_cast<const f16x8v*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
}
template <>
__device__ __forceinline__ void load_vec<f16_t, 1>(const f16_t* p, float (&o)[1]) { o[0] = (float)p->v; }
template <>
__device__ __forceinline__ void load_vec<float, 1>(const float* p, float (&o)[1]) { o[0] = *p; }

template <typename T, int V>
__device__ __forceinline__ void store_vec(T* p, const float (&o)[V]);
template <>
__device__ __forceinline__ void store_vec<bf16_t, 8>(bf16_t* p, const float (&o)[8]) {
  s16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(o[j]);
  *reinterpret_cast<s16x8*>(p) = v;
}
template <>
__device__ __forceinline__ void store_vec<float, 4>(float* p, const float (&o)[4]) {
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = o[j];
  *reinterpret_cast<f32x4*>(p) = v;
}
template <>
__device__ __forceinline__ void store_vec<bf16_t, 1>(bf16_t* p, const float (&o)[1]) { *p = f2bf(o[0]); }
template <>
__device__ __forceinline__ void store_vec<f16_t, 8>(f16_t* p, const float (&o)[8]) {
  f16x8v v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (_Float16)o[j];
  *reinterpret_cast<f16x8v*>(p) = v;
}
template <>
__device__ __forceinline__ void store_vec<f16_t, 1>(f16_t* p, const float (&o)[1]) { p->v = (_Float16)o[0]; }
template <>
__device__ __forceinline__ void store_vec<float, 1>(float* p, const float (&o)[1]) { *p = o[0]; }

// Per-channel fp32 coefficients for V consecutive channels: 16-byte loads (V=8: two, V=4: one). With one scalar load per
// channel a single-iteration thread issued 4-5x more (poorly coalesced) coefficient loads than data loads, which is what
// bounded the small late-stage layers.
template <int V>
__device__ __forceinline__ void load_coef(const float* __restrict__ p, int c0, float (&o)[V]) {
  if constexpr (V == 8) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + c0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + c0 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
  } else if constexpr (V == 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = a[j];
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = p[c0 + j];
  }
}

// ---------------------------------------------------------------------------------------
// Pass A: per-channel column reductions of up to two quantities.
//   MODE 0: q0 = x, q1 = x*x                      (forward statistics)
//   MODE 1: q0 = dz, q1 = dz * (x-mean)*invstd    (backward dbeta / dgamma), dz = dy*[out>0]
//           the ReLU mask comes from ``out`` or, when out == nullptr and mscale != nullptr (no residual was added),
//           is recomputed as [x*mscale + mshift > 0] so that the activation tensor is not re-read.
// block = 256 threads: cpb channel-vectors x rpi rows per iteration.
// ---------------------------------------------------------------------------------------
template <typename T, int V, int MODE, bool FUSED>
__global__ void __launch_bounds__(256) bn_colreduce_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ out,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ mscale, const float* __restrict__ mshift,
    float* __restrict__ partial, int64_t M, int C, int cpb, int rpi,
    float* __restrict__ sums, float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ gpart,
    unsigned* __restrict__ cnt, int G, const uint8_t* __restrict__ bits) {
  __shared__ float red[2 * V * 256];
  const int cv = C / V;
  const int tid = threadIdx.x;
  const int r_in = tid / cpb, c_in = tid - r_in * cpb;
  const int cvec = blockIdx.x * cpb + c_in;
  const bool active = (r_in < rpi) && (cvec < cv);
  float a0[V], a1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
  if (active) {
    const int c0 = cvec * V;
    float mu[V], is[V], ms[V], mt[V];
    const bool xmask = (MODE == 1) && (out == nullptr) && (bits == nullptr) && (mscale != nullptr);
    if (MODE == 1) {
      load_coef<V>(mean, c0, mu);
      load_coef<V>(invstd, c0, is);
      if (xmask) { load_coef<V>(mscale, c0, ms); load_coef<V>(mshift, c0, mt); }
    }
#pragma unroll 2
    for (int64_t r = (int64_t)blockIdx.y * rpi + r_in; r < M; r += (int64_t)gridDim.y * rpi) {
      const int64_t off = r * C + c0;
      float xv[V];
      load_vec<T, V>(x + off, xv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) { a0[j] += xv[j]; a1[j] += xv[j] * xv[j]; }
      } else {
        float dv[V];
        load_vec<T, V>(dy + off, dv);
        if (bits != nullptr) {
          // ReLU mask saved by the forward apply as one byte per V-channel vector (1/16 of re-reading the bf16 output)
          const unsigned mb = bits[r * cv + cvec];
#pragma unroll
          for (int j = 0; j < V; ++j) dv[j] = ((mb >> j) & 1u) ? dv[j] : 0.f;
        } else if (out != nullptr) {
          float ov[V];
          load_vec<T, V>(out + off, ov);
#pragma unroll
          for (int j = 0; j < V; ++j) dv[j] = ov[j] > 0.f ? dv[j] : 0.f;
        } else if (xmask) {
#pragma unroll
          for (int j = 0; j < V; ++j) dv[j] = (xv[j] * ms[j] + mt[j]) > 0.f ? dv[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) { a0[j] += dv[j]; a1[j] += dv[j] * (xv[j] - mu[j]) * is[j]; }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) { red[j * 256 + tid] = a0[j]; red[(V + j) * 256 + tid] = a1[j]; }
  __syncthreads();
  if (r_in == 0 && cvec < cv) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float s0 = 0.f, s1 = 0.f;
      for (int r = 0; r < rpi; ++r) { s0 += red[j * 256 + r * cpb + c_in]; s1 += red[(V + j) * 256 + r * cpb + c_in]; }
      float* d0 = partial + ((int64_t)blockIdx.y * 2) * C + cvec * V + j;
      float* d1 = partial + ((int64_t)blockIdx.y * 2 + 1) * C + cvec * V + j;
      if (FUSED && cnt != nullptr) { ticket::st_agent(d0, s0); ticket::st_agent(d1, s1); } else { *d0 = s0; *d1 = s1; }
    }
  }
  if (!FUSED || cnt == nullptr) return;
  // ---------------- second stage inside this launch (ticket.h): the last block of each group of G row-blocks sums the group's
  // partial rows, the last group sums the group results into sums[2C] (+ the dbeta / dgamma gradient views)
  __shared__ int s_last;
  const int gy = gridDim.y, ngroups = (gy + G - 1) / G;
  const int g = blockIdx.y / G, gbeg = g * G, gend = (gbeg + G < gy) ? gbeg + G : gy;
  const int cbeg = blockIdx.x * cpb * V;
  const int ncol = (C - cbeg < cpb * V) ? C - cbeg : cpb * V;       // channels owned by this column block
  if (!ticket::arrive_last(cnt + gridDim.x + blockIdx.x * ngroups + g, (unsigned)(gend - gbeg), &s_last)) return;
  for (int idx = tid; idx < 2 * ncol; idx += 256) {
    const int q = idx / ncol, c = cbeg + idx - q * ncol;
    ticket::st_agent(gpart + ((int64_t)g * 2 + q) * C + c, ticket::sum_rows(partial + ((int64_t)gbeg * 2 + q) * C + c, gend - gbeg, 2 * (int64_t)C));
  }
  if (!ticket::arrive_last(cnt + blockIdx.x, (unsigned)ngroups, &s_last)) return;
  for (int idx = tid; idx < 2 * ncol; idx += 256) {
    const int q = idx / ncol, c = cbeg + idx - q * ncol;
    const float a = ticket::sum_rows(gpart + (int64_t)q * C + c, ngroups, 2 * (int64_t)C);
    sums[q * C + c] = a;
    if (q == 0) { if (out0) out0[c] = a; }
    else if (out1) out1[c] = a;
  }
}

// Stage 2: sums[q*C + c] = sum over row-blocks of partial[(y*2+q)*C + c]; optional extra outputs
// (gradient views for dbeta / dgamma) are written in the same pass.
// 1024 threads = 32 columns x 32 row groups: the per-thread chain over the row blocks is the whole cost of this tiny kernel
// (it ran ~7.6 us with 8 row groups), so it is kept short and four loads are in flight per thread.
__global__ void __launch_bounds__(1024) bn_sum_partials_kernel(const float* __restrict__ partial, int gy, int C,
                                                               float* __restrict__ sums, float* __restrict__ out0,
                                                               float* __restrict__ out1) {
  __shared__ float red[1024];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;  // 32 columns x 32 row groups
  const int col = blockIdx.x * 32 + cl;                     // column in [0, 2C)
  float a = 0.f;
  if (col < 2 * C) {
    const int q = col / C, c = col - q * C;
    const float* p = partial + (int64_t)q * C + c;
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int y = rg;
    for (; y + 96 < gy; y += 128) {
      a += p[(int64_t)y * 2 * C]; a1 += p[(int64_t)(y + 32) * 2 * C];
      a2 += p[(int64_t)(y + 64) * 2 * C]; a3 += p[(int64_t)(y + 96) * 2 * C];
    }
    for (; y < gy; y += 32) a += p[(int64_t)y * 2 * C];
    a += a1 + a2 + a3;
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if (rg == 0 && col < 2 * C) {
    a = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) a += red[k * 32 + cl];
    sums[col] = a;
    if (col < C) { if (out0) out0[col] = a; }
    else if (out1) out1[col - C] = a;
  }
}

#define BN_MAX_ROWBLOCKS 2048                 // workspace bound
#define BN_TICKET_COUNTERS 4096               // entries of the self-resetting ticket counter array (dl4j_bn_ticket_counters)
#define BN_TICKET_GROUPS ((BN_MAX_ROWBLOCKS + 31) / 32)
// grid policy of the column reductions (tunable for tools/bn_microbench.py through the environment)
static int bn_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
static int bn_rowblocks() { static int v = bn_env_int("DL4J_BN_ROWBLOCKS", 1024); return v < 1 ? 1 : (v > BN_MAX_ROWBLOCKS ? BN_MAX_ROWBLOCKS : v); }
static int bn_min_iters() { static int v = bn_env_int("DL4J_BN_MIN_ITERS", 8); return v < 1 ? 1 : v; }
// workspace floats needed for the partials of one call
static inline int64_t colreduce_ws_floats(int C) { return (int64_t)BN_MAX_ROWBLOCKS * 2 * C; }

template <typename T, int MODE>
static int launch_colreduce(const void* x, const void* dy, const void* out, const float* mean,
                            const float* invstd, const float* mscale, const float* mshift, float* partial, float* sums,
                            float* out0, float* out1, int64_t M, int C, hipStream_t st, float* gpart = nullptr,
                            unsigned* cnt = nullptr, const uint8_t* bits = nullptr) {
  constexpr int VFULL = VecOf<T>::N;
  const bool vec_ok = (C % VFULL) == 0;
  const int V = vec_ok ? VFULL : 1;
  const int cv = C / V;
  const int cpb = cv < 256 ? cv : 256;
  const int rpi = 256 / cpb;
  const int gx = (cv + cpb - 1) / cpb;
  const int64_t rows_pb = (int64_t)rpi * bn_min_iters();             // at least this many row-iterations per block
  int64_t gy = (M + rows_pb - 1) / rows_pb;
  if (gy < 1) gy = 1;
  int64_t cap = bn_rowblocks() / gx; if (cap < 1) cap = 1;
  if (gy > cap) gy = cap;
  dim3 grid(gx, (unsigned)gy);
  constexpr int G = 32;
  // fused second stage: counters laid out as [gx] level-2 + [gx][ngroups] level-1 (BN_TICKET_COUNTERS entries available)
  const bool fused = cnt != nullptr && gpart != nullptr && (int64_t)gx * (1 + (gy + G - 1) / G) <= BN_TICKET_COUNTERS;
  unsigned* c = fused ? cnt : nullptr;
#define COLREDUCE_LAUNCH(VV, FF)                                                                                             \
  hipLaunchKernelGGL((bn_colreduce_kernel<T, VV, MODE, FF>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, (const T*)out, mean, \
                     invstd, mscale, mshift, partial, M, C, cpb, rpi, sums, out0, out1, gpart, c, G, bits)
  // the hand-off code is a template parameter: carried in the plain kernel it costs registers the bandwidth-bound loop wants
  if (vec_ok) { if (fused) COLREDUCE_LAUNCH(VFULL, true); else COLREDUCE_LAUNCH(VFULL, false); }
  else { if (fused) COLREDUCE_LAUNCH(1, true); else COLREDUCE_LAUNCH(1, false); }
#undef COLREDUCE_LAUNCH
  if (!fused)
    hipLaunchKernelGGL(bn_sum_partials_kernel, dim3((2 * C + 31) / 32), dim3(1024), 0, st, partial, (int)gy, C, sums, out0,
                       out1);
  DL4J_RET_LAST();
}

// ---------------------------------------------------------------------------------------
// Finalize: sums -> mean / invstd / scale / shift (+ running-stat gradient entries).
// stat_mode: 0 = none, 1 = variance parameter, 2 = log10(stdev) parameter (DL4J useLogStd).
// conv_bias (optional): bias of the producing conv that was folded out of x; it only shifts
// the batch mean seen by the running statistics.
// ---------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const float* __restrict__ conv_bias,
                                   const float* __restrict__ gmean, const float* __restrict__ gvar,
                                   float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ dgmean, float* __restrict__ dgvar,
                                   int C, float invM, float eps, float decay, int stat_mode) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float m = sums[c] * invM;
  float var = sums[C + c] * invM - m * m;
  var = var < 0.f ? 0.f : var;
  const float is = rsqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean[c] = m; invstd[c] = is;
  const float sc = g * is;
  scale[c] = sc; shift[c] = b - m * sc;
  if (stat_mode != 0) {
    const float bm = m + (conv_bias ? conv_bias[c] : 0.f);
    dgmean[c] = (gmean[c] - bm) * (1.f - decay);
    if (stat_mode == 1) {
      dgvar[c] = (gvar[c] - var) * (1.f - decay);
    } else {
      // parameter is log10(std); d = log10(var_t / (decay*var_t + (1-decay)*batchVar)) / 2
      const float vt = powf(10.f, 2.f * gvar[c]);
      const float vnew = decay * vt + (1.f - decay) * var;
      dgvar[c] = logf(vt / vnew) * (0.5f / 2.302585092994046f);
    }
  }
}

// Fused "reduce the conv-epilogue partials + finalize": partial is [gy][2][C] (row 0 sum, row 1 sum of squares per M tile).
// One block owns 8 channels; its 32 row groups stride over the gy tiles (4 independent loads in flight per thread), an LDS
// tree-less reduction (32 adds by 8 threads) finishes the sums and the same threads run the finalize math. Replaces two
// latency-bound launches per BatchNorm (bn_sum_partials + bn_finalize) by one with 2-8x more blocks.
__global__ __launch_bounds__(1024) void bn_finalize_partials_kernel(
    const float* __restrict__ partial, int gy, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ conv_bias, const float* __restrict__ gmean, const float* __restrict__ gvar, float* __restrict__ mean,
    float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ dgmean,
    float* __restrict__ dgvar, int C, float invM, float eps, float decay, int stat_mode, int shifted) {
  // 1024 threads = 8 channels x 128 row groups (was 32): the serial chain over the M tiles was the whole cost (~7 us)
  __shared__ float s0[128][9], s1[128][9];
  const int cl = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
  if (c < C) {
    const int64_t C2 = 2 * (int64_t)C;
    int r = rg;
    for (; r + 128 < gy; r += 256) {
      a0 += partial[r * C2 + c];
      b0 += partial[r * C2 + C + c];
      a1 += partial[(r + 128) * C2 + c];
      b1 += partial[(r + 128) * C2 + C + c];
    }
    if (r < gy) { a0 += partial[r * C2 + c]; b0 += partial[r * C2 + C + c]; }
  }
  s0[rg][cl] = a0 + a1;
  s1[rg][cl] = b0 + b1;
  __syncthreads();
  if (rg < 32) {
    s0[rg][cl] += s0[rg + 32][cl] + s0[rg + 64][cl] + s0[rg + 96][cl];
    s1[rg][cl] += s1[rg + 32][cl] + s1[rg + 64][cl] + s1[rg + 96][cl];
  }
  __syncthreads();
  if (rg != 0 || c >= C) return;
  float su = 0.f, sq = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) { su += s0[k][cl]; sq += s1[k][cl]; }
  // shifted: the partials are sums of (x - pivot), pivot = running mean - folded conv bias (conv_igemm.hip IGemmP::piv_mean)
  const float piv = shifted ? gmean[c] - (conv_bias ? conv_bias[c] : 0.f) : 0.f;
  const float md = su * invM;
  const float m = piv + md;
  float var = sq * invM - md * md;
  var = var < 0.f ? 0.f : var;
  const float is = rsqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean[c] = m; invstd[c] = is;
  const float sc = g * is;
  scale[c] = sc; shift[c] = b - m * sc;
  if (stat_mode != 0) {
    const float bm = m + (conv_bias ? conv_bias[c] : 0.f);
    dgmean[c] = (gmean[c] - bm) * (1.f - decay);
    if (stat_mode == 1) {
      dgvar[c] = (gvar[c] - var) * (1.f - decay);
    } else {
      const float vt = powf(10.f, 2.f * gvar[c]);
      const float vnew = decay * vt + (1.f - decay) * var;
      dgvar[c] = logf(vt / vnew) * (0.5f / 2.302585092994046f);
    }
  }
}

// Inference-time scale/shift from global statistics.
__global__ void bn_infer_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ gmean, const float* __restrict__ gvar,
                                       const float* __restrict__ conv_bias,
                                       float* __restrict__ scale, float* __restrict__ shift, int C, float eps,
                                       int log_std) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float var = log_std ? powf(10.f, 2.f * gvar[c]) : gvar[c];
  const float sc = (gamma ? gamma[c] : 1.f) * rsqrtf(var + eps);
  const float m = gmean[c] - (conv_bias ? conv_bias[c] : 0.f);
  scale[c] = sc; shift[c] = (beta ? beta[c] : 0.f) - m * sc;
}

// ---------------------------------------------------------------------------------------
// Pass B kernels are grid-stride over 16-byte vectors. When the total thread count is a multiple
// of the number of channel-vectors (always true for power-of-two channel counts), a thread sees
// the SAME channels in every iteration, so the per-channel coefficients are hoisted into
// registers (HOIST): the loop body is then pure streaming (loads, FMAs, stores) instead of
// issuing 16-40 scalar coefficient loads per vector.
// ---------------------------------------------------------------------------------------
// Grid of the streaming passes. In isolation (tools/stream_lab: y = relu(x * s + b), bf16, rotating buffers) 8192 blocks stream at
// 5.65 TB/s against 5.2 at 2048 blocks (206 MB), but inside the training step the larger grids are slower (8.59 ms at 2048, 8.62 at
// 8192, 8.66 at 16384): the default stays at 8 blocks per CU.
static int bn_stream_blocks() { static int v = bn_env_int("DL4J_BN_STREAM_BLOCKS", 2048); return v < 256 ? 256 : v; }
static inline int hoist_grid(int64_t nvec, int cv, bool* hoist) {
  int64_t want = (nvec + 255) / 256;
  if (want < 1) want = 1;
  if (want > bn_stream_blocks()) want = bn_stream_blocks();
  int g = launch_grid(LS_BATCHNORM, (int)want);
  *hoist = false;
  if (cv <= 256) {
    if (256 % cv == 0) *hoist = true;
  } else if (cv % 256 == 0) {
    const int m = cv / 256;
    g = (g + m - 1) / m * m;
    *hoist = true;
  }
  return g;
}

// Pass B (forward): y = act( x*scale + shift [+ res (*rscale + rshift)] )
template <typename T, int V, bool HOIST>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const T* __restrict__ res,
                                                       const float* __restrict__ rscale,
                                                       const float* __restrict__ rshift, T* __restrict__ y,
                                                       int64_t nvec, int cv, int relu, uint8_t* __restrict__ mbits) {
  const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float sc[V], sh[V], rs[V], rt[V];
  const bool has_rs = (res != nullptr) && (rscale != nullptr);
  auto coeffs = [&](int c0) {
    load_coef<V>(scale, c0, sc);
    load_coef<V>(shift, c0, sh);
    if (has_rs) { load_coef<V>(rscale, c0, rs); load_coef<V>(rshift, c0, rt); }
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) { rs[j] = 1.f; rt[j] = 0.f; }
    }
  };
  if (HOIST) coeffs((int)(tid0 % cv) * V);
  for (int64_t i = tid0; i < nvec; i += stride) {
    if (!HOIST) coeffs((int)(i % cv) * V);
    float xv[V];
    load_vec<T, V>(x + i * V, xv);
#pragma unroll
    for (int j = 0; j < V; ++j) xv[j] = xv[j] * sc[j] + sh[j];
    if (res != nullptr) {
      float rv[V];
      load_vec<T, V>(res + i * V, rv);
#pragma unroll
      for (int j = 0; j < V; ++j) xv[j] += rv[j] * rs[j] + rt[j];
    }
    if (relu) {
      if (mbits != nullptr) {
        // one mask byte per vector for the backward pass (bit j: channel j of the vector passed the ReLU)
        unsigned mb = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) mb |= (xv[j] > 0.f ? 1u : 0u) << j;
        mbits[i] = (uint8_t)mb;
      }
#pragma unroll
      for (int j = 0; j < V; ++j) xv[j] = fmaxf(xv[j], 0.f);
    }
    store_vec<T, V>(y + i * V, xv);
  }
}

template <typename T>
static int launch_apply(const void* x, const float* scale, const float* shift, const void* res, const float* rscale,
                        const float* rshift, void* y, int64_t M, int C, int relu, hipStream_t st, uint8_t* mbits = nullptr) {
  constexpr int VFULL = VecOf<T>::N;
  if (C % VFULL == 0) {
    const int64_t nvec = M * C / VFULL;
    bool hoist;
    const int g = hoist_grid(nvec, C / VFULL, &hoist);
    if (hoist)
      hipLaunchKernelGGL((bn_apply_kernel<T, VFULL, true>), dim3(g), dim3(256), 0, st, (const T*)x, scale, shift,
                         (const T*)res, rscale, rshift, (T*)y, nvec, C / VFULL, relu, mbits);
    else
      hipLaunchKernelGGL((bn_apply_kernel<T, VFULL, false>), dim3(g), dim3(256), 0, st, (const T*)x, scale, shift,
                         (const T*)res, rscale, rshift, (T*)y, nvec, C / VFULL, relu, mbits);
  } else {
    const int64_t nvec = M * C;
    hipLaunchKernelGGL((bn_apply_kernel<T, 1, false>), dim3(launch_grid(LS_BATCHNORM, grid_for(nvec, 256))), dim3(256), 0, st, (const T*)x, scale,
                       shift, (const T*)res, rscale, rshift, (T*)y, nvec, C, relu, mbits);
  }
  DL4J_RET_LAST();
}

// ---------------------------------------------------------------------------------------
// Pass B (backward): dx = g*is*(dz - dbeta/M - xhat*dgamma/M) = A*dz + B*x + K per channel with
//   A = g*is, B = -A*is*dgamma/M, K = -A*dbeta/M - B*mean;   dz = dy*[out>0]
// optionally also writes dz (gradient for the residual branch).
// ---------------------------------------------------------------------------------------
template <typename T, int V, bool HOIST>
__global__ void __launch_bounds__(256) bn_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        const T* __restrict__ out, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ sums,
                                                        const float* __restrict__ mscale,
                                                        const float* __restrict__ mshift, T* __restrict__ dx,
                                                        T* __restrict__ dz_out, int64_t nvec, int cv, int C,
                                                        float invM, const uint8_t* __restrict__ bits) {
  const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float A[V], B[V], K[V], ms[V], mt[V];
  const bool xmask = (out == nullptr) && (bits == nullptr) && (mscale != nullptr);
  auto coeffs = [&](int c0) {
    float is[V], g[V], s0[V], s1[V], mu[V];
    load_coef<V>(invstd, c0, is);
    load_coef<V>(sums, c0, s0);
    load_coef<V>(sums + C, c0, s1);
    load_coef<V>(mean, c0, mu);
    if (gamma) load_coef<V>(gamma, c0, g);
    if (xmask) { load_coef<V>(mscale, c0, ms); load_coef<V>(mshift, c0, mt); }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float a = (gamma ? g[j] : 1.f) * is[j];
      const float bq = -a * is[j] * s1[j] * invM;
      A[j] = a; B[j] = bq; K[j] = -a * s0[j] * invM - bq * mu[j];
    }
  };
  if (HOIST) coeffs((int)(tid0 % cv) * V);
  for (int64_t i = tid0; i < nvec; i += stride) {
    if (!HOIST) coeffs((int)(i % cv) * V);
    float xv[V], dv[V];
    load_vec<T, V>(x + i * V, xv);
    load_vec<T, V>(dy + i * V, dv);
    if (bits != nullptr) {
      const unsigned mb = bits[i];
#pragma unroll
      for (int j = 0; j < V; ++j) dv[j] = ((mb >> j) & 1u) ? dv[j] : 0.f;
    } else if (out != nullptr) {
      float ov[V];
      load_vec<T, V>(out + i * V, ov);
#pragma unroll
      for (int j = 0; j < V; ++j) dv[j] = ov[j] > 0.f ? dv[j] : 0.f;
    } else if (xmask) {
#pragma unroll
      for (int j = 0; j < V; ++j) dv[j] = (xv[j] * ms[j] + mt[j]) > 0.f ? dv[j] : 0.f;
    }
    if (dz_out != nullptr) store_vec<T, V>(dz_out + i * V, dv);
    float r[V];
#pragma unroll
    for (int j = 0; j < V; ++j) r[j] = A[j] * dv[j] + B[j] * xv[j] + K[j];
    store_vec<T, V>(dx + i * V, r);
  }
}

template <typename T>
static int launch_bwd_dx(const void* x, const void* dy, const void* out, const float* mean, const float* invstd,
                         const float* gamma, const float* sums, const float* mscale, const float* mshift, void* dx,
                         void* dz, int64_t M, int C, hipStream_t st, const uint8_t* bits = nullptr) {
  constexpr int VFULL = VecOf<T>::N;
  const float invM = 1.f / (float)M;
  if (C % VFULL == 0) {
    const int64_t nvec = M * C / VFULL;
    bool hoist;
    const int g = hoist_grid(nvec, C / VFULL, &hoist);
    if (hoist)
      hipLaunchKernelGGL((bn_bwd_dx_kernel<T, VFULL, true>), dim3(g), dim3(256), 0, st, (const T*)x, (const T*)dy,
                         (const T*)out, mean, invstd, gamma, sums, mscale, mshift, (T*)dx, (T*)dz, nvec, C / VFULL, C, invM, bits);
    else
      hipLaunchKernelGGL((bn_bwd_dx_kernel<T, VFULL, false>), dim3(g), dim3(256), 0, st, (const T*)x, (const T*)dy,
                         (const T*)out, mean, invstd, gamma, sums, mscale, mshift, (T*)dx, (T*)dz, nvec, C / VFULL, C, invM, bits);
  } else {
    const int64_t nvec = M * C;
    hipLaunchKernelGGL((bn_bwd_dx_kernel<T, 1, false>), dim3(launch_grid(LS_BATCHNORM, grid_for(nvec, 256))), dim3(256), 0, st, (const T*)x,
                       (const T*)dy, (const T*)out, mean, invstd, gamma, sums, mscale, mshift, (T*)dx, (T*)dz, nvec, C, C, invM, bits);
  }
  DL4J_RET_LAST();
}

// dtype codes: 0 = fp32, 1 = bf16, 2 = fp16
DL4J_API int64_t dl4j_bn_ws_floats(int C) { return colreduce_ws_floats(C); }

// Reduce [gy][2][C] partials (e.g. produced by the conv epilogue) to sums[2C].
DL4J_API int dl4j_bn_sum_partials(const float* partial, int gy, int C, float* sums, float* out0, float* out1,
                                  hipStream_t st) {
  hipLaunchKernelGGL(bn_sum_partials_kernel, dim3((2 * C + 31) / 32), dim3(1024), 0, st, partial, gy, C, sums, out0,
                     out1);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_bn_stats(const void* x, float* partial_ws, float* sums, int64_t M, int C, int dtype,
                           hipStream_t st) {
  if (dtype == 1)
    return launch_colreduce<bf16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, partial_ws, sums, nullptr,
                                       nullptr, M, C, st);
  if (dtype == 2)
    return launch_colreduce<f16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, partial_ws, sums, nullptr,
                                      nullptr, M, C, st);
  return launch_colreduce<float, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, partial_ws, sums, nullptr, nullptr,
                                    M, C, st);
}

DL4J_API int dl4j_bn_finalize(const float* sums, const float* gamma, const float* beta, const float* conv_bias,
                              const float* gmean, const float* gvar, float* mean, float* invstd, float* scale,
                              float* shift, float* dgmean, float* dgvar, int C, int64_t M, float eps, float decay,
                              int stat_mode, hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, gamma, beta, conv_bias, gmean,
                     gvar, mean, invstd, scale, shift, dgmean, dgvar, C, 1.f / (float)M, eps, decay, stat_mode);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_bn_finalize_partials(const float* partial, int gy, const float* gamma, const float* beta, const float* conv_bias,
                                       const float* gmean, const float* gvar, float* mean, float* invstd, float* scale, float* shift,
                                       float* dgmean, float* dgvar, int C, int64_t M, float eps, float decay, int stat_mode,
                                       hipStream_t st) {
  if (C <= 0 || gy <= 0) return 0;
  hipLaunchKernelGGL(bn_finalize_partials_kernel, dim3((C + 7) / 8), dim3(1024), 0, st, partial, gy, gamma, beta, conv_bias, gmean,
                     gvar, mean, invstd, scale, shift, dgmean, dgvar, C, 1.f / (float)M, eps, decay, stat_mode, 0);
  DL4J_RET_LAST();
}

// Same for partial sums of (x - pivot), pivot = gmean - conv_bias (needs gmean).
DL4J_API int dl4j_bn_finalize_partials_shifted(const float* partial, int gy, const float* gamma, const float* beta, const float* conv_bias,
                                               const float* gmean, const float* gvar, float* mean, float* invstd, float* scale, float* shift,
                                               float* dgmean, float* dgvar, int C, int64_t M, float eps, float decay, int stat_mode,
                                               hipStream_t st) {
  if (C <= 0 || gy <= 0) return 0;
  if (gmean == nullptr) return -2;
  hipLaunchKernelGGL(bn_finalize_partials_kernel, dim3((C + 7) / 8), dim3(1024), 0, st, partial, gy, gamma, beta, conv_bias, gmean,
                     gvar, mean, invstd, scale, shift, dgmean, dgvar, C, 1.f / (float)M, eps, decay, stat_mode, 1);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_bn_infer_coeffs(const float* gamma, const float* beta, const float* gmean, const float* gvar,
                                  const float* conv_bias, float* scale, float* shift, int C, float eps, int log_std,
                                  hipStream_t st) {
  hipLaunchKernelGGL(bn_infer_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, gmean, gvar,
                     conv_bias, scale, shift, C, eps, log_std);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_bn_apply(const void* x, const float* scale, const float* shift, const void* res,
                           const float* rscale, const float* rshift, void* y, int64_t M, int C, int relu, int dtype,
                           hipStream_t st) {
  if (dtype == 1) return launch_apply<bf16_t>(x, scale, shift, res, rscale, rshift, y, M, C, relu, st);
  if (dtype == 2) return launch_apply<f16_t>(x, scale, shift, res, rscale, rshift, y, M, C, relu, st);
  return launch_apply<float>(x, scale, shift, res, rscale, rshift, y, M, C, relu, st);
}

DL4J_API int dl4j_bn_ticket_counters() { return BN_TICKET_COUNTERS; }
DL4J_API int64_t dl4j_bn_ticket_gpart_floats(int C) { return (int64_t)BN_TICKET_GROUPS * 2 * C; }

// Same as dl4j_bn_bwd_reduce, second reduction stage fused into the launch (ticket.h). cnt: BN_TICKET_COUNTERS zero-initialised
// uint32 (self-resetting, may be shared by all layers on a stream); gpart: dl4j_bn_ticket_gpart_floats(C) floats.
DL4J_API int dl4j_bn_bwd_reduce_fused(const void* x, const void* dy, const void* out, const float* mean,
                                      const float* invstd, const float* mscale, const float* mshift, float* partial_ws,
                                      float* sums, float* dbeta_out, float* dgamma_out, int64_t M, int C, int dtype,
                                      float* gpart, unsigned* cnt, hipStream_t st) {
  if (dtype == 1)
    return launch_colreduce<bf16_t, 1>(x, dy, out, mean, invstd, mscale, mshift, partial_ws, sums, dbeta_out, dgamma_out,
                                       M, C, st, gpart, cnt);
  if (dtype == 2)
    return launch_colreduce<f16_t, 1>(x, dy, out, mean, invstd, mscale, mshift, partial_ws, sums, dbeta_out, dgamma_out,
                                      M, C, st, gpart, cnt);
  return launch_colreduce<float, 1>(x, dy, out, mean, invstd, mscale, mshift, partial_ws, sums, dbeta_out, dgamma_out, M, C,
                                    st, gpart, cnt);
}

// mscale/mshift (optional, used when out == nullptr): recompute the fused-ReLU mask from x instead of reading the output.
DL4J_API int dl4j_bn_bwd_reduce(const void* x, const void* dy, const void* out, const float* mean,
                                const float* invstd, const float* mscale, const float* mshift, float* partial_ws,
                                float* sums, float* dbeta_out, float* dgamma_out, int64_t M, int C, int dtype,
                                hipStream_t st) {
  if (dtype == 1)
    return launch_colreduce<bf16_t, 1>(x, dy, out, mean, invstd, mscale, mshift, partial_ws, sums, dbeta_out, dgamma_out,
                                       M, C, st);
  if (dtype == 2)
    return launch_colreduce<f16_t, 1>(x, dy, out, mean, invstd, mscale, mshift, partial_ws, sums, dbeta_out, dgamma_out,
                                      M, C, st);
  return launch_colreduce<float, 1>(x, dy, out, mean, invstd, mscale, mshift, partial_ws, sums, dbeta_out, dgamma_out, M, C,
                                    st);
}

DL4J_API int dl4j_bn_bwd_dx(const void* x, const void* dy, const void* out, const float* mean, const float* invstd,
                            const float* gamma, const float* sums, const float* mscale, const float* mshift, void* dx,
                            void* dz, int64_t M, int C, int dtype, hipStream_t st) {
  if (dtype == 1) return launch_bwd_dx<bf16_t>(x, dy, out, mean, invstd, gamma, sums, mscale, mshift, dx, dz, M, C, st);
  if (dtype == 2) return launch_bwd_dx<f16_t>(x, dy, out, mean, invstd, gamma, sums, mscale, mshift, dx, dz, M, C, st);
  return launch_bwd_dx<float>(x, dy, out, mean, invstd, gamma, sums, mscale, mshift, dx, dz, M, C, st);
}

// ---- ReLU mask as bits: one byte per V-channel vector (V = 8 for bf16 / fp16, 4 for fp32, 1 when C is not a multiple of that),
// written by the forward apply and read by both backward passes instead of the stored activation: 1/16 of the bytes for bf16.
DL4J_API int dl4j_bn_mask_vec(int C, int dtype) {
  const int v = dtype == 0 ? 4 : 8;
  return (C % v == 0) ? v : 1;
}

DL4J_API int dl4j_bn_apply_mask(const void* x, const float* scale, const float* shift, const void* res,
                                const float* rscale, const float* rshift, void* y, void* mask_bits, int64_t M, int C, int relu,
                                int dtype, hipStream_t st) {
  if (dtype == 1) return launch_apply<bf16_t>(x, scale, shift, res, rscale, rshift, y, M, C, relu, st, (uint8_t*)mask_bits);
  if (dtype == 2) return launch_apply<f16_t>(x, scale, shift, res, rscale, rshift, y, M, C, relu, st, (uint8_t*)mask_bits);
  return launch_apply<float>(x, scale, shift, res, rscale, rshift, y, M, C, relu, st, (uint8_t*)mask_bits);
}

DL4J_API int dl4j_bn_bwd_reduce_mask(const void* x, const void* dy, const void* mask_bits, const float* mean,
                                     const float* invstd, float* partial_ws, float* sums, float* dbeta_out, float* dgamma_out,
                                     int64_t M, int C, int dtype, hipStream_t st) {
  const uint8_t* b = (const uint8_t*)mask_bits;
  if (dtype == 1)
    return launch_colreduce<bf16_t, 1>(x, dy, nullptr, mean, invstd, nullptr, nullptr, partial_ws, sums, dbeta_out, dgamma_out, M, C, st,
                                       nullptr, nullptr, b);
  if (dtype == 2)
    return launch_colreduce<f16_t, 1>(x, dy, nullptr, mean, invstd, nullptr, nullptr, partial_ws, sums, dbeta_out, dgamma_out, M, C, st,
                                      nullptr, nullptr, b);
  return launch_colreduce<float, 1>(x, dy, nullptr, mean, invstd, nullptr, nullptr, partial_ws, sums, dbeta_out, dgamma_out, M, C, st,
                                    nullptr, nullptr, b);
}

DL4J_API int dl4j_bn_bwd_dx_mask(const void* x, const void* dy, const void* mask_bits, const float* mean, const float* invstd,
                                 const float* gamma, const float* sums, void* dx, void* dz, int64_t M, int C, int dtype,
                                 hipStream_t st) {
  const uint8_t* b = (const uint8_t*)mask_bits;
  if (dtype == 1) return launch_bwd_dx<bf16_t>(x, dy, nullptr, mean, invstd, gamma, sums, nullptr, nullptr, dx, dz, M, C, st, b);
  if (dtype == 2) return launch_bwd_dx<f16_t>(x, dy, nullptr, mean, invstd, gamma, sums, nullptr, nullptr, dx, dz, M, C, st, b);
  return launch_bwd_dx<float>(x, dy, nullptr, mean, invstd, gamma, sums, nullptr, nullptr, dx, dz, M, C, st, b);
}
