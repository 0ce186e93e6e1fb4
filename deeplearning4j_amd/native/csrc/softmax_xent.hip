// Fused softmax + multi-class cross entropy (LossMCXENT / LossNegativeLogLikelihood) forward,
// score and gradient in one pass per row:  p = softmax(z), grad = p - y,
// score_row = -sum(y * log(clip(p, 1e-10, 1)))   (reference: nd4j-api/.../lossfunctions/impl/LossMCXENT.java:96-164).
// Also stand-alone softmax / log-softmax (reference kernel: helpers/cuda/activations.cu softMaxCuda).
// One workgroup (256 threads = 4 waves) per row; wave64 shuffles + one LDS hop for block reductions.
#include "common.h"

template <typename T, typename YT>
__global__ void __launch_bounds__(256) softmax_xent_kernel(const T* __restrict__ z, const YT* __restrict__ y,
                                                           const int64_t* __restrict__ ylab,
                                                           T* __restrict__ probs, T* __restrict__ grad,
                                                           float* __restrict__ row_loss, float* __restrict__ score,
                                                           int N) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const T* zr = z + row * N;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 256) mx = fmaxf(mx, Cvt<T>::ld(zr, i));
  mx = block_max(mx, red);
  float se = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) se += __expf(Cvt<T>::ld(zr, i) - mx);
  se = block_sum(se, red);
  const float inv = 1.f / se;
  float loss = 0.f;
  const int64_t lab = ylab ? ylab[row] : -1;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float p = __expf(Cvt<T>::ld(zr, i) - mx) * inv;
    const float yv = ylab ? (i == lab ? 1.f : 0.f) : Cvt<YT>::ld(y + row * N, i);
    if (probs) Cvt<T>::st(probs + row * N, i, p);
    if (grad) Cvt<T>::st(grad + row * N, i, p - yv);
    if (yv != 0.f) loss -= yv * __logf(fminf(fmaxf(p, 1e-10f), 1.f));
  }
  loss = block_sum(loss, red);
  if (threadIdx.x == 0) {
    if (row_loss) row_loss[row] = loss;
    if (score) atomicAdd(score, loss);
  }
}

// mode 0: softmax, 1: log-softmax
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const T* __restrict__ z, T* __restrict__ out, int N,
                                                           int mode) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const T* zr = z + row * N;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 256) mx = fmaxf(mx, Cvt<T>::ld(zr, i));
  mx = block_max(mx, red);
  float se = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) se += __expf(Cvt<T>::ld(zr, i) - mx);
  se = block_sum(se, red);
  const float inv = 1.f / se, lse = __logf(se);
  for (int i = threadIdx.x; i < N; i += 256) {
    const float v = Cvt<T>::ld(zr, i) - mx;
    Cvt<T>::st(out + row * N, i, mode == 0 ? __expf(v) * inv : v - lse);
  }
}

// The same with the row held in registers: one 16-byte load per 8 bf16 / 4 fp32 elements, one read and one write of the row in total
// (the three-pass form above reads it three times with 2- / 4-byte accesses). NV = 16-byte pieces per thread; rows up to 256 * NV pieces.
template <typename T> struct Pack;
template <> struct Pack<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void ld(const float* p, float* x) { const float4 v = *(const float4*)p; x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
  static __device__ __forceinline__ void st(float* p, const float* x) { *(float4*)p = make_float4(x[0], x[1], x[2], x[3]); }
};
template <> struct Pack<bf16_t> {
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

template <typename T, int NV>
__global__ void __launch_bounds__(256) softmax_rows_vec_kernel(const T* __restrict__ z, T* __restrict__ out, int N, int mode) {
  constexpr int V = Pack<T>::V;
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const T* zr = z + row * N;
  T* orow = out + row * N;
  const int nvec = N / V;
  float x[NV][V];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vi = threadIdx.x + k * 256;
    if (vi < nvec) {
      Pack<T>::ld(zr + (int64_t)vi * V, x[k]);
#pragma unroll
      for (int j = 0; j < V; ++j) mx = fmaxf(mx, x[k][j]);
    }
  }
  mx = block_max(mx, red);
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (threadIdx.x + k * 256 < nvec) {
#pragma unroll
      for (int j = 0; j < V; ++j) { x[k][j] -= mx; se += __expf(x[k][j]); }
    }
  se = block_sum(se, red);
  const float inv = 1.f / se, lse = __logf(se);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vi = threadIdx.x + k * 256;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < V; ++j) x[k][j] = mode == 0 ? __expf(x[k][j]) * inv : x[k][j] - lse;
      Pack<T>::st(orow + (int64_t)vi * V, x[k]);
    }
  }
}

template <typename T, int NV>
__global__ void __launch_bounds__(256) softmax_bwd_vec_kernel(const T* __restrict__ p, const T* __restrict__ dy, T* __restrict__ dz, int N) {
  constexpr int V = Pack<T>::V;
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const int nvec = N / V;
  float pv[NV][V], gv[NV][V];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vi = threadIdx.x + k * 256;
    if (vi < nvec) {
      Pack<T>::ld(p + row * N + (int64_t)vi * V, pv[k]);
      Pack<T>::ld(dy + row * N + (int64_t)vi * V, gv[k]);
#pragma unroll
      for (int j = 0; j < V; ++j) s += pv[k][j] * gv[k][j];
    }
  }
  s = block_sum(s, red);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vi = threadIdx.x + k * 256;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < V; ++j) pv[k][j] *= gv[k][j] - s;
      Pack<T>::st(dz + row * N + (int64_t)vi * V, pv[k]);
    }
  }
}

// 16-byte pieces per thread for a row of N elements, 0 when the register-resident form does not apply
template <typename T>
static int softmax_vec_nv(const void* a, const void* b, const void* c, int N, int max_nv) {
  constexpr int V = Pack<T>::V;
  if (N % V != 0 || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)c & 15)) return 0;
  const int per_thread = (N / V + 255) / 256;
  for (int nv = 1; nv <= max_nv; nv *= 2)
    if (per_thread <= nv) return nv;
  return 0;
}

template <typename T>
static bool launch_softmax_rows_vec(const T* z, T* out, int64_t M, int N, int mode, hipStream_t st) {
  switch (softmax_vec_nv<T>(z, out, nullptr, N, 8)) {
    case 1: hipLaunchKernelGGL((softmax_rows_vec_kernel<T, 1>), dim3((unsigned)M), dim3(256), 0, st, z, out, N, mode); return true;
    case 2: hipLaunchKernelGGL((softmax_rows_vec_kernel<T, 2>), dim3((unsigned)M), dim3(256), 0, st, z, out, N, mode); return true;
    case 4: hipLaunchKernelGGL((softmax_rows_vec_kernel<T, 4>), dim3((unsigned)M), dim3(256), 0, st, z, out, N, mode); return true;
    case 8: hipLaunchKernelGGL((softmax_rows_vec_kernel<T, 8>), dim3((unsigned)M), dim3(256), 0, st, z, out, N, mode); return true;
    default: return false;
  }
}

template <typename T>
static bool launch_softmax_bwd_vec(const T* p, const T* dy, T* dz, int64_t M, int N, hipStream_t st) {
  switch (softmax_vec_nv<T>(p, dy, dz, N, 4)) {
    case 1: hipLaunchKernelGGL((softmax_bwd_vec_kernel<T, 1>), dim3((unsigned)M), dim3(256), 0, st, p, dy, dz, N); return true;
    case 2: hipLaunchKernelGGL((softmax_bwd_vec_kernel<T, 2>), dim3((unsigned)M), dim3(256), 0, st, p, dy, dz, N); return true;
    case 4: hipLaunchKernelGGL((softmax_bwd_vec_kernel<T, 4>), dim3((unsigned)M), dim3(256), 0, st, p, dy, dz, N); return true;
    default: return false;
  }
}

// softmax backward: dz = p * (dy - sum(dy*p))
template <typename T>
__global__ void __launch_bounds__(256) softmax_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dy,
                                                          T* __restrict__ dz, int N) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) s += Cvt<T>::ld(p + row * N, i) * Cvt<T>::ld(dy + row * N, i);
  s = block_sum(s, red);
  for (int i = threadIdx.x; i < N; i += 256) {
    const float pv = Cvt<T>::ld(p + row * N, i);
    Cvt<T>::st(dz + row * N, i, pv * (Cvt<T>::ld(dy + row * N, i) - s));
  }
}

// Zeroing through a kernel, not hipMemsetAsync: a 4-byte memset NODE inside a captured HIP graph was observed to be dropped /
// mis-ordered on replay when a second graph had been instantiated earlier in the process (the score then accumulated on top
// of whatever the recycled block held; tools/repro_two_captured_nets.py).
__global__ void zero_f32_kernel(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

DL4J_API int dl4j_softmax_xent(const void* z, const void* y, const int64_t* ylab, void* probs, void* grad,
                               float* row_loss, float* score, int64_t M, int N, int dtype, hipStream_t st) {
  if (score) hipLaunchKernelGGL(zero_f32_kernel, dim3(1), dim3(64), 0, st, score, 1);
  if (M <= 0) return 0;
  // dtype: low byte = logits / probs / grad type, second byte = label type (0 fp32, 1 bf16); labels keep the network dtype
  const int zt = dtype & 0xff, yt = (dtype >> 8) & 0xff;
  if (zt == 1 && yt == 1)
    hipLaunchKernelGGL((softmax_xent_kernel<bf16_t, bf16_t>), dim3((unsigned)M), dim3(256), 0, st, (const bf16_t*)z,
                       (const bf16_t*)y, ylab, (bf16_t*)probs, (bf16_t*)grad, row_loss, score, N);
  else if (zt == 1)
    hipLaunchKernelGGL((softmax_xent_kernel<bf16_t, float>), dim3((unsigned)M), dim3(256), 0, st, (const bf16_t*)z,
                       (const float*)y, ylab, (bf16_t*)probs, (bf16_t*)grad, row_loss, score, N);
  else if (yt == 0)
    hipLaunchKernelGGL((softmax_xent_kernel<float, float>), dim3((unsigned)M), dim3(256), 0, st, (const float*)z,
                       (const float*)y, ylab, (float*)probs, (float*)grad, row_loss, score, N);
  else
    return -2;
  DL4J_RET_LAST();
}

DL4J_API int dl4j_softmax_rows(const void* z, void* out, int64_t M, int N, int mode, int dtype, hipStream_t st) {
  if (M <= 0) return 0;
  if (dtype == 1) {
    if (!launch_softmax_rows_vec<bf16_t>((const bf16_t*)z, (bf16_t*)out, M, N, mode, st))
      hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, dim3((unsigned)M), dim3(256), 0, st, (const bf16_t*)z, (bf16_t*)out, N, mode);
  } else {
    if (!launch_softmax_rows_vec<float>((const float*)z, (float*)out, M, N, mode, st))
      hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3((unsigned)M), dim3(256), 0, st, (const float*)z, (float*)out, N, mode);
  }
  DL4J_RET_LAST();
}

DL4J_API int dl4j_softmax_bwd(const void* p, const void* dy, void* dz, int64_t M, int N, int dtype,
                              hipStream_t st) {
  if (M <= 0) return 0;
  if (dtype == 1) {
    if (!launch_softmax_bwd_vec<bf16_t>((const bf16_t*)p, (const bf16_t*)dy, (bf16_t*)dz, M, N, st))
      hipLaunchKernelGGL(softmax_bwd_kernel<bf16_t>, dim3((unsigned)M), dim3(256), 0, st, (const bf16_t*)p, (const bf16_t*)dy, (bf16_t*)dz, N);
  } else {
    if (!launch_softmax_bwd_vec<float>((const float*)p, (const float*)dy, (float*)dz, M, N, st))
      hipLaunchKernelGGL(softmax_bwd_kernel<float>, dim3((unsigned)M), dim3(256), 0, st, (const float*)p, (const float*)dy, (float*)dz, N);
  }
  DL4J_RET_LAST();
}
