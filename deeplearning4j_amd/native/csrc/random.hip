// Device RNG fills for gfx950, bit-identical to the host routines in rt_random.cpp (philox.h).
//
// Covers the legacy random op family (opNums: SURVEY Appendix E "RANDOM"; reference kernels
// libnd4j/include/loops/cuda/random.cu, ops/random_ops.h, ops/special_random_ops.h) and the custom-op
// RNG helpers (helpers/cuda/random.cu: fillUniform/Gamma/Poisson, dropout.cu).
// Every element is generated from (key, stream, offset + element index), so results do not depend on
// the launch geometry and a HIP-graph replay with a bumped device-side offset stays reproducible.
#include "common.h"
#include "philox.h"

enum RandOp {
  R_UNIFORM = 0, R_DROPOUT = 1, R_DROPOUT_INVERTED = 2, R_PROB_MERGE = 3, R_LINSPACE = 4, R_CHOICE = 5,
  R_GAUSSIAN = 6, R_BERNOULLI = 7, R_BINOMIAL = 8, R_BINOMIAL_EX = 9, R_LOGNORMAL = 10, R_TRUNC_NORMAL = 11,
  R_ALPHA_DROPOUT = 12, R_EXPONENTIAL = 13, R_EXPONENTIAL_INV = 14, R_POISSON = 15, R_GAMMA = 16
};

struct RandArgs {
  uint64_t key, stream, offset;
  float a, b, c, d;     // op-dependent (see switch)
  int64_t n;
  int op;
  int trials;
};

__device__ __forceinline__ float u_at(const RandArgs& r, uint64_t i) { return philox::u01(philox::bits_at(r.key, r.stream, r.offset + i)); }

__device__ __forceinline__ float gauss_at(const RandArgs& r, uint64_t i) {
  uint64_t p = i >> 1;
  float u1 = u_at(r, 2 * p), u2 = u_at(r, 2 * p + 1);
  float rad = sqrtf(-2.0f * logf(u1)), t = 6.28318530717958647692f * u2;
  return (i & 1) ? rad * sinf(t) : rad * cosf(t);
}

// secondary independent draws for rejection loops: same key, stream advanced by the attempt number
__device__ __forceinline__ float u_try(const RandArgs& r, uint64_t i, uint32_t attempt, uint32_t k) {
  return philox::u01(philox::bits_at(r.key, r.stream + 0x9E3779B97F4A7C15ull * (attempt + 1), r.offset + 4 * i + k));
}
__device__ __forceinline__ float gauss_try(const RandArgs& r, uint64_t i, uint32_t attempt) {
  float u1 = u_try(r, i, attempt, 0), u2 = u_try(r, i, attempt, 1);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

template <typename T>
__global__ void random_fill_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ z, RandArgs r) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.n; i += (int64_t)gridDim.x * blockDim.x) {
    float out;
    switch (r.op) {
      case R_UNIFORM: out = r.a + (r.b - r.a) * u_at(r, i); break;                       // a=from, b=to
      case R_DROPOUT: out = u_at(r, i) < r.a ? (float)Cvt<T>::ld(x, i) : 0.f; break;      // a = keep prob
      case R_DROPOUT_INVERTED: out = u_at(r, i) < r.a ? (float)Cvt<T>::ld(x, i) / r.a : 0.f; break;
      case R_PROB_MERGE: out = u_at(r, i) < r.a ? (float)Cvt<T>::ld(x, i) : (float)Cvt<T>::ld(y, i); break;
      case R_LINSPACE: out = r.n > 1 ? r.a + (r.b - r.a) * ((float)i / (float)(r.n - 1)) : r.a; break;
      case R_GAUSSIAN: out = r.a + r.b * gauss_at(r, i); break;                          // a=mean, b=std
      case R_BERNOULLI: out = u_at(r, i) < r.a ? 1.f : 0.f; break;
      case R_BINOMIAL:
      case R_BINOMIAL_EX: {
        int k = 0;
        for (int t = 0; t < r.trials; ++t) k += u_try(r, i, t >> 2, t & 3) < r.a;
        out = (float)k;
      } break;
      case R_LOGNORMAL: out = expf(r.a + r.b * gauss_at(r, i)); break;
      case R_TRUNC_NORMAL: {                                                            // |z| <= 2 sigma, resample
        float g = gauss_at(r, i);
        for (uint32_t att = 0; att < 16 && fabsf(g) > 2.f; ++att) g = gauss_try(r, i, att);
        out = r.a + r.b * fminf(fmaxf(g, -2.f), 2.f);
      } break;
      case R_ALPHA_DROPOUT: {   // a = keep prob, b = alpha', c = a-coef, d = b-coef
        float v = u_at(r, i) < r.a ? (float)Cvt<T>::ld(x, i) : r.b;
        out = r.c * v + r.d;
      } break;
      case R_EXPONENTIAL: out = -logf(u_at(r, i)) / r.a; break;                          // a = lambda
      case R_EXPONENTIAL_INV: out = 1.f - expf(-r.a * (float)Cvt<T>::ld(x, i)); break;
      case R_POISSON: {                                                                 // a = lambda
        float lam = x ? (float)Cvt<T>::ld(x, i) : r.a;
        if (lam < 30.f) {
          float L = expf(-lam), p = 1.f;
          int k = 0;
          for (uint32_t t = 0; t < 256; ++t) {
            p *= u_try(r, i, t >> 2, t & 3);
            if (p <= L) break;
            ++k;
          }
          out = (float)k;
        } else {
          out = fmaxf(0.f, floorf(lam + sqrtf(lam) * gauss_at(r, i) + 0.5f));
        }
      } break;
      case R_GAMMA: {   // Marsaglia-Tsang; a = alpha (shape), b = beta (rate)
        float alpha = x ? (float)Cvt<T>::ld(x, i) : r.a;
        float boost = 1.f;
        if (alpha < 1.f) {
          boost = powf(u_try(r, i, 63, 3), 1.f / alpha);
          alpha += 1.f;
        }
        float d = alpha - 1.f / 3.f, c = rsqrtf(9.f * d), res = d;
        for (uint32_t att = 0; att < 32; ++att) {
          float g = gauss_try(r, i, att), v = 1.f + c * g;
          if (v <= 0.f) continue;
          v = v * v * v;
          float u = u_try(r, i, att, 2);
          if (logf(u) < 0.5f * g * g + d - d * v + d * logf(v)) {
            res = d * v;
            break;
          }
        }
        out = res * boost / (r.b > 0.f ? r.b : 1.f);
      } break;
      default: out = 0.f;
    }
    Cvt<T>::st(z, i, out);
  }
}

// Choice: z[i] = source[k] where k = first index with cdf[k] >= u  (probs given as cumulative array)
template <typename T>
__global__ void random_choice_kernel(const T* __restrict__ src, const float* __restrict__ cdf, int m, T* __restrict__ z, RandArgs r) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.n; i += (int64_t)gridDim.x * blockDim.x) {
    float u = u_at(r, i) * cdf[m - 1];
    int lo = 0, hi = m - 1;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (cdf[mid] >= u) hi = mid; else lo = mid + 1;
    }
    z[i] = src[lo];
  }
}

// args: {a, b, c, d}; dtype 0 fp32, 1 bf16, 2 fp64
DL4J_API int dl4j_random_fill(int op, const void* x, const void* y, void* z, int64_t n, uint64_t key, uint64_t stream,
                              uint64_t offset, const float* args, int trials, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  RandArgs r{key, stream, offset, args[0], args[1], args[2], args[3], n, op, trials};
  int g = grid_for(n, 256 * 4);
  if (dtype == 0) random_fill_kernel<float><<<g, 256, 0, st>>>((const float*)x, (const float*)y, (float*)z, r);
  else if (dtype == 1) random_fill_kernel<bf16_t><<<g, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)y, (bf16_t*)z, r);
  else if (dtype == 2) random_fill_kernel<double><<<g, 256, 0, st>>>((const double*)x, (const double*)y, (double*)z, r);
  else return 1;
  DL4J_RET_LAST();
}

DL4J_API int dl4j_random_choice(const float* src, const float* cdf, int m, float* z, int64_t n, uint64_t key, uint64_t stream,
                                uint64_t offset, hipStream_t st) {
  if (n <= 0 || m <= 0) return 0;
  RandArgs r{key, stream, offset, 0, 0, 0, 0, n, R_CHOICE, 0};
  random_choice_kernel<float><<<grid_for(n, 256 * 4), 256, 0, st>>>(src, cdf, m, z, r);
  DL4J_RET_LAST();
}
