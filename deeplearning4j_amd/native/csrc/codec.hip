// Gradient-compression codecs and multi-array DP primitives for gfx950 (wave64).
//
// Wire formats = reference (libnd4j/include/loops/cuda/type_conversions.cu:215-499): see rt_codec.cpp.
// The reference's threshold encoder is written around 32-lane ballots and hard-coded 5-step warp
// prefix loops (type_conversions.cu:223-299) and launches one thread per element. Here:
//  * P1: one block per 4096-element tile, per-wave popcount of a 64-bit ballot, block total to
//        counts[tile] (no atomics, no cross-block hand-off).
//  * scan: single-block exclusive scan over tile counts (tiles = n/4096 — 6k for ResNet-50).
//  * P3: same tiling; position inside the tile = running block offset + wave offsets (LDS) +
//        mbcnt (lanes below me in the 64-bit ballot). Output is in index order, identical to the
//        host codec.
// Multi-array average / accumulate (loops/cuda/specials/averagingKernel.cu:29-76,
// accumulateKernel.cu:38-68): pointer table in device memory, float4 grid-stride; used by
// Nd4j.averageAndPropagate / accumulate for arrays that live on this GPU (cross-GPU reduction goes
// through RCCL, parallel/comm.py).
#include "common.h"

#define TILE 4096
#define TPB 256

__device__ __forceinline__ int lanes_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__global__ __launch_bounds__(TPB) void thr_count_kernel(const float* __restrict__ x, int64_t n, float tau,
                                                        int* __restrict__ counts) {
  __shared__ int wsum[TPB / 64];
  const int64_t base = (int64_t)blockIdx.x * TILE;
  int c = 0;
#pragma unroll 4
  for (int k = 0; k < TILE / TPB; ++k) {
    int64_t i = base + k * TPB + threadIdx.x;
    bool pass = i < n && fabsf(x[i]) >= tau;
    c += __popcll(__ballot(pass));     // wave-uniform
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of counts[0..m) in place; total -> counts[m]; also writes the message header
__global__ __launch_bounds__(1024) void thr_scan_kernel(int* __restrict__ counts, int m, int* __restrict__ out,
                                                        int64_t n, float tau, int max_elems) {
  __shared__ int part[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < m; base += 1024) {
    int i = base + threadIdx.x;
    int v = i < m ? counts[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {     // Hillis-Steele inclusive scan
      int t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    int incl = part[threadIdx.x];
    if (i < m) counts[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts[m] = carry;
    out[0] = carry < max_elems ? carry : max_elems;
    out[1] = (int)n;
    out[2] = __float_as_int(tau);
    out[3] = 0;
  }
}

__global__ __launch_bounds__(TPB) void thr_emit_kernel(float* __restrict__ x, int64_t n, float tau,
                                                       const int* __restrict__ offsets, int* __restrict__ out,
                                                       int max_elems) {
  __shared__ int wcnt[TPB / 64];
  const int64_t base = (int64_t)blockIdx.x * TILE;
  int running = offsets[blockIdx.x];
  const int wid = threadIdx.x >> 6;
  for (int k = 0; k < TILE / TPB; ++k) {
    int64_t i = base + k * TPB + threadIdx.x;
    float v = i < n ? x[i] : 0.f;
    bool pass = i < n && fabsf(v) >= tau;
    uint64_t b = __ballot(pass);
    int below = lanes_below(b);
    if ((threadIdx.x & 63) == 0) wcnt[wid] = __popcll(b);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) {
      int c = wcnt[w];
      woff += w < wid ? c : 0;
      tot += c;
    }
    int pos = running + woff + below;
    if (pass && pos < max_elems) {
      out[4 + pos] = v > 0.f ? (int)(i + 1) : -(int)(i + 1);
      x[i] = v > 0.f ? v - tau : v + tau;
    }
    running += tot;
    __syncthreads();
  }
}

__global__ void thr_decode_kernel(const int* __restrict__ enc, float* __restrict__ z) {
  const int count = enc[0];
  const float tau = __int_as_float(enc[2]);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
    int el = enc[4 + k];
    int idx = (el < 0 ? -el : el) - 1;
    z[idx] += el < 0 ? -tau : tau;        // indices are unique within a message
  }
}

__global__ __launch_bounds__(TPB) void bitmap_encode_kernel(float* __restrict__ x, int64_t n, float tau,
                                                            int* __restrict__ out, int* __restrict__ block_counts) {
  __shared__ int wsum[TPB / 64];
  const float half = tau * 0.5f;
  int cnt = 0;
  const int64_t rounds = (n + (int64_t)gridDim.x * TPB - 1) / ((int64_t)gridDim.x * TPB);
  for (int64_t r = 0; r < rounds; ++r) {
    int64_t i = (r * gridDim.x + blockIdx.x) * TPB + threadIdx.x;
    float v = i < n ? x[i] : 0.f;
    float a = fabsf(v);
    bool has = i < n && a >= tau;
    bool halfstep = i < n && !has && a >= half && v < 0.f;
    bool sign = (has && v < 0.f) || halfstep;
    uint64_t hb = __ballot(has), sb = __ballot(sign);
    if (has) x[i] = v < 0.f ? v + tau : v - tau;
    else if (halfstep) x[i] = v + half;
    cnt += __popcll(hb) + __popcll(__ballot(halfstep));
    int lane = threadIdx.x & 63;
    if ((lane & 15) == 0 && i < n) {
      int sh = lane;    // 0,16,32,48
      out[4 + (i >> 4)] = (int)((hb >> sh) & 0xffffull) | (int)(((sb >> sh) & 0xffffull) << 16);
    }
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void bitmap_header_kernel(const int* __restrict__ block_counts, int m,
                                                             int* __restrict__ out, int64_t n, float tau) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < m; i += blockDim.x) s += (float)block_counts[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    out[0] = (int)s;
    out[1] = (int)n;
    out[2] = __float_as_int(tau);
    out[3] = 1;
  }
}

__global__ void bitmap_decode_kernel(const int* __restrict__ enc, float* __restrict__ z) {
  const int64_t n = enc[1];
  const float tau = __int_as_float(enc[2]), half = tau * 0.5f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int word = enc[4 + (i >> 4)];
    int bit = (int)(i & 15);
    bool has = word & (1 << bit), sign = word & (1 << (bit + 16));
    if (has) z[i] += sign ? -tau : tau;
    else if (sign) z[i] -= half;
  }
}

// ---- multi-array average / accumulate ------------------------------------------------------------
__global__ void multi_reduce_kernel(const float* const* __restrict__ ptrs, int narr, int64_t len, float* __restrict__ z,
                                    float scale, int propagate) {
  const int64_t n4 = len >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < narr; ++a) acc += reinterpret_cast<const f32x4*>(ptrs[a])[i];
    acc *= scale;
    if (z) reinterpret_cast<f32x4*>(z)[i] = acc;
    if (propagate)
      for (int a = 0; a < narr; ++a) reinterpret_cast<f32x4*>(const_cast<float*>(ptrs[a]))[i] = acc;
  }
  if (blockIdx.x == 0) {
    for (int64_t i = (n4 << 2) + threadIdx.x; i < len; i += blockDim.x) {
      float acc = 0.f;
      for (int a = 0; a < narr; ++a) acc += ptrs[a][i];
      acc *= scale;
      if (z) z[i] = acc;
      if (propagate)
        for (int a = 0; a < narr; ++a) const_cast<float*>(ptrs[a])[i] = acc;
    }
  }
}

// pullRows: z[r, :] = x[idx[r], :]   (row-major, contiguous rows of `width`)
template <typename T>
__global__ void pull_rows_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx, T* __restrict__ z, int64_t rows,
                                 int64_t width, int64_t x_rows) {
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    int64_t s = idx[r];
    if (s < 0 || s >= x_rows) continue;     // validated on host too; never read out of bounds
    for (int64_t c = threadIdx.x; c < width; c += blockDim.x) z[r * width + c] = x[s * width + c];
  }
}

// ---- C API ----------------------------------------------------------------------------------------
DL4J_API int64_t dl4j_threshold_ws_ints(int64_t n) { return (n + TILE - 1) / TILE + 1; }

// counts_ws: dl4j_threshold_ws_ints(n) ints. out: 4 + max_elems ints. Total count readable at out[0].
DL4J_API int dl4j_threshold_encode(float* x, int64_t n, float tau, int* out, int max_elems, int* counts_ws, hipStream_t st) {
  if (n <= 0 || n >= 2147483647LL) return 1;
  int tiles = (int)((n + TILE - 1) / TILE);
  thr_count_kernel<<<tiles, TPB, 0, st>>>(x, n, tau, counts_ws);
  thr_scan_kernel<<<1, 1024, 0, st>>>(counts_ws, tiles, out, n, tau, max_elems);
  thr_emit_kernel<<<tiles, TPB, 0, st>>>(x, n, tau, counts_ws, out, max_elems);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_threshold_decode(const int* enc, int max_count, float* z, hipStream_t st) {
  int g = grid_for(max_count > 0 ? max_count : 1, 256);
  thr_decode_kernel<<<g, 256, 0, st>>>(enc, z);
  DL4J_RET_LAST();
}

// block_ws: >= 2048 ints
DL4J_API int dl4j_bitmap_encode(float* x, int64_t n, float tau, int* out, int* block_ws, hipStream_t st) {
  if (n <= 0 || n >= 2147483647LL) return 1;
  int g = grid_for(n, TPB);
  bitmap_encode_kernel<<<g, TPB, 0, st>>>(x, n, tau, out, block_ws);
  bitmap_header_kernel<<<1, 1024, 0, st>>>(block_ws, g, out, n, tau);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_bitmap_decode(const int* enc, int64_t n, float* z, hipStream_t st) {
  bitmap_decode_kernel<<<grid_for(n, 256), 256, 0, st>>>(enc, z);
  DL4J_RET_LAST();
}

// ptrs: device array of narr float pointers (16-byte aligned arrays). z may be null.
DL4J_API int dl4j_multi_reduce(const float* const* ptrs, int narr, int64_t len, float* z, float scale, int propagate,
                               hipStream_t st) {
  multi_reduce_kernel<<<grid_for(len / 4 + 1, 256), 256, 0, st>>>(ptrs, narr, len, z, scale, propagate);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_pull_rows(const void* x, const int64_t* idx, void* z, int64_t rows, int64_t width, int64_t x_rows,
                            int elem_size, hipStream_t st) {
  int g = grid_for(rows, 1);
  if (elem_size == 4) pull_rows_kernel<float><<<g, 256, 0, st>>>((const float*)x, idx, (float*)z, rows, width, x_rows);
  else if (elem_size == 2) pull_rows_kernel<uint16_t><<<g, 256, 0, st>>>((const uint16_t*)x, idx, (uint16_t*)z, rows, width, x_rows);
  else if (elem_size == 8) pull_rows_kernel<double><<<g, 256, 0, st>>>((const double*)x, idx, (double*)z, rows, width, x_rows);
  else return 1;
  DL4J_RET_LAST();
}
