// Aggregate ops: MANY small independent BLAS-1 / GEMM problems in ONE launch.
//
// Reference: libnd4j/include/loops/aggregates.h + ops/aggregate_ops.h (opNums legacy_ops.h:26-28: 0 HierarchicSoftmax, 1 Dot, 2 Axpy,
// 3 SkipGram, 4 CBOW, 5 GEMM), entered through NativeOps::execAggregate / execAggregateBatch (NativeOps.h:931-942): the Java side packs
// up to 512 aggregates into one parameter surface and the CUDA side runs one block per aggregate. Here the batch is described by
// pointer / parameter tables in device memory and one workgroup (or a few, for GEMM tiles) serves each aggregate; SkipGram / CBOW /
// HierarchicSoftmax batches go to the batched word2vec kernel of helpers_misc.hip (dl4j_sg_cb_batch).
#include "common.h"

namespace {

struct AxpyItem { const float* x; float* y; float alpha; int n; };
struct DotItem { const float* x; const float* y; int n; int pad; };
struct GemmItem { const float* a; const float* b; float* c; int m, n, k, lda, ldb, ldc, ta, tb; float alpha, beta; };

__global__ __launch_bounds__(256) void aggregate_axpy_kernel(const AxpyItem* __restrict__ items, int count) {
  for (int it = blockIdx.x; it < count; it += gridDim.x) {
    const AxpyItem p = items[it];
    for (int i = threadIdx.x; i < p.n; i += blockDim.x) p.y[i] = fmaf(p.alpha, p.x[i], p.y[i]);
  }
}

__global__ __launch_bounds__(256) void aggregate_dot_kernel(const DotItem* __restrict__ items, float* __restrict__ out, int count) {
  __shared__ float part[4];
  for (int it = blockIdx.x; it < count; it += gridDim.x) {
    const DotItem p = items[it];
    float acc = 0.f;
    for (int i = threadIdx.x; i < p.n; i += blockDim.x) acc = fmaf(p.x[i], p.y[i], acc);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[it] = part[0] + part[1] + part[2] + part[3];
  }
}

// C = alpha * op(A) . op(B) + beta * C for small column-major problems (the reference's aggregate GEMM is cblas-style, Order = 'f'):
// blockIdx.y = aggregate, blockIdx.x walks 16 x 16 output tiles, operands staged through LDS in 16-deep K slices.
__global__ __launch_bounds__(256) void aggregate_gemm_kernel(const GemmItem* __restrict__ items, int count) {
  __shared__ float sa[16][17], sb[16][17];
  const GemmItem p = items[blockIdx.y];
  const int tiles_m = (p.m + 15) / 16, tiles_n = (p.n + 15) / 16;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  for (int tile = blockIdx.x; tile < tiles_m * tiles_n; tile += gridDim.x) {
    const int row0 = (tile % tiles_m) * 16, col0 = (tile / tiles_m) * 16;
    float acc = 0.f;
    for (int k0 = 0; k0 < p.k; k0 += 16) {
      const int ar = row0 + tx, ak = k0 + ty;           // sa[kk][r] = op(A)[row0 + r, k0 + kk]
      sa[ty][tx] = (ar < p.m && ak < p.k) ? (p.ta ? p.a[(size_t)ar * p.lda + ak] : p.a[(size_t)ak * p.lda + ar]) : 0.f;
      const int bk = k0 + tx, bc = col0 + ty;           // sb[c][kk] = op(B)[k0 + kk, col0 + c]
      sb[ty][tx] = (bk < p.k && bc < p.n) ? (p.tb ? p.b[(size_t)bk * p.ldb + bc] : p.b[(size_t)bc * p.ldb + bk]) : 0.f;
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc = fmaf(sa[kk][tx], sb[ty][kk], acc);
      __syncthreads();
    }
    const int r = row0 + tx, c = col0 + ty;
    if (r < p.m && c < p.n) {
      float* dst = p.c + (size_t)c * p.ldc + r;
      *dst = p.beta == 0.f ? p.alpha * acc : fmaf(p.alpha, acc, p.beta * *dst);
    }
  }
}

}  // namespace

DL4J_API int dl4j_aggregate_item_bytes(int op) { return op == 2 ? (int)sizeof(AxpyItem) : op == 1 ? (int)sizeof(DotItem) : op == 5 ? (int)sizeof(GemmItem) : 0; }

// items: device table of `count` packed descriptors (layout = the structs above, dl4j_aggregate_item_bytes each)
DL4J_API int dl4j_aggregate_batch(int op, const void* items, void* out, int count, int max_tiles, hipStream_t st) {
  if (count <= 0) return 0;
  if (!items) return -3;
  switch (op) {
    case 2: aggregate_axpy_kernel<<<min(count, 4096), 256, 0, st>>>((const AxpyItem*)items, count); break;
    case 1: if (!out) return -3; aggregate_dot_kernel<<<min(count, 4096), 256, 0, st>>>((const DotItem*)items, (float*)out, count); break;
    case 5: {
      if (count > 65535) return -2;
      aggregate_gemm_kernel<<<dim3(max(1, min(max_tiles, 1024)), count), 256, 0, st>>>((const GemmItem*)items, count);
    } break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
