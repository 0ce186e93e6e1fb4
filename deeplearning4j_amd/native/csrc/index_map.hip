// One data-movement kernel for the whole "index remapping" family: pad (constant / reflect / symmetric), reverse, tile,
// roll, strided slice, permute and concat are all out[c] = in[f(c)] where f acts per dimension.
//
// Reference: separate kernels per op in libnd4j/include/ops/declarable/helpers/cuda — pad.cu (padCuda, mirrorPadKernel),
// reverse.cu, transforms.cu (tile), roll.cu, concat.cu, s_t_d.cu / d_t_s.cu, each with coordinate <-> offset conversion through
// shapeInfo in shared memory. Here the per-dimension map is a tiny by-value descriptor (rank <= 8), the index math is
// 32-bit when the region fits, and consecutive lanes walk the innermost output dimension so stores are coalesced.
#include "common.h"

namespace {

enum DimMode { M_SHIFT = 0, M_REVERSE = 1, M_WRAP = 2, M_REFLECT = 3, M_SYMMETRIC = 4, M_STRIDE = 5 };

struct DimMap {
  int64_t out_stride;   // elements
  int64_t in_stride;
  int32_t region;       // extent of this dim in the iterated region
  int32_t in_dim;       // input extent
  int32_t a;            // shift: in = c + a (a may be negative: padding); wrap: in = (c + a) mod in_dim; stride: in = a + c * b
  int32_t b;
  int32_t mode;
  int32_t pad_;
};

struct MapP {
  DimMap d[8];
  int rank;
  int64_t out_base, total;
};

template <typename T>
__global__ void __launch_bounds__(256) index_map_kernel(const T* __restrict__ in, T* __restrict__ out, const MapP p, T pad_value) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (int64_t)gridDim.x * 256) {
    int64_t rem = i, io = 0, oo = p.out_base;
    bool valid = true;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
      if (k >= p.rank) continue;
      const DimMap& m = p.d[k];
      const int c = (int)(rem % m.region);
      rem /= m.region;
      int s;
      switch (m.mode) {
        case M_REVERSE: s = m.in_dim - 1 - c; break;
        case M_WRAP: s = (c + m.a) % m.in_dim; if (s < 0) s += m.in_dim; break;
        case M_REFLECT: {          // edge NOT repeated: -1 -> 1, n -> n-2
          s = c + m.a;
          const int period = 2 * (m.in_dim - 1);
          if (period > 0) { s %= period; if (s < 0) s += period; if (s >= m.in_dim) s = period - s; } else s = 0;
          break;
        }
        case M_SYMMETRIC: {        // edge repeated: -1 -> 0, n -> n-1
          s = c + m.a;
          const int period = 2 * m.in_dim;
          s %= period; if (s < 0) s += period; if (s >= m.in_dim) s = period - 1 - s;
          break;
        }
        case M_STRIDE: s = m.a + c * m.b; break;
        default: s = c + m.a;
      }
      if ((unsigned)s >= (unsigned)m.in_dim) valid = false;
      io += (int64_t)s * m.in_stride;
      oo += (int64_t)c * m.out_stride;
    }
    out[oo] = valid ? in[io] : pad_value;
  }
}

}  // namespace

// dims: rank x {out_stride, in_stride, region, in_dim, a, b, mode} as int64; pad_bits: raw bits of the pad value (low esize bytes)
DL4J_API int dl4j_index_map(const void* in, void* out, int rank, const int64_t* dims, int64_t out_base, uint64_t pad_bits, int esize,
                            hipStream_t st) {
  if (rank < 1 || rank > 8) return -2;
  MapP p;
  p.rank = rank; p.out_base = out_base; p.total = 1;
  for (int k = 0; k < rank; ++k) {
    const int64_t* d = dims + 7 * k;
    if (d[2] <= 0) return 0;                                   // empty region
    if (d[2] > 2147483647LL || d[3] > 2147483647LL || d[3] <= 0) return -2;
    p.d[k].out_stride = d[0]; p.d[k].in_stride = d[1]; p.d[k].region = (int32_t)d[2]; p.d[k].in_dim = (int32_t)d[3];
    p.d[k].a = (int32_t)d[4]; p.d[k].b = (int32_t)d[5]; p.d[k].mode = (int32_t)d[6]; p.d[k].pad_ = 0;
    p.total *= d[2];
  }
  const int g = grid_for(p.total, 1024);
  switch (esize) {
    case 1: hipLaunchKernelGGL(index_map_kernel<uint8_t>, dim3(g), dim3(256), 0, st, (const uint8_t*)in, (uint8_t*)out, p, (uint8_t)pad_bits); break;
    case 2: hipLaunchKernelGGL(index_map_kernel<uint16_t>, dim3(g), dim3(256), 0, st, (const uint16_t*)in, (uint16_t*)out, p, (uint16_t)pad_bits); break;
    case 4: hipLaunchKernelGGL(index_map_kernel<uint32_t>, dim3(g), dim3(256), 0, st, (const uint32_t*)in, (uint32_t*)out, p, (uint32_t)pad_bits); break;
    case 8: hipLaunchKernelGGL(index_map_kernel<uint64_t>, dim3(g), dim3(256), 0, st, (const uint64_t*)in, (uint64_t*)out, p, (uint64_t)pad_bits); break;
    default: return -2;
  }
  DL4J_RET_LAST();
}
