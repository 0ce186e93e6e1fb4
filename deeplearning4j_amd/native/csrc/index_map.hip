// One data-movement kernel for the whole "index remapping" family: pad (constant / reflect / symmetric), reverse, tile,
// roll, strided slice, permute and concat are all out[c] = in[f(c)] where f acts per dimension.
//
// Reference: separate kernels per op in libnd4j/include/ops/declarable/helpers/cuda — pad.cu (padCuda, mirrorPadKernel),
// reverse.cu, transforms.cu (tile), roll.cu, concat.cu, s_t_d.cu / d_t_s.cu, each with coordinate <-> offset conversion through
// shapeInfo in shared memory. Here the per-dimension map is a tiny by-value descriptor (rank <= 8), the index math is
// 32-bit when the region fits, and every thread moves 16 bytes of the innermost output dimension (vector load / store when the run is
// contiguous and aligned), consecutive lanes taking consecutive 16-byte pieces.
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
  int32_t div_s;        // region as an invariant divisor: q = (umulhi(n, div_m) + n) >> div_s for n < 2^31
  uint32_t div_m;
  int32_t pad_;
};

// division by a launch-invariant 32-bit divisor as multiply-high + shift (exact for n < 2^31): the coordinate decomposition is a chain of
// these per 16 bytes moved, and a hardware-less integer division costs an order of magnitude more ALU work than the copy itself
static inline void fastdiv_init(uint32_t d, uint32_t* m, int32_t* s) {
  int sh = 0;
  while ((1ull << sh) < d) ++sh;
  *s = sh;
  *m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << sh) - d)) / d + 1);
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, uint32_t m, int s) { return (__umulhi(n, m) + n) >> s; }

struct MapP {
  DimMap d[8];
  int rank;
  int64_t out_base, total;
  uint32_t seg_m;       // fast divisor: segments per innermost row (long rows are cut into segments of seg_vecs pieces)
  int32_t seg_s;
  int32_t segs_per_row, seg_vecs;
  uint32_t slab_m;      // fast divisor: slabs of the second-innermost dimension (short rows: one visit covers `slab` of its coordinates)
  int32_t slab_s;
  int32_t slab, slabs;
};

__device__ __forceinline__ int map_dim(const DimMap& m, int c) {
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
  return s;
}

// One thread = V consecutive elements of the innermost output dimension (V * sizeof(T) = 16 bytes): the coordinate decomposition of the
// outer dimensions (a chain of integer divisions, the expensive part of an element-per-thread mapping) is done once per 16 bytes, in
// 32-bit arithmetic when the vector count fits. When the V source elements are one contiguous, dword-aligned run (shift / wrap / unit
// stride / reverse that does not cross an edge) they are fetched with one 16-byte load, and stored with one 16-byte store when the
// destination run is 16-byte aligned; everything else falls back to per-element accesses inside the same thread.
template <typename T>
struct alignas(16) IVec { T v[16 / (int)sizeof(T)]; };

// Work decomposition: a group of L lanes (L = power of two <= 64, about the number of 16-byte pieces of an innermost row) owns one row of
// the region at a time. The coordinate decomposition of the outer dimensions (a chain of divisions, done as multiply-high with
// launch-invariant magic numbers) happens once per row and lane; the lanes then walk the row in 16-byte pieces (V elements), where only the
// innermost dimension's map is evaluated. When the V source elements of a piece are one contiguous, dword-aligned run (shift / wrap / unit
// stride / reverse that does not cross an edge) they are fetched with one 16-byte load, and stored with one 16-byte store when the
// destination is 16-byte aligned; everything else falls back to per-element accesses inside the same thread.
template <typename T>
__global__ void __launch_bounds__(256) index_map_kernel(const T* __restrict__ in, T* __restrict__ out, const MapP p, T pad_value,
                                                        int64_t rows, int vec_per_row, int lg2_L) {
  constexpr int V = 16 / (int)sizeof(T);
  const int L = 1 << lg2_L;
  const int lane = threadIdx.x & (L - 1);
  const int rows_per_block = 256 >> lg2_L;
  const DimMap m0 = p.d[p.rank - 1];
  const bool unit = m0.mode == M_SHIFT || m0.mode == M_WRAP || (m0.mode == M_STRIDE && m0.b == 1);
  const bool in_unit = m0.in_stride == 1, out_unit = m0.out_stride == 1;
  // visits: (outer coordinates) x (slabs of the second-innermost dimension) x (segments of the innermost row)
  const DimMap m1 = p.d[p.rank >= 2 ? p.rank - 2 : 0];
  const int region1 = p.rank >= 2 ? m1.region : 1;
  const int64_t vrows = (rows / region1) * p.slabs * p.segs_per_row;
  const bool small = vrows < 2147483647LL;
  for (int64_t vrow = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x >> lg2_L); vrow < vrows; vrow += (int64_t)gridDim.x * rows_per_block) {
    int64_t io0 = 0, oo0 = p.out_base;
    bool outer_valid = true;
    int seg = 0, slab = 0;
    if (small) {
      uint32_t rem = (uint32_t)vrow;
      if (p.segs_per_row > 1) {
        const uint32_t q = fastdiv(rem, p.seg_m, p.seg_s);
        seg = (int)(rem - q * (uint32_t)p.segs_per_row);
        rem = q;
      }
      if (p.slabs > 1) {
        const uint32_t q = fastdiv(rem, p.slab_m, p.slab_s);
        slab = (int)(rem - q * (uint32_t)p.slabs);
        rem = q;
      }
#pragma unroll 1
      for (int k = p.rank - 3; k >= 0; --k) {
        const DimMap& m = p.d[k];
        const uint32_t q = fastdiv(rem, m.div_m, m.div_s);
        const int c = (int)(rem - q * (uint32_t)m.region);
        rem = q;
        const int s = m.mode == M_SHIFT ? c + m.a : map_dim(m, c);
        if ((unsigned)s >= (unsigned)m.in_dim) outer_valid = false;
        io0 += (int64_t)s * m.in_stride;
        oo0 += (int64_t)c * m.out_stride;
      }
    } else {
      int64_t rem = vrow;
      seg = (int)(rem % p.segs_per_row);
      rem /= p.segs_per_row;
      slab = (int)(rem % p.slabs);
      rem /= p.slabs;
#pragma unroll 1
      for (int k = p.rank - 3; k >= 0; --k) {
        const DimMap& m = p.d[k];
        const int c = (int)(rem % m.region);
        rem /= m.region;
        const int s = map_dim(m, c);
        if ((unsigned)s >= (unsigned)m.in_dim) outer_valid = false;
        io0 += (int64_t)s * m.in_stride;
        oo0 += (int64_t)c * m.out_stride;
      }
    }
    const int v_begin = seg * p.seg_vecs, v_end = min(vec_per_row, v_begin + p.seg_vecs);
    const int c1_begin = slab * p.slab, c1_end = min(region1, c1_begin + p.slab);
    for (int c1 = c1_begin; c1 < c1_end; ++c1) {
    int64_t io = io0, oo = oo0;
    bool row_valid = outer_valid;
    if (p.rank >= 2) {
      const int s1 = m1.mode == M_SHIFT ? c1 + m1.a : map_dim(m1, c1);
      if ((unsigned)s1 >= (unsigned)m1.in_dim) row_valid = false;
      io += (int64_t)s1 * m1.in_stride;
      oo += (int64_t)c1 * m1.out_stride;
    }
    const T* irow = in + io;
    T* orow = out + oo;
#pragma unroll 4
    for (int v = v_begin + lane; v < v_end; v += L) {
      const int c0 = v * V;
      const int nv = min(V, m0.region - c0);
      IVec<T> val;
      bool loaded = false;
      if (row_valid && nv == V && in_unit) {
        const int sf = map_dim(m0, c0);
        const int sl = unit ? sf + V - 1 : map_dim(m0, c0 + V - 1);
        if (sl - sf == V - 1 && sf >= 0 && sl < m0.in_dim) {                      // ascending contiguous run
          const T* src = irow + sf;
          if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {
            __builtin_memcpy(&val, __builtin_assume_aligned(src, 4), 16);
            loaded = true;
          }
        } else if (m0.mode == M_REVERSE && sf - sl == V - 1 && sl >= 0 && sf < m0.in_dim) {   // descending run: load and flip
          const T* src = irow + sl;
          if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {
            IVec<T> t;
            __builtin_memcpy(&t, __builtin_assume_aligned(src, 4), 16);
#pragma unroll
            for (int j = 0; j < V; ++j) val.v[j] = t.v[V - 1 - j];
            loaded = true;
          }
        }
      }
      if (!loaded) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          T x = pad_value;
          if (j < nv && row_valid) {
            const int s = map_dim(m0, c0 + j);
            if ((unsigned)s < (unsigned)m0.in_dim) x = irow[(int64_t)s * m0.in_stride];
          }
          val.v[j] = x;
        }
      }
      T* dst = orow + (int64_t)c0 * m0.out_stride;
      if (nv == V && out_unit && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        *reinterpret_cast<IVec<T>*>(dst) = val;
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j)
          if (j < nv) dst[(int64_t)j * m0.out_stride] = val.v[j];
      }
    }
    }   // c1
  }
}

template <typename T>
static void launch_index_map(const void* in, void* out, MapP& p, uint64_t pad_bits, hipStream_t st) {
  constexpr int V = 16 / (int)sizeof(T);
  const int region0 = p.d[p.rank - 1].region;
  const int vec_per_row = (region0 + V - 1) / V;
  const int64_t rows = p.total / region0;
  for (int k = 0; k < p.rank; ++k) fastdiv_init((uint32_t)p.d[k].region, &p.d[k].div_m, &p.d[k].div_s);
  p.seg_vecs = vec_per_row > 512 ? 512 : vec_per_row;          // long rows: segments of 512 pieces (8 per lane)
  p.segs_per_row = (vec_per_row + p.seg_vecs - 1) / p.seg_vecs;
  fastdiv_init((uint32_t)p.segs_per_row, &p.seg_m, &p.seg_s);
  int lg = 0;
  while ((1 << lg) < p.seg_vecs && lg < 6) ++lg;              // lanes per row segment: 1 .. 64
  const int rows_per_block = 256 >> lg;
  // short rows: one visit walks up to 8 coordinates of the second-innermost dimension, so the outer decomposition is paid once per slab
  const int region1 = p.rank >= 2 ? p.d[p.rank - 2].region : 1;
  p.slab = (p.rank >= 2 && p.segs_per_row == 1 && vec_per_row <= 64) ? (region1 < 8 ? region1 : 8) : 1;
  p.slabs = (region1 + p.slab - 1) / p.slab;
  fastdiv_init((uint32_t)p.slabs, &p.slab_m, &p.slab_s);
  const int64_t vrows = (rows / region1) * p.slabs * p.segs_per_row;
  // about 8 pieces per thread before the grid-stride loop repeats
  const int64_t pieces_per_block = (int64_t)rows_per_block * p.seg_vecs * p.slab;
  int64_t blocks = (vrows + rows_per_block - 1) / rows_per_block;
  if (pieces_per_block < 2048) {
    const int64_t fold = 2048 / (pieces_per_block > 0 ? pieces_per_block : 1);
    blocks = (blocks + fold - 1) / fold;
  }
  if (blocks < 1) blocks = 1;
  if (blocks > DL4J_MAX_GRID) blocks = DL4J_MAX_GRID;
  hipLaunchKernelGGL((index_map_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)in, (T*)out, p, (T)pad_bits, rows, vec_per_row, lg);
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
  switch (esize) {
    case 1: launch_index_map<uint8_t>(in, out, p, pad_bits, st); break;
    case 2: launch_index_map<uint16_t>(in, out, p, pad_bits, st); break;
    case 4: launch_index_map<uint32_t>(in, out, p, pad_bits, st); break;
    case 8: launch_index_map<uint64_t>(in, out, p, pad_bits, st); break;
    default: return -2;
  }
  DL4J_RET_LAST();
}
