// Memory / device / stream / event / launch-context / sort groups of the C ABI, natively on the HIP runtime and rocPRIM.
//
// Reference: libnd4j/include/legacy/NativeOps.h — memory :540-565 and :710-755, devices :597-699 and :767, P2P :856-867,
// LaunchContext accessors :1665-1672, sort family :1411-1441, ravel / unravel :1456-1464 (host forms live in rt_shape.cpp).
// The reference implements these in legacy/cuda/NativeOps.cu on cudaMalloc / cudaStream_t and a hand-written bitonic + odd-even
// sort (loops/cuda/specials/bitonicSortStep.cu, oesTad.cu); here memory and streams are plain HIP runtime objects that any foreign
// binding can own without torch in the process, and sorting is rocPRIM's radix sort (one pass per 8 key bits at memory bandwidth,
// segmented form for sorts along a dimension).
//
// Conventions kept from the reference ABI: `extraPointers[1]` is the stream (may be null = the null stream), a data pointer pair is
// (host, device) and the device one wins when it is set, shapeInfo = [rank, shape.., strides.., extras(dtype flags), ews, order].
// Every function records failures in this library's error slot (dl4j_rt_last_error) and returns a non-zero code where the
// reference signature has a return value.
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bfloat16.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#define ABI extern "C" __attribute__((visibility("default")))

typedef int64_t LongType;
typedef void* Pointer;

namespace {

thread_local int g_err_code = 0;
thread_local char g_err_msg[512] = {0};

int fail(int code, const char* what, hipError_t e = hipSuccess) {
  g_err_code = code;
  if (e != hipSuccess) snprintf(g_err_msg, sizeof(g_err_msg), "%s: %s", what, hipGetErrorString(e));
  else snprintf(g_err_msg, sizeof(g_err_msg), "%s", what);
  return code;
}
#define HIP_TRY(expr, what) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return fail(1, what, e__); } while (0)
#define HIP_TRY_P(expr, what) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { fail(1, what, e__); return nullptr; } } while (0)

// ---- shapeInfo helpers (array/ArrayOptions.h:36-134 data type flags)
constexpr LongType F_BOOL = 524288, F_HALF = 4096, F_BF16 = 2048, F_FLOAT = 8192, F_DOUBLE = 16384, F_INT8 = 32768, F_INT16 = 65536,
                   F_INT32 = 131072, F_INT64 = 262144, F_UNSIGNED = 8388608;
enum Ty { T_NONE = 0, T_F32, T_F64, T_F16, T_BF16, T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_I64, T_U64, T_BOOL };

int si_rank(const LongType* si) { return (int)si[0]; }
LongType si_extras(const LongType* si) { return si[1 + 2 * si[0]]; }
LongType si_length(const LongType* si) { LongType n = 1; for (int i = 0; i < si_rank(si); ++i) n *= si[1 + i]; return n; }
bool si_c_contiguous(const LongType* si) {
  LongType expect = 1;
  const int r = si_rank(si);
  for (int i = r - 1; i >= 0; --i) { if (si[1 + i] != 1 && si[1 + r + i] != expect) return false; expect *= si[1 + i]; }
  return true;
}
Ty type_of(const LongType* si) {
  const LongType e = si_extras(si);
  const bool u = (e & F_UNSIGNED) != 0;
  if (e & F_FLOAT) return T_F32;
  if (e & F_DOUBLE) return T_F64;
  if (e & F_HALF) return T_F16;
  if (e & F_BF16) return T_BF16;
  if (e & F_INT8) return u ? T_U8 : T_I8;
  if (e & F_INT16) return u ? T_U16 : T_I16;
  if (e & F_INT32) return u ? T_U32 : T_I32;
  if (e & F_INT64) return u ? T_U64 : T_I64;
  if (e & F_BOOL) return T_BOOL;
  return T_NONE;
}
int size_of(Ty t) {
  switch (t) { case T_F64: case T_I64: case T_U64: return 8; case T_F32: case T_I32: case T_U32: return 4;
               case T_F16: case T_BF16: case T_I16: case T_U16: return 2; case T_I8: case T_U8: case T_BOOL: return 1; default: return 0; }
}
hipStream_t stream_of(Pointer* extra) { return extra ? (hipStream_t)extra[1] : nullptr; }

// ---- per-thread launch context: streams and the small scratch buffers the legacy executors expect (LaunchContext.h)
struct LaunchCtx {
  int device = -1;
  hipStream_t exec = nullptr, copy = nullptr;
  void *scalar = nullptr, *reduction = nullptr, *allocation = nullptr;
};
constexpr size_t SCALAR_BYTES = 4096, REDUCTION_BYTES = 4u << 20, ALLOCATION_BYTES = 1u << 20;
thread_local std::vector<LaunchCtx*> t_ctx;

// ---- gather / strided copies used by the key-value and along-dimension sorts
template <typename V>
__global__ void gather_kernel(const V* __restrict__ in, const uint32_t* __restrict__ idx, V* __restrict__ out, LongType n) {
  for (LongType i = blockIdx.x * (LongType)blockDim.x + threadIdx.x; i < n; i += (LongType)gridDim.x * blockDim.x) out[i] = in[idx[i]];
}
__global__ void iota_kernel(uint32_t* idx, LongType n) {
  for (LongType i = blockIdx.x * (LongType)blockDim.x + threadIdx.x; i < n; i += (LongType)gridDim.x * blockDim.x) idx[i] = (uint32_t)i;
}
__global__ void iota_rows_kernel(uint32_t* idx, LongType rows, LongType cols) {
  const LongType n = rows * cols;
  for (LongType i = blockIdx.x * (LongType)blockDim.x + threadIdx.x; i < n; i += (LongType)gridDim.x * blockDim.x) idx[i] = (uint32_t)(i % cols);
}
// gather inside rows: out[r, j] = in[r, perm[r, j]]
template <typename V>
__global__ void gather_rows_kernel(const V* __restrict__ in, const uint32_t* __restrict__ perm, V* __restrict__ out, LongType rows, LongType cols) {
  const LongType n = rows * cols;
  for (LongType i = blockIdx.x * (LongType)blockDim.x + threadIdx.x; i < n; i += (LongType)gridDim.x * blockDim.x) {
    const LongType r = i / cols;
    out[i] = in[r * cols + perm[i]];
  }
}

struct TadGeom {            // sub-arrays along one dimension of a strided array
  int outer_rank;
  LongType outer_shape[8], outer_stride[8];
  LongType tad_len, tad_stride, num_tads;
};
template <typename V, bool TO_DENSE>
__global__ void tad_copy_kernel(V* __restrict__ arr, V* __restrict__ dense, TadGeom g) {
  const LongType n = g.num_tads * g.tad_len;
  for (LongType i = blockIdx.x * (LongType)blockDim.x + threadIdx.x; i < n; i += (LongType)gridDim.x * blockDim.x) {
    LongType t = i / g.tad_len, j = i - t * g.tad_len, off = j * g.tad_stride;
    for (int d = g.outer_rank - 1; d >= 0; --d) { const LongType c = t % g.outer_shape[d]; t /= g.outer_shape[d]; off += c * g.outer_stride[d]; }
    if (TO_DENSE) dense[i] = arr[off]; else arr[off] = dense[i];
  }
}
int blocks_for(LongType n) { return (int)std::min<LongType>(4096, std::max<LongType>(1, (n + 255) / 256)); }

struct SegBegin { LongType len; __host__ __device__ LongType operator()(LongType i) const { return i * len; } };

struct DevTmp {            // stream-ordered temporary
  void* p = nullptr;
  hipStream_t st;
  explicit DevTmp(hipStream_t s) : st(s) {}
  hipError_t get(size_t bytes) { return hipMallocAsync(&p, bytes ? bytes : 16, st); }
  ~DevTmp() { if (p) (void)hipFreeAsync(p, st); }
};

// Keys are sorted as order-preserving unsigned integers of their width (one radix-sort instantiation per width instead of one per
// data type and direction): floating types flip the sign bit of non-negative values and all bits of negative ones, signed integers
// flip the sign bit, a descending sort complements the code. Encoding happens in the sort's input iterator, decoding in the
// copy-back pass, so neither costs a pass over memory.
enum Codec { C_UNSIGNED = 0, C_SIGNED = 1, C_FLOAT = 2 };
template <typename U>
struct KeyCode {
  int codec;
  bool desc;
  __host__ __device__ U operator()(U v) const {
    constexpr U SIGN = (U)1 << (8 * sizeof(U) - 1);
    if (codec == C_SIGNED) v ^= SIGN;
    else if (codec == C_FLOAT) v = (v & SIGN) ? (U)~v : (U)(v | SIGN);
    return desc ? (U)~v : v;
  }
  __host__ __device__ U decode(U v) const {
    constexpr U SIGN = (U)1 << (8 * sizeof(U) - 1);
    if (desc) v = (U)~v;
    if (codec == C_SIGNED) v ^= SIGN;
    else if (codec == C_FLOAT) v = (v & SIGN) ? (U)(v & ~SIGN) : (U)~v;
    return v;
  }
};
template <typename U>
__global__ void decode_kernel(const U* __restrict__ in, U* __restrict__ out, LongType n, KeyCode<U> kc) {
  for (LongType i = blockIdx.x * (LongType)blockDim.x + threadIdx.x; i < n; i += (LongType)gridDim.x * blockDim.x) out[i] = kc.decode(in[i]);
}
int codec_of(Ty t) {
  switch (t) { case T_F32: case T_F64: case T_F16: case T_BF16: return C_FLOAT; case T_I8: case T_I16: case T_I32: case T_I64: return C_SIGNED; default: return C_UNSIGNED; }
}

// whole-array sort of `n` keys in place on the device; optional permutation output (uint32 positions of the sorted keys)
template <typename U>
int device_sort(U* keys, LongType n, int codec, bool desc, uint32_t* perm, hipStream_t st) {
  if (n <= 1) { if (perm && n == 1) HIP_TRY(hipMemsetAsync(perm, 0, 4, st), "sort"); return 0; }
  if (n >= (LongType)1 << 32) return fail(3, "sort: more than 2^32 - 1 elements");
  DevTmp kout(st), vin(st), tmp(st);
  HIP_TRY(kout.get(n * sizeof(U)), "sort: temporary keys");
  const KeyCode<U> kc{codec, desc};
  auto in = rocprim::make_transform_iterator(keys, kc);
  size_t bytes = 0;
  if (perm) {
    HIP_TRY(vin.get(n * 4), "sort: temporary positions");
    iota_kernel<<<blocks_for(n), 256, 0, st>>>((uint32_t*)vin.p, n);
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, in, (U*)kout.p, (uint32_t*)vin.p, perm, (size_t)n, 0, 8 * sizeof(U), st), "sort");
    HIP_TRY(tmp.get(bytes), "sort: temporary storage");
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, bytes, in, (U*)kout.p, (uint32_t*)vin.p, perm, (size_t)n, 0, 8 * sizeof(U), st), "sort");
  } else {
    HIP_TRY(rocprim::radix_sort_keys(nullptr, bytes, in, (U*)kout.p, (size_t)n, 0, 8 * sizeof(U), st), "sort");
    HIP_TRY(tmp.get(bytes), "sort: temporary storage");
    HIP_TRY(rocprim::radix_sort_keys(tmp.p, bytes, in, (U*)kout.p, (size_t)n, 0, 8 * sizeof(U), st), "sort");
  }
  decode_kernel<U><<<blocks_for(n), 256, 0, st>>>((const U*)kout.p, keys, n, kc);
  HIP_TRY(hipGetLastError(), "sort: launch");
  return 0;
}

// `rows` independent sorts of `cols` contiguous keys each, in place, with the per-row permutation (callers without values drop it)
template <typename U>
int device_sort_rows(U* keys, LongType rows, LongType cols, int codec, bool desc, uint32_t* perm, hipStream_t st) {
  const LongType n = rows * cols;
  if (n == 0 || cols <= 1) { if (perm && n) HIP_TRY(hipMemsetAsync(perm, 0, n * 4, st), "sortTad"); return 0; }
  if (n >= (LongType)1 << 32) return fail(3, "sortTad: more than 2^32 - 1 elements");
  DevTmp kout(st), vin(st), vout(st), tmp(st);
  HIP_TRY(kout.get(n * sizeof(U)), "sortTad: temporary keys");
  HIP_TRY(vin.get(n * 4), "sortTad: temporary positions");
  if (!perm) { HIP_TRY(vout.get(n * 4), "sortTad: temporary positions"); perm = (uint32_t*)vout.p; }
  const KeyCode<U> kc{codec, desc};
  auto in = rocprim::make_transform_iterator(keys, kc);
  auto begin = rocprim::make_transform_iterator(rocprim::make_counting_iterator<LongType>(0), SegBegin{cols});
  auto end = rocprim::make_transform_iterator(rocprim::make_counting_iterator<LongType>(1), SegBegin{cols});
  size_t bytes = 0;
  iota_rows_kernel<<<blocks_for(n), 256, 0, st>>>((uint32_t*)vin.p, rows, cols);
  HIP_TRY(rocprim::segmented_radix_sort_pairs(nullptr, bytes, in, (U*)kout.p, (uint32_t*)vin.p, perm, (size_t)n, (unsigned)rows, begin, end, 0, 8 * sizeof(U), st), "sortTad");
  HIP_TRY(tmp.get(bytes), "sortTad: temporary storage");
  HIP_TRY(rocprim::segmented_radix_sort_pairs(tmp.p, bytes, in, (U*)kout.p, (uint32_t*)vin.p, perm, (size_t)n, (unsigned)rows, begin, end, 0, 8 * sizeof(U), st), "sortTad");
  decode_kernel<U><<<blocks_for(n), 256, 0, st>>>((const U*)kout.p, keys, n, kc);
  HIP_TRY(hipGetLastError(), "sortTad: launch");
  return 0;
}

int device_gather(void* data, int elem, const uint32_t* perm, LongType n, hipStream_t st) {
  DevTmp out(st);
  HIP_TRY(out.get(n * elem), "sort: temporary values");
  const int b = blocks_for(n);
  switch (elem) {
    case 1: gather_kernel<uint8_t><<<b, 256, 0, st>>>((const uint8_t*)data, perm, (uint8_t*)out.p, n); break;
    case 2: gather_kernel<uint16_t><<<b, 256, 0, st>>>((const uint16_t*)data, perm, (uint16_t*)out.p, n); break;
    case 4: gather_kernel<uint32_t><<<b, 256, 0, st>>>((const uint32_t*)data, perm, (uint32_t*)out.p, n); break;
    case 8: gather_kernel<uint64_t><<<b, 256, 0, st>>>((const uint64_t*)data, perm, (uint64_t*)out.p, n); break;
    default: return fail(2, "sort: unsupported value type");
  }
  HIP_TRY(hipMemcpyAsync(data, out.p, n * elem, hipMemcpyDeviceToDevice, st), "sort: copy back");
  return 0;
}

#define WIDTH_SWITCH(bytes, CALL)                                                                 \
  switch (bytes) {                                                                                \
    case 1: { typedef uint8_t U; CALL; } break;                                                   \
    case 2: { typedef uint16_t U; CALL; } break;                                                  \
    case 4: { typedef uint32_t U; CALL; } break;                                                  \
    case 8: { typedef uint64_t U; CALL; } break;                                                  \
    default: rc = fail(2, "sort: unsupported key type");                                          \
  }
#define KEY_SWITCH(ty, CALL)                                                                      \
  switch (ty) {                                                                                   \
    case T_F32: { typedef float K; CALL; } break;                                                 \
    case T_F64: { typedef double K; CALL; } break;                                                \
    case T_F16: { typedef __half K; CALL; } break;                                                \
    case T_BF16: { typedef hip_bfloat16 K; CALL; } break;                                         \
    case T_I8: { typedef int8_t K; CALL; } break;                                                 \
    case T_U8: case T_BOOL: { typedef uint8_t K; CALL; } break;                                   \
    case T_I16: { typedef int16_t K; CALL; } break;                                               \
    case T_U16: { typedef uint16_t K; CALL; } break;                                              \
    case T_I32: { typedef int32_t K; CALL; } break;                                               \
    case T_U32: { typedef uint32_t K; CALL; } break;                                              \
    case T_I64: { typedef int64_t K; CALL; } break;                                               \
    case T_U64: { typedef uint64_t K; CALL; } break;                                              \
    default: rc = fail(2, "sort: unsupported key type");                                          \
  }

// ---- host forms (arrays without a device buffer)
template <typename K> struct HostKey { static double v(K x) { return (double)x; } };
template <> struct HostKey<__half> { static double v(__half x) { return (double)__half2float(x); } };
template <> struct HostKey<hip_bfloat16> { static double v(hip_bfloat16 x) { return (double)(float)x; } };

template <typename K>
void host_perm(const K* keys, LongType n, bool desc, std::vector<uint32_t>& perm) {
  perm.resize(n);
  std::iota(perm.begin(), perm.end(), 0u);
  if (desc) std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return HostKey<K>::v(keys[a]) > HostKey<K>::v(keys[b]); });
  else std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return HostKey<K>::v(keys[a]) < HostKey<K>::v(keys[b]); });
}
void host_apply(void* data, int elem, const std::vector<uint32_t>& perm) {
  const size_t n = perm.size();
  std::vector<uint8_t> tmp(n * elem);
  const uint8_t* src = (const uint8_t*)data;
  for (size_t i = 0; i < n; ++i) memcpy(&tmp[i * elem], src + (size_t)perm[i] * elem, elem);
  memcpy(data, tmp.data(), n * elem);
}

bool tad_geom(const LongType* si, LongType dim, TadGeom& g) {
  const int r = si_rank(si);
  if (r < 1 || r > 9) return false;
  if (dim < 0) dim += r;
  if (dim < 0 || dim >= r) return false;
  g.outer_rank = 0;
  g.num_tads = 1;
  for (int i = 0; i < r; ++i) {
    if (i == dim) continue;
    g.outer_shape[g.outer_rank] = si[1 + i];
    g.outer_stride[g.outer_rank] = si[1 + r + i];
    g.num_tads *= si[1 + i];
    ++g.outer_rank;
  }
  g.tad_len = si[1 + dim];
  g.tad_stride = si[1 + r + dim];
  return true;
}
bool tads_are_rows(const LongType* si, const TadGeom& g) { return si_c_contiguous(si) && g.tad_stride == 1; }

// sort of the sub-arrays along one dimension; `other` (optional) is reordered with the keys (same shape / strides)
int sort_tads(void* hx, void* dx, const LongType* si, void* hy, void* dy, const LongType* ysi, LongType dim, bool desc, bool by_value,
              hipStream_t st) {
  // by_value: the roles swap - y holds the keys, x follows
  void *hk = by_value ? hy : hx, *dk = by_value ? dy : dx, *hv = by_value ? hx : hy, *dv = by_value ? dx : dy;
  const LongType* ksi = by_value ? ysi : si;
  const LongType* vsi = by_value ? si : ysi;
  const bool pairs = vsi != nullptr;
  TadGeom g, gv;
  if (!tad_geom(ksi, dim, g)) return fail(2, "sortTad: rank 1..9 and one valid dimension are required");
  if (pairs && (!tad_geom(vsi, dim, gv) || gv.num_tads != g.num_tads || gv.tad_len != g.tad_len))
    return fail(2, "sortTad: keys and values must have the same shape");
  const Ty kt = type_of(ksi);
  const int ke = size_of(kt), ve = pairs ? size_of(type_of(vsi)) : 0;
  if (!ke || (pairs && !ve)) return fail(2, "sortTad: unsupported data type");
  const LongType n = g.num_tads * g.tad_len;
  int rc = 0;
  if (n == 0) return 0;
  if (dk) {
    if (pairs && !dv) return fail(2, "sortTad: keys on the device but values on the host");
    DevTmp dense(st), vdense(st), perm(st);
    void* keys = dk;
    const bool rows = tads_are_rows(ksi, g);
    if (!rows) {
      HIP_TRY(dense.get(n * ke), "sortTad: dense keys");
      keys = dense.p;
      switch (ke) {
        case 1: tad_copy_kernel<uint8_t, true><<<blocks_for(n), 256, 0, st>>>((uint8_t*)dk, (uint8_t*)keys, g); break;
        case 2: tad_copy_kernel<uint16_t, true><<<blocks_for(n), 256, 0, st>>>((uint16_t*)dk, (uint16_t*)keys, g); break;
        case 4: tad_copy_kernel<uint32_t, true><<<blocks_for(n), 256, 0, st>>>((uint32_t*)dk, (uint32_t*)keys, g); break;
        default: tad_copy_kernel<uint64_t, true><<<blocks_for(n), 256, 0, st>>>((uint64_t*)dk, (uint64_t*)keys, g); break;
      }
    }
    if (pairs) HIP_TRY(perm.get(n * 4), "sortTad: positions");
    WIDTH_SWITCH(ke, rc = device_sort_rows<U>((U*)keys, g.num_tads, g.tad_len, codec_of(kt), desc, (uint32_t*)perm.p, st));
    if (rc) return rc;
    if (!rows) {
      switch (ke) {
        case 1: tad_copy_kernel<uint8_t, false><<<blocks_for(n), 256, 0, st>>>((uint8_t*)dk, (uint8_t*)keys, g); break;
        case 2: tad_copy_kernel<uint16_t, false><<<blocks_for(n), 256, 0, st>>>((uint16_t*)dk, (uint16_t*)keys, g); break;
        case 4: tad_copy_kernel<uint32_t, false><<<blocks_for(n), 256, 0, st>>>((uint32_t*)dk, (uint32_t*)keys, g); break;
        default: tad_copy_kernel<uint64_t, false><<<blocks_for(n), 256, 0, st>>>((uint64_t*)dk, (uint64_t*)keys, g); break;
      }
    }
    if (pairs) {
      // values: to dense rows, permute inside each row, back
      HIP_TRY(vdense.get(2 * n * ve), "sortTad: dense values");
      uint8_t* a = (uint8_t*)vdense.p;
      uint8_t* b = a + n * ve;
#define VAL_PASS(V)                                                                                                          \
  { tad_copy_kernel<V, true><<<blocks_for(n), 256, 0, st>>>((V*)dv, (V*)a, gv);                                              \
    hipLaunchKernelGGL((gather_rows_kernel<V>), dim3(blocks_for(n)), dim3(256), 0, st, (const V*)a, (const uint32_t*)perm.p, (V*)b, g.num_tads, g.tad_len); \
    tad_copy_kernel<V, false><<<blocks_for(n), 256, 0, st>>>((V*)dv, (V*)b, gv); }
      switch (ve) { case 1: VAL_PASS(uint8_t) break; case 2: VAL_PASS(uint16_t) break; case 4: VAL_PASS(uint32_t) break; default: VAL_PASS(uint64_t) break; }
#undef VAL_PASS
    }
    HIP_TRY(hipGetLastError(), "sortTad: launch");
    return 0;
  }
  if (!hk) return fail(2, "sortTad: no buffer");
  if (pairs && !hv) return fail(2, "sortTad: keys on the host but values on the device");
  std::vector<uint8_t> row(g.tad_len * ke), vrow(pairs ? g.tad_len * ve : 0);
  std::vector<uint32_t> perm;
  for (LongType t = 0; t < g.num_tads; ++t) {
    LongType rem = t, off = 0, voff = 0;
    for (int d = g.outer_rank - 1; d >= 0; --d) {
      const LongType c = rem % g.outer_shape[d];
      rem /= g.outer_shape[d];
      off += c * g.outer_stride[d];
      if (pairs) voff += c * gv.outer_stride[d];
    }
    for (LongType j = 0; j < g.tad_len; ++j) memcpy(&row[j * ke], (uint8_t*)hk + (off + j * g.tad_stride) * ke, ke);
    KEY_SWITCH(kt, host_perm<K>((const K*)row.data(), g.tad_len, desc, perm));
    if (rc) return rc;
    for (LongType j = 0; j < g.tad_len; ++j) memcpy((uint8_t*)hk + (off + j * g.tad_stride) * ke, &row[(size_t)perm[j] * ke], ke);
    if (pairs) {
      for (LongType j = 0; j < g.tad_len; ++j) memcpy(&vrow[j * ve], (uint8_t*)hv + (voff + j * gv.tad_stride) * ve, ve);
      for (LongType j = 0; j < g.tad_len; ++j) memcpy((uint8_t*)hv + (voff + j * gv.tad_stride) * ve, &vrow[(size_t)perm[j] * ve], ve);
    }
  }
  return 0;
}

int sort_flat(void* hx, void* dx, const LongType* si, void* hy, void* dy, const LongType* ysi, bool desc, bool by_value, hipStream_t st) {
  void *hk = by_value ? hy : hx, *dk = by_value ? dy : dx, *hv = by_value ? hx : hy, *dv = by_value ? dx : dy;
  const LongType* ksi = by_value ? ysi : si;
  const LongType* vsi = by_value ? si : ysi;
  const bool pairs = vsi != nullptr;
  if (!si_c_contiguous(ksi) || (pairs && !si_c_contiguous(vsi))) return fail(2, "sort: dense (element-wise stride 1) arrays are required");
  const LongType n = si_length(ksi);
  if (pairs && si_length(vsi) != n) return fail(2, "sort: keys and values must have the same length");
  const Ty kt = type_of(ksi);
  const int ke = size_of(kt), ve = pairs ? size_of(type_of(vsi)) : 0;
  if (!ke || (pairs && !ve)) return fail(2, "sort: unsupported data type");
  int rc = 0;
  if (n == 0) return 0;
  if (dk) {
    if (pairs && !dv) return fail(2, "sort: keys on the device but values on the host");
    DevTmp perm(st);
    if (pairs) HIP_TRY(perm.get(std::max<LongType>(n, 1) * 4), "sort: positions");
    WIDTH_SWITCH(ke, rc = device_sort<U>((U*)dk, n, codec_of(kt), desc, (uint32_t*)perm.p, st));
    if (rc) return rc;
    if (pairs && n > 1) rc = device_gather(dv, ve, (const uint32_t*)perm.p, n, st);
    return rc;
  }
  if (!hk) return fail(2, "sort: no buffer");
  if (pairs && !hv) return fail(2, "sort: keys on the host but values on the device");
  if (n >= (LongType)1 << 32) return fail(3, "sort: more than 2^32 - 1 elements");
  std::vector<uint32_t> perm;
  KEY_SWITCH(kt, host_perm<K>((const K*)hk, n, desc, perm));
  if (rc) return rc;
  host_apply(hk, ke, perm);
  if (pairs) host_apply(hv, ve, perm);
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------ error slot
ABI int dl4j_rt_last_error_code() { return g_err_code; }
ABI const char* dl4j_rt_last_error_message() { return g_err_msg; }
ABI void dl4j_rt_clear_error() { g_err_code = 0; g_err_msg[0] = 0; }

// ------------------------------------------------------------------------------------------------------------------ memory
ABI Pointer mallocHost(LongType bytes, int flags) {
  void* p = nullptr;
  HIP_TRY_P(hipHostMalloc(&p, bytes > 0 ? (size_t)bytes : 1, hipHostMallocDefault), "mallocHost");     // pinned: async copies really overlap
  return p;
}
ABI Pointer mallocDevice(LongType bytes, int deviceId, int flags) {
  int cur = 0;
  HIP_TRY_P(hipGetDevice(&cur), "mallocDevice");
  if (deviceId >= 0 && deviceId != cur) HIP_TRY_P(hipSetDevice(deviceId), "mallocDevice: setDevice");
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes > 0 ? (size_t)bytes : 1);
  if (deviceId >= 0 && deviceId != cur) (void)hipSetDevice(cur);
  if (e != hipSuccess) { fail(1, "mallocDevice", e); return nullptr; }
  return p;
}
ABI int freeHost(Pointer p) { if (p) HIP_TRY(hipHostFree(p), "freeHost"); return 1; }
ABI int freeDevice(Pointer p, int deviceId) { if (p) HIP_TRY(hipFree(p), "freeDevice"); return 1; }

namespace {
hipMemcpyKind kind_of(int flags) {
  switch (flags) { case 0: return hipMemcpyHostToHost; case 1: return hipMemcpyHostToDevice; case 2: return hipMemcpyDeviceToHost;
                   case 3: return hipMemcpyDeviceToDevice; default: return hipMemcpyDefault; }
}
}
// flags: 0 host->host, 1 host->device, 2 device->host, 3 device->device, anything else: decided from the pointers; `reserved` = stream
ABI int memcpySync(Pointer dst, Pointer src, LongType size, int flags, Pointer reserved) {
  if (size <= 0) return 1;
  HIP_TRY(hipMemcpy(dst, src, (size_t)size, kind_of(flags)), "memcpySync");
  return 1;
}
ABI int memcpyAsync(Pointer dst, Pointer src, LongType size, int flags, Pointer reserved) {
  if (size <= 0) return 1;
  HIP_TRY(hipMemcpyAsync(dst, src, (size_t)size, kind_of(flags), (hipStream_t)reserved), "memcpyAsync");
  return 1;
}
ABI int memsetSync(Pointer dst, int value, LongType size, int flags, Pointer reserved) {
  if (size <= 0) return 1;
  HIP_TRY(hipMemset(dst, value, (size_t)size), "memsetSync");
  return 1;
}
ABI int memsetAsync(Pointer dst, int value, LongType size, int flags, Pointer reserved) {
  if (size <= 0) return 1;
  HIP_TRY(hipMemsetAsync(dst, value, (size_t)size, (hipStream_t)reserved), "memsetAsync");
  return 1;
}

// ------------------------------------------------------------------------------------------------------------------ devices
ABI int getAvailableDevices() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
ABI int setDevice(int deviceId) { HIP_TRY(hipSetDevice(deviceId), "setDevice"); return 1; }
ABI int getDevice() { int d = 0; if (hipGetDevice(&d) != hipSuccess) return -1; return d; }
ABI LongType getDeviceFreeMemory(int deviceId) {
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return 0;
  if (deviceId >= 0 && deviceId != cur && hipSetDevice(deviceId) != hipSuccess) return 0;
  size_t f = 0, t = 0;
  hipError_t e = hipMemGetInfo(&f, &t);
  if (deviceId >= 0 && deviceId != cur) (void)hipSetDevice(cur);
  return e == hipSuccess ? (LongType)f : 0;
}
ABI LongType getDeviceFreeMemoryDefault() { return getDeviceFreeMemory(-1); }
ABI LongType getDeviceTotalMemory(int deviceId) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, deviceId < 0 ? getDevice() : deviceId) != hipSuccess) return 0;
  return (LongType)p.totalGlobalMem;
}
ABI int getDeviceMajor(int deviceId) { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, deviceId) != hipSuccess) return 0; return p.major; }
ABI int getDeviceMinor(int deviceId) { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, deviceId) != hipSuccess) return 0; return p.minor; }
ABI const char* getDeviceName(int deviceId) {
  static thread_local char name[256];
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, deviceId) != hipSuccess) return "";
  snprintf(name, sizeof(name), "%s (%s)", p.name, p.gcnArchName);
  return name;
}
// compute units, wavefront size, LDS per workgroup, L2 bytes, memory clock kHz, bus width: what launch heuristics need
ABI int dl4j_device_properties(int deviceId, LongType* out6) {
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, deviceId), "deviceProperties");
  out6[0] = p.multiProcessorCount; out6[1] = p.warpSize; out6[2] = (LongType)p.sharedMemPerBlock; out6[3] = p.l2CacheSize;
  out6[4] = p.memoryClockRate; out6[5] = p.memoryBusWidth;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------ P2P
namespace { bool g_p2p = false; std::mutex g_p2p_mu; }
ABI bool isP2PAvailable() {
  const int n = getAvailableDevices();
  if (n < 2) return false;
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      if (a == b) continue;
      int ok = 0;
      if (hipDeviceCanAccessPeer(&ok, a, b) != hipSuccess || !ok) return false;
    }
  return true;
}
ABI void enableP2P(bool enable) {
  std::lock_guard<std::mutex> lk(g_p2p_mu);
  const int n = getAvailableDevices();
  int cur = getDevice();
  for (int a = 0; a < n; ++a) {
    if (hipSetDevice(a) != hipSuccess) continue;
    for (int b = 0; b < n; ++b) {
      if (a == b) continue;
      int ok = 0;
      if (hipDeviceCanAccessPeer(&ok, a, b) != hipSuccess || !ok) continue;
      hipError_t e = enable ? hipDeviceEnablePeerAccess(b, 0) : hipDeviceDisablePeerAccess(b);
      if (e == hipErrorPeerAccessAlreadyEnabled || e == hipErrorPeerAccessNotEnabled) (void)hipGetLastError();
    }
  }
  if (cur >= 0) (void)hipSetDevice(cur);
  g_p2p = enable && n > 1;
}
ABI bool checkP2P() { return g_p2p; }

// ------------------------------------------------------------------------------------------------------------------ streams / events
ABI Pointer createContext() { return nullptr; }     // the reference returns 0 as well (NativeOps.cu): contexts are implicit
ABI Pointer createStream() {
  hipStream_t s = nullptr;
  HIP_TRY_P(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "createStream");
  return (Pointer)s;
}
ABI int dl4j_destroy_stream(Pointer stream) { if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream), "destroyStream"); return 1; }
ABI Pointer createEvent() {
  hipEvent_t e = nullptr;
  HIP_TRY_P(hipEventCreateWithFlags(&e, hipEventDisableTiming), "createEvent");
  return (Pointer)e;
}
ABI int registerEvent(Pointer event, Pointer stream) { HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream), "registerEvent"); return 1; }
ABI int destroyEvent(Pointer event) { if (event) HIP_TRY(hipEventDestroy((hipEvent_t)event), "destroyEvent"); return 1; }
ABI int streamSynchronize(Pointer stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream), "streamSynchronize"); return 1; }
ABI int eventSynchronize(Pointer event) { HIP_TRY(hipEventSynchronize((hipEvent_t)event), "eventSynchronize"); return 1; }
ABI int dl4j_stream_wait_event(Pointer stream, Pointer event) { HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0), "streamWaitEvent"); return 1; }
ABI int dl4j_event_query(Pointer event) { hipError_t e = hipEventQuery((hipEvent_t)event); if (e == hipSuccess) return 1; if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; } return -fail(1, "eventQuery", e); }

// ------------------------------------------------------------------------------------------------------------------ launch context
ABI Pointer defaultLaunchContext() {
  int dev = getDevice();
  if (dev < 0) { fail(1, "defaultLaunchContext: no device"); return nullptr; }
  if ((int)t_ctx.size() <= dev) t_ctx.resize(dev + 1, nullptr);
  LaunchCtx*& c = t_ctx[dev];
  if (!c) {
    LaunchCtx* n = new LaunchCtx();
    n->device = dev;
    if (hipStreamCreateWithFlags(&n->exec, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&n->copy, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&n->scalar, SCALAR_BYTES) != hipSuccess || hipMalloc(&n->reduction, REDUCTION_BYTES) != hipSuccess ||
        hipMalloc(&n->allocation, ALLOCATION_BYTES) != hipSuccess) {
      fail(1, "defaultLaunchContext: stream / scratch creation failed", hipGetLastError());
      delete n;
      return nullptr;
    }
    c = n;
  }
  return c;
}
ABI Pointer lcScalarPointer(Pointer lc) { return lc ? ((LaunchCtx*)lc)->scalar : nullptr; }
ABI Pointer lcReductionPointer(Pointer lc) { return lc ? ((LaunchCtx*)lc)->reduction : nullptr; }
ABI Pointer lcAllocationPointer(Pointer lc) { return lc ? ((LaunchCtx*)lc)->allocation : nullptr; }
ABI Pointer lcExecutionStream(Pointer lc) { return lc ? (Pointer)((LaunchCtx*)lc)->exec : nullptr; }
ABI Pointer lcCopyStream(Pointer lc) { return lc ? (Pointer)((LaunchCtx*)lc)->copy : nullptr; }
ABI Pointer lcBlasHandle(Pointer lc) { return nullptr; }      // GEMMs are in-tree kernels launched on lcExecutionStream: there is no vendor handle to hand out
ABI Pointer lcSolverHandle(Pointer lc) { return nullptr; }
ABI LongType dl4j_lc_scratch_bytes(int which) { return which == 0 ? SCALAR_BYTES : which == 1 ? REDUCTION_BYTES : ALLOCATION_BYTES; }

// ------------------------------------------------------------------------------------------------------------------ sort family
ABI void sort(Pointer* extra, void* x, const LongType* xShapeInfo, void* dx, const LongType* dxShapeInfo, bool descending) {
  sort_flat(x, dx, xShapeInfo, nullptr, nullptr, nullptr, descending, false, stream_of(extra));
}
ABI void sortByKey(Pointer* extra, void* x, const LongType* xShapeInfo, void* dx, const LongType* dxShapeInfo, void* y,
                   const LongType* yShapeInfo, void* dy, const LongType* dyShapeInfo, bool descending) {
  sort_flat(x, dx, xShapeInfo, y, dy, yShapeInfo, descending, false, stream_of(extra));
}
ABI void sortByValue(Pointer* extra, void* x, const LongType* xShapeInfo, void* dx, const LongType* dxShapeInfo, void* y,
                     const LongType* yShapeInfo, void* dy, const LongType* dyShapeInfo, bool descending) {
  sort_flat(x, dx, xShapeInfo, y, dy, yShapeInfo, descending, true, stream_of(extra));
}
ABI void sortTad(Pointer* extra, void* hX, const LongType* hXShapeInfo, void* dX, const LongType* dXShapeInfo, LongType* dimension,
                 LongType dimensionLength, const LongType* tadShapeInfo, const LongType* tadOffsets, bool descending) {
  if (dimensionLength != 1) { fail(2, "sortTad: exactly one dimension"); return; }
  sort_tads(hX, dX, hXShapeInfo, nullptr, nullptr, nullptr, dimension[0], descending, false, stream_of(extra));
}
ABI void sortTadByKey(Pointer* extra, void* x, const LongType* xShapeInfo, void* dX, const LongType* dXShapeInfo, void* y,
                      const LongType* yShapeInfo, void* dy, const LongType* dyShapeInfo, LongType* dimension, long long dimensionLength,
                      bool descending) {
  if (dimensionLength != 1) { fail(2, "sortTadByKey: exactly one dimension"); return; }
  sort_tads(x, dX, xShapeInfo, y, dy, yShapeInfo, dimension[0], descending, false, stream_of(extra));
}
ABI void sortTadByValue(Pointer* extra, void* x, const LongType* xShapeInfo, void* dx, const LongType* dxShapeInfo, void* y,
                        const LongType* yShapeInfo, void* dy, const LongType* dyShapeInfo, LongType* dimension, LongType dimensionLength,
                        bool descending) {
  if (dimensionLength != 1) { fail(2, "sortTadByValue: exactly one dimension"); return; }
  sort_tads(x, dx, xShapeInfo, y, dy, yShapeInfo, dimension[0], descending, true, stream_of(extra));
}

namespace {
__global__ void coo_flat_kernel(const LongType* __restrict__ indices, LongType* __restrict__ flat, LongType n, int rank, TadGeom shape) {
  for (LongType i = blockIdx.x * (LongType)blockDim.x + threadIdx.x; i < n; i += (LongType)gridDim.x * blockDim.x) {
    LongType f = 0;
    for (int d = 0; d < rank; ++d) f = f * shape.outer_shape[d] + indices[i * rank + d];
    flat[i] = f;
  }
}
__global__ void coo_rows_kernel(const LongType* __restrict__ in, const uint32_t* __restrict__ perm, LongType* __restrict__ out, LongType n, int rank) {
  for (LongType i = blockIdx.x * (LongType)blockDim.x + threadIdx.x; i < n * rank; i += (LongType)gridDim.x * blockDim.x) {
    const LongType r = i / rank, d = i - r * rank;
    out[i] = in[(LongType)perm[r] * rank + d];
  }
}
}
// COO entries (indices [length, rank] row-major, values x) into row-major order of the full array described by xShapeInfo.
// Device form when extraPointers[0] is non-null (= "buffers are device memory"), stream in extraPointers[1]; host form otherwise.
ABI void sortCooIndices(Pointer* extra, LongType* indices, void* x, LongType length, const LongType* xShapeInfo) {
  const int rank = si_rank(xShapeInfo);
  const int ve = size_of(type_of(xShapeInfo));
  if (rank < 1 || rank > 8 || !ve) { fail(2, "sortCooIndices: rank 1..8 and a known value type are required"); return; }
  if (length <= 1) return;
  if (length >= (LongType)1 << 32) { fail(3, "sortCooIndices: too many entries"); return; }
  const bool device = extra && extra[0];
  if (device) {
    hipStream_t st = stream_of(extra);
    TadGeom shape{};
    for (int d = 0; d < rank; ++d) shape.outer_shape[d] = xShapeInfo[1 + d];
    DevTmp flat(st), perm(st), rows(st);
    if (flat.get(length * 8) != hipSuccess || perm.get(length * 4) != hipSuccess || rows.get(length * rank * 8) != hipSuccess) { fail(1, "sortCooIndices: temporaries"); return; }
    coo_flat_kernel<<<blocks_for(length), 256, 0, st>>>(indices, (LongType*)flat.p, length, rank, shape);
    if (device_sort<uint64_t>((uint64_t*)flat.p, length, C_SIGNED, false, (uint32_t*)perm.p, st)) return;
    coo_rows_kernel<<<blocks_for(length * rank), 256, 0, st>>>(indices, (const uint32_t*)perm.p, (LongType*)rows.p, length, rank);
    if (hipMemcpyAsync(indices, rows.p, length * rank * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) { fail(1, "sortCooIndices: copy back"); return; }
    device_gather(x, ve, (const uint32_t*)perm.p, length, st);
    return;
  }
  std::vector<int64_t> flat(length);
  for (LongType i = 0; i < length; ++i) { LongType f = 0; for (int d = 0; d < rank; ++d) f = f * xShapeInfo[1 + d] + indices[i * rank + d]; flat[i] = f; }
  std::vector<uint32_t> perm;
  host_perm<int64_t>(flat.data(), length, false, perm);
  std::vector<LongType> rows((size_t)length * rank);
  for (LongType i = 0; i < length; ++i) memcpy(&rows[i * rank], indices + (LongType)perm[i] * rank, rank * 8);
  memcpy(indices, rows.data(), rows.size() * 8);
  host_apply(x, ve, perm);
}
