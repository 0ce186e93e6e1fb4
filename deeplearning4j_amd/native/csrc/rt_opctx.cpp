// Native op-execution ABI: OpaqueContext + execCustomOp2 + calculateOutputShapes2, the entry points a foreign binding (the
// reference's JavaCPP layer, NativeOps.h:1488-1508 execCustomOp2 / calculateOutputShapes2 and :1610-1646 the OpContext setters)
// uses to run a declarable op without any interpreter in between.
//
// Layout of this implementation
//   * OpaqueContext carries (buffer, shapeInfo, specialBuffer, specialShapeInfo) per input / output array plus t/i/b/d arguments,
//     exactly the fields graph::Context exposes through the C ABI.
//   * A native executor table (op hash -> function) covers the core declarable ops in C++: host arrays are computed here with
//     rt::parallel_for loops; arrays that carry a device ("special") buffer are dispatched to the hand-written gfx950 kernels of
//     libdl4j_hip.so, resolved once with dlopen/dlsym (the runtime library itself stays free of a link-time HIP dependency, as
//     libnd4j's cpu and cuda backends are separate builds of one ABI).
//   * Ops without a native executor go to an optional fallback callback registered by the embedding runtime
//     (registerCustomOpFallback); without one execCustomOp2 reports "not implemented natively" through the error channel.
//   * Shape functions are native for every op in the table (calculateOutputShapes2 -> OpaqueShapeList).
#include "rt_common.h"
#include "rt_exceptions.h"

#include <algorithm>
#include <cmath>
#include <dlfcn.h>
#include <map>
#include <mutex>

RT_API LongType ops_hash(const char* name);
RT_API int nd4j_shape_buffer(int rank, const LongType* shape, char order, LongType extras, LongType* out);

namespace {

enum DType { DT_UNKNOWN = 0, DT_BF16, DT_HALF, DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64, DT_BOOL };
// ArrayOptions data type flags (array/ArrayOptions.h:36-134)
constexpr LongType F_BF16 = 2048, F_HALF = 4096, F_FLOAT = 8192, F_DOUBLE = 16384, F_INT = 131072, F_LONG = 262144, F_BOOL = 524288;

DType dtype_of(const LongType* si) {
  const LongType e = rt::si_extras(si);
  if (e & F_FLOAT) return DT_FLOAT;
  if (e & F_DOUBLE) return DT_DOUBLE;
  if (e & F_BF16) return DT_BF16;
  if (e & F_HALF) return DT_HALF;
  if (e & F_INT) return DT_INT32;
  if (e & F_LONG) return DT_INT64;
  if (e & F_BOOL) return DT_BOOL;
  return DT_UNKNOWN;
}
LongType flag_of(DType d) {
  switch (d) { case DT_FLOAT: return F_FLOAT; case DT_DOUBLE: return F_DOUBLE; case DT_BF16: return F_BF16; case DT_HALF: return F_HALF;
               case DT_INT32: return F_INT; case DT_INT64: return F_LONG; case DT_BOOL: return F_BOOL; default: return 0; }
}

struct Arr {
  void* host = nullptr;
  const LongType* si = nullptr;
  void* dev = nullptr;
  const LongType* dsi = nullptr;
  std::vector<LongType> own_si;      // copy, so the context owns its descriptors
  int rank() const { return rt::si_rank(si); }
  LongType dim(int i) const { return rt::si_shape(si)[i]; }
  LongType stride(int i) const { return rt::si_strides(si)[i]; }
  LongType length() const { return rt::si_length(si); }
  bool c_contiguous() const {
    LongType expect = 1;
    for (int i = rank() - 1; i >= 0; --i) { if (dim(i) != 1 && stride(i) != expect) return false; expect *= dim(i); }
    return true;
  }
};

}  // namespace

struct OpaqueContext {
  int node_id = 0;
  std::vector<Arr> in, out;
  std::vector<double> t;
  std::vector<LongType> i;
  std::vector<uint8_t> b;
  std::vector<int> d;
  void* stream = nullptr;
  bool allow_helpers = true, inplace = false;
  int exec_mode = 0;
};

struct OpaqueShapeList {
  std::vector<std::vector<LongType>> shapes;
};

namespace {

// ---------------------------------------------------------------------------------------------- device kernels (dlopen)
struct Strides3 { int rank; int contiguous; int64_t shape[8]; int64_t sx[8], sy[8], sz[8]; };
typedef int (*pairwise_fn)(int, const void*, const void*, void*, int64_t, const Strides3*, const double*, int, double, int, int, void*);
typedef int (*transform_fn)(int, const void*, void*, int64_t, const Strides3*, const double*, int, int, void*);
typedef int (*softmax_fn)(const void*, void*, int64_t, int, int, int, void*);
typedef int (*gemm_fn)(const void*, const void*, void*, const float*, const void*, float, float*, const float*, int, const int64_t*, void*);
typedef int (*biasadd_fn)(const void*, const float*, void*, int64_t, int, int64_t, int, void*);

struct Hip {
  void* handle = nullptr;
  pairwise_fn pairwise = nullptr;
  transform_fn transform = nullptr;
  softmax_fn softmax = nullptr;
  gemm_fn gemm = nullptr;
  biasadd_fn biasadd = nullptr;
  bool tried = false;
};
Hip g_hip;
std::mutex g_hip_mu;

Hip* hip() {
  std::lock_guard<std::mutex> lk(g_hip_mu);
  if (!g_hip.tried) {
    g_hip.tried = true;
    Dl_info info;
    std::string dir = ".";
    if (dladdr((void*)&ops_hash, &info) && info.dli_fname) {
      dir = info.dli_fname;
      const size_t p = dir.find_last_of('/');
      dir = p == std::string::npos ? "." : dir.substr(0, p);
    }
    g_hip.handle = dlopen((dir + "/libdl4j_hip.so").c_str(), RTLD_NOW | RTLD_LOCAL);
    if (g_hip.handle) {
      g_hip.pairwise = (pairwise_fn)dlsym(g_hip.handle, "dl4j_legacy_pairwise");
      g_hip.transform = (transform_fn)dlsym(g_hip.handle, "dl4j_legacy_transform");
      g_hip.softmax = (softmax_fn)dlsym(g_hip.handle, "dl4j_softmax_rows");
      g_hip.gemm = (gemm_fn)dlsym(g_hip.handle, "dl4j_gemm_nt_ex");
      g_hip.biasadd = (biasadd_fn)dlsym(g_hip.handle, "dl4j_bias_add");
    }
  }
  return g_hip.handle ? &g_hip : nullptr;
}

// dtype codes of the HIP entry points: generic kernels (legacy) 0 f32 1 bf16 2 f64 3 i32 4 i64; NN kernels 0 f32 1 bf16 2 f16
int legacy_code(DType d) { return d == DT_FLOAT ? 0 : d == DT_BF16 ? 1 : d == DT_DOUBLE ? 2 : d == DT_INT32 ? 3 : d == DT_INT64 ? 4 : -1; }
int nn_code(DType d) { return d == DT_FLOAT ? 0 : d == DT_BF16 ? 1 : d == DT_HALF ? 2 : -1; }

// functor ids of legacy_ops.hip (same tables as nd4j/ops/legacy.py)
enum { B_ADD = 0, B_SUB = 1, B_MUL = 2, B_DIV = 3, B_RSUB = 4, B_RDIV = 5, B_POW = 7, B_MAX = 8, B_MIN = 9, B_FMOD = 16, B_FLOORDIV = 19,
       B_FLOORMOD = 20, B_SQSUB = 21, B_ELU = 40, B_LRELU = 43, B_RELU = 46, B_RELU6 = 47, B_EQ = 53, B_GT = 54, B_LT = 55, B_GE = 57, B_NE = 59,
       B_LE = 64 };
enum { U_CUBE = 6, U_SQUARE = 9, U_IDENTITY = 11, U_EXP = 46, U_SIGMOID = 49, U_SOFTPLUS = 51, U_TANH = 52, U_HARDTANH = 56, U_SOFTSIGN = 57,
       U_HARDSIGMOID = 58, U_RATIONALTANH = 59, U_RECTIFIEDTANH = 60, U_SELU = 64 };

// ---------------------------------------------------------------------------------------------- helpers
bool all_device(const OpaqueContext& c) {
  for (auto& a : c.in) if (!a.dev) return false;
  for (auto& a : c.out) if (!a.dev) return false;
  return !c.in.empty();
}
bool all_host(const OpaqueContext& c) {
  for (auto& a : c.in) if (!a.host) return false;
  for (auto& a : c.out) if (!a.host) return false;
  return true;
}

template <typename T> inline T* hp(const Arr& a) { return reinterpret_cast<T*>(a.host); }

int fail(int code, const std::string& m) { rt::set_error(code, m); return code; }

// broadcast index of a (row-major, c-contiguous) operand of shape `s` inside the output shape `o`
struct Bcast {
  int rank;
  LongType oshape[rt::MAX_RANK], astride[rt::MAX_RANK], bstride[rt::MAX_RANK];
};
bool make_bcast(const Arr& a, const Arr& b, const Arr& o, Bcast& bc) {
  bc.rank = o.rank();
  if (bc.rank > rt::MAX_RANK || a.rank() > bc.rank || b.rank() > bc.rank || bc.rank < 0) return false;
  LongType sa = 1, sb = 1;
  for (int k = bc.rank - 1, ka = a.rank() - 1, kb = b.rank() - 1; k >= 0; --k, --ka, --kb) {
    const LongType od = o.dim(k), ad = ka >= 0 ? a.dim(ka) : 1, bd = kb >= 0 ? b.dim(kb) : 1;
    if ((ad != od && ad != 1) || (bd != od && bd != 1)) return false;
    bc.oshape[k] = od;
    bc.astride[k] = ad == 1 ? 0 : sa;
    bc.bstride[k] = bd == 1 ? 0 : sb;
    sa *= ad; sb *= bd;
  }
  return true;
}

template <typename T, typename F>
void host_binary(const Arr& a, const Arr& b, const Arr& o, const Bcast& bc, F f) {
  const LongType n = o.length();
  const bool same = a.length() == n && b.length() == n;
  const T* pa = hp<T>(a); const T* pb = hp<T>(b); T* po = hp<T>(o);
  rt::parallel_for(0, n, 1 << 14, [&](LongType lo, LongType hi) {
    if (same) { for (LongType i = lo; i < hi; ++i) po[i] = f(pa[i], pb[i]); return; }
    for (LongType i = lo; i < hi; ++i) {
      LongType r = i, ia = 0, ib = 0;
      for (int k = bc.rank - 1; k >= 0; --k) { const LongType c = r % bc.oshape[k]; r /= bc.oshape[k]; ia += c * bc.astride[k]; ib += c * bc.bstride[k]; }
      po[i] = f(pa[ia], pb[ib]);
    }
  });
}

template <typename T, typename F>
void host_unary(const Arr& a, const Arr& o, F f) {
  const T* pa = hp<T>(a); T* po = hp<T>(o);
  rt::parallel_for(0, o.length(), 1 << 14, [&](LongType lo, LongType hi) { for (LongType i = lo; i < hi; ++i) po[i] = f(pa[i]); });
}

bool contiguous_all(const OpaqueContext& c) {
  for (auto& a : c.in) if (!a.c_contiguous()) return false;
  for (auto& a : c.out) if (!a.c_contiguous()) return false;
  return true;
}

void fill_strides3(const Arr& x, const Arr* y, const Arr& z, Strides3& s) {
  s.rank = 0; s.contiguous = 1;
  for (int k = 0; k < 8; ++k) { s.shape[k] = 1; s.sx[k] = s.sy[k] = s.sz[k] = 0; }
  const bool same = (!y || y->length() == z.length()) && x.length() == z.length();
  if (same) return;
  s.contiguous = 0;
  s.rank = z.rank();
  if (s.rank > 8) { s.rank = 0; s.contiguous = 1; return; }     // callers reject rank > 8 before launching; never write past [8]
  LongType sx = 1, sy = 1, sz = 1;
  for (int k = z.rank() - 1, kx = x.rank() - 1, ky = y ? y->rank() - 1 : -1; k >= 0; --k, --kx, --ky) {
    const LongType od = z.dim(k), xd = kx >= 0 ? x.dim(kx) : 1, yd = (y && ky >= 0) ? y->dim(ky) : 1;
    s.shape[k] = od;
    s.sx[k] = xd == 1 ? 0 : sx; s.sy[k] = yd == 1 ? 0 : sy; s.sz[k] = sz;
    sx *= xd; sy *= yd; sz *= od;
  }
}

// ---------------------------------------------------------------------------------------------- executors
typedef int (*ExecFn)(OpaqueContext&);
typedef int (*ShapeFn)(const std::vector<const LongType*>& in, const OpaqueContext& args, OpaqueShapeList& out);

template <typename T, int FUNCTOR>
inline T host_bin_eval(T x, T y) {
  switch (FUNCTOR) {
    case B_ADD: return (T)(x + y);
    case B_SUB: return (T)(x - y);
    case B_MUL: return (T)(x * y);
    case B_DIV: return (T)(x / y);
    case B_RSUB: return (T)(y - x);
    case B_RDIV: return (T)(y / x);
    case B_POW: return (T)std::pow((double)x, (double)y);
    case B_MAX: return x > y ? x : y;
    case B_MIN: return x < y ? x : y;
    case B_FMOD: return (T)std::fmod((double)x, (double)y);
    case B_FLOORDIV: return (T)std::floor((double)x / (double)y);
    case B_FLOORMOD: { const double r = std::fmod((double)x, (double)y); return (T)((r != 0.0 && ((r < 0.0) != ((double)y < 0.0))) ? r + (double)y : r); }
    case B_SQSUB: return (T)((x - y) * (x - y));
    default: return x;
  }
}

template <int FUNCTOR>
int exec_binary(OpaqueContext& c) {
  if (c.in.size() < 2 || c.out.size() < 1) return fail(20, "binary op: 2 inputs and 1 output are required");
  Arr &a = c.in[0], &b = c.in[1], &o = c.out[0];
  const DType dt = dtype_of(a.si);
  if (dtype_of(b.si) != dt || dtype_of(o.si) != dt) return fail(21, "binary op: operands and output must share one data type");
  Bcast bc;
  if (!make_bcast(a, b, o, bc)) return fail(22, "binary op: shapes are not broadcastable to the output shape");
  if (!contiguous_all(c)) return fail(23, "native executors take c-contiguous arrays");
  if (all_device(c)) {
    Hip* h = hip();
    if (!h || !h->pairwise || legacy_code(dt) < 0) return fail(24, "device execution unavailable for this op / data type");
    if (o.rank() > 8 || a.rank() > 8 || b.rank() > 8) return fail(24, "device broadcast: rank above 8 is not handled natively");
    Strides3 s; fill_strides3(a, &b, o, s);
    return h->pairwise(FUNCTOR, a.dev, b.dev, o.dev, o.length(), &s, nullptr, 0, 0.0, legacy_code(dt), 0, c.stream) == 0 ? 0
           : fail(25, "dl4j_legacy_pairwise failed");
  }
  if (!all_host(c)) return fail(26, "arrays must be all host or all device");
  auto run = [&](auto tag) {
    typedef decltype(tag) T;
    host_binary<T>(a, b, o, bc, [](T x, T y) { return host_bin_eval<T, FUNCTOR>(x, y); });
  };
  switch (dt) {
    case DT_FLOAT: run(float()); return 0;
    case DT_DOUBLE: run(double()); return 0;
    case DT_INT32: run(int32_t()); return 0;
    case DT_INT64: run(int64_t()); return 0;
    default: return fail(27, "host execution: unsupported data type");
  }
}

template <int KIND>   // 0 relu 1 sigmoid 2 tanh 3 identity
int exec_unary(OpaqueContext& c) {
  if (c.in.size() < 1 || c.out.size() < 1) return fail(20, "transform op: 1 input and 1 output are required");
  Arr &a = c.in[0], &o = c.out[0];
  const DType dt = dtype_of(a.si);
  if (dtype_of(o.si) != dt || a.length() != o.length()) return fail(21, "transform op: output must match the input");
  if (!contiguous_all(c)) return fail(23, "native executors take c-contiguous arrays");
  if (all_device(c)) {
    Hip* h = hip();
    if (!h || legacy_code(dt) < 0) return fail(24, "device execution unavailable for this op / data type");
    if (o.rank() > 8 || a.rank() > 8) return fail(24, "device transform: rank above 8 is not handled natively");
    Strides3 s; fill_strides3(a, nullptr, o, s);
    if (KIND == 0) return h->pairwise(B_RELU, a.dev, nullptr, o.dev, o.length(), &s, nullptr, 1, c.t.empty() ? 0.0 : c.t[0], legacy_code(dt), 0, c.stream) == 0 ? 0 : fail(25, "dl4j_legacy_pairwise failed");
    const int f = KIND == 1 ? U_SIGMOID : KIND == 2 ? U_TANH : U_IDENTITY;
    return h->transform(f, a.dev, o.dev, o.length(), &s, nullptr, legacy_code(dt), 0, c.stream) == 0 ? 0 : fail(25, "dl4j_legacy_transform failed");
  }
  if (!all_host(c)) return fail(26, "arrays must be all host or all device");
  const double cut = c.t.empty() ? 0.0 : c.t[0];
  auto run = [&](auto tag) {
    typedef decltype(tag) T;
    if (KIND == 0) host_unary<T>(a, o, [cut](T x) { return x > (T)cut ? x : (T)cut; });
    else if (KIND == 1) host_unary<T>(a, o, [](T x) { return (T)(1.0 / (1.0 + std::exp(-(double)x))); });
    else if (KIND == 2) host_unary<T>(a, o, [](T x) { return (T)std::tanh((double)x); });
    else host_unary<T>(a, o, [](T x) { return x; });
  };
  if (dt == DT_FLOAT) { run(float()); return 0; }
  if (dt == DT_DOUBLE) { run(double()); return 0; }
  return fail(27, "host execution: unsupported data type");
}

int exec_softmax(OpaqueContext& c) {
  if (c.in.size() < 1 || c.out.size() < 1) return fail(20, "softmax: 1 input and 1 output are required");
  Arr &a = c.in[0], &o = c.out[0];
  const DType dt = dtype_of(a.si);
  const int rank = a.rank();
  int dim = c.i.empty() ? rank - 1 : (int)c.i[0];
  if (dim < 0) dim += rank;
  if (dim != rank - 1) return fail(28, "softmax: the native executor normalises the last dimension");
  if (!contiguous_all(c) || a.length() != o.length()) return fail(23, "native executors take c-contiguous arrays");
  const LongType N = a.dim(rank - 1), M = a.length() / (N ? N : 1);
  if (all_device(c)) {
    Hip* h = hip();
    if (!h || !h->softmax || nn_code(dt) < 0) return fail(24, "device execution unavailable for this op / data type");
    return h->softmax(a.dev, o.dev, M, (int)N, 0, nn_code(dt), c.stream) == 0 ? 0 : fail(25, "dl4j_softmax_rows failed");
  }
  if (!all_host(c)) return fail(26, "arrays must be all host or all device");
  auto run = [&](auto tag) {
    typedef decltype(tag) T;
    const T* x = hp<T>(a); T* y = hp<T>(o);
    rt::parallel_for(0, M, 64, [&](LongType lo, LongType hi) {
      for (LongType r = lo; r < hi; ++r) {
        double mx = -INFINITY, s = 0;
        for (LongType k = 0; k < N; ++k) mx = std::max(mx, (double)x[r * N + k]);
        for (LongType k = 0; k < N; ++k) s += std::exp((double)x[r * N + k] - mx);
        for (LongType k = 0; k < N; ++k) y[r * N + k] = (T)(std::exp((double)x[r * N + k] - mx) / s);
      }
    });
  };
  if (dt == DT_FLOAT) { run(float()); return 0; }
  if (dt == DT_DOUBLE) { run(double()); return 0; }
  return fail(27, "host execution: unsupported data type");
}

int abi_matmul(OpaqueContext& c, bool ta, bool tb, bool tc, double alpha, double beta);      // rt_opctx_nn.inc: generic strided fp32 GEMM

// matmul: iArgs [transA, transB, transC]; 2-D operands (the batched / higher-rank forms stay with the fallback)
int exec_matmul(OpaqueContext& c) {
  if (c.in.size() < 2 || c.out.size() < 1) return fail(20, "matmul: 2 inputs and 1 output are required");
  Arr &a = c.in[0], &b = c.in[1], &o = c.out[0];
  if (a.rank() != 2 || b.rank() != 2 || o.rank() != 2) return fail(29, "matmul: the native executor takes rank-2 operands");
  const bool ta = c.i.size() > 0 && c.i[0], tb = c.i.size() > 1 && c.i[1], tc = c.i.size() > 2 && c.i[2];
  const DType dt = dtype_of(a.si);
  if (dtype_of(b.si) != dt || !contiguous_all(c)) return fail(21, "matmul: same-type c-contiguous operands are required");
  const LongType M = ta ? a.dim(1) : a.dim(0), K = ta ? a.dim(0) : a.dim(1), N = tb ? b.dim(0) : b.dim(1);
  if ((tb ? b.dim(1) : b.dim(0)) != K || o.dim(tc ? 1 : 0) != M || o.dim(tc ? 0 : 1) != N) return fail(22, "matmul: inner / output dimensions do not match");
  const double alpha = c.t.size() > 0 ? c.t[0] : 1.0, beta = c.t.size() > 1 ? c.t[1] : 0.0;
  {
    // the MFMA kernel is the "NT" form on device buffers; every other fp32 form (and fp32 host arrays with a transposed result)
    // runs the generic strided GEMM
    const int chunk = dt == DT_FLOAT ? 4 : 8;
    const bool mfma = all_device(c) && !ta && tb && !tc && beta == 0.0 && nn_code(dt) >= 0 && K % chunk == 0 && dtype_of(o.si) == dt;
    if (!mfma && dt == DT_FLOAT && dtype_of(o.si) == DT_FLOAT && (all_device(c) || tc)) return abi_matmul(c, ta, tb, tc, alpha, beta);
  }
  if (tc) return fail(29, "matmul: transposed result is handled natively for fp32 only");
  if (all_device(c)) {
    Hip* h = hip();
    // the MFMA kernel is the "NT" form: A [M,K] row-major, B [N,K] row-major
    if (!h || !h->gemm || nn_code(dt) < 0 || ta || !tb || beta != 0.0 || dtype_of(o.si) != dt)
      return fail(24, "device matmul: natively only C = alpha * A . B^T (transA=0, transB=1, beta=0); other forms use the fallback");
    const int ch = dt == DT_FLOAT ? 4 : 8;
    if (K % ch) return fail(24, "device matmul: K must be a multiple of the 16-byte chunk");
    const int64_t d[14] = {M, N, K, K, K, N, 0, 0, 0, 1, (M >= 128 && N >= 128) ? 0 : 2, nn_code(dt) == 0 ? 2 : (nn_code(dt) == 1 ? 0 : 1), 0, 0};
    return h->gemm(a.dev, b.dev, o.dev, nullptr, nullptr, (float)alpha, nullptr, nullptr, 0, d, c.stream) == 0 ? 0 : fail(25, "dl4j_gemm_nt_ex failed");
  }
  if (!all_host(c)) return fail(26, "arrays must be all host or all device");
  auto run = [&](auto tag) {
    typedef decltype(tag) T;
    const T* A = hp<T>(a); const T* B = hp<T>(b); T* C = hp<T>(o);
    const LongType lda = a.dim(1), ldb = b.dim(1);
    rt::parallel_for(0, M, 8, [&](LongType lo, LongType hi) {
      for (LongType m = lo; m < hi; ++m)
        for (LongType n = 0; n < N; ++n) {
          double s = 0;
          for (LongType k = 0; k < K; ++k) s += (double)(ta ? A[k * lda + m] : A[m * lda + k]) * (double)(tb ? B[n * ldb + k] : B[k * ldb + n]);
          C[m * N + n] = (T)(alpha * s + (beta != 0.0 ? beta * (double)C[m * N + n] : 0.0));
        }
    });
  };
  if (dt == DT_FLOAT) { run(float()); return 0; }
  if (dt == DT_DOUBLE) { run(double()); return 0; }
  return fail(27, "host execution: unsupported data type");
}

// biasadd: x [.., C] (NHWC / 2-D) or NCHW with bArgs[0] = true
int exec_biasadd(OpaqueContext& c) {
  if (c.in.size() < 2 || c.out.size() < 1) return fail(20, "biasadd: 2 inputs and 1 output are required");
  Arr &x = c.in[0], &bi = c.in[1], &o = c.out[0];
  const bool nchw = !c.b.empty() && c.b[0];
  const int rank = x.rank();
  const LongType C = bi.length();
  const int caxis = nchw ? 1 : rank - 1;
  if (x.dim(caxis) != C || !contiguous_all(c) || x.length() != o.length()) return fail(22, "biasadd: bias length must equal the channel dimension");
  LongType inner = 1;
  for (int k = caxis + 1; k < rank; ++k) inner *= x.dim(k);
  const DType dt = dtype_of(x.si);
  if (all_device(c)) {
    Hip* h = hip();
    if (!h || !h->biasadd || nn_code(dt) < 0 || dtype_of(bi.si) != DT_FLOAT) return fail(24, "device biasadd: fp32 bias and fp32/bf16/fp16 data are required");
    return h->biasadd(x.dev, (const float*)bi.dev, o.dev, x.length(), (int)C, inner, nn_code(dt), c.stream) == 0 ? 0 : fail(25, "dl4j_bias_add failed");
  }
  if (!all_host(c)) return fail(26, "arrays must be all host or all device");
  auto run = [&](auto tag) {
    typedef decltype(tag) T;
    const T* X = hp<T>(x); const T* Bv = hp<T>(bi); T* Y = hp<T>(o);
    rt::parallel_for(0, x.length(), 1 << 14, [&](LongType lo, LongType hi) { for (LongType i = lo; i < hi; ++i) Y[i] = X[i] + Bv[(i / inner) % C]; });
  };
  if (dtype_of(bi.si) != dt) return fail(21, "host biasadd: bias must have the data type of x");
  if (dt == DT_FLOAT) { run(float()); return 0; }
  if (dt == DT_DOUBLE) { run(double()); return 0; }
  return fail(27, "host execution: unsupported data type");
}

// reduce_sum / reduce_mean over iArgs dimensions (all when empty), bArgs[0] = keepDims; host only
template <int MEAN>
int exec_reduce(OpaqueContext& c) {
  if (c.in.size() < 1 || c.out.size() < 1) return fail(20, "reduce: 1 input and 1 output are required");
  Arr &x = c.in[0], &o = c.out[0];
  if (!all_host(c)) return fail(24, "reduce: device arrays use the legacy reduce entry points (dl4j_legacy_reduce)");
  if (!contiguous_all(c)) return fail(23, "native executors take c-contiguous arrays");
  const int rank = x.rank();
  std::vector<char> red((size_t)rank, c.i.empty() ? 1 : 0);
  for (LongType dmn : c.i) {
    if (dmn < -rank || dmn >= rank) return fail(22, "reduce: dimension out of range for the input rank");
    red[(size_t)(dmn < 0 ? dmn + rank : dmn)] = 1;
  }
  std::vector<LongType> ostride((size_t)rank, 0);
  LongType s = 1, cnt = 1;
  for (int k = rank - 1; k >= 0; --k) { if (!red[(size_t)k]) { ostride[(size_t)k] = s; s *= x.dim(k); } else cnt *= x.dim(k); }
  if (s != o.length()) return fail(22, "reduce: output length does not match the kept dimensions");
  const DType dt = dtype_of(x.si);
  auto run = [&](auto tag) {
    typedef decltype(tag) T;
    std::vector<double> acc((size_t)s, 0.0);
    const T* X = hp<T>(x);
    const LongType n = x.length();
    for (LongType i = 0; i < n; ++i) {
      LongType r = i, oi = 0;
      for (int k = rank - 1; k >= 0; --k) { const LongType cc = r % x.dim(k); r /= x.dim(k); oi += cc * ostride[(size_t)k]; }
      acc[(size_t)oi] += (double)X[i];
    }
    T* O = hp<T>(o);
    for (LongType i = 0; i < s; ++i) O[i] = (T)(MEAN ? acc[(size_t)i] / (double)cnt : acc[(size_t)i]);
  };
  if (dt == DT_FLOAT) { run(float()); return 0; }
  if (dt == DT_DOUBLE) { run(double()); return 0; }
  return fail(27, "host execution: unsupported data type");
}

// ---------------------------------------------------------------------------------------------- shape functions
void push_shape(OpaqueShapeList& out, const std::vector<LongType>& shape, DType dt) {
  std::vector<LongType> si((size_t)rt::si_len((int)shape.size()));
  nd4j_shape_buffer((int)shape.size(), shape.data(), 'c', flag_of(dt), si.data());
  out.shapes.push_back(std::move(si));
}

int shape_same(const std::vector<const LongType*>& in, const OpaqueContext&, OpaqueShapeList& out) {
  if (in.empty()) return fail(30, "shape function: no input shape");
  std::vector<LongType> s(rt::si_shape(in[0]), rt::si_shape(in[0]) + rt::si_rank(in[0]));
  push_shape(out, s, dtype_of(in[0]));
  return 0;
}
int shape_broadcast(const std::vector<const LongType*>& in, const OpaqueContext&, OpaqueShapeList& out) {
  if (in.size() < 2) return fail(30, "shape function: two input shapes are required");
  const int ra = rt::si_rank(in[0]), rb = rt::si_rank(in[1]), r = std::max(ra, rb);
  std::vector<LongType> s((size_t)r);
  for (int k = 0; k < r; ++k) {
    const LongType a = k < r - ra ? 1 : rt::si_shape(in[0])[k - (r - ra)], b = k < r - rb ? 1 : rt::si_shape(in[1])[k - (r - rb)];
    if (a != b && a != 1 && b != 1) return fail(31, "shape function: shapes are not broadcastable");
    s[(size_t)k] = a == 1 ? b : a;
  }
  push_shape(out, s, dtype_of(in[0]));
  return 0;
}
int shape_matmul(const std::vector<const LongType*>& in, const OpaqueContext& a, OpaqueShapeList& out) {
  if (in.size() < 2 || rt::si_rank(in[0]) != 2 || rt::si_rank(in[1]) != 2) return fail(30, "matmul shape function: two rank-2 shapes");
  const bool ta = a.i.size() > 0 && a.i[0], tb = a.i.size() > 1 && a.i[1], tc = a.i.size() > 2 && a.i[2];
  const LongType M = rt::si_shape(in[0])[ta ? 1 : 0], N = rt::si_shape(in[1])[tb ? 0 : 1];
  push_shape(out, tc ? std::vector<LongType>{N, M} : std::vector<LongType>{M, N}, dtype_of(in[0]));
  return 0;
}
int shape_reduce(const std::vector<const LongType*>& in, const OpaqueContext& a, OpaqueShapeList& out) {
  if (in.empty()) return fail(30, "shape function: no input shape");
  const int rank = rt::si_rank(in[0]);
  const bool keep = !a.b.empty() && a.b[0];
  std::vector<char> red((size_t)rank, a.i.empty() ? 1 : 0);
  for (LongType d : a.i) {
    if (d < -rank || d >= rank) return fail(31, "reduce shape function: dimension out of range for the input rank");
    red[(size_t)(d < 0 ? d + rank : d)] = 1;
  }
  std::vector<LongType> s;
  for (int k = 0; k < rank; ++k) { if (!red[(size_t)k]) s.push_back(rt::si_shape(in[0])[k]); else if (keep) s.push_back(1); }
  push_shape(out, s, dtype_of(in[0]));
  return 0;
}

#include "rt_opctx_nn.inc"

struct NativeOp { ExecFn exec; ShapeFn shape; std::string name; };
std::map<LongType, NativeOp>& table() {
  static std::map<LongType, NativeOp> t = [] {
    std::map<LongType, NativeOp> m;
    auto reg = [&](const char* n, ExecFn e, ShapeFn s) { m[ops_hash(n)] = NativeOp{e, s, n}; };
    reg("add", exec_binary<B_ADD>, shape_broadcast); reg("subtract", exec_binary<B_SUB>, shape_broadcast);
    reg("multiply", exec_binary<B_MUL>, shape_broadcast); reg("divide", exec_binary<B_DIV>, shape_broadcast);
    reg("maximum", exec_binary<B_MAX>, shape_broadcast); reg("minimum", exec_binary<B_MIN>, shape_broadcast);
    reg("relu", exec_unary<0>, shape_same); reg("sigmoid", exec_unary<1>, shape_same); reg("tanh", exec_unary<2>, shape_same);
    reg("identity", exec_unary<3>, shape_same);
    reg("softmax", exec_softmax, shape_same); reg("matmul", exec_matmul, shape_matmul); reg("biasadd", exec_biasadd, shape_same);
    reg("reduce_sum", exec_reduce_any<0>, shape_reduce); reg("reduce_mean", exec_reduce_any<1>, shape_reduce);
    // ---- rt_opctx_nn.inc
    reg("reversesubtract", exec_binary<B_RSUB>, shape_broadcast); reg("reversedivide", exec_binary<B_RDIV>, shape_broadcast);
    reg("squaredsubtract", exec_binary<B_SQSUB>, shape_broadcast); reg("realdiv", exec_binary<B_DIV>, shape_broadcast);
    reg("floordiv", exec_binary<B_FLOORDIV>, shape_broadcast); reg("floormod", exec_binary<B_FLOORMOD>, shape_broadcast);
    reg("mod", exec_binary<B_FMOD>, shape_broadcast); reg("Pow", exec_binary<B_POW>, shape_broadcast);
    reg("equals", exec_compare<B_EQ>, shape_broadcast_bool); reg("not_equals", exec_compare<B_NE>, shape_broadcast_bool);
    reg("greater", exec_compare<B_GT>, shape_broadcast_bool); reg("greater_equal", exec_compare<B_GE>, shape_broadcast_bool);
    reg("less", exec_compare<B_LT>, shape_broadcast_bool); reg("less_equal", exec_compare<B_LE>, shape_broadcast_bool);
    reg("square", exec_act<U_SQUARE, false, 0>, shape_same); reg("cube", exec_act<U_CUBE, false, 0>, shape_same);
    reg("softplus", exec_act<U_SOFTPLUS, false, 0>, shape_same); reg("softsign", exec_act<U_SOFTSIGN, false, 0>, shape_same);
    reg("hardtanh", exec_act<U_HARDTANH, false, 0>, shape_same); reg("hardsigmoid", exec_act<U_HARDSIGMOID, false, 0>, shape_same);
    reg("selu", exec_act<U_SELU, false, 0>, shape_same); reg("rationaltanh", exec_act<U_RATIONALTANH, false, 0>, shape_same);
    reg("rectifiedtanh", exec_act<U_RECTIFIEDTANH, false, 0>, shape_same);
    reg("relu6", exec_act<B_RELU6, true, 0>, shape_same); reg("lrelu", exec_act<B_LRELU, true, 10>, shape_same);
    reg("elu", exec_act<B_ELU, true, 1000>, shape_same);
    reg("relu_bp", exec_act_bp<0, 0>, shape_same); reg("sigmoid_bp", exec_act_bp<1, 0>, shape_same); reg("tanh_bp", exec_act_bp<2, 0>, shape_same);
    reg("relu6_bp", exec_act_bp<3, 0>, shape_same); reg("lrelu_bp", exec_act_bp<4, 10>, shape_same); reg("softplus_bp", exec_act_bp<5, 0>, shape_same);
    reg("softsign_bp", exec_act_bp<6, 0>, shape_same); reg("cube_bp", exec_act_bp<7, 0>, shape_same); reg("hardtanh_bp", exec_act_bp<8, 0>, shape_same);
    reg("hardsigmoid_bp", exec_act_bp<9, 0>, shape_same); reg("elu_bp", exec_act_bp<10, 1000>, shape_same); reg("selu_bp", exec_act_bp<11, 0>, shape_same);
    reg("softmax_bp", exec_softmax_bp<0>, shape_same); reg("log_softmax", exec_log_softmax, shape_same); reg("log_softmax_bp", exec_softmax_bp<1>, shape_same);
    reg("reduce_max", exec_reduce_abi<2>, shape_reduce); reg("reduce_min", exec_reduce_abi<3>, shape_reduce);
    reg("reduce_sqnorm", exec_reduce_abi<4>, shape_reduce); reg("reduce_norm1", exec_reduce_abi<5>, shape_reduce);
    reg("reduce_norm2", exec_reduce_abi<6>, shape_reduce); reg("reduce_prod", exec_reduce_abi<7>, shape_reduce);
    reg("argmax", exec_reduce_abi<8>, shape_reduce_index); reg("argmin", exec_reduce_abi<9>, shape_reduce_index);
    reg("conv2d", exec_conv2d, shape_conv2d); reg("conv2d_bp", exec_conv2d_bp, shape_conv2d_bp);
    reg("maxpool2d", exec_pool2d<1>, shape_pool2d<1>); reg("avgpool2d", exec_pool2d<0>, shape_pool2d<0>);
    reg("maxpool2d_bp", exec_pool2d_bp<1>, shape_same); reg("avgpool2d_bp", exec_pool2d_bp<0>, shape_same);
    reg("batchnorm", exec_batchnorm, shape_same); reg("batchnorm_bp", exec_batchnorm_bp, shape_like_inputs_but_last);
    reg("layer_norm", exec_layer_norm, shape_same); reg("layer_norm_bp", exec_layer_norm_bp, shape_like_inputs_but_last);
    reg("softmax_cross_entropy_loss", exec_sce, shape_sce); reg("softmax_cross_entropy_loss_grad", exec_sce_grad, shape_like_all_inputs);
    reg("sgd_updater", exec_updater<1, 0>, shape_updater<0>); reg("nesterovs_updater", exec_updater<2, 1>, shape_updater<1>);
    reg("adam_updater", exec_updater<3, 2>, shape_updater<2>); reg("rms_prop_updater", exec_updater<4, 1>, shape_updater<1>);
    reg("ada_grad_updater", exec_updater<5, 1>, shape_updater<1>); reg("ada_delta_updater", exec_updater<6, 2>, shape_updater<2>);
    reg("ada_max_updater", exec_updater<7, 2>, shape_updater<2>); reg("nadam_updater", exec_updater<8, 2>, shape_updater<2>);
    reg("ams_grad_updater", exec_updater<9, 3>, shape_updater<3>); reg("adabelief_updater", exec_updater<10, 2>, shape_updater<2>);
    reg("matmul_bp", exec_matmul_bp, shape_first_two); reg("xw_plus_b", exec_xw_plus_b, shape_xw_plus_b);
    reg("xw_plus_b_bp", exec_xw_plus_b_bp, shape_first_three); reg("biasadd_bp", exec_biasadd_bp, shape_first_two);
    reg("gather", exec_gather, shape_gather); reg("scatter_add", exec_scatter_add, shape_same); reg("concat", exec_concat, shape_concat);
    reg("cast", exec_cast, shape_cast); reg("reshape", exec_reshape, shape_reshape); reg("permute", exec_permute, shape_permute);
    reg("transpose", exec_permute, shape_permute); reg("onehot", exec_onehot, shape_onehot); reg("assign", exec_assign, shape_same);
    return m;
  }();
  return t;
}

typedef int (*FallbackFn)(LongType hash, OpaqueContext* ctx, void* user);
FallbackFn g_fallback = nullptr;
void* g_fallback_user = nullptr;

void set_arrays(std::vector<Arr>& dst, int n, void** buffers, void** shapeInfos, void** specials, void** specialShapes) {
  dst.assign((size_t)n, Arr());
  for (int k = 0; k < n; ++k) {
    Arr& a = dst[(size_t)k];
    a.host = buffers ? buffers[k] : nullptr;
    a.dev = specials ? specials[k] : nullptr;
    const LongType* si = shapeInfos ? (const LongType*)shapeInfos[k] : nullptr;
    if (si) { a.own_si.assign(si, si + rt::si_len(rt::si_rank(si))); a.si = a.own_si.data(); }
    a.dsi = specialShapes ? (const LongType*)specialShapes[k] : nullptr;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------- C ABI (names of NativeOps.h)
RT_API OpaqueContext* createGraphContext(int nodeId) { auto* c = new OpaqueContext(); c->node_id = nodeId; return c; }
RT_API void deleteGraphContext(OpaqueContext* ctx) { delete ctx; }
RT_API void ctxPurge(OpaqueContext* ctx) { if (ctx) { ctx->in.clear(); ctx->out.clear(); ctx->t.clear(); ctx->i.clear(); ctx->b.clear(); ctx->d.clear(); } }
RT_API void ctxAllowHelpers(OpaqueContext* ctx, bool reallyAllow) { ctx->allow_helpers = reallyAllow; }
RT_API void ctxSetExecutionMode(OpaqueContext* ctx, int execMode) { ctx->exec_mode = execMode; }
RT_API void ctxShapeFunctionOverride(OpaqueContext*, bool) {}
RT_API void markGraphContextInplace(OpaqueContext* ctx, bool reallyInplace) { ctx->inplace = reallyInplace; }
RT_API void setGraphContextCudaContext(OpaqueContext* ctx, void* stream, void* /*reductionPointer*/, void* /*allocationPointer*/) { ctx->stream = stream; }
RT_API void setGraphContextInputArrays(OpaqueContext* ctx, int numArrays, void** buffer, void** shapeInfo, void** specialBuffer,
                                       void** specialShapeInfo) {
  set_arrays(ctx->in, numArrays, buffer, shapeInfo, specialBuffer, specialShapeInfo);
}
RT_API void setGraphContextOutputArrays(OpaqueContext* ctx, int numArrays, void** buffer, void** shapeInfo, void** specialBuffer,
                                        void** specialShapeInfo) {
  set_arrays(ctx->out, numArrays, buffer, shapeInfo, specialBuffer, specialShapeInfo);
}
RT_API void setGraphContextTArguments(OpaqueContext* ctx, double* arguments, int numberOfArguments) { ctx->t.assign(arguments, arguments + numberOfArguments); }
RT_API void setGraphContextIArguments(OpaqueContext* ctx, LongType* arguments, int numberOfArguments) { ctx->i.assign(arguments, arguments + numberOfArguments); }
RT_API void setGraphContextBArguments(OpaqueContext* ctx, bool* arguments, int numberOfArguments) { ctx->b.assign(arguments, arguments + numberOfArguments); }
RT_API void setGraphContextDArguments(OpaqueContext* ctx, int* arguments, int numberOfArguments) { ctx->d.assign(arguments, arguments + numberOfArguments); }
RT_API int ctxNumInputs(OpaqueContext* ctx) { return (int)ctx->in.size(); }
RT_API int ctxNumOutputs(OpaqueContext* ctx) { return (int)ctx->out.size(); }
// accessors the fallback needs to read a context from the other side of the ABI
namespace {
template <typename V> inline bool in_range(const V& v, int k) { return k >= 0 && (size_t)k < v.size(); }
}
RT_API void* ctxInputBuffer(OpaqueContext* ctx, int k, int special) {
  if (!ctx || !in_range(ctx->in, k)) { fail(14, "ctxInputBuffer: index out of range"); return nullptr; }
  return special ? ctx->in[(size_t)k].dev : ctx->in[(size_t)k].host;
}
RT_API const LongType* ctxInputShapeInfo(OpaqueContext* ctx, int k) {
  if (!ctx || !in_range(ctx->in, k)) { fail(14, "ctxInputShapeInfo: index out of range"); return nullptr; }
  return ctx->in[(size_t)k].si;
}
RT_API void* ctxOutputBuffer(OpaqueContext* ctx, int k, int special) {
  if (!ctx || !in_range(ctx->out, k)) { fail(14, "ctxOutputBuffer: index out of range"); return nullptr; }
  return special ? ctx->out[(size_t)k].dev : ctx->out[(size_t)k].host;
}
RT_API const LongType* ctxOutputShapeInfo(OpaqueContext* ctx, int k) {
  if (!ctx || !in_range(ctx->out, k)) { fail(14, "ctxOutputShapeInfo: index out of range"); return nullptr; }
  return ctx->out[(size_t)k].si;
}
RT_API int ctxNumTArgs(OpaqueContext* ctx) { return (int)ctx->t.size(); }
RT_API int ctxNumIArgs(OpaqueContext* ctx) { return (int)ctx->i.size(); }
RT_API int ctxNumBArgs(OpaqueContext* ctx) { return (int)ctx->b.size(); }
RT_API double ctxTArg(OpaqueContext* ctx, int k) {
  if (!ctx || !in_range(ctx->t, k)) { fail(14, "ctxTArg: index out of range"); return 0.0; }
  return ctx->t[(size_t)k];
}
RT_API LongType ctxIArg(OpaqueContext* ctx, int k) {
  if (!ctx || !in_range(ctx->i, k)) { fail(14, "ctxIArg: index out of range"); return 0; }
  return ctx->i[(size_t)k];
}
RT_API int ctxBArg(OpaqueContext* ctx, int k) {
  if (!ctx || !in_range(ctx->b, k)) { fail(14, "ctxBArg: index out of range"); return 0; }
  return ctx->b[(size_t)k];
}

RT_API void registerCustomOpFallback(void* fn, void* user) { g_fallback = (FallbackFn)fn; g_fallback_user = user; }
RT_API int nativeOpImplemented(LongType hash) { return table().count(hash) ? 1 : 0; }
RT_API int nativeOpCount() { return (int)table().size(); }
// names of the ops with a native executor, '\n'-separated (the native half of a getAllCustomOps-style listing; the embedding
// runtime marks every other registered op as served by its fallback)
RT_API const char* nativeOpList() {
  static std::string s = [] {
    std::vector<std::string> names;
    for (auto& kv : table()) names.push_back(kv.second.name);
    std::sort(names.begin(), names.end());
    std::string r;
    for (auto& n : names) { r += n; r += '\n'; }
    return r;
  }();
  return s.c_str();
}

// Status execCustomOp2(Pointer* extraPointers, LongType hash, OpaqueContext* opContext)   (NativeOps.h:1488)
RT_API int execCustomOp2(void** /*extraPointers*/, LongType hash, OpaqueContext* ctx) {
  try {
    if (!ctx) return fail(10, "execCustomOp2: null context");
    auto it = table().find(hash);
    if (it != table().end()) {
      for (auto& a : ctx->in) if (!a.si) return fail(11, "execCustomOp2: an input array has no shapeInfo");
      for (auto& a : ctx->out) if (!a.si) return fail(11, "execCustomOp2: an output array has no shapeInfo");
      const int rc = it->second.exec(*ctx);
      if (rc == 0 || !g_fallback || rc < 24 || rc > 29) return rc;   // codes 24 / 28 / 29: "not handled natively" -> fallback
    }
    if (g_fallback) return g_fallback(hash, ctx, g_fallback_user);
    return fail(12, "execCustomOp2: op is not implemented natively and no fallback is registered");
  } catch (...) {
    return rt::report_current_exception("execCustomOp2");
  }
}

// OpaqueShapeList* calculateOutputShapes2(Pointer*, LongType hash, Pointer* inputBuffers, Pointer* inputShapes, int numInputShapes,
//                                         double* tArgs, int numTArgs, LongType* iArgs, int numIArgs, bool* bArgs, int numBArgs,
//                                         int* dArgs, int numDArgs)                      (NativeOps.h:1498)
RT_API OpaqueShapeList* calculateOutputShapes2(void** /*extraPointers*/, LongType hash, void** /*inputBuffers*/, void** inputShapes,
                                               int numInputShapes, double* tArgs, int numTArgs, LongType* iArgs, int numIArgs, bool* bArgs,
                                               int numBArgs, int* dArgs, int numDArgs) {
  try {
    auto it = table().find(hash);
    if (it == table().end()) { fail(12, "calculateOutputShapes2: op has no native shape function"); return nullptr; }
    OpaqueContext args;
    if (tArgs) args.t.assign(tArgs, tArgs + numTArgs);
    if (iArgs) args.i.assign(iArgs, iArgs + numIArgs);
    if (bArgs) args.b.assign(bArgs, bArgs + numBArgs);
    if (dArgs) args.d.assign(dArgs, dArgs + numDArgs);
    std::vector<const LongType*> in;
    for (int k = 0; k < numInputShapes; ++k) in.push_back((const LongType*)inputShapes[k]);
    auto* out = new OpaqueShapeList();
    if (it->second.shape(in, args, *out) != 0) { delete out; return nullptr; }
    return out;
  } catch (...) {
    rt::report_current_exception("calculateOutputShapes2");
    return nullptr;
  }
}
RT_API LongType getShapeListSize(OpaqueShapeList* list) { return list ? (LongType)list->shapes.size() : 0; }
RT_API const LongType* getShape(OpaqueShapeList* list, LongType i) { return list->shapes[(size_t)i].data(); }
RT_API void deleteShapeList(OpaqueShapeList* list) { delete list; }
