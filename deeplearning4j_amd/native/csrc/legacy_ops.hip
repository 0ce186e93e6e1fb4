// "Legacy" op families for gfx950: pairwise / broadcast / scalar / transform / reduce / reduce3 /
// indexreduce / summarystats, dispatched by functor id.
//
// Reference role: libnd4j loops/cuda/{pairwise,broadcasting,scalar,transform/*,reduce/*,reduce3,
// indexreduce,summarystatsreduce}.* + ops/ops.h functors + legacy/cuda/NativeOpExecutioner.cu
// (SURVEY §2.2). The opNum tables (ABI, Appendix E) live on the Python side
// (nd4j/ops/legacy.py) and map each family's opNum to the functor ids below.
//
// Own design (not a port of the TAD/shapeInfo-pointer kernels):
//  * strides travel BY VALUE in the kernel argument block (rank <= 8 after dimension coalescing on
//    the host), so there are no device-side shapeInfo buffers, no constant caches to sync;
//  * broadcast == pairwise with zero strides; scalar == pairwise with a by-value scalar;
//  * reductions never use the "last block done" threadfence idiom of reduce_float.chpp:142-223
//    (not safe across XCD L2s): long reductions are split into segments -> partial accumulators ->
//    a second merge kernel. Short TADs get one wave64 each, long ones a 256-thread block.
//  * 64-bit indexing throughout (288 GB HBM makes > 2^31 elements reachable).
#include "common.h"
#include <type_traits>

#define MAXR 8

struct Strides3 {
  int rank;
  int contiguous;                 // 1: x,y,z all dense with identical order -> linear index
  int64_t shape[MAXR];
  int64_t sx[MAXR], sy[MAXR], sz[MAXR];
};

struct TadDesc {
  int irank, orank;               // inner (reduced) dims, outer (kept) dims
  int64_t ishape[MAXR], istride[MAXR], iystride[MAXR];
  int64_t oshape[MAXR], ostride[MAXR], oystride[MAXR];
  int64_t tad_len, num_tads;
};

// ---- binary functors ------------------------------------------------------------------------------
enum BinF {
  B_ADD = 0, B_SUB, B_MUL, B_DIV, B_RSUB, B_RDIV, B_COPY, B_POW, B_MAX, B_MIN, B_AMAX, B_AMIN, B_AXPY, B_CAS, B_CAR,
  B_REMAINDER, B_FMOD, B_ATAN2, B_TRUNCDIV, B_FLOORDIV, B_FLOORMOD, B_SQSUB, B_RMOD, B_SAFEDIV, B_MOD, B_RELERR,
  B_BINRELERR, B_BINMINABSRELERR, B_LOR, B_LXOR, B_LNOT, B_LAND, B_POWDERIV, B_LOGPOISSON, B_LOGPOISSONFULL,
  B_TRUNCMOD, B_REPLACENANS, B_DIVNONAN, B_IGAMMA, B_IGAMMAC,
  // scalar-only
  B_ELU, B_ELUDERIV, B_SXE_SMOOTH, B_LRELU, B_LRELUDERIV, B_LOGX, B_RELU, B_RELU6, B_STEP, B_LSTMCLIP, B_SQRSUB,
  B_RPOW, B_RELUDERIV,
  // boolean-result
  B_EQ, B_GT, B_LT, B_EPS, B_GTE, B_MATCH, B_NEQ, B_AND, B_OR, B_XOR, B_NOT, B_LTE,
  // integer
  B_SHL, B_SHR, B_CSHL, B_CSHR, B_IAND, B_IOR, B_IXOR
};

// MatchCondition modes (nd4j Conditions): extra[0]=compare value, extra[1]=eps, extra[2]=mode
template <typename A>
__device__ __forceinline__ bool match_cond(A x, A cmp, A eps, int mode) {
  switch (mode) {
    case 0: return fabs((double)(x - cmp)) <= (double)eps;   // epsEquals
    case 1: return fabs((double)(x - cmp)) > (double)eps;    // epsNotEquals
    case 2: return x < cmp;
    case 3: return x > cmp;
    case 4: return x <= cmp;
    case 5: return x >= cmp;
    case 6: return fabs((double)x) < (double)cmp;            // absLessThan
    case 7: return fabs((double)x) > (double)cmp;
    case 8: return isinf((double)x);
    case 9: return isnan((double)x);
    case 10: return x == cmp;
    case 11: return x != cmp;
    case 12: return fabs((double)x) >= (double)cmp;
    case 13: return fabs((double)x) <= (double)cmp;
    case 14: return isfinite((double)x);
    case 15: return !isfinite((double)x);
  }
  return false;
}

// regularized incomplete gamma P(a,x) (series for x<a+1, continued fraction otherwise)
__device__ double igamma_p(double a, double x) {
  if (x <= 0.0 || a <= 0.0) return 0.0;
  double gln = lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 200; ++n) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (fabs(del) < fabs(sum) * 1e-15) break;
    }
    return sum * exp(-x + a * log(x) - gln);
  }
  double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, h = d;
  for (int i = 1; i < 200; ++i) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < 1e-300) d = 1e-300;
    c = b + an / c;
    if (fabs(c) < 1e-300) c = 1e-300;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-15) break;
  }
  return 1.0 - exp(-x + a * log(x) - gln) * h;
}

template <typename A>
__device__ __forceinline__ A rel_err(A a, A b) {
  if (a == (A)0 && b == (A)0) return (A)0;
  return (A)(fabs((double)(a - b)) / (fabs((double)a) + fabs((double)b)));
}

template <typename A>
__device__ __forceinline__ A bin_op(int f, A x, A y, const double* e) {
  switch (f) {
    case B_ADD: return x + y;
    case B_SUB: return x - y;
    case B_MUL: return x * y;
    case B_DIV: return x / y;
    case B_RSUB: return y - x;
    case B_RDIV: return y / x;
    case B_COPY: return y;
    case B_POW: return (A)pow((double)x, (double)y);
    case B_RPOW: return (A)pow((double)y, (double)x);
    case B_MAX: return x > y ? x : y;
    case B_MIN: return x < y ? x : y;
    case B_AMAX: return fabs((double)x) > fabs((double)y) ? x : y;
    case B_AMIN: return fabs((double)x) < fabs((double)y) ? x : y;
    case B_AXPY: return (A)(e[0] * (double)x) + y;
    case B_CAS: {   // CompareAndSet: where cond(x, e0) holds take y (pairwise) / e[3] (set value)
      return match_cond<A>(x, (A)e[0], (A)e[1], (int)e[2]) ? y : x;
    }
    case B_CAR: return match_cond<A>(y, (A)e[0], (A)e[1], (int)e[2]) ? y : x;   // CompareAndReplace
    case B_REMAINDER: return (A)remainder((double)x, (double)y);
    case B_FMOD: return (A)fmod((double)x, (double)y);
    case B_ATAN2: return (A)atan2((double)x, (double)y);
    case B_TRUNCDIV: return (A)trunc((double)x / (double)y);
    case B_FLOORDIV: return (A)floor((double)x / (double)y);
    case B_FLOORMOD: {
      double m = fmod((double)x, (double)y);
      return (A)((m != 0.0 && ((m < 0.0) != ((double)y < 0.0))) ? m + (double)y : m);
    }
    case B_SQSUB: return (x - y) * (x - y);
    case B_SQRSUB: return (y - x) * (y - x);
    case B_RMOD: return (A)((long long)y % ((long long)x == 0 ? 1 : (long long)x));
    case B_SAFEDIV: return y == (A)0 ? (A)0 : x / y;
    case B_MOD:
    case B_TRUNCMOD: return (A)((long long)x % ((long long)y == 0 ? 1 : (long long)y));
    case B_RELERR: return rel_err<A>(x, y);
    case B_BINRELERR: return rel_err<A>(x, y) > (A)e[0] ? (A)1 : (A)0;
    case B_BINMINABSRELERR: {
      if (fabs((double)(x - y)) < e[1]) return (A)0;
      return rel_err<A>(x, y) > (A)e[0] ? (A)1 : (A)0;
    }
    case B_LOR: return (x != (A)0 || y != (A)0) ? (A)1 : (A)0;
    case B_LXOR: return ((x != (A)0) != (y != (A)0)) ? (A)1 : (A)0;
    case B_LNOT: return (x != (A)0 && y == (A)0) ? (A)1 : ((x == (A)0) ? (A)1 : (A)0);
    case B_LAND: return (x != (A)0 && y != (A)0) ? (A)1 : (A)0;
    case B_POWDERIV: return (A)((double)y * pow((double)x, (double)y - 1.0));
    case B_LOGPOISSON: return (A)(exp((double)y) - (double)x * (double)y);
    case B_LOGPOISSONFULL: {
      double z = (double)x, c = (double)y;
      return (A)(exp(c) - z * c + (z * log(z) - z + 0.5 * log(6.283185307179586 * z)));
    }
    case B_REPLACENANS: return isnan((double)x) ? y : x;
    case B_DIVNONAN: return y == (A)0 ? (A)0 : x / y;
    case B_IGAMMA: return (A)igamma_p((double)x, (double)y);
    case B_IGAMMAC: return (A)(1.0 - igamma_p((double)x, (double)y));
    case B_ELU: return x >= (A)0 ? x : (A)((double)y * (exp((double)x) - 1.0));
    case B_ELUDERIV: return x >= (A)0 ? (A)1 : (A)((double)y * exp((double)x));
    case B_SXE_SMOOTH: return (A)((double)x * (1.0 - (double)y) + 0.5 * (double)y);
    case B_LRELU: return x < (A)0 ? (A)((double)y * (double)x) : x;
    case B_LRELUDERIV: return x >= (A)0 ? (A)1 : y;
    case B_LOGX: return (A)(log((double)x) / log((double)y));
    case B_RELU: return x > y ? x : y;
    case B_RELU6: {
      A r = x > y ? x : y;
      return r < (A)6 ? r : (A)6;
    }
    case B_STEP: return x > y ? (A)1 : (A)0;
    case B_LSTMCLIP: return x > y ? y : (x < -y ? -y : x);
    case B_RELUDERIV: return x > y ? (A)1 : (A)0;
    case B_EQ: return x == y ? (A)1 : (A)0;
    case B_GT: return x > y ? (A)1 : (A)0;
    case B_LT: return x < y ? (A)1 : (A)0;
    case B_EPS: {
      double d = fabs((double)x - (double)y);
      double eps = e ? e[0] : 0.0;
      return d <= (eps > 0.0 ? eps : 1e-5) ? (A)1 : (A)0;
    }
    case B_GTE: return x >= y ? (A)1 : (A)0;
    case B_MATCH: return match_cond<A>(x, (A)e[0], (A)e[1], (int)e[2]) ? (A)1 : (A)0;
    case B_NEQ: return x != y ? (A)1 : (A)0;
    case B_AND: return (x != (A)0 && y != (A)0) ? (A)1 : (A)0;
    case B_OR: return (x != (A)0 || y != (A)0) ? (A)1 : (A)0;
    case B_XOR: return ((x != (A)0) != (y != (A)0)) ? (A)1 : (A)0;
    case B_NOT: return x == (A)0 ? (A)1 : (A)0;
    case B_LTE: return x <= y ? (A)1 : (A)0;
    case B_SHL: return (A)((long long)x << ((long long)y & 63));
    case B_SHR: return (A)((long long)x >> ((long long)y & 63));
    case B_CSHL: {
      uint32_t v = (uint32_t)(long long)x;
      int s = (int)((long long)y & 31);
      return (A)(int32_t)((v << s) | (v >> ((32 - s) & 31)));
    }
    case B_CSHR: {
      uint32_t v = (uint32_t)(long long)x;
      int s = (int)((long long)y & 31);
      return (A)(int32_t)((v >> s) | (v << ((32 - s) & 31)));
    }
    case B_IAND: return (A)((long long)x & (long long)y);
    case B_IOR: return (A)((long long)x | (long long)y);
    case B_IXOR: return (A)((long long)x ^ (long long)y);
  }
  return x;
}

// ---- unary functors -----------------------------------------------------------------------------------
enum UnF {
  U_ABS = 0, U_SIGN, U_ONES, U_NEG, U_ROUND, U_TIMESONEMINUS, U_CUBE, U_ONEMINUS, U_RECIPROCAL, U_SQUARE, U_CAS,
  U_IDENTITY, U_CEIL, U_FLOOR, U_CLIP, U_COPY, U_SQRT, U_RSQRT,
  U_ISINF, U_ISNAN, U_ISFINITE, U_ISINFORNAN, U_MATCHBOOL, U_ISPOS, U_NOT, U_ISNEG,
  U_SCALEDTANH, U_AFFINE, U_TANHD, U_HARDTANHD, U_SIGMOIDD, U_SOFTSIGND, U_TAND, U_SELUD, U_HARDSIGMOIDD, U_RATTANHD,
  U_RECTTANHD, U_SWISHD, U_ACOSHD, U_ASINHD, U_SINHD, U_LOGSIGMOIDD, U_SPECIALD, U_STABILIZE, U_CUBED, U_COS, U_EXP,
  U_LOG, U_SETRANGE, U_SIGMOID, U_SIN, U_SOFTPLUS, U_TANH, U_ACOS, U_ASIN, U_ATAN, U_HARDTANH, U_SOFTSIGN,
  U_HARDSIGMOID, U_RATTANH, U_RECTTANH, U_SINH, U_COSH, U_TAN, U_SELU, U_SWISH, U_LOG1P, U_ERF, U_ACOSH, U_ASINH,
  U_RINT, U_LOGSIGMOID, U_ERFC, U_EXPM1, U_ATANH, U_GELU, U_GELUD, U_PGELU, U_PGELUD, U_MISH, U_MISHD
};

__device__ __forceinline__ double sigm(double x) { return 1.0 / (1.0 + exp(-x)); }
__device__ __forceinline__ double rat_tanh(double x) {
  double d = x * (2.0 / 3.0);
  double a = fabs(d), sg = d < 0 ? -1.0 : 1.0;
  double t = sg * (1.0 - 1.0 / (1.0 + a + d * d + 1.41645 * d * d * d * d));
  return 1.7159 * t;
}

__device__ __forceinline__ double un_op(int f, double x, const double* e) {
  const double SELU_A = 1.6732632423543772, SELU_L = 1.0507009873554805;
  switch (f) {
    case U_ABS: return fabs(x);
    case U_SIGN: return (x > 0.0) - (x < 0.0);
    case U_ONES: return 1.0;
    case U_NEG: return -x;
    case U_ROUND: return round(x);
    case U_TIMESONEMINUS: return x * (1.0 - x);
    case U_CUBE: return x * x * x;
    case U_ONEMINUS: return 1.0 - x;
    case U_RECIPROCAL: return 1.0 / x;
    case U_SQUARE: return x * x;
    case U_CAS: return match_cond<double>(x, e[0], e[1], (int)e[2]) ? e[3] : x;
    case U_IDENTITY:
    case U_COPY: return x;
    case U_CEIL: return ceil(x);
    case U_FLOOR: return floor(x);
    case U_CLIP: return x < e[0] ? e[0] : (x > e[1] ? e[1] : x);
    case U_SQRT: return sqrt(x);
    case U_RSQRT: return 1.0 / sqrt(x);
    case U_ISINF: return isinf(x) ? 1.0 : 0.0;
    case U_ISNAN: return isnan(x) ? 1.0 : 0.0;
    case U_ISFINITE: return isfinite(x) ? 1.0 : 0.0;
    case U_ISINFORNAN: return isfinite(x) ? 0.0 : 1.0;
    case U_MATCHBOOL: return match_cond<double>(x, e[0], e[1], (int)e[2]) ? 1.0 : 0.0;
    case U_ISPOS: return x > 0.0 ? 1.0 : 0.0;
    case U_NOT: return x == 0.0 ? 1.0 : 0.0;
    case U_ISNEG: return x < 0.0 ? 1.0 : 0.0;
    case U_SCALEDTANH: return e[0] * tanh(e[1] * x);
    case U_AFFINE: return e[0] * x + e[1];
    case U_TANHD: { double t = tanh(x); return 1.0 - t * t; }
    case U_HARDTANHD: return (x >= -1.0 && x <= 1.0) ? 1.0 : 0.0;
    case U_SIGMOIDD: { double s = sigm(x); return s * (1.0 - s); }
    case U_SOFTSIGND: { double d = 1.0 + fabs(x); return 1.0 / (d * d); }
    case U_TAND: { double c = cos(x); return 1.0 / (c * c); }
    case U_SELUD: return x > 0.0 ? SELU_L : SELU_L * SELU_A * exp(x);
    case U_HARDSIGMOIDD: return (x < -2.5 || x > 2.5) ? 0.0 : 0.2;
    case U_RATTANHD: {
      double d = x * (2.0 / 3.0), a = fabs(d);
      double den = 1.0 + a + d * d + 1.41645 * d * d * d * d;
      double td = (1.0 + (d < 0 ? -1.0 : 1.0) * (2.0 * d + 4.0 * 1.41645 * d * d * d)) / (den * den);
      return 1.7159 * (2.0 / 3.0) * td;
    }
    case U_RECTTANHD: { double t = tanh(x); return x > 0.0 ? 1.0 - t * t : 0.0; }
    case U_SWISHD: { double s = sigm(x); return s + x * s * (1.0 - s); }
    case U_ACOSHD: return 1.0 / (sqrt(x - 1.0) * sqrt(x + 1.0));
    case U_ASINHD: return 1.0 / sqrt(x * x + 1.0);
    case U_SINHD: return cosh(x);
    case U_LOGSIGMOIDD: return 1.0 / (exp(x) + 1.0);
    case U_SPECIALD: return x * (1.0 - x);
    case U_STABILIZE: {
      double k = e[0];
      if (x * k > 88.0) return 88.0 / k;
      if (x * k < -88.0) return -88.0 / k;
      return x;
    }
    case U_CUBED: return 3.0 * x * x;
    case U_COS: return cos(x);
    case U_EXP: return exp(x);
    case U_LOG: return log(x);
    case U_SETRANGE: {
      double mn = e[0], mx = e[1];
      if (x >= mn && x <= mx) return x;
      if (mn == 0.0 && mx == 1.0) return floor(sigm(x) * (mx - mn)) + mn;
      return floor(x * (mx - mn)) + mn;
    }
    case U_SIGMOID: return sigm(x);
    case U_SIN: return sin(x);
    case U_SOFTPLUS: return x > 30.0 ? x : log1p(exp(x));
    case U_TANH: return tanh(x);
    case U_ACOS: return acos(x);
    case U_ASIN: return asin(x);
    case U_ATAN: return atan(x);
    case U_HARDTANH: return x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x);
    case U_SOFTSIGN: return x / (1.0 + fabs(x));
    case U_HARDSIGMOID: return fmin(1.0, fmax(0.0, 0.2 * x + 0.5));
    case U_RATTANH: return rat_tanh(x);
    case U_RECTTANH: return fmax(0.0, tanh(x));
    case U_SINH: return sinh(x);
    case U_COSH: return cosh(x);
    case U_TAN: return tan(x);
    case U_SELU: return x > 0.0 ? SELU_L * x : SELU_L * SELU_A * (exp(x) - 1.0);
    case U_SWISH: return x * sigm(x);
    case U_LOG1P: return log1p(x);
    case U_ERF: return erf(x);
    case U_ACOSH: return acosh(x);
    case U_ASINH: return asinh(x);
    case U_RINT: return rint(x);
    case U_LOGSIGMOID: return -(x < -30.0 ? -x : log1p(exp(-x)));
    case U_ERFC: return erfc(x);
    case U_EXPM1: return expm1(x);
    case U_ATANH: return atanh(x);
    case U_GELU: return x * sigm(1.702 * x);
    case U_GELUD: { double s = sigm(1.702 * x); return s + 1.702 * x * s * (1.0 - s); }
    case U_PGELU: { double u = 0.7978845608028654 * (x + 0.044715 * x * x * x); return 0.5 * x * (1.0 + tanh(u)); }
    case U_PGELUD: {
      double u = 0.7978845608028654 * (x + 0.044715 * x * x * x), t = tanh(u);
      return 0.5 * (1.0 + t) + 0.5 * x * (1.0 - t * t) * 0.7978845608028654 * (1.0 + 3.0 * 0.044715 * x * x);
    }
    case U_MISH: { double sp = x > 30.0 ? x : log1p(exp(x)); return x * tanh(sp); }
    case U_MISHD: {
      double sp = x > 30.0 ? x : log1p(exp(x)), t = tanh(sp), s = sigm(x);
      return t + x * (1.0 - t * t) * s;
    }
  }
  return x;
}

// float fast path for the hot activations (avoids fp64 math on the common fp32/bf16 tensors)
__device__ __forceinline__ bool un_op_f32(int f, float x, float& out) {
  switch (f) {
    case U_ABS: out = fabsf(x); return true;
    case U_NEG: out = -x; return true;
    case U_SQUARE: out = x * x; return true;
    case U_SQRT: out = sqrtf(x); return true;
    case U_RSQRT: out = rsqrtf(x); return true;
    case U_EXP: out = __expf(x); return true;
    case U_LOG: out = __logf(x); return true;
    case U_TANH: out = tanhf(x); return true;
    case U_SIGMOID: out = 1.f / (1.f + __expf(-x)); return true;
    case U_IDENTITY:
    case U_COPY: out = x; return true;
    case U_RECIPROCAL: out = 1.f / x; return true;
    case U_ONEMINUS: out = 1.f - x; return true;
    case U_SIN: out = sinf(x); return true;
    case U_COS: out = cosf(x); return true;
    case U_SIGN: out = (float)((x > 0.f) - (x < 0.f)); return true;
    case U_CUBE: out = x * x * x; return true;
    case U_CEIL: out = ceilf(x); return true;
    case U_FLOOR: out = floorf(x); return true;
    case U_ROUND: out = roundf(x); return true;
    case U_RINT: out = rintf(x); return true;
    case U_TIMESONEMINUS: out = x * (1.f - x); return true;
    case U_SOFTPLUS: out = x > 30.f ? x : log1pf(__expf(x)); return true;
    case U_SOFTSIGN: out = x / (1.f + fabsf(x)); return true;
    case U_HARDTANH: out = x < -1.f ? -1.f : (x > 1.f ? 1.f : x); return true;
    case U_HARDSIGMOID: out = fminf(1.f, fmaxf(0.f, 0.2f * x + 0.5f)); return true;
    case U_SWISH: out = x / (1.f + __expf(-x)); return true;
    case U_GELU: out = x / (1.f + __expf(-1.702f * x)); return true;
    case U_PGELU: { const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x); out = 0.5f * x * (1.f + tanhf(u)); return true; }
    case U_SELU: out = x > 0.f ? 1.0507009873554805f * x : 1.0507009873554805f * 1.6732632423543772f * (__expf(x) - 1.f); return true;
    case U_LOG1P: out = log1pf(x); return true;
    case U_EXPM1: out = expm1f(x); return true;
    case U_ERF: out = erff(x); return true;
    case U_TANHD: { const float t = tanhf(x); out = 1.f - t * t; return true; }
    case U_SIGMOIDD: { const float sg = 1.f / (1.f + __expf(-x)); out = sg * (1.f - sg); return true; }
    case U_LOGSIGMOID: out = -(x < -30.f ? -x : log1pf(__expf(-x))); return true;
    case U_SINH: out = sinhf(x); return true;
    case U_COSH: out = coshf(x); return true;
    case U_TAN: out = tanf(x); return true;
    case U_ATAN: out = atanf(x); return true;
  }
  return false;
}

template <typename T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };
template <> struct Acc<int32_t> { typedef long long type; };
template <> struct Acc<int64_t> { typedef long long type; };

template <typename T> __device__ __forceinline__ typename Acc<T>::type ldv(const T* p, int64_t i) { return (typename Acc<T>::type)p[i]; }
template <> __device__ __forceinline__ float ldv<bf16_t>(const bf16_t* p, int64_t i) { return bf2f(p[i]); }
template <typename T, typename A> __device__ __forceinline__ void stv(T* p, int64_t i, A v) { p[i] = (T)v; }
template <> __device__ __forceinline__ void stv<bf16_t, float>(bf16_t* p, int64_t i, float v) { p[i] = f2bf(v); }
template <> __device__ __forceinline__ void stv<bf16_t, double>(bf16_t* p, int64_t i, double v) { p[i] = f2bf((float)v); }

__device__ __forceinline__ void offsets3(const Strides3& s, int64_t idx, int64_t& ox, int64_t& oy, int64_t& oz) {
  ox = oy = oz = 0;
#pragma unroll
  for (int d = MAXR - 1; d >= 0; --d) {
    if (d < s.rank) {
      int64_t c = idx % s.shape[d];
      idx /= s.shape[d];
      ox += c * s.sx[d];
      oy += c * s.sy[d];
      oz += c * s.sz[d];
    }
  }
}

struct Extra {
  double e[4];
};

// Z = type of output (T, or uint8 for bool-result families)
template <typename T, typename Z>
__global__ void pairwise_kernel(const T* __restrict__ x, const T* __restrict__ y, Z* __restrict__ z, int64_t n, int f,
                                Strides3 s, Extra ex, int use_scalar, double scalar) {
  typedef typename Acc<T>::type A;
  const A sc = (A)scalar;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t ox, oy, oz;
    if (s.contiguous) ox = oy = oz = i;
    else offsets3(s, i, ox, oy, oz);
    A a = ldv<T>(x, ox);
    A b = use_scalar ? sc : ldv<T>(y, oy);
    A r = bin_op<A>(f, a, b, ex.e);
    stv<Z, A>(z, oz, r);
  }
}

template <typename T, typename Z>
__global__ void transform_kernel(const T* __restrict__ x, Z* __restrict__ z, int64_t n, int f, Strides3 s, Extra ex) {
  typedef typename Acc<T>::type A;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t ox, oy, oz;
    if (s.contiguous) ox = oz = i;
    else offsets3(s, i, ox, oy, oz);
    A a = ldv<T>(x, ox);
    float fo;
    if (sizeof(A) == 4 && un_op_f32(f, (float)a, fo)) stv<Z, A>(z, oz, (A)fo);
    else stv<Z, A>(z, oz, (A)un_op(f, (double)a, ex.e));
  }
}


// ---- 16-byte vector fast paths ----------------------------------------------------------------------
// The generic kernels above move one element per thread per iteration (2- or 4-byte accesses) and recompute rank-8 offsets for every
// element of a broadcast; measured on MI355X that is 0.8-3.8 TB/s where the same math through 16-byte accesses reaches ~5.5 TB/s
// (profiles/r2_legacy_bandwidth.md). For fp32 / bf16 operands that are dense, 16-byte aligned and either the same shape, a scalar, a
// row vector (y index = i % mid) or a "one y per 16 bytes" broadcast (column vector, per-channel: y index = (i / inner) % mid with
// inner a multiple of the vector width) a thread moves whole 16-byte vectors; the functor id stays a (wave-uniform) runtime switch.
template <typename T> struct VecIO;
template <> struct VecIO<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  }
};
template <> struct VecIO<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
    const s16x8 t = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf2f((bf16_t)t[i]);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
    s16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (short)f2bf(v[i]);
    *reinterpret_cast<s16x8*>(p) = t;
  }
};

enum { YM_SAME = 0, YM_SCALAR = 1, YM_ONE_PER_VEC = 2, YM_ROW = 3 };

// F >= 0: the functor id is a compile-time constant (the switch folds away: hot functors); F < 0: runtime id `f`
template <typename T, int F>
__global__ void __launch_bounds__(256) pairwise_vec_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ z,
                                                           int64_t nvec, int f_rt, Extra ex, int ymode, float scalar, int64_t inner,
                                                           int64_t mid) {
  constexpr int N = VecIO<T>::N;
  const int f = F >= 0 ? F : f_rt;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    const int64_t i0 = v * N;
    float a[N], b[N], r[N];
    VecIO<T>::ld(x + i0, a);
    if (ymode == YM_SAME) VecIO<T>::ld(y + i0, b);
    else if (ymode == YM_ROW) VecIO<T>::ld(y + (i0 % mid), b);
    else {
      const float yv = ymode == YM_SCALAR ? scalar : (float)ldv<T>(y, (i0 / inner) % mid);
#pragma unroll
      for (int k = 0; k < N; ++k) b[k] = yv;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = bin_op<float>(f, a[k], b[k], ex.e);
    VecIO<T>::st(z + i0, r);
  }
}

template <typename T, int F>
__global__ void __launch_bounds__(256) transform_vec_kernel(const T* __restrict__ x, T* __restrict__ z, int64_t nvec, int f_rt, Extra ex) {
  constexpr int N = VecIO<T>::N;
  const int f = F >= 0 ? F : f_rt;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    float a[N], r[N];
    VecIO<T>::ld(x + v * N, a);
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (!un_op_f32(f, a[k], r[k])) r[k] = (float)un_op(f, (double)a[k], ex.e);
    VecIO<T>::st(z + v * N, r);
  }
}

#define HOT_BIN(X) X(B_ADD) X(B_SUB) X(B_MUL) X(B_DIV) X(B_RSUB) X(B_RDIV) X(B_COPY) X(B_MAX) X(B_MIN) X(B_SQSUB) X(B_RELU) X(B_LRELU)
#define HOT_UN(X) X(U_ABS) X(U_NEG) X(U_SQUARE) X(U_SQRT) X(U_RSQRT) X(U_EXP) X(U_LOG) X(U_TANH) X(U_SIGMOID) X(U_IDENTITY) X(U_COPY) \
  X(U_RECIPROCAL) X(U_ONEMINUS) X(U_SIGN) X(U_CUBE) X(U_FLOOR) X(U_CEIL) X(U_SOFTPLUS) X(U_SWISH) X(U_GELU) X(U_PGELU) X(U_HARDTANH) \
  X(U_HARDSIGMOID) X(U_SOFTSIGN) X(U_TANHD) X(U_SIGMOIDD) X(U_SIN) X(U_COS) X(U_ERF) X(U_SELU)

template <typename T>
static void launch_pairwise_vec(int g, hipStream_t st, const T* x, const T* y, T* z, int64_t nvec, int f, const Extra& ex, int ymode,
                                float scalar, int64_t inner, int64_t mid) {
  switch (f) {
#define X(ID) case ID: pairwise_vec_kernel<T, ID><<<g, 256, 0, st>>>(x, y, z, nvec, f, ex, ymode, scalar, inner, mid); return;
    HOT_BIN(X)
#undef X
    default: pairwise_vec_kernel<T, -1><<<g, 256, 0, st>>>(x, y, z, nvec, f, ex, ymode, scalar, inner, mid);
  }
}

template <typename T>
static void launch_transform_vec(int g, hipStream_t st, const T* x, T* z, int64_t nvec, int f, const Extra& ex) {
  switch (f) {
#define X(ID) case ID: transform_vec_kernel<T, ID><<<g, 256, 0, st>>>(x, z, nvec, f, ex); return;
    HOT_UN(X)
#undef X
    default: transform_vec_kernel<T, -1><<<g, 256, 0, st>>>(x, z, nvec, f, ex);
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Classifies y's addressing for the vector kernel: returns the ymode or -1 (use the generic kernel).
static int classify_y(const Strides3& s, int vecn, int64_t& inner, int64_t& mid) {
  inner = 1; mid = 1;
  // x and z must be dense in the (coalesced) iteration order
  int64_t dense = 1;
  for (int d = s.rank - 1; d >= 0; --d) {
    if (s.shape[d] != 1 && (s.sx[d] != dense || s.sz[d] != dense)) return -1;
    dense *= s.shape[d];
  }
  // y: a run of dims with dense strides ("mid"), zero strides outside it
  int lo = -1, hi = -1;
  for (int d = 0; d < s.rank; ++d)
    if (s.shape[d] != 1 && s.sy[d] != 0) { if (lo < 0) lo = d; hi = d; }
  if (lo < 0) return -1;
  int64_t yd = 1;
  for (int d = hi; d >= lo; --d) {
    if (s.shape[d] != 1 && s.sy[d] != yd) return -1;
    yd *= s.shape[d];
  }
  mid = yd;
  for (int d = s.rank - 1; d > hi; --d) inner *= s.shape[d];
  if (inner == 1) return (mid % vecn == 0) ? YM_ROW : -1;
  return (inner % vecn == 0) ? YM_ONE_PER_VEC : -1;
}

// ---- reductions -----------------------------------------------------------------------------------
enum RedF {
  R_MEAN = 0, R_AMEAN, R_NORM1, R_NORM2, R_NORMMAX, R_NORMFROB, R_NORMP, R_SQNORM, R_ENTROPY, R_LOGENTROPY, R_SHANNON,
  R_SUM, R_MAX, R_MIN, R_PROD, R_ASUM, R_AMAX, R_AMIN,
  R_ANY, R_ALL, R_ISFINITE, R_ISINFORNAN, R_ISNAN, R_ISINF, R_ISPOS, R_ISNEG,
  R_COUNTNZ, R_COUNTZ, R_MATCH,
  // reduce3 (two operands; up to 3 running sums)
  R3_MANHATTAN, R3_EUCLID, R3_COSSIM, R3_DOT, R3_EQEPS, R3_COSDIST, R3_JACCARD, R3_HAMMING,
  // summary stats
  RS_VAR, RS_STD,
  // index reduce
  RI_MAX, RI_MIN, RI_AMAX, RI_AMIN, RI_FIRST, RI_LAST
};

enum { K_SUM = 0, K_MAX = 1, K_MIN = 2, K_PROD = 3 };

__device__ __forceinline__ int red_kind(int f) {
  switch (f) {
    case R_NORMMAX: case R_MAX: case R_AMAX: case R_ANY: case R_ISINFORNAN: case R_ISNAN: case R_ISINF: return K_MAX;
    case R_MIN: case R_AMIN: case R_ALL: case R_ISFINITE: case R_ISPOS: case R_ISNEG: return K_MIN;
    case R_PROD: return K_PROD;
  }
  return K_SUM;
}
__device__ __forceinline__ double red_init(int kind) {
  return kind == K_MAX ? -INFINITY : kind == K_MIN ? INFINITY : kind == K_PROD ? 1.0 : 0.0;
}
__device__ __forceinline__ double red_merge(int kind, double a, double b) {
  return kind == K_SUM ? a + b : kind == K_MAX ? fmax(a, b) : kind == K_MIN ? fmin(a, b) : a * b;
}
__device__ __forceinline__ double red_pre(int f, double x, const double* e) {
  switch (f) {
    case R_MEAN: case R_SUM: case R_MAX: case R_MIN: case R_PROD: return x;
    case R_AMEAN: case R_NORM1: case R_ASUM: case R_NORMMAX: case R_AMAX: case R_AMIN: return fabs(x);
    case R_NORM2: case R_NORMFROB: case R_SQNORM: return x * x;
    case R_NORMP: return pow(fabs(x), e[0]);
    case R_ENTROPY: case R_LOGENTROPY: return x > 0.0 ? x * log(x) : 0.0;
    case R_SHANNON: { double p = x * x; return p > 0.0 ? p * log2(p) : 0.0; }
    case R_ANY: return x != 0.0 ? 1.0 : 0.0;
    case R_ALL: return x != 0.0 ? 1.0 : 0.0;
    case R_ISFINITE: return isfinite(x) ? 1.0 : 0.0;
    case R_ISINFORNAN: return isfinite(x) ? 0.0 : 1.0;
    case R_ISNAN: return isnan(x) ? 1.0 : 0.0;
    case R_ISINF: return isinf(x) ? 1.0 : 0.0;
    case R_ISPOS: return x > 0.0 ? 1.0 : 0.0;
    case R_ISNEG: return x < 0.0 ? 1.0 : 0.0;
    case R_COUNTNZ: return x != 0.0 ? 1.0 : 0.0;
    case R_COUNTZ: return x == 0.0 ? 1.0 : 0.0;
    case R_MATCH: return match_cond<double>(x, e[0], e[1], (int)e[2]) ? 1.0 : 0.0;
  }
  return x;
}
__device__ __forceinline__ double red_post(int f, double acc, double n, const double* e) {
  switch (f) {
    case R_MEAN: case R_AMEAN: return acc / n;
    case R_NORM2: case R_NORMFROB: return sqrt(acc);
    case R_NORMP: return pow(acc, 1.0 / e[0]);
    case R_ENTROPY: case R_SHANNON: return -acc;
    case R_LOGENTROPY: return log(-acc);
  }
  return acc;
}

__device__ __forceinline__ double wave_merge(int kind, double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = red_merge(kind, v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ int64_t tad_base(const int64_t* oshape, const int64_t* ostride, int orank, int64_t t) {
  int64_t off = 0;
  for (int d = orank - 1; d >= 0; --d) {
    off += (t % oshape[d]) * ostride[d];
    t /= oshape[d];
  }
  return off;
}
__device__ __forceinline__ int64_t inner_off(const int64_t* ishape, const int64_t* istride, int irank, int64_t j) {
  if (irank == 1) return j * istride[0];
  int64_t off = 0;
  for (int d = irank - 1; d >= 0; --d) {
    off += (j % ishape[d]) * istride[d];
    j /= ishape[d];
  }
  return off;
}

// group = 64 (one wave per TAD) or 256 (one block per TAD). segs = segments per TAD (two-stage when > 1).
// out: [num_tads * segs] accumulators (double) when segs > 1, else final values into z.
template <typename T, typename Z>
__global__ __launch_bounds__(256) void reduce_kernel(const T* __restrict__ x, Z* __restrict__ z, double* __restrict__ partial,
                                                     TadDesc d, int f, Extra ex, int group, int segs) {
  __shared__ double red[4];
  const int kind = red_kind(f);
  const int groups_per_block = 256 / group;
  const int g_in_block = threadIdx.x / group, lane = threadIdx.x % group;
  const int64_t total_groups = d.num_tads * segs;
  // group == 256 -> one group per block, so g (and the trip count) is block-uniform around the barriers
  for (int64_t g = (int64_t)blockIdx.x * groups_per_block + g_in_block; g < total_groups;
       g += (int64_t)gridDim.x * groups_per_block) {
    const bool live = g < total_groups;
    double acc = red_init(kind);
    int64_t t = 0, sidx = 0;
    if (live) {
      t = g / segs;
      sidx = g % segs;
      const int64_t seg_len = (d.tad_len + segs - 1) / segs;
      const int64_t j0 = sidx * seg_len, j1 = (j0 + seg_len < d.tad_len) ? j0 + seg_len : d.tad_len;
      const int64_t base = tad_base(d.oshape, d.ostride, d.orank, t);
      for (int64_t j = j0 + lane; j < j1; j += group) {
        double v = (double)ldv<T>(x, base + inner_off(d.ishape, d.istride, d.irank, j));
        acc = red_merge(kind, acc, red_pre(f, v, ex.e));
      }
    }
    acc = wave_merge(kind, acc);
    if (group == 256) {
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
      __syncthreads();
      acc = red_merge(kind, red_merge(kind, red[0], red[1]), red_merge(kind, red[2], red[3]));
    }
    if (live && lane == 0) {
      if (segs > 1) partial[g] = acc;
      else stv<Z, double>(z, t, red_post(f, acc, (double)d.tad_len, ex.e));
    }
  }
}

// one wave per TAD: lanes stride over the TAD's partial accumulators (contiguous: coalesced), wave_merge, lane 0 writes the result
template <typename Z>
__global__ void __launch_bounds__(256) reduce_finish_kernel(const double* __restrict__ partial, Z* __restrict__ z, int64_t num_tads, int segs,
                                                            int64_t tad_len, int f, Extra ex) {
  const int kind = red_kind(f);
  const int lane = threadIdx.x & 63;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < num_tads; t += (int64_t)gridDim.x * 4) {
    double acc = red_init(kind);
    int s = lane;
    // 8 loads in flight per lane: with 8k segments of one TAD a lane walks 128 partials, and one dependent load per step cost ~35 us
    for (; s + 7 * 64 < segs; s += 8 * 64) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[t * segs + s + u * 64];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = red_merge(kind, acc, v[u]);
    }
    for (; s < segs; s += 64) acc = red_merge(kind, acc, partial[t * segs + s]);
    acc = wave_merge(kind, acc);
    if (lane == 0) stv<Z, double>(z, t, red_post(f, acc, (double)tad_len, ex.e));
  }
}

// few TADs with many segments (a full reduction: 1 TAD x 8k partials): one 256-thread block per TAD instead of one wave
template <typename Z>
__global__ void __launch_bounds__(256) reduce_finish_block_kernel(const double* __restrict__ partial, Z* __restrict__ z, int64_t num_tads,
                                                                  int segs, int64_t tad_len, int f, Extra ex) {
  __shared__ double red[4];
  const int kind = red_kind(f);
  const int64_t t = blockIdx.x;
  if (t >= num_tads) return;
  double acc = red_init(kind);
  int s = threadIdx.x;
  for (; s + 7 * 256 < segs; s += 8 * 256) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[t * segs + s + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = red_merge(kind, acc, v[u]);
  }
  for (; s < segs; s += 256) acc = red_merge(kind, acc, partial[t * segs + s]);
  acc = wave_merge(kind, acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    stv<Z, double>(z, t, red_post(f, red_merge(kind, red_merge(kind, red[0], red[1]), red_merge(kind, red[2], red[3])), (double)tad_len, ex.e));
}

// Dense-TAD variant of reduce_kernel for fp32 / bf16: the reduced elements of a TAD are contiguous and every TAD starts on a 16-byte
// boundary, so a lane moves 16-byte vectors (same grouping / segment scheme, same double accumulators).
template <typename T, typename Z, int F>
__global__ __launch_bounds__(256) void reduce_vec_kernel(const T* __restrict__ x, Z* __restrict__ z, double* __restrict__ partial,
                                                         TadDesc d, int f_rt, Extra ex, int group, int segs) {
  constexpr int N = VecIO<T>::N;
  __shared__ double red[4];
  const int f = F >= 0 ? F : f_rt;
  const int kind = red_kind(f);
  const int groups_per_block = 256 / group;
  const int g_in_block = threadIdx.x / group, lane = threadIdx.x % group;
  const int64_t total_groups = d.num_tads * segs;
  const int64_t nvec = d.tad_len / N;
  for (int64_t g = (int64_t)blockIdx.x * groups_per_block + g_in_block; g < total_groups;
       g += (int64_t)gridDim.x * groups_per_block) {
    double acc = red_init(kind);
    const int64_t t = g / segs, sidx = g % segs;
    const int64_t seg_len = (nvec + segs - 1) / segs;
    const int64_t j0 = sidx * seg_len, j1 = (j0 + seg_len < nvec) ? j0 + seg_len : nvec;
    const T* base = x + tad_base(d.oshape, d.ostride, d.orank, t);
    int64_t j = j0 + lane;
    // four 16-byte loads in flight per lane and two accumulator chains: with one load per iteration the loop is bound by HBM latency
    // (8 dependent round trips per thread at 64M elements), not bandwidth
    double acc2 = red_init(kind);
    for (; j + 3 * (int64_t)group < j1; j += 4 * (int64_t)group) {
      float v0[N], v1[N], v2[N], v3[N];
      VecIO<T>::ld(base + j * N, v0);
      VecIO<T>::ld(base + (j + group) * N, v1);
      VecIO<T>::ld(base + (j + 2 * (int64_t)group) * N, v2);
      VecIO<T>::ld(base + (j + 3 * (int64_t)group) * N, v3);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        acc = red_merge(kind, acc, red_pre(f, (double)v0[k], ex.e));
        acc2 = red_merge(kind, acc2, red_pre(f, (double)v1[k], ex.e));
      }
#pragma unroll
      for (int k = 0; k < N; ++k) {
        acc = red_merge(kind, acc, red_pre(f, (double)v2[k], ex.e));
        acc2 = red_merge(kind, acc2, red_pre(f, (double)v3[k], ex.e));
      }
    }
    for (; j < j1; j += group) {
      float v[N];
      VecIO<T>::ld(base + j * N, v);
#pragma unroll
      for (int k = 0; k < N; ++k) acc = red_merge(kind, acc, red_pre(f, (double)v[k], ex.e));
    }
    acc = red_merge(kind, acc, acc2);
    acc = wave_merge(kind, acc);
    if (group == 256) {
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
      __syncthreads();
      acc = red_merge(kind, red_merge(kind, red[0], red[1]), red_merge(kind, red[2], red[3]));
    }
    if (lane == 0) {
      if (segs > 1) partial[g] = acc;
      else stv<Z, double>(z, t, red_post(f, acc, (double)d.tad_len, ex.e));
    }
  }
}

// Column reduction (the reduced dimension is the SLOW one: x viewed as [rows][cols], one result per column). A block owns 64 * CPT
// consecutive columns and a slab of rows; lanes run along the columns so every row is read with coalesced accesses, 4 row lanes per
// block, partial[col * segs + slab] -> reduce_finish_kernel. (One block per TAD would read a 2- or 4-byte element per 16 KB stride.)
template <typename T, int CPT, int F>
__global__ __launch_bounds__(256) void reduce_col_kernel(const T* __restrict__ x, double* __restrict__ partial, int64_t rows,
                                                         int64_t cols, int64_t pitch, int f_rt, Extra ex, int segs, int64_t rows_per_seg) {
  __shared__ double red[4][64 * CPT];
  const int f = F >= 0 ? F : f_rt;
  const int kind = red_kind(f);
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 64 + cl) * CPT;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_seg;
  const int64_t r1 = r0 + rows_per_seg < rows ? r0 + rows_per_seg : rows;
  double acc[CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) acc[k] = red_init(kind);
  if (c0 < cols) {
    int64_t r = r0 + rl;
    // four rows in flight per lane (loads first, then the accumulate chain): one row per iteration left the loop latency-bound
    for (; r + 12 < r1; r += 16) {
      double v[4][CPT];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const T* p = x + (r + 4 * u) * pitch + c0;
        if (std::is_same<T, bf16_t>::value && CPT == 2 && c0 + 1 < cols && (reinterpret_cast<uintptr_t>(p) & 3) == 0) {
          const uint32_t w = *reinterpret_cast<const uint32_t*>(p);        // both bf16 columns of the lane in one dword load
          v[u][0] = (double)__uint_as_float(w << 16);
          v[u][CPT - 1] = (double)__uint_as_float(w & 0xffff0000u);
        } else {
#pragma unroll
          for (int k = 0; k < CPT; ++k) v[u][k] = (c0 + k < cols) ? (double)ldv<T>(p, k) : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < CPT; ++k)
          if (c0 + k < cols) acc[k] = red_merge(kind, acc[k], red_pre(f, v[u][k], ex.e));
    }
    for (; r < r1; r += 4) {
      const T* p = x + r * pitch + c0;
#pragma unroll
      for (int k = 0; k < CPT; ++k)
        if (c0 + k < cols) acc[k] = red_merge(kind, acc[k], red_pre(f, (double)ldv<T>(p, k), ex.e));
    }
  }
#pragma unroll
  for (int k = 0; k < CPT; ++k) red[rl][cl * CPT + k] = acc[k];
  __syncthreads();
  for (int c = threadIdx.x; c < 64 * CPT; c += 256) {
    const int64_t col = (int64_t)blockIdx.x * 64 * CPT + c;
    if (col < cols)
      partial[col * segs + blockIdx.y] = red_merge(kind, red_merge(kind, red[0][c], red[1][c]), red_merge(kind, red[2][c], red[3][c]));
  }
}

// reduce3 / summarystats / indexreduce: one 256-thread block per TAD (three running values)
struct Tri {
  double a, b, c;
};
__device__ __forceinline__ Tri tri_merge(int f, Tri p, Tri q) {
  if (f == RS_VAR || f == RS_STD) {           // Welford/Chan merge of (n, mean, M2)
    if (q.a == 0.0) return p;
    if (p.a == 0.0) return q;
    double n = p.a + q.a, dl = q.b - p.b;
    return Tri{n, p.b + dl * q.a / n, p.c + q.c + dl * dl * p.a * q.a / n};
  }
  if (f >= RI_MAX) {                          // (value, index): keep better value, ties -> lower index
    bool take_q;
    if (q.b < 0.0) take_q = false;
    else if (p.b < 0.0) take_q = true;
    else if (f == RI_FIRST) take_q = q.b < p.b;
    else if (f == RI_LAST) take_q = q.b > p.b;
    else if (f == RI_MAX || f == RI_AMAX) take_q = q.a > p.a || (q.a == p.a && q.b < p.b);
    else take_q = q.a < p.a || (q.a == p.a && q.b < p.b);
    return take_q ? q : p;
  }
  return Tri{p.a + q.a, p.b + q.b, p.c + q.c};
}
__device__ __forceinline__ Tri tri_shfl(Tri v, int o) {
  return Tri{__shfl_xor(v.a, o, 64), __shfl_xor(v.b, o, 64), __shfl_xor(v.c, o, 64)};
}

__device__ __forceinline__ Tri tri_pre(int f, double a, double b, int64_t j, const Extra& ex) {
  switch (f) {
    case R3_MANHATTAN: return Tri{fabs(a - b), 0, 0};
    case R3_EUCLID: return Tri{(a - b) * (a - b), 0, 0};
    case R3_COSSIM: case R3_COSDIST: return Tri{a * b, a * a, b * b};
    case R3_DOT: return Tri{a * b, 0, 0};
    case R3_EQEPS: return Tri{fabs(a - b) > (ex.e[0] > 0 ? ex.e[0] : 1e-5) ? 1.0 : 0.0, 0, 0};
    case R3_JACCARD: return Tri{fmin(a, b), fmax(a, b), 0};
    case R3_HAMMING: return Tri{a != b ? 1.0 : 0.0, 0, 0};
    case RS_VAR: case RS_STD: return Tri{1.0, a, 0.0};
    case RI_MAX: case RI_MIN: return Tri{a, (double)j, 0};
    case RI_AMAX: case RI_AMIN: return Tri{fabs(a), (double)j, 0};
    default: return Tri{a, match_cond<double>(a, ex.e[0], ex.e[1], (int)ex.e[2]) ? (double)j : -1.0, 0};   // FIRST / LAST
  }
}
__device__ __forceinline__ Tri tri_init(int f) { return (f >= RI_MAX) ? Tri{0.0, -1.0, 0.0} : Tri{0.0, 0.0, 0.0}; }
__device__ __forceinline__ double tri_post(int f, Tri acc, double tad_len, const Extra& ex) {
  switch (f) {
    case R3_EUCLID: return sqrt(acc.a);
    case R3_COSSIM: return acc.a / (sqrt(acc.b) * sqrt(acc.c));
    case R3_COSDIST: return 1.0 - acc.a / (sqrt(acc.b) * sqrt(acc.c));
    case R3_JACCARD: return acc.b != 0.0 ? 1.0 - acc.a / acc.b : 0.0;
    case R3_HAMMING: return acc.a / tad_len;
    case RS_VAR: case RS_STD: {
      double den = ex.e[0] != 0.0 ? acc.a - 1.0 : acc.a;     // e[0] = biasCorrected
      double var = den > 0.0 ? acc.c / den : 0.0;
      return f == RS_STD ? sqrt(var) : var;
    }
    case RI_MAX: case RI_MIN: case RI_AMAX: case RI_AMIN: case RI_FIRST: case RI_LAST: return acc.b;
    default: return acc.a;
  }
}

// reduce3 / summarystats / indexreduce: a 256-thread block per (TAD, segment) with three running values. segs > 1 (few, long TADs —
// a full reduction of a large array used to run in ONE block) writes Tri partials that reduce_tri_finish_kernel merges; tri_merge is
// associative for every family (sums, Chan's variance merge, value/index pairs with index tie-breaks).
template <typename T, typename Z, int F>
__global__ __launch_bounds__(256) void reduce_tri_kernel(const T* __restrict__ x, const T* __restrict__ y, Z* __restrict__ z,
                                                         TadDesc d, int f_rt, Extra ex, int segs, double* __restrict__ partial) {
  __shared__ Tri red[4];
  const int f = F >= 0 ? F : f_rt;
  const int64_t total = d.num_tads * segs;
  const int64_t seg_len = (d.tad_len + segs - 1) / segs;
  for (int64_t g = blockIdx.x; g < total; g += gridDim.x) {
    const int64_t t = g / segs, sidx = g - t * segs;
    const int64_t j0 = sidx * seg_len, j1 = (j0 + seg_len < d.tad_len) ? j0 + seg_len : d.tad_len;
    const int64_t bx = tad_base(d.oshape, d.ostride, d.orank, t);
    const int64_t by = y ? tad_base(d.oshape, d.oystride, d.orank, t) : 0;
    Tri acc = tri_init(f);
    for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) {
      double a = (double)ldv<T>(x, bx + inner_off(d.ishape, d.istride, d.irank, j));
      double b = y ? (double)ldv<T>(y, by + inner_off(d.ishape, d.iystride, d.irank, j)) : 0.0;
      acc = tri_merge(f, acc, tri_pre(f, a, b, j, ex));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc = tri_merge(f, acc, tri_shfl(acc, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      acc = tri_merge(f, tri_merge(f, red[0], red[1]), tri_merge(f, red[2], red[3]));
      if (segs > 1) {
        partial[g * 3 + 0] = acc.a; partial[g * 3 + 1] = acc.b; partial[g * 3 + 2] = acc.c;
      } else {
        stv<Z, double>(z, t, tri_post(f, acc, (double)d.tad_len, ex));
      }
    }
  }
}

// one wave per TAD (see reduce_finish_kernel)
template <typename Z>
__global__ void __launch_bounds__(256) reduce_tri_finish_kernel(const double* __restrict__ partial, Z* __restrict__ z, int64_t num_tads,
                                                                int segs, int64_t tad_len, int f, Extra ex) {
  const int lane = threadIdx.x & 63;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < num_tads; t += (int64_t)gridDim.x * 4) {
    Tri acc = tri_init(f);
    for (int s = lane; s < segs; s += 64) {
      const double* p = partial + (t * segs + s) * 3;
      acc = tri_merge(f, acc, Tri{p[0], p[1], p[2]});
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc = tri_merge(f, acc, tri_shfl(acc, o));
    if (lane == 0) stv<Z, double>(z, t, tri_post(f, acc, (double)tad_len, ex));
  }
}

// ---- C API ------------------------------------------------------------------------------------------
// dtype: 0 f32, 1 bf16, 2 f64, 3 i32, 4 i64.  zkind: 0 = same as input, 1 = uint8 (bool), 2 = int64, 3 = f32, 4 = f64
#define DISPATCH_T(dtype, CALL)                      \
  switch (dtype) {                                   \
    case 0: { typedef float T; CALL; } break;        \
    case 1: { typedef bf16_t T; CALL; } break;       \
    case 2: { typedef double T; CALL; } break;       \
    case 3: { typedef int32_t T; CALL; } break;      \
    case 4: { typedef int64_t T; CALL; } break;      \
    default: return 1;                               \
  }

DL4J_API int dl4j_legacy_pairwise(int f, const void* x, const void* y, void* z, int64_t n, const Strides3* s, const double* extra,
                                  int use_scalar, double scalar, int dtype, int zbool, hipStream_t st) {
  if (n <= 0) return 0;
  Extra ex{{extra ? extra[0] : 0, extra ? extra[1] : 0, extra ? extra[2] : 0, extra ? extra[3] : 0}};
  if (!zbool && (dtype == 0 || dtype == 1) && aligned16(x) && aligned16(z) && (use_scalar || aligned16(y))) {
    const int vecn = dtype == 0 ? 4 : 8;
    int64_t inner = 1, mid = 1;
    int ymode = -1;
    if (n % vecn == 0) {
      if (use_scalar) ymode = s->contiguous ? YM_SCALAR : -1;
      else if (s->contiguous) ymode = YM_SAME;
      else ymode = classify_y(*s, vecn, inner, mid);
    }
    if (ymode >= 0) {
      const int64_t nvec = n / vecn;
      const int gv = launch_grid(LS_PAIRWISE, grid_for(nvec, 256 * 2));
      if (dtype == 0)
        launch_pairwise_vec<float>(gv, st, (const float*)x, (const float*)y, (float*)z, nvec, f, ex, ymode, (float)scalar, inner, mid);
      else
        launch_pairwise_vec<bf16_t>(gv, st, (const bf16_t*)x, (const bf16_t*)y, (bf16_t*)z, nvec, f, ex, ymode, (float)scalar, inner, mid);
      DL4J_RET_LAST();
    }
  }
  const int blk = launch_block(LS_PAIRWISE, 256);
  int g = launch_grid(LS_PAIRWISE, grid_for(n, blk * 4));
  if (zbool) {
    DISPATCH_T(dtype, (pairwise_kernel<T, uint8_t><<<g, blk, 0, st>>>((const T*)x, (const T*)y, (uint8_t*)z, n, f, *s, ex, use_scalar, scalar)))
  } else {
    DISPATCH_T(dtype, (pairwise_kernel<T, T><<<g, blk, 0, st>>>((const T*)x, (const T*)y, (T*)z, n, f, *s, ex, use_scalar, scalar)))
  }
  DL4J_RET_LAST();
}

DL4J_API int dl4j_legacy_transform(int f, const void* x, void* z, int64_t n, const Strides3* s, const double* extra, int dtype,
                                   int zbool, hipStream_t st) {
  if (n <= 0) return 0;
  Extra ex{{extra ? extra[0] : 0, extra ? extra[1] : 0, extra ? extra[2] : 0, extra ? extra[3] : 0}};
  if (!zbool && (dtype == 0 || dtype == 1) && s->contiguous && aligned16(x) && aligned16(z) && n % (dtype == 0 ? 4 : 8) == 0) {
    const int64_t nvec = n / (dtype == 0 ? 4 : 8);
    const int gv = launch_grid(LS_TRANSFORM, grid_for(nvec, 256 * 2));
    if (dtype == 0) launch_transform_vec<float>(gv, st, (const float*)x, (float*)z, nvec, f, ex);
    else launch_transform_vec<bf16_t>(gv, st, (const bf16_t*)x, (bf16_t*)z, nvec, f, ex);
    DL4J_RET_LAST();
  }
  const int blk = launch_block(LS_TRANSFORM, 256);
  int g = launch_grid(LS_TRANSFORM, grid_for(n, blk * 4));
  if (zbool) {
    DISPATCH_T(dtype, (transform_kernel<T, uint8_t><<<g, blk, 0, st>>>((const T*)x, (uint8_t*)z, n, f, *s, ex)))
  } else {
    DISPATCH_T(dtype, (transform_kernel<T, T><<<g, blk, 0, st>>>((const T*)x, (T*)z, n, f, *s, ex)))
  }
  DL4J_RET_LAST();
}

DL4J_API int64_t dl4j_legacy_reduce_segs(int64_t num_tads, int64_t tad_len) {
  // few, long TADs: split each into segments so that ~8k blocks (32 per CU: enough waves in flight to hide HBM latency behind
  // 16-byte loads) share the work; measured: a 64M-element full reduction went from 1.9 to >4 TB/s moving from 1k to 8k segments
  if (num_tads >= 2048 || tad_len < (1 << 15)) return 1;
  int64_t want = 8192 / num_tads;
  int64_t maxs = tad_len / (1 << 12);
  int64_t s = want < maxs ? want : maxs;
  return s < 1 ? 1 : s;
}

#define HOT_RED(X) X(R_SUM) X(R_MEAN) X(R_MAX) X(R_MIN) X(R_NORM2) X(R_NORM1) X(R_SQNORM) X(R_ASUM) X(R_AMAX) X(R_PROD)

template <typename T> struct HasVec { static constexpr bool value = false; };
template <> struct HasVec<float> { static constexpr bool value = true; };
template <> struct HasVec<bf16_t> { static constexpr bool value = true; };

// rows-per-slab choice of the column reduction: enough (column block, slab) pairs to fill the chip, slabs of at least 64 rows
static int col_segs(int64_t rows, int64_t col_blocks) {
  // 8 blocks per CU (a full complement of 2048 threads): the kernel is bound by the loads in flight, 4 per CU left bf16 at 2.6 TB/s
  int64_t want = (8 * DL4J_NUM_CU + col_blocks - 1) / col_blocks;
  const int64_t maxs = (rows + 63) / 64;
  if (want > maxs) want = maxs;
  if (want > 256) want = 256;
  return (int)(want < 1 ? 1 : want);
}

static bool col_path(const TadDesc& d) {
  return d.irank == 1 && d.orank == 1 && d.ostride[0] == 1 && d.istride[0] >= d.num_tads && d.tad_len >= 8 && d.num_tads >= 64;
}

// number of double accumulators dl4j_legacy_reduce needs in `partial` for this descriptor (0: none)
DL4J_API int64_t dl4j_legacy_reduce_partial_len(const TadDesc* d, int elem_size) {
  if (col_path(*d)) {
    const int cpt = elem_size == 2 ? 2 : 1;
    const int64_t col_blocks = (d->num_tads + 64 * cpt - 1) / (64 * cpt);
    return d->num_tads * col_segs(d->tad_len, col_blocks);
  }
  const int64_t segs = dl4j_legacy_reduce_segs(d->num_tads, d->tad_len);
  return segs > 1 ? d->num_tads * segs : 0;
}

template <typename T, typename Z>
static int run_reduce(const void* x, void* z, double* partial, const TadDesc& d, int f, const Extra& ex, hipStream_t st) {
  // column reduction: reduced dim strided, kept dim dense
  if (partial && col_path(d)) {
    constexpr int CPT = sizeof(T) == 2 ? 2 : 1;
    const int64_t col_blocks = (d.num_tads + 64 * CPT - 1) / (64 * CPT);
    const int segs = col_segs(d.tad_len, col_blocks);
    const int64_t rps = (((d.tad_len + segs - 1) / segs) + 3) / 4 * 4;
    const int gy = (int)((d.tad_len + rps - 1) / rps);
    if (col_blocks <= 65535 * 16) {
      const dim3 cg((unsigned)col_blocks, (unsigned)gy);
      switch (f) {
#define X(ID) case ID: reduce_col_kernel<T, CPT, ID><<<cg, 256, 0, st>>>((const T*)x, partial, d.tad_len, d.num_tads, d.istride[0], f, ex, gy, rps); break;
        HOT_RED(X)
#undef X
        default: reduce_col_kernel<T, CPT, -1><<<cg, 256, 0, st>>>((const T*)x, partial, d.tad_len, d.num_tads, d.istride[0], f, ex, gy, rps);
      }
      reduce_finish_kernel<Z><<<grid_for((d.num_tads + 3) / 4, 1), 256, 0, st>>>(partial, (Z*)z, d.num_tads, gy, d.tad_len, f, ex);
      return (int)hipGetLastError();
    }
  }
  int64_t segs = dl4j_legacy_reduce_segs(d.num_tads, d.tad_len);
  if (segs > 1 && !partial) segs = 1;
  int group = d.tad_len <= 512 ? 64 : 256;
  int64_t groups = d.num_tads * segs;
  int per_block = 256 / group;
  int g = grid_for((groups + per_block - 1) / per_block, 1);
  bool vec = false;
  if constexpr (HasVec<T>::value) {
    constexpr int N = VecIO<T>::N;
    vec = d.irank == 1 && d.istride[0] == 1 && d.tad_len % N == 0 && d.tad_len >= 4 * N * group && aligned16(x);
    for (int k = 0; vec && k < d.orank; ++k) vec = d.ostride[k] % N == 0;
    if (vec && groups % per_block == 0 && (int64_t)g * per_block <= groups) {
      switch (f) {
#define X(ID) case ID: reduce_vec_kernel<T, Z, ID><<<g, 256, 0, st>>>((const T*)x, (Z*)z, partial, d, f, ex, group, (int)segs); break;
        HOT_RED(X)
#undef X
        default: reduce_vec_kernel<T, Z, -1><<<g, 256, 0, st>>>((const T*)x, (Z*)z, partial, d, f, ex, group, (int)segs);
      }
    } else {
      vec = false;
    }
  }
  if (!vec) reduce_kernel<T, Z><<<g, 256, 0, st>>>((const T*)x, (Z*)z, partial, d, f, ex, group, (int)segs);
  if (segs > 1)
    if (d.num_tads <= 64 && segs >= 1024)
      reduce_finish_block_kernel<Z><<<(unsigned)d.num_tads, 256, 0, st>>>(partial, (Z*)z, d.num_tads, (int)segs, d.tad_len, f, ex);
    else
      reduce_finish_kernel<Z><<<grid_for((d.num_tads + 3) / 4, 1), 256, 0, st>>>(partial, (Z*)z, d.num_tads, (int)segs, d.tad_len, f, ex);
  return (int)hipGetLastError();
}

// zkind: 0 same dtype as x, 1 uint8, 2 int64, 3 f32, 4 f64.  partial: >= num_tads*segs doubles (or null)
DL4J_API int dl4j_legacy_reduce(int f, const void* x, void* z, double* partial, const TadDesc* d, const double* extra, int dtype,
                                int zkind, hipStream_t st) {
  if (d->num_tads <= 0) return 0;
  Extra ex{{extra ? extra[0] : 0, extra ? extra[1] : 0, extra ? extra[2] : 0, extra ? extra[3] : 0}};
  switch (zkind) {
    case 0: DISPATCH_T(dtype, return (run_reduce<T, T>(x, z, partial, *d, f, ex, st))) break;
    case 1: DISPATCH_T(dtype, return (run_reduce<T, uint8_t>(x, z, partial, *d, f, ex, st))) break;
    case 2: DISPATCH_T(dtype, return (run_reduce<T, int64_t>(x, z, partial, *d, f, ex, st))) break;
    case 3: DISPATCH_T(dtype, return (run_reduce<T, float>(x, z, partial, *d, f, ex, st))) break;
    case 4: DISPATCH_T(dtype, return (run_reduce<T, double>(x, z, partial, *d, f, ex, st))) break;
  }
  return 1;
}

template <typename T, typename Z>
static int run_tri(const void* x, const void* y, void* z, double* partial, const TadDesc& d, int f, const Extra& ex, hipStream_t st) {
  int64_t segs = partial ? dl4j_legacy_reduce_segs(d.num_tads, d.tad_len) : 1;
  if (segs < 1) segs = 1;
  const int g = grid_for(d.num_tads * segs, 1);
  switch (f) {
#define X(ID) case ID: reduce_tri_kernel<T, Z, ID><<<g, 256, 0, st>>>((const T*)x, (const T*)y, (Z*)z, d, f, ex, (int)segs, partial); break;
    X(R3_COSSIM) X(R3_DOT) X(R3_EUCLID) X(R3_MANHATTAN) X(RS_VAR) X(RS_STD) X(RI_MAX) X(RI_MIN)
#undef X
    default: reduce_tri_kernel<T, Z, -1><<<g, 256, 0, st>>>((const T*)x, (const T*)y, (Z*)z, d, f, ex, (int)segs, partial);
  }
  if (segs > 1)
    reduce_tri_finish_kernel<Z><<<grid_for((d.num_tads + 3) / 4, 1), 256, 0, st>>>(partial, (Z*)z, d.num_tads, (int)segs, d.tad_len, f, ex);
  return (int)hipGetLastError();
}

// partial: >= 3 * num_tads * dl4j_legacy_reduce_segs(num_tads, tad_len) doubles, or null (one block per TAD)
DL4J_API int dl4j_legacy_reduce_tri_p(int f, const void* x, const void* y, void* z, double* partial, const TadDesc* d,
                                      const double* extra, int dtype, int zkind, hipStream_t st) {
  if (d->num_tads <= 0) return 0;
  Extra ex{{extra ? extra[0] : 0, extra ? extra[1] : 0, extra ? extra[2] : 0, extra ? extra[3] : 0}};
  switch (zkind) {
    case 0: DISPATCH_T(dtype, return (run_tri<T, T>(x, y, z, partial, *d, f, ex, st))) break;
    case 2: DISPATCH_T(dtype, return (run_tri<T, int64_t>(x, y, z, partial, *d, f, ex, st))) break;
    case 3: DISPATCH_T(dtype, return (run_tri<T, float>(x, y, z, partial, *d, f, ex, st))) break;
    case 4: DISPATCH_T(dtype, return (run_tri<T, double>(x, y, z, partial, *d, f, ex, st))) break;
  }
  return 1;
}

DL4J_API int dl4j_legacy_reduce_tri(int f, const void* x, const void* y, void* z, const TadDesc* d, const double* extra, int dtype,
                                    int zkind, hipStream_t st) {
  return dl4j_legacy_reduce_tri_p(f, x, y, z, nullptr, d, extra, dtype, zkind, st);
}
