// Fused GRU sequence kernels for gfx950 (bf16 activations / weights, fp32 gate math): ONE launch per time step does the recurrent
// GEMM of all three gates, the gate non-linearities and the state blend - forward - and the recurrent gradient GEMM plus the whole
// cell backward - backward. Between launches only h_t (and, backwards, the pre-activation gradients) exist in memory.
//
// Reference: libnd4j's gru layer (ops/declarable/helpers/impl/gru.cpp:123-160, gruTimeLoop -> gruCell per step: mmul(h, Wh), ~10
// elementwise transforms, 3 temporaries per step) and its gruCellBp (:162-380, four more mmuls per step). The previous form here
// (rnn_seq.hip: MFMA GEMM launch + cell launch per step) was launch-bound at [mb 64..128, n 512]: two launches and a [mb, 3n]
// round trip per step. The decomposition below is the one of lstm_persist.hip's step kernels with three gates instead of four.
//
// Gate order [r | u | c] (libnd4j): r = sigmoid(xz_r + hz_r), u = sigmoid(xz_u + hz_u), c = tanh(xz_c + r * hz_c),
// h = u * h_prev + (1 - u) * c, with xz = x . Wx + b (one big GEMM outside) and hz = h_prev . Wh.
//
// Forward step: a workgroup owns 8 hidden units = 24 gate columns {g * n + 8w + k} (rows 24..31 of the 32-row MFMA tile are fed
// zeros) x 32 batch rows; its waves split K; v_mfma_f32_32x32x16_bf16 with A = the gate rows of Wh^T [3n, n]
// (K-contiguous) and B = h_{t-1} rows, both straight from L2 into registers. In the accumulator layout lane l holds, for batch row
// l % 32, the three gates (register group = gate) of units 4 * (l / 32) + j, so the cell runs in registers and h_t leaves as one
// 8-byte store per lane. Waves 1.. hand their partial sums to wave 0 through LDS.
//
// Backward step: dh_t = E_t + u_{t+1} * dh_{t+1} (carried in fp32) + dHZ_{t+1} . Wh^T (v_mfma_f32_16x16x32_bf16, K = 3n), then the
// cell backward in registers writes dXZ_t (gradient of the input projection, = of x . Wx + b) and dHZ_t (gradient of h_{t-1} . Wh).
// Weight gradients are two large GEMMs over all steps afterwards (dWh = Hprev^T . dHZ, dWx = X^T . dXZ).
//
// Host-checked requirements: bf16, n % 128 == 0 (whole MFMA K-steps per K quarter in both directions), mb <= 128, 16-byte aligned
// operands. Everything else stays on the generic loop.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float g_sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float g_tanh(float x) {
  const float e = __expf(-2.f * fabsf(x));
  const float t = (1.f - e) / (1.f + e);
  return x < 0.f ? -t : t;
}
__device__ __forceinline__ uint64_t g_pack4(float a, float b, float c, float d) {
  return (uint64_t)f2bf(a) | ((uint64_t)f2bf(b) << 16) | ((uint64_t)f2bf(c) << 32) | ((uint64_t)f2bf(d) << 48);
}
__device__ __forceinline__ void g_unpack4(uint64_t v, float o[4]) {
  o[0] = bf2f((bf16_t)(v & 0xffff)); o[1] = bf2f((bf16_t)((v >> 16) & 0xffff));
  o[2] = bf2f((bf16_t)((v >> 32) & 0xffff)); o[3] = bf2f((bf16_t)(v >> 48));
}


// element-type traits (as lstm_persist.hip's DiagBf16 / DiagF32): bf16 on MFMA 32x32x16 / 16x16x32, fp32 on 32x32x2 / 16x16x4 with one
// 16-byte load (4 K values per lane) feeding 4 MFMAs
struct GruBf16 {
  typedef bf16_t T;
  typedef bf16x8_t vec;
  static constexpr int FSTEP = 16, FLANE = 8, BSTEP = 32, BLANE = 8;
  static __device__ __forceinline__ void mfma_f(f32x16& acc, const vec& a, const vec& b) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); }
  static __device__ __forceinline__ void mfma_b(f32x4& acc, const vec& a, const vec& b) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); }
  static __device__ __forceinline__ void ld4(const T* p, float o[4]) { g_unpack4(*(const uint64_t*)p, o); }
  static __device__ __forceinline__ void st4(T* p, float a, float b, float c, float d) { *(uint64_t*)p = g_pack4(a, b, c, d); }
};
struct GruF32 {
  typedef float T;
  typedef f32x4 vec;
  static constexpr int FSTEP = 8, FLANE = 4, BSTEP = 16, BLANE = 4;
  static __device__ __forceinline__ void mfma_f(f32x16& acc, const vec& a, const vec& b) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], b[c], acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void mfma_b(f32x4& acc, const vec& a, const vec& b) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[c], acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void ld4(const T* p, float o[4]) {
    const f32x4 v = *(const f32x4*)p;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
  static __device__ __forceinline__ void st4(T* p, float a, float b, float c, float d) { *(f32x4*)p = f32x4{a, b, c, d}; }
};

template <typename E>
struct GruFwdP {
  const E* xz;        // [mb, 3n] input projection + bias of THIS step
  const E* wh_t;      // [3n, ld] recurrent weights transposed: row = gate column, K = n contiguous
  int64_t ld;
  const E* hprev;     // [mb, n]
  E* H;               // [mb, n]
  E* SV;              // [mb, 4n] saved r, u, c, hz_c - or null (inference)
  int mb, n;
};

// Decomposition (both kernels): ONE MFMA tile per workgroup, its K range split over the GRU_WAVES waves in whole k-steps, partial
// tiles summed through LDS by wave 0, which runs the cell. These kernels are bound by operand bytes per CU (every byte comes from L2);
// the first form (8 units x all batch rows per workgroup, 16 waves = K quarter x batch tile, backward K loop left rolled by the
// compiler into load, load, wait, MFMA) moved three times the bytes per workgroup - the same finding as lstm_persist.hip's
// lstm_diag_*_kernel, measured in profiles/r3_lstm_layer_microbench.md.
#define GRU_WAVES 8
#define GRU_LOG2W 3
#define GRU_THREADS (64 * GRU_WAVES)
#define GRU_KSTEPS 8         // k-steps per wave held in registers at once (n <= 1024 forward, n <= 680 backward in one pass)

// Forward: workgroup = hidden units [8 bx, 8 bx + 8) = 24 gate rows {g * n + 8 bx + k} (rows 24..31 of the 32-row MFMA tile are fed
// zeros) x batch rows [32 by, 32 by + 32).
template <typename TR>
__global__ void __launch_bounds__(GRU_THREADS) gru_step_fwd_kernel(GruFwdP<typename TR::T> p) {
  typedef typename TR::T T;
  typedef typename TR::vec vec;
  // partial tiles as [wave][4 register groups][64 lanes] of 16 bytes: a lane stride of 16 bytes keeps the 128-bit LDS accesses free of
  // bank conflicts (as [wave][lane] of 64 bytes they ran at 6x the conflict-free cycle count, profiles/r3_pmc_sq_lstm.md - though
  // the hand-off is short enough that the step time did not change)
  __shared__ __attribute__((aligned(16))) f32x4 red[(GRU_WAVES - 1) * 4 * 64];
  const int n = p.n, mb = p.mb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 8;
  const int row = blockIdx.y * 32 + (lane & 31);
  const int kq = lane >> 5;
  const int uq = u0 + kq * 4;
  const bool row_ok = row < mb;
  const bool cell_lane = wave == 0 && row_ok;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  {
    const int gc = lane & 31;                                            // row of the MFMA tile: gate * 8 + unit; gate 3 is padding
    const bool a_on = gc < 24;
    const T* ap = p.wh_t + ((int64_t)(a_on ? (gc >> 3) : 0) * n + u0 + (gc & 7)) * p.ld + kq * TR::FLANE;
    const T* bp = p.hprev + (int64_t)(row_ok ? row : 0) * n + kq * TR::FLANE;
    const vec zero = {};
    const int S = n / TR::FSTEP, s0 = (wave * S) >> GRU_LOG2W, s1 = ((wave + 1) * S) >> GRU_LOG2W;
    for (int sb = s0; sb < s1; sb += GRU_KSTEPS) {
      vec a[GRU_KSTEPS], b[GRU_KSTEPS];
#pragma unroll
      for (int i = 0; i < GRU_KSTEPS; ++i) {
        const int k = (sb + i) * TR::FSTEP;
        a[i] = (sb + i < s1 && a_on) ? *(const vec*)(ap + k) : zero;
        b[i] = (sb + i < s1 && row_ok) ? *(const vec*)(bp + k) : zero;
      }
#pragma unroll
      for (int i = 0; i < GRU_KSTEPS; ++i) TR::mfma_f(acc, a[i], b[i]);
    }
  }
  // wave 0 owns the cell: its operands travel while the other waves hand their partial tiles over
  float fr[4] = {0, 0, 0, 0}, fu[4] = {0, 0, 0, 0}, fc[4] = {0, 0, 0, 0}, hp[4] = {0, 0, 0, 0};
  if (cell_lane) {
    const T* xp = p.xz + (int64_t)row * 3 * n + uq;
    TR::ld4(xp, fr); TR::ld4(xp + n, fu); TR::ld4(xp + 2 * (int64_t)n, fc);
    TR::ld4(p.hprev + (int64_t)row * n + uq, hp);
  }
  if (wave > 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) red[((wave - 1) * 4 + g) * 64 + lane] = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
  }
  __syncthreads();
  if (!cell_lane) return;
#pragma unroll 3
  for (int w = 0; w < GRU_WAVES - 1; ++w) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 part = red[(w * 4 + g) * 64 + lane];
      acc[4 * g] += part[0]; acc[4 * g + 1] += part[1]; acc[4 * g + 2] += part[2]; acc[4 * g + 3] += part[3];
    }
  }
  float hn[4], vr[4], vu[4], vc[4], vh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float r = g_sigm(fr[j] + acc[j]);
    const float u = g_sigm(fu[j] + acc[4 + j]);
    const float hc = acc[8 + j];
    const float c = g_tanh(fc[j] + r * hc);
    vr[j] = r; vu[j] = u; vc[j] = c; vh[j] = hc;
    hn[j] = u * hp[j] + (1.f - u) * c;
  }
  TR::st4(p.H + (int64_t)row * n + uq, hn[0], hn[1], hn[2], hn[3]);
  if (p.SV) {
    T* sv = p.SV + (int64_t)row * 4 * n + uq;
    TR::st4(sv, vr[0], vr[1], vr[2], vr[3]);
    TR::st4(sv + n, vu[0], vu[1], vu[2], vu[3]);
    TR::st4(sv + 2 * (int64_t)n, vc[0], vc[1], vc[2], vc[3]);
    TR::st4(sv + 3 * (int64_t)n, vh[0], vh[1], vh[2], vh[3]);
  }
}

template <typename EL>
struct GruBwdP {
  const EL* E;         // [mb, n] dL/dh_t from the layers above (this step)
  const EL* wh;        // [n, ld] row j = the 3n gate weights leaving hidden unit j (K = 3n contiguous)
  int64_t ld;
  const EL* SV;        // [mb, 4n] of this step
  const EL* hprev;     // [mb, n]
  const EL* dhz_next;  // [mb, 3n] gradient of hz of the step processed before this one, or null
  const float* dhp_in;     // [mb, n] fp32 u_{t+1} * dh_{t+1}, or null
  EL* dxz;             // [mb, 3n]
  EL* dhz;             // [mb, 3n]
  float* dhp_out;          // [mb, n] fp32
  int mb, n;
};

// Backward: workgroup = hidden units [16 bx, 16 bx + 16) x batch rows [16 by, 16 by + 16): one 16x16 MFMA tile, K = 3n.
template <typename TR>
__global__ void __launch_bounds__(GRU_THREADS) gru_step_bwd_kernel(GruBwdP<typename TR::T> p) {
  typedef typename TR::T T;
  typedef typename TR::vec vec;
  __shared__ __attribute__((aligned(16))) f32x4 red[(GRU_WAVES - 1) * 64];
  const int n = p.n, mb = p.mb, K = 3 * p.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 16, r0 = blockIdx.y * 16;
  const int bcol = lane & 15, q = lane >> 4;
  const int row = r0 + bcol;
  const bool row_ok = row < mb;
  const int uq = u0 + q * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (p.dhz_next != nullptr) {
    const T* ap = p.wh + (int64_t)(u0 + bcol) * p.ld + q * TR::BLANE;
    const T* bp = p.dhz_next + (int64_t)(row_ok ? row : 0) * K + q * TR::BLANE;
    const vec zero = {};
    const int S = K / TR::BSTEP, s0 = (wave * S) >> GRU_LOG2W, s1 = ((wave + 1) * S) >> GRU_LOG2W;      // 32-wide k-steps of this wave
    for (int sb = s0; sb < s1; sb += GRU_KSTEPS) {
      vec a[GRU_KSTEPS], b[GRU_KSTEPS];
#pragma unroll
      for (int i = 0; i < GRU_KSTEPS; ++i) {
        const int k = (sb + i) * TR::BSTEP;
        a[i] = sb + i < s1 ? *(const vec*)(ap + k) : zero;
        b[i] = (sb + i < s1 && row_ok) ? *(const vec*)(bp + k) : zero;
      }
#pragma unroll
      for (int i = 0; i < GRU_KSTEPS; ++i) TR::mfma_b(acc, a[i], b[i]);
    }
  }
  float e4[4] = {0, 0, 0, 0}, r4[4] = {0, 0, 0, 0}, u4[4] = {0, 0, 0, 0}, c4[4] = {0, 0, 0, 0}, h4[4] = {0, 0, 0, 0}, hp4[4] = {0, 0, 0, 0};
  f32x4 carry = {0.f, 0.f, 0.f, 0.f};
  const bool cell_lane = wave == 0 && row_ok;
  if (cell_lane) {
    const int64_t hidx = (int64_t)row * n + uq;
    TR::ld4(p.E + hidx, e4);
    const T* sv = p.SV + (int64_t)row * 4 * n + uq;
    TR::ld4(sv, r4); TR::ld4(sv + n, u4); TR::ld4(sv + 2 * (int64_t)n, c4); TR::ld4(sv + 3 * (int64_t)n, h4);
    TR::ld4(p.hprev + hidx, hp4);
    if (p.dhp_in) carry = *(const f32x4*)(p.dhp_in + hidx);
  }
  if (p.dhz_next != nullptr && wave > 0) red[(wave - 1) * 64 + lane] = acc;
  __syncthreads();
  if (!cell_lane) return;
  if (p.dhz_next != nullptr) {
#pragma unroll
    for (int w = 0; w < GRU_WAVES - 1; ++w) {
      const f32x4 part = red[w * 64 + lane];
      acc += part;
    }
  }
  float dr[4], du[4], dc[4], dcr[4];
  f32x4 dpo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float d = e4[j] + acc[j] + carry[j];
    const float r = r4[j], u = u4[j], c = c4[j];
    dc[j] = d * (1.f - u) * (1.f - c * c);
    du[j] = d * (hp4[j] - c) * u * (1.f - u);
    dr[j] = dc[j] * h4[j] * r * (1.f - r);
    dcr[j] = dc[j] * r;
    dpo[j] = d * u;
  }
  const int64_t hidx = (int64_t)row * n + uq;
  *(f32x4*)(p.dhp_out + hidx) = dpo;
  T* gx = p.dxz + (int64_t)row * K + uq;
  T* gh = p.dhz + (int64_t)row * K + uq;
  TR::st4(gx, dr[0], dr[1], dr[2], dr[3]); TR::st4(gx + n, du[0], du[1], du[2], du[3]); TR::st4(gx + 2 * (int64_t)n, dc[0], dc[1], dc[2], dc[3]);
  TR::st4(gh, dr[0], dr[1], dr[2], dr[3]); TR::st4(gh + n, du[0], du[1], du[2], du[3]); TR::st4(gh + 2 * (int64_t)n, dcr[0], dcr[1], dcr[2], dcr[3]);
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

DL4J_API int dl4j_gru_steps_supported(int mb, int n) { return (mb > 0 && mb <= 128 && n > 0 && n % 128 == 0 && n / 8 <= 4096) ? 1 : 0; }

template <typename TR>
static int gru_fwd_t(const void* xz, const void* wh_t, int64_t ld, const void* h0, void* H, void* SV, int T, int mb, int n, hipStream_t st) {
  typedef typename TR::T E;
  const size_t hn = (size_t)mb * n;
  const E* hp = (const E*)h0;
  for (int t = 0; t < T; ++t) {
    GruFwdP<E> p{(const E*)xz + 3 * t * hn, (const E*)wh_t, ld, hp, (E*)H + t * hn, SV ? (E*)SV + 4 * t * hn : nullptr, mb, n};
    hipLaunchKernelGGL((gru_step_fwd_kernel<TR>), dim3(n / 8, (mb + 31) / 32), dim3(GRU_THREADS), 0, st, p);
    hp = (const E*)H + t * hn;
  }
  DL4J_RET_LAST();
}

// xz [T, mb, 3n], wh_t [3n, ld], h0 [mb, n] -> H [T, mb, n], SV [T, mb, 4n] (null for inference). dtype: 0 = bf16, 1 = fp32.
DL4J_API int dl4j_gru_steps_fwd(const void* xz, const void* wh_t, int64_t ld, const void* h0, void* H, void* SV, int T, int mb, int n,
                                int dtype, hipStream_t st) {
  if (T <= 0) return 0;
  if (!xz || !wh_t || !h0 || !H) return -3;
  if (!dl4j_gru_steps_supported(mb, n) || ld < n || ld % 8 != 0 || !al16(xz) || !al16(wh_t) || !al16(h0) || !al16(H) || (SV && !al16(SV)))
    return -2;
  if (dtype == 0) return gru_fwd_t<GruBf16>(xz, wh_t, ld, h0, H, SV, T, mb, n, st);
  if (dtype == 1) return gru_fwd_t<GruF32>(xz, wh_t, ld, h0, H, SV, T, mb, n, st);
  return -2;
}

template <typename TR>
static int gru_bwd_t(const void* Eg, const void* wh, int64_t ld, const void* SV, const void* H, const void* h0, void* dXZ, void* dHZ,
                     float* carry, int T, int mb, int n, hipStream_t st) {
  typedef typename TR::T E;
  const size_t hn = (size_t)mb * n;
  int last = 0;
  for (int t = T - 1; t >= 0; --t) {
    const int half = (T - 1 - t) & 1;
    GruBwdP<E> p{(const E*)Eg + t * hn, (const E*)wh, ld, (const E*)SV + 4 * t * hn, t > 0 ? (const E*)H + (t - 1) * hn : (const E*)h0,
                 t == T - 1 ? nullptr : (const E*)dHZ + 3 * (t + 1) * hn, t == T - 1 ? nullptr : carry + (size_t)(half ^ 1) * hn,
                 (E*)dXZ + 3 * t * hn, (E*)dHZ + 3 * t * hn, carry + (size_t)half * hn, mb, n};
    hipLaunchKernelGGL((gru_step_bwd_kernel<TR>), dim3(n / 16, (mb + 15) / 16), dim3(GRU_THREADS), 0, st, p);
    last = half;
  }
  const int rc = (int)hipGetLastError();
  return rc ? -100 - rc : last;
}

// E [T, mb, n] (dL/dH), wh [n, ld] (ld >= 3n), SV [T, mb, 4n], H [T, mb, n], h0 [mb, n] -> dXZ, dHZ [T, mb, 3n]; carry [2, mb, n] fp32
// scratch: on return one half of it holds u_0 * dh_0, the direct part of dL/dh0 (the caller adds dHZ_0 . Wh^T).
// Returns the index (0 / 1) of the carry half that holds it, or a negative error code. dtype: 0 = bf16, 1 = fp32.
DL4J_API int dl4j_gru_steps_bwd(const void* E, const void* wh, int64_t ld, const void* SV, const void* H, const void* h0, void* dXZ, void* dHZ,
                                float* carry, int T, int mb, int n, int dtype, hipStream_t st) {
  if (T <= 0) return 0;
  if (!E || !wh || !SV || !H || !h0 || !dXZ || !dHZ || !carry) return -3;
  if (!dl4j_gru_steps_supported(mb, n) || ld < 3 * (int64_t)n || ld % 8 != 0 || !al16(E) || !al16(wh) || !al16(SV) || !al16(H) || !al16(h0) ||
      !al16(dXZ) || !al16(dHZ) || !al16(carry))
    return -2;
  if (dtype == 0) return gru_bwd_t<GruBf16>(E, wh, ld, SV, H, h0, dXZ, dHZ, carry, T, mb, n, st);
  if (dtype == 1) return gru_bwd_t<GruF32>(E, wh, ld, SV, H, h0, dXZ, dHZ, carry, T, mb, n, st);
  return -2;
}
