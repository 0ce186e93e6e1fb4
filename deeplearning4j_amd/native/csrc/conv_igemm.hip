// Implicit-GEMM 2-D convolution for gfx950 (MI355X / CDNA4): bf16 NHWC activations, OHWI filters,
// fp32 accumulation on the matrix cores (v_mfma_f32_32x32x16_bf16), no im2col buffer.
//
// The reference conv2d is im2col -> cuBLAS GEMM -> permute -> addBias with a materialised column
// buffer and a stream sync per GEMM (libnd4j/include/ops/declarable/helpers/cuda/convolutions_conv2d.cu:78-100,
// helpers/cuda_off/MmulHelper.cu:347); backprop recomputes im2col (ConvolutionLayer.java:132-138).
// Here:
//   * forward  : C[M=N*OH*OW, K] = A[M, R*S*C] x B[K, R*S*C]^T with A gathered on the fly (zero
//                fill for padding). Epilogue fuses bias, bf16 rounding, and — for a following
//                BatchNorm — the per-channel sum / sum-of-squares partials (one pass over y saved).
//   * dgrad    : the same kernel on dy with spatially flipped, transposed filters ([C][R][S][K]);
//                1x1 stride-s convs scatter rows into a pre-zeroed dx.
//   * wgrad    : dW[K, R*S*C] += dy[P, K]^T x gather(x)[P, C] per filter tap; both operands have the
//                reduction (pixel) index as the slow memory index, so tiles are staged row-major in
//                LDS and read with the hardware transposing load ds_read_b64_tr_b16. Split over
//                pixels across workgroups, fp32 atomics into the (pre-zeroed) gradient.
// Tiling: 256 threads = 4 waves (2x2), wave tile (BM/2)x(BN/2) of 32x32 MFMA blocks, BK = 64,
// double-buffered LDS with register staging (global loads for tile t+1 issued before the MFMAs of
// tile t, LDS write after), XOR-swizzled 16-byte chunks so ds_read_b128 / tr reads are bank
// conflict free, XCD-aware workgroup->tile map so tiles sharing an A panel share an L2.
#include <type_traits>
#include "common.h"
#include "ticket.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// Division by a launch-invariant divisor (Granlund-Montgomery): q = (t + ((n - t) >> s1)) >> s2 with t = mulhi(n, mul); exact for every
// 32-bit n. The pixel -> (image, row, column) decompositions of the loaders and epilogues are 2 divisions per tile row: as runtime
// divisions they were ~25 VALU instructions each, a large part of the instruction count of the short-reduction layers.
struct FastDiv {
  uint32_t mul, s1, s2;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.mul = (uint32_t)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l < 1 ? 0 : l - 1;
  return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
  const uint32_t un = (uint32_t)n, t = __umulhi(un, f.mul);
  return (int)((t + ((un - t) >> f.s1)) >> f.s2);
}

struct IGemmP {
  const bf16_t* x;
  const bf16_t* w;
  bf16_t* y;
  const float* bias;
  float* stats;       // [mtiles][2][K] partial sums (or null)
  int N, H, W, C;     // input tensor
  int OH, OW, K;      // logical output pixels grid / out channels
  int R, S, sh, sw, ph, pw, dh, dw;
  int OHs, OWs, oys, oxs;  // physical output tensor spatial dims and pixel strides (scatter for strided dgrad)
  int M;              // N*OH*OW
  int mtiles, ntiles;
  int nbuf;           // LDS stages: 2 (double buffered) or 1 when the reduction is a single 64-wide tile (1x1 conv, C = 64):
                      // halves the LDS footprint so twice as many workgroups are resident to hide the load latency
  int accum;          // 1: y += result (in-place gradient accumulation at graph fan-out points, saves a separate add pass)
  ticket::BnFin fin;  // fin.cnt != null: the last workgroup of every N tile reduces the statistics and runs the BatchNorm finalize
  // EPI == 1 (data-gradient launches): the output tile is the gradient flowing INTO a BatchNorm(+ReLU) whose input was `bnb_y`.
  // The epilogue (a) optionally adds another gradient for the same tensor, masked by the ReLU bits of the block it came through
  // (identity shortcut: dz = d_out * [out > 0] is never materialised), and (b) emits the per-tile partial sums of the BatchNorm
  // backward reductions (sum dz, sum dz * xhat), so the separate column-reduction pass over dy and x disappears.
  const bf16_t* acc_src;     // gradient to add (same physical layout as y), or null
  const uint8_t* acc_bits;   // one byte per 8 channels: bit j set = element passes; null = plain add
  const bf16_t* bnb_y;       // BatchNorm input (raw producer output), same layout as y; null = no partial sums
  const float* bnb_mean; const float* bnb_invstd;
  const float* bnb_ms; const float* bnb_mt;   // ReLU mask recomputed as [y*ms + mt > 0] (BatchNorm+ReLU without residual), or null
  const uint8_t* bnb_bits;   // ReLU mask bits saved by the forward apply (residual blocks), or null
  float* bnb_partial;        // [mtiles][2][K]
  int bnb_mode;              // 0: no mask, 1: bnb_bits, 2: bnb_ms / bnb_mt (set by the host entry point)
  // statistics pivot: the partial sums are of (y - pivot) with pivot[c] = piv_mean[c] - piv_bias[c] (the consuming BatchNorm's
  // running mean, minus the folded conv bias): E[(y-p)^2] - E[y-p]^2 does not cancel when |mean| >> std. Null = no shift.
  const float* piv_mean; const float* piv_bias;
  FastDiv d_ohw, d_ow;      // divisions by OH*OW and OW (set in launch_fwd)
};

// 256 bytes of zeros: out-of-range operand pieces (padding taps, rows past M, channel tails) are LOADED from here. The loaders are
// straight-line code - per-lane source select, unconditional 16-byte load. Written as "v = 0; if (ok) v = load" every load sat in its
// own EXEC-masked block behind a zero initialisation of its destination registers, and the compiler guarded those writes with
// s_waitcnt vmcnt(1): the first loads of a k-tile had to RETURN before the remaining ones were issued (two exposed round trips per
// k-tile instead of one).
static __device__ __attribute__((aligned(256))) unsigned char g_conv_zero_page[256];

// Loads the compiler must not see.  The two-stage loaders below keep a k-tile's loads in flight across a whole loop iteration, which the
// compiler's wait-count bookkeeping cannot express (it drained the queue at the loop header).  An inline-asm load into an ordinary
// variable is NOT safe either: the compiler believes the value exists as soon as the asm statement has executed and is free to copy
// the register before the data has landed (it did: `v_mov_b64 v[2:3], v[4:5]` straight after the load, i.e. stale data, in the
// 128x64 data-gradient instantiation).  So the in-flight data lives in ACCUMULATION registers that only inline asm ever names:
// global_load_dwordx4 a[..] -> counted s_waitcnt -> ds_write_b128 from a[..].  The kernels keep their MFMA accumulators in VGPRs and
// reserve a0..a(N-1) with a clobber list (acc_reserve), and tools/check_async_loads.py (run by tests/test_isa_lint_cpu.py) checks on
// the generated ISA that no compiler-emitted instruction touches an accumulation register in these kernels.
template <int R>
struct AccSlot;                                    // a[R:R+3]; every statement names its registers in the clobber list
#define ACC_SLOT(R0, R1, R2, R3)                                                                                                \
  template <>                                                                                                                   \
  struct AccSlot<R0> {                                                                                                          \
    static __device__ __forceinline__ void load16(const void* src) {                                                            \
      asm volatile("global_load_dwordx4 a[" #R0 ":" #R3 "], %0, off" ::"v"(src) : "memory", "a" #R0, "a" #R1, "a" #R2, "a" #R3); \
    }                                                                                                                           \
    static __device__ __forceinline__ void lds_write16(uint32_t lds_addr) {                                                     \
      asm volatile("ds_write_b128 %0, a[" #R0 ":" #R3 "]" ::"v"(lds_addr) : "memory", "a" #R0, "a" #R1, "a" #R2, "a" #R3);      \
    }                                                                                                                           \
    static __device__ __forceinline__ uint32_t read(int q) {                                                                    \
      uint32_t v;                                                                                                               \
      if (q == 0) asm volatile("v_accvgpr_read_b32 %0, a" #R0 : "=v"(v)::"a" #R0);                                              \
      else if (q == 1) asm volatile("v_accvgpr_read_b32 %0, a" #R1 : "=v"(v)::"a" #R1);                                         \
      else if (q == 2) asm volatile("v_accvgpr_read_b32 %0, a" #R2 : "=v"(v)::"a" #R2);                                         \
      else asm volatile("v_accvgpr_read_b32 %0, a" #R3 : "=v"(v)::"a" #R3);                                                     \
      return v;                                                                                                                 \
    }                                                                                                                           \
  };
ACC_SLOT(0, 1, 2, 3)
ACC_SLOT(4, 5, 6, 7)
ACC_SLOT(8, 9, 10, 11)
ACC_SLOT(12, 13, 14, 15)
ACC_SLOT(16, 17, 18, 19)
ACC_SLOT(20, 21, 22, 23)
ACC_SLOT(24, 25, 26, 27)
ACC_SLOT(28, 29, 30, 31)
ACC_SLOT(32, 33, 34, 35)
ACC_SLOT(36, 37, 38, 39)
ACC_SLOT(40, 41, 42, 43)
ACC_SLOT(44, 45, 46, 47)
ACC_SLOT(48, 49, 50, 51)
ACC_SLOT(52, 53, 54, 55)
ACC_SLOT(56, 57, 58, 59)
ACC_SLOT(60, 61, 62, 63)
#undef ACC_SLOT
template <int R>
__device__ __forceinline__ void acc_load16(const void* src) { AccSlot<R>::load16(src); }
template <int R>
__device__ __forceinline__ void acc_lds_write16(uint32_t lds_addr) { AccSlot<R>::lds_write16(lds_addr); }
// the wait that retires a set names the set's registers too, so nothing the compiler keeps in them can live across it
template <int N>
__device__ __forceinline__ void acc_reserve() {
  static_assert(N <= 64, "at most a0..a63 are reserved");
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17",
               "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
  if constexpr (N > 32)
    asm volatile("" ::: "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
  if constexpr (N > 48)
    asm volatile("" ::: "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  typedef const char __attribute__((address_space(3))) * lds_cptr_t;
  return (uint32_t)(uintptr_t)(lds_cptr_t)p;
}
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
}

// BIG: the gathered operand has >= 2^31 elements (288 GB parts hold such activations): row base offsets are 64-bit. A template
// parameter so the usual instantiation keeps its 32-bit address arithmetic (fewer VGPRs, no 64-bit adds per load).
template <int BM, int BN, bool FIN, int EPI = 0, bool BIG = false>
__global__ void __launch_bounds__(256) igemm_fwd_kernel(const IGemmP p) {
  typedef typename std::conditional<BIG, int64_t, int>::type aoff_t;
  constexpr int TM = BM / 64, TN = BN / 64;   // 32x32 MFMA blocks per wave in M / N
  constexpr int AI = BM / 32, BI = BN / 32;   // 16-byte loads per thread per tile for A / B
  constexpr int PITCH_C = BN * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + p.nbuf * BM * 128;

  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / p.ntiles, nt = lid - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int chunk = tid & 7, lrow = tid >> 3;

  aoff_t a_base[AI];
  int a_ih0[AI], a_iw0[AI];
  const int OHW = p.OH * p.OW;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + lrow + 32 * i;
    if (m < p.M) {
      const int n = fdiv(m, p.d_ohw), rem = m - n * OHW;
      const int oh = fdiv(rem, p.d_ow), ow = rem - oh * p.OW;
      a_ih0[i] = oh * p.sh - p.ph;
      a_iw0[i] = ow * p.sw - p.pw;
      a_base[i] = (((aoff_t)n * p.H + a_ih0[i]) * p.W + a_iw0[i]) * p.C;
    } else {
      a_ih0[i] = -(1 << 20); a_iw0[i] = -(1 << 20); a_base[i] = 0;
    }
  }
  const int CT = p.C >> 6;
  const int KT = p.R * p.S * CT;
  const int64_t wrow = (int64_t)p.R * p.S * p.C;

  // Two register stages: the loads of k-tile t+2 are issued at the top of iteration t and written to LDS at the end of iteration
  // t+1, so every load has two iterations to land. The deep-reduction layers (3x3 taps, C >= 512: 18-72 k-tiles of only 4-16 MFMAs
  // per wave) run at one exposed round trip per k-tile with a single stage. Even tiles travel through set 0, odd tiles through set 1:
  // set S, slot i is a[(S * (AI + BI) + i) * 4 ..+3] (see acc_load16).
  constexpr int SLOTS = AI + BI;
  acc_reserve<2 * SLOTS * 4>();
  auto gload = [&](int t, auto set) {
    constexpr int S = decltype(set)::value;
    const int rs = t / CT, c0 = (t - rs * CT) << 6;
    const int r = rs / p.S, s = rs - r * p.S;
    const int doff = ((r * p.dh) * p.W + s * p.dw) * p.C + c0 + chunk * 8;
    sfor<AI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int ih = a_ih0[i] + r * p.dh, iw = a_iw0[i] + s * p.dw;
      const bool ok = ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
      const bf16_t* src = ok ? p.x + ((int64_t)a_base[i] + doff) : reinterpret_cast<const bf16_t*>(g_conv_zero_page);
      acc_load16<(S * SLOTS + i) * 4>(src);
    });
    sfor<BI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      acc_load16<(S * SLOTS + AI + i) * 4>(p.w + (int64_t)(n0 + lrow + 32 * i) * wrow + rs * p.C + c0 + chunk * 8);
    });
  };
  // retire a set by hand, leaving the `newer` set's loads in flight, and move it to LDS
  const uint32_t swz_off = (uint32_t)(lrow * 128 + ((chunk ^ ((lrow >> 1) & 7)) << 4));     // row + 32 i: the swizzle term is the same
  const uint32_t ldsA = lds_addr_of(sA) + swz_off, ldsB = lds_addr_of(sB) + swz_off;
  auto retire_write = [&](int buf, auto set, bool newer) {
    constexpr int S = decltype(set)::value;
    if (newer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc_reserve<2 * SLOTS * 4>();
    const uint32_t wa = ldsA + buf * (BM * 128), wb = ldsB + buf * (BN * 128);
    sfor<AI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      acc_lds_write16<(S * SLOTS + i) * 4>(wa + i * 32 * 128);
    });
    sfor<BI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      acc_lds_write16<(S * SLOTS + AI + i) * 4>(wb + i * 32 * 128);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the compiler does not track these writes: drain before the barrier
  };
  constexpr std::integral_constant<int, 0> SET0{};
  constexpr std::integral_constant<int, 1> SET1{};

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  gload(0, SET0);
  if (KT > 1) gload(1, SET1);
  retire_write(0, SET0, KT > 1);
  __syncthreads();
  const int l31 = lane & 31, lh = lane >> 5;
  // statistics pivot of this lane's output columns, requested before the K loop so the epilogue does not wait for it
  float pvq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * (BN / 2) + j * 32 + l31;
    pvq[j] = p.piv_mean ? p.piv_mean[col] - (p.piv_bias ? p.piv_bias[col] : 0.f) : 0.f;
  }
  auto compute = [&](int cur) {
    const char* cA = sA + cur * (BM * 128);
    const char* cB = sB + cur * (BN * 128);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t af[TM], bfr[TN];
      const int q = ks * 2 + lh;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / 2) + i * 32 + l31;
        af[i] = *reinterpret_cast<const bf16x8_t*>(cA + row * 128 + ((q ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (BN / 2) + j * 32 + l31;
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(cB + row * 128 + ((q ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };
  const int bmask = p.nbuf - 1;
  for (int t = 0; t < KT; t += 2) {
    if (t + 2 < KT) gload(t + 2, SET0);              // set 0 is free: tile t went to LDS in the previous iteration (or the prologue)
    compute(t & bmask);
    if (t + 1 < KT) retire_write((t + 1) & bmask, SET1, t + 2 < KT);
    __syncthreads();
    if (t + 1 < KT) {                                // workgroup-uniform: every wave takes the same barriers
      if (t + 3 < KT) gload(t + 3, SET1);
      compute((t + 1) & bmask);
      if (t + 2 < KT) retire_write((t + 2) & bmask, SET0, t + 3 < KT);
      __syncthreads();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // EPI == 1: everything the fused epilogue reads from global memory is requested HERE, for all of this thread's rows at once,
  // so the latencies overlap each other and the accumulator staging below (issued row by row inside the store loop they were
  // exposed once per row: +11 us per launch)
  constexpr int E_CPR = BN / 8, E_RPP = 256 / E_CPR, E_NIT = EPI == 1 ? BM / E_RPP : 1;
  int64_t e_o[E_NIT];
  s16x8 e_y[E_NIT], e_prev[E_NIT];
  unsigned e_mb[E_NIT], e_ab[E_NIT];
  if constexpr (EPI == 1) {
    // Straight-line code: every load is UNCONDITIONAL (absent operands are redirected to memory that is valid to read and their
    // values discarded by selects on workgroup-uniform flags; rows past M read row M-1). With a branch per optional operand the
    // compiler wrapped each load in its own block and waited vmcnt(0) inside it, i.e. the rows' round trips ran one after another.
    const int ch = tid % E_CPR, r0 = tid / E_CPR;
    const bool want = p.bnb_partial != nullptr;
    const bool has_prev = p.acc_src != nullptr || p.accum;
    const bf16_t* prev_ptr = p.acc_src != nullptr ? p.acc_src : p.y;                 // same layout as y; y itself when accumulating in place
    const bool has_ab = p.acc_src != nullptr && p.acc_bits != nullptr;
    const uint8_t* ab_ptr = has_ab ? p.acc_bits : reinterpret_cast<const uint8_t*>(prev_ptr);
    const bf16_t* y_ptr = want ? p.bnb_y : prev_ptr;
    const bool has_mb = want && p.bnb_mode == 1;
    const uint8_t* mb_ptr = has_mb ? p.bnb_bits : reinterpret_cast<const uint8_t*>(y_ptr);
    const s16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < E_NIT; ++i) {
      const int m = m0 + r0 + i * E_RPP;
      const int mc = m < p.M ? m : p.M - 1;
      const int n = fdiv(mc, p.d_ohw), rem = mc - n * OHW;
      const int oh = fdiv(rem, p.d_ow), ow = rem - oh * p.OW;
      const int64_t o = (((int64_t)n * p.OHs + oh * p.oys) * p.OWs + ow * p.oxs) * p.K + n0 + ch * 8;
      e_o[i] = m < p.M ? o : (int64_t)-1;
      s16x8 pv = zero8;
      unsigned ab = 0xFFu;
      if (has_prev) {                                  // uniform for the whole grid and around BOTH loads of the row
        pv = *reinterpret_cast<const s16x8*>(prev_ptr + o);
        ab = ab_ptr[o >> 3];
      }
      const s16x8 yv = *reinterpret_cast<const s16x8*>(y_ptr + o);
      const unsigned mb = mb_ptr[o >> 3];
      e_prev[i] = pv;
      e_ab[i] = has_ab ? ab : 0xFFu;
      e_y[i] = yv;
      e_mb[i] = has_mb ? mb : 0xFFu;
    }
  }

  // ---------------- epilogue: bias, bf16 rounding, BN-stat partials, LDS transpose, coalesced store
  char* sC = smem;
  float* sS = reinterpret_cast<float*>(smem + ((BM * PITCH_C + 255) & ~255));
  float csum[TN], csq[TN];
  // two copies of the accumulator pass, chosen by a workgroup-uniform branch: launches without a statistics consumer (data gradients,
  // convolutions not followed by a BatchNorm) skip the 5 VALU instructions per element of the column sums
  auto acc_pass = [&](auto with_stats) {
    constexpr bool WS = decltype(with_stats)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wn * (BN / 2) + j * 32 + l31;
      const float bv = p.bias ? p.bias[n0 + col] : 0.f;
      const float pv = pvq[j];
      float s = 0.f, q2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          const bf16_t b = f2bf(acc[i][j][e] + bv);
          *reinterpret_cast<bf16_t*>(sC + row * PITCH_C + col * 2) = b;
          if constexpr (WS) {
            const float v = (m0 + row < p.M) ? bf2f(b) - pv : 0.f;
            s += v; q2 += v * v;
          }
        }
      }
      if constexpr (WS) {
        csum[j] = s + __shfl_xor(s, 32, 64);
        csq[j] = q2 + __shfl_xor(q2, 32, 64);
      } else {
        csum[j] = 0.f; csq[j] = 0.f;
      }
    }
  };
  if (EPI == 0 && p.stats != nullptr) acc_pass(std::true_type{}); else acc_pass(std::false_type{});
  if (p.stats != nullptr && lane < 32) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wn * (BN / 2) + j * 32 + lane;
      sS[(wm * 2 + 0) * BN + col] = csum[j];
      sS[(wm * 2 + 1) * BN + col] = csq[j];
    }
  }
  __syncthreads();
  if (p.stats != nullptr && tid < 2 * BN) {
    const int qs = tid / BN, col = tid - qs * BN;
    const float v = sS[(0 * 2 + qs) * BN + col] + sS[(1 * 2 + qs) * BN + col];
    float* dst = p.stats + ((int64_t)mt * 2 + qs) * p.K + n0 + col;
    if (FIN && p.fin.cnt != nullptr) ticket::st_agent(dst, v); else *dst = v;      // write-through only when another block reads it
  }
  // in-kernel BatchNorm finalize (ticket.h): the ticket is taken BEFORE the output tile is stored, so only the 2*BN statistic
  // stores are waited for and the atomic's round trip overlaps the tile stores
  // FIN is a template parameter: the default instantiation carries none of the hand-off code (it cost the plain kernel ~5 %)
  __shared__ int s_last;
  const ticket::BnFin& f = p.fin;
  const bool fin = FIN && f.cnt != nullptr;
  const int G = fin ? f.G : 1, ngroups = (p.mtiles + G - 1) / G;
  const int g = mt / G, gbeg = g * G, gend = (gbeg + G < p.mtiles) ? gbeg + G : p.mtiles;
  unsigned tk = 0;
  if (fin) {
    ticket::publish();
    if (tid == 0) tk = ticket::take(f.cnt + p.ntiles + nt * ngroups + g);
  }
  constexpr int CPR = BN / 8;
  if constexpr (EPI == 1) {
    // thread -> fixed 8-channel chunk `ch`, rows r0, r0 + RPP, ...: per-channel partial sums stay in registers over the rows
    constexpr int RPP = 256 / CPR;
    const int ch = tid % CPR, r0 = tid / CPR;
    const int cbase = n0 + ch * 8;
    const bool want = p.bnb_partial != nullptr;
    // coefficient pointers are always valid when `want` (the host substitutes the mean pointer for absent mask coefficients and
    // says which mask form applies in bnb_mode): unconditional 16-byte loads, everything stays in registers
    const int mode = p.bnb_mode;             // 0: no mask, 1: saved bits, 2: recompute [y*ms + mt > 0]
    f32x4 mu0, mu1, is0, is1, ms0, ms1, mt0, mt1;
    mu0 = mu1 = is0 = is1 = ms0 = ms1 = mt0 = mt1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (want) {
      mu0 = *reinterpret_cast<const f32x4*>(p.bnb_mean + cbase);   mu1 = *reinterpret_cast<const f32x4*>(p.bnb_mean + cbase + 4);
      is0 = *reinterpret_cast<const f32x4*>(p.bnb_invstd + cbase); is1 = *reinterpret_cast<const f32x4*>(p.bnb_invstd + cbase + 4);
      ms0 = *reinterpret_cast<const f32x4*>(p.bnb_ms + cbase);     ms1 = *reinterpret_cast<const f32x4*>(p.bnb_ms + cbase + 4);
      mt0 = *reinterpret_cast<const f32x4*>(p.bnb_mt + cbase);     mt1 = *reinterpret_cast<const f32x4*>(p.bnb_mt + cbase + 4);
    }
    f32x4 s0a = {0.f, 0.f, 0.f, 0.f}, s0b = s0a, s1a = s0a, s1b = s0a;      // sum dz / sum dz*xhat for channels 0-3 / 4-7
    const bool have_prev = p.acc_src != nullptr || p.accum;
#pragma unroll
    for (int i = 0; i < E_NIT; ++i) {
      const int row = r0 + i * RPP;
      const int64_t o = e_o[i];
      if (o < 0) continue;
      s16x8 v = *reinterpret_cast<const s16x8*>(sC + row * PITCH_C + ch * 16);
      if (have_prev) {
        const s16x8 prev = e_prev[i];
        const unsigned ab = e_ab[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(bf2f((bf16_t)v[j]) + (((ab >> j) & 1u) ? bf2f((bf16_t)prev[j]) : 0.f));
      }
      *reinterpret_cast<s16x8*>(p.y + o) = v;
      if (want) {
        const s16x8 yv = e_y[i];
        const unsigned mb = e_mb[i];
#define BNB_LANE(J, MU, IS, MS, MT, S0, S1, E)                                  \
        {                                                                      \
          const float yf = bf2f((bf16_t)yv[J]);                                \
          bool pass = (mb >> J) & 1u;                                          \
          if (mode == 2) pass = (yf * MS[E] + MT[E]) > 0.f;                    \
          const float dzv = pass ? bf2f((bf16_t)v[J]) : 0.f;                   \
          S0[E] += dzv;                                                        \
          S1[E] += dzv * (yf - MU[E]) * IS[E];                                 \
        }
        BNB_LANE(0, mu0, is0, ms0, mt0, s0a, s1a, 0) BNB_LANE(1, mu0, is0, ms0, mt0, s0a, s1a, 1)
        BNB_LANE(2, mu0, is0, ms0, mt0, s0a, s1a, 2) BNB_LANE(3, mu0, is0, ms0, mt0, s0a, s1a, 3)
        BNB_LANE(4, mu1, is1, ms1, mt1, s0b, s1b, 0) BNB_LANE(5, mu1, is1, ms1, mt1, s0b, s1b, 1)
        BNB_LANE(6, mu1, is1, ms1, mt1, s0b, s1b, 2) BNB_LANE(7, mu1, is1, ms1, mt1, s0b, s1b, 3)
#undef BNB_LANE
      }
    }
    if (!want) return;
    __syncthreads();                         // every thread is done reading the staged C tile: the buffer is reused below
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[j * 256 + tid] = s0a[j]; red[(4 + j) * 256 + tid] = s0b[j];
      red[(8 + j) * 256 + tid] = s1a[j]; red[(12 + j) * 256 + tid] = s1b[j];
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int qs = tid / BN, col = tid - qs * BN;
      const int chx = col >> 3, j = col & 7;
      float acc = 0.f;
#pragma unroll 8
      for (int r = 0; r < RPP; ++r) acc += red[(qs * 8 + j) * 256 + r * CPR + chx];
      p.bnb_partial[((int64_t)mt * 2 + qs) * p.K + n0 + col] = acc;
    }
    return;
  }
  for (int idx = tid; idx < BM * CPR; idx += 256) {
    const int row = idx / CPR, ch = idx - row * CPR;
    const int m = m0 + row;
    if (m < p.M) {
      const int n = fdiv(m, p.d_ohw), rem = m - n * OHW;
      const int oh = fdiv(rem, p.d_ow), ow = rem - oh * p.OW;
      const int64_t o = (((int64_t)n * p.OHs + oh * p.oys) * p.OWs + ow * p.oxs) * p.K + n0 + ch * 8;
      s16x8 v = *reinterpret_cast<const s16x8*>(sC + row * PITCH_C + ch * 16);
      if (p.accum) {
        const s16x8 prev = *reinterpret_cast<const s16x8*>(p.y + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(bf2f((bf16_t)v[j]) + bf2f((bf16_t)prev[j]));
      }
      *reinterpret_cast<s16x8*>(p.y + o) = v;
    }
  }
  if (!fin) return;
  // two levels so no block reads more than ~G + mtiles/G partial rows
  if (!ticket::resolve(f.cnt + p.ntiles + nt * ngroups + g, tk, (unsigned)(gend - gbeg), &s_last)) return;
  if (tid < 2 * BN) {
    const int qs = tid / BN, col = tid - qs * BN;
    ticket::st_agent(f.gpart + ((int64_t)g * 2 + qs) * p.K + n0 + col,
                     ticket::sum_rows(p.stats + ((int64_t)gbeg * 2 + qs) * p.K + n0 + col, gend - gbeg, 2 * (int64_t)p.K));
  }
  if (!ticket::arrive_last(f.cnt + nt, (unsigned)ngroups, &s_last)) return;
  if (tid < BN) {
    const float su = ticket::sum_rows(f.gpart + n0 + tid, ngroups, 2 * (int64_t)p.K);
    const float sq = ticket::sum_rows(f.gpart + p.K + n0 + tid, ngroups, 2 * (int64_t)p.K);
    ticket::bn_finalize_channel(f, n0 + tid, su, sq);
  }
}

template <int BM, int BN>
static int launch_fwd(IGemmP& p, hipStream_t st) {
  p.mtiles = (p.M + BM - 1) / BM;
  p.ntiles = p.K / BN;
  p.d_ohw = make_fastdiv((uint32_t)(p.OH * p.OW));
  p.d_ow = make_fastdiv((uint32_t)p.OW);
  const int KT = p.R * p.S * (p.C >> 6);
  p.nbuf = KT == 1 ? 1 : 2;
  constexpr int PITCH_C = BN * 2 + 16;
  constexpr int epi = ((BM * PITCH_C + 255) & ~255) + 4 * BN * (int)sizeof(float);   // epilogue staging (C tile + stat partials)
  int smem = p.nbuf * (BM + BN) * 128;
  if (smem < epi) smem = epi;
  static DeviceOnceLds lds_attr;
  const int want = 2 * (BM + BN) * 128 > epi ? 2 * (BM + BN) * 128 : epi;
  const bool big = (int64_t)p.N * p.H * p.W * p.C > 0x7fffffffLL;
  if (big && (p.fin.cnt != nullptr || p.acc_src != nullptr || p.bnb_partial != nullptr)) return -2;   // 32-bit variants only
  if ((int64_t)p.M * 1 > 0x7fffffffLL) return -2;
  if (p.fin.cnt != nullptr) {
    static DeviceOnceLds lds_attr_fin;
    lds_attr_fin.ensure(reinterpret_cast<const void*>(&igemm_fwd_kernel<BM, BN, true>), want);
    hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, true>), dim3(p.mtiles * p.ntiles), dim3(256), smem, st, p);
    DL4J_RET_LAST();
  }
  if (p.acc_src != nullptr || p.bnb_partial != nullptr) {
    static DeviceOnceLds lds_attr_epi;
    const int want1 = want > 16384 ? want : 16384;        // the partial-sum reduction needs 2 x 8 x 256 floats
    if (smem < 16384) smem = 16384;
    lds_attr_epi.ensure(reinterpret_cast<const void*>(&igemm_fwd_kernel<BM, BN, false, 1>), want1);
    hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, false, 1>), dim3(p.mtiles * p.ntiles), dim3(256), smem, st, p);
    DL4J_RET_LAST();
  }
  if ((int64_t)p.N * p.H * p.W * p.C > 0x7fffffffLL) {
    static DeviceOnceLds lds_attr_big;
    lds_attr_big.ensure(reinterpret_cast<const void*>(&igemm_fwd_kernel<BM, BN, false, 0, true>), want);
    hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, false, 0, true>), dim3(p.mtiles * p.ntiles), dim3(256), smem, st, p);
    DL4J_RET_LAST();
  }
  lds_attr.ensure(reinterpret_cast<const void*>(&igemm_fwd_kernel<BM, BN, false>), want);
  hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, false>), dim3(p.mtiles * p.ntiles), dim3(256), smem, st, p);
  DL4J_RET_LAST();
}

DL4J_API int dl4j_igemm_conv_fin(const void* x, const void* w, void* y, const float* bias, float* stats,
                                 const int* g, int tile, int accum, int* mtiles_out, void* const* ptrs, const float* fparams,
                                 float* gpart, unsigned* cnt, hipStream_t st);

// geom = [N,H,W,C, OH,OW,K, R,S, sh,sw, ph,pw, dh,dw, OHs,OWs,oys,oxs]
// returns the number of M tiles used (needed to reduce the BN-stat partials) via *mtiles_out.
DL4J_API int dl4j_igemm_conv(const void* x, const void* w, void* y, const float* bias, float* stats,
                             const int* g, int tile, int accum, int* mtiles_out, hipStream_t st) {
  return dl4j_igemm_conv_fin(x, w, y, bias, stats, g, tile, accum, mtiles_out, nullptr, nullptr, nullptr, nullptr, st);
}

// Same, with the BatchNorm finalize of the consuming layer fused into the launch (ticket.h). ptrs: gamma, beta, conv_bias, gmean,
// gvar (inputs, may be null), mean, invstd, scale, shift, dgmean, dgvar (outputs); fparams: eps, decay; stat_mode as bn_finalize.
// cnt: zero-initialised counters (self-resetting), at least ntiles * (1 + ceil(mtiles / 32)) entries; gpart: >= ceil(mtiles/32)*2*K.
static thread_local const float* g_piv_mean = nullptr;
static thread_local const float* g_piv_bias = nullptr;

DL4J_API int dl4j_igemm_conv_fin(const void* x, const void* w, void* y, const float* bias, float* stats,
                                 const int* g, int tile, int accum, int* mtiles_out, void* const* ptrs, const float* fparams,
                                 float* gpart, unsigned* cnt, hipStream_t st) {
  IGemmP p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.bias = bias; p.stats = stats; p.accum = accum;
  p.fin = ticket::BnFin{};
  p.acc_src = nullptr; p.acc_bits = nullptr; p.bnb_y = nullptr; p.bnb_mean = p.bnb_invstd = p.bnb_ms = p.bnb_mt = nullptr;
  p.bnb_bits = nullptr; p.bnb_partial = nullptr; p.bnb_mode = 0;
  p.piv_mean = g_piv_mean; p.piv_bias = g_piv_bias;
  p.N = g[0]; p.H = g[1]; p.W = g[2]; p.C = g[3]; p.OH = g[4]; p.OW = g[5]; p.K = g[6]; p.R = g[7]; p.S = g[8];
  p.sh = g[9]; p.sw = g[10]; p.ph = g[11]; p.pw = g[12]; p.dh = g[13]; p.dw = g[14];
  p.OHs = g[15]; p.OWs = g[16]; p.oys = g[17]; p.oxs = g[18];
  p.M = p.N * p.OH * p.OW;
  if (p.C % 64 != 0 || p.K % 64 != 0) return -2;
  if (cnt != nullptr && stats != nullptr && ptrs != nullptr) {
    ticket::BnFin& f = p.fin;
    f.gamma = (const float*)ptrs[0]; f.beta = (const float*)ptrs[1]; f.conv_bias = (const float*)ptrs[2];
    f.gmean = (const float*)ptrs[3]; f.gvar = (const float*)ptrs[4];
    f.mean = (float*)ptrs[5]; f.invstd = (float*)ptrs[6]; f.scale = (float*)ptrs[7]; f.shift = (float*)ptrs[8];
    f.dgmean = (float*)ptrs[9]; f.dgvar = (float*)ptrs[10];
    f.gpart = gpart; f.cnt = cnt; f.invM = 1.f / (float)p.M; f.eps = fparams[0]; f.decay = fparams[1];
    f.stat_mode = (int)fparams[2]; f.G = 32;
  }
  int rc;
  if (tile == 0) { if (p.K % 128) return -2; rc = launch_fwd<128, 128>(p, st); }
  else if (tile == 1) rc = launch_fwd<128, 64>(p, st);
  else rc = launch_fwd<64, 64>(p, st);
  if (mtiles_out) *mtiles_out = p.mtiles;
  return rc;
}

// dl4j_igemm_conv with the statistics pivot (see IGemmP::piv_mean): the partial rows hold sums of (y - pivot).
DL4J_API int dl4j_igemm_conv_piv(const void* x, const void* w, void* y, const float* bias, float* stats, const int* g, int tile,
                                 int accum, int* mtiles_out, const float* piv_mean, const float* piv_bias, hipStream_t st) {
  g_piv_mean = piv_mean; g_piv_bias = piv_bias;
  const int rc = dl4j_igemm_conv_fin(x, w, y, bias, stats, g, tile, accum, mtiles_out, nullptr, nullptr, nullptr, nullptr, st);
  g_piv_mean = nullptr; g_piv_bias = nullptr;
  return rc;
}

// Data-gradient launch with the fused epilogue described at IGemmP (EPI == 1). ptrs = [acc_src, acc_bits, bnb_y, bnb_mean,
// bnb_invstd, bnb_ms, bnb_mt, bnb_bits, bnb_partial] (any may be null; bnb_partial needs bnb_y / mean / invstd).
// geom as dl4j_igemm_conv. bnb_partial holds dl4j_igemm_mtiles(M, tile) * 2 * K floats.
DL4J_API int dl4j_igemm_dgrad_bn(const void* x, const void* w, void* y, const int* g, int tile, int accum, void* const* ptrs,
                                 hipStream_t st) {
  IGemmP p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.bias = nullptr; p.stats = nullptr; p.accum = accum;
  p.fin = ticket::BnFin{};
  p.acc_src = (const bf16_t*)ptrs[0]; p.acc_bits = (const uint8_t*)ptrs[1]; p.bnb_y = (const bf16_t*)ptrs[2];
  p.bnb_mean = (const float*)ptrs[3]; p.bnb_invstd = (const float*)ptrs[4]; p.bnb_ms = (const float*)ptrs[5];
  p.bnb_mt = (const float*)ptrs[6]; p.bnb_bits = (const uint8_t*)ptrs[7]; p.bnb_partial = (float*)ptrs[8];
  p.piv_mean = nullptr; p.piv_bias = nullptr;
  if (p.bnb_partial != nullptr && (p.bnb_y == nullptr || p.bnb_mean == nullptr || p.bnb_invstd == nullptr)) return -2;
  if ((p.bnb_ms == nullptr) != (p.bnb_mt == nullptr)) return -2;
  if (p.acc_src == nullptr && p.bnb_partial == nullptr) return -2;
  p.bnb_mode = p.bnb_ms != nullptr ? 2 : (p.bnb_bits != nullptr ? 1 : 0);
  if (p.bnb_ms == nullptr) { p.bnb_ms = p.bnb_mean; p.bnb_mt = p.bnb_mean; }     // always loadable (values unused in modes 0 / 1)
  // 16-byte loads of the per-channel coefficients, 16-byte accesses of the tensors
  for (const void* q : {(const void*)p.bnb_mean, (const void*)p.bnb_invstd, (const void*)p.bnb_ms, (const void*)p.bnb_mt,
                        (const void*)p.acc_src, (const void*)p.bnb_y, (const void*)y})
    if ((reinterpret_cast<uintptr_t>(q) & 15) != 0) return -2;
  p.N = g[0]; p.H = g[1]; p.W = g[2]; p.C = g[3]; p.OH = g[4]; p.OW = g[5]; p.K = g[6]; p.R = g[7]; p.S = g[8];
  p.sh = g[9]; p.sw = g[10]; p.ph = g[11]; p.pw = g[12]; p.dh = g[13]; p.dw = g[14];
  p.OHs = g[15]; p.OWs = g[16]; p.oys = g[17]; p.oxs = g[18];
  p.M = p.N * p.OH * p.OW;
  if (p.C % 64 != 0 || p.K % 64 != 0) return -2;
  if (tile == 0) { if (p.K % 128) return -2; return launch_fwd<128, 128>(p, st); }
  if (tile == 1) return launch_fwd<128, 64>(p, st);
  return launch_fwd<64, 64>(p, st);
}

// number of M tiles for a tile id (so the caller can size the stats workspace before launching)
DL4J_API int dl4j_igemm_mtiles(int64_t M, int tile) {
  const int bm = (tile == 2) ? 64 : 128;
  return (int)((M + bm - 1) / bm);
}

// =====================================================================================
// wgrad
// =====================================================================================
struct WGradP {
  const bf16_t* dy;   // [P][K]
  const bf16_t* x;    // [N][H][W][C]
  float* dw;          // [K][R*S][C] fp32 (accumulated)
  int N, H, W, C, OH, OW, K, R, S, sh, sw, ph, pw, dh, dw_;
  int P;              // N*OH*OW
  int ktiles, ctiles, splits, pps;   // pps: pixels per split (multiple of 64)
  int use_atomic;
  FastDiv d_ohw, d_ow;
  // Piggy-backed second stage of a column reduction (BatchNorm backward sums, layer-norm gain / bias gradients): every workgroup of
  // the weight-gradient launch first sums its slice of the [tail_gy][2][tail_C] partial rows that an earlier launch emitted
  // (bn_sum_partials_kernel's work). The two jobs are independent and adjacent in the backward pass, so one launch does both and the
  // 5-6 us dependent launch per reduction disappears (52 per ResNet-50 step). The grid does not grow: it is sized to fill the chip
  // exactly once and extra workgroups would wait for a second round.
  // Column sums of dy (the bias gradient of a dense layer = 1x1 convolution over "pixels") accumulated from the operand tiles that pass
  // through the registers anyway: db[k] += sum over pixels of dy[p][k] (fp32 atomics into a pre-zeroed / accumulating vector). Every
  // (k-tile, pixel step) is summed by exactly one of the workgroups that load it (the one with ct == step % ctiles). Replaces one
  // 5-6 us launch per linear layer of the BERT step.
  float* db;
  int db_mode;       // 1: column sums of the dy operand -> db[K];  2: column sums of the x operand -> db[C] (callers that pass the
                     // activations in the dy slot and the gradient in the x slot to get the transposed dW: ops/linear.py)
  int main_blocks, tail_blocks, tail_gy, tail_C;
  const float* tail_partial;
  float* tail_sums; float* tail_out0; float* tail_out1;
};

template <int PITCH>
__device__ __forceinline__ int tr_swz(int pixel) {
  return PITCH == 256 ? ((pixel & 3) << 2) : (((pixel >> 1) & 1) << 2);
}

// LDS tile [64 pixels][BC channels] (row pitch BC*2 bytes); fragment for the 32 channels starting at
// cb, pixels ks*16 .. ks*16+15 as a 32x32x16 MFMA A/B operand via two transposing reads.
template <int PITCH>
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int cb, int ks, int lane) {
  const int g = lane >> 4, h = g >> 1;
  const int q = (lane & 15) >> 2, pp = lane & 3;
  const int ch = cb + 16 * (g & 1) + 4 * pp;
  const int p0 = ks * 16 + 8 * h + q;
  const int a0 = p0 * PITCH + ((((ch >> 3) ^ tr_swz<PITCH>(p0))) << 4) + (ch & 7) * 2;
  const int p1 = p0 + 4;
  const int a1 = p1 * PITCH + ((((ch >> 3) ^ tr_swz<PITCH>(p1))) << 4) + (ch & 7) * 2;
  typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
  const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + a0));
  const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + a1));
  s16x8 r;
  r[0] = v0[0]; r[1] = v0[1]; r[2] = v0[2]; r[3] = v0[3];
  r[4] = v1[0]; r[5] = v1[1]; r[6] = v1[2]; r[7] = v1[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

// Split-K atomics are the expensive part of this kernel (fp32 atomic adds retire at ~2 TB/s, measured with
// tools/wgrad_sweep.py), so a workgroup is TWO 4-wave groups: each group runs the pipelined loop over its own half of the
// workgroup's pixel range with its own LDS buffers, the halves are summed through LDS and only one group issues the
// atomics — half the atomic traffic for the same number of waves in flight.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <int F16>
__device__ __forceinline__ f32x16 wg_mfma(const bf16x8_t& a, const bf16x8_t& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// F16: operands are IEEE half instead of bfloat16 (same bytes, same lane maps, the f16 MFMA). Channel counts need only be multiples
// of 8: 16-byte chunks past K / C are fed zeros and the stores are masked.
template <int BK_, int BC_, int F16>   // BK_: out-channel tile (dy), BC_: in-channel tile (x)
__global__ void __launch_bounds__(512) igemm_wgrad_kernel(const WGradP p) {
  constexpr int PA = BK_ * 2, PB = BC_ * 2;          // row pitches in bytes
  constexpr int CPA = BK_ / 8, CPB = BC_ / 8;        // 16-byte chunks per pixel row
  constexpr int RPA = 256 / CPA, RPB = 256 / CPB;    // pixel rows covered per pass
  constexpr int NA = 64 / RPA, NB = 64 / RPB;        // passes per 64-pixel tile
  constexpr int TM = BK_ / 64, TN = BC_ / 64;
  constexpr int GBYTES = 2 * 64 * (PA + PB);         // double-buffered operands of one group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.tail_partial != nullptr) {
    // rider: this workgroup's slice of the [tail_gy][2 C] partial rows -> sums (columns [0, C) and [C, 2C) are the two quantities).
    // Every workgroup of the launch takes ceil(2C / blocks) columns, so the grid (sized to fill the chip exactly once) does not grow.
    float* red = reinterpret_cast<float*>(smem);
    const int C2 = 2 * p.tail_C, nb = (int)gridDim.x;
    const int ncw = (C2 + nb - 1) / nb;                          // columns of this workgroup
    int ncl = 1;
    while (ncl < ncw && ncl < 16) ncl <<= 1;                     // column lanes (power of two <= 16), 512 / ncl row lanes
    const int nrl = 512 / ncl;
    const int cl = threadIdx.x & (ncl - 1), rl = threadIdx.x / ncl;
    for (int cb = (int)blockIdx.x * ncw; cb < (int)blockIdx.x * ncw + ncw && cb < C2; cb += ncl) {
      const int col = cb + cl;
      const bool ok = cl < ncw && col < (int)blockIdx.x * ncw + ncw && col < C2;
      float a = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      if (ok) {
        const float* q = p.tail_partial + col;
        int y = rl;
        for (; y + 3 * nrl < p.tail_gy; y += 4 * nrl) {
          a += q[(int64_t)y * C2]; a1 += q[(int64_t)(y + nrl) * C2];
          a2 += q[(int64_t)(y + 2 * nrl) * C2]; a3 += q[(int64_t)(y + 3 * nrl) * C2];
        }
        for (; y < p.tail_gy; y += nrl) a += q[(int64_t)y * C2];
      }
      red[threadIdx.x] = (a + a1) + (a2 + a3);
      __syncthreads();
      if (rl == 0 && ok) {
        float t = 0.f;
        for (int k = 0; k < nrl; ++k) t += red[k * ncl + cl];
        p.tail_sums[col] = t;
        if (col < p.tail_C) { if (p.tail_out0) p.tail_out0[col] = t; }
        else if (p.tail_out1) p.tail_out1[col - p.tail_C] = t;
      }
      __syncthreads();
    }
  }
  const int grp = threadIdx.x >> 8;
  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
  char* sA = smem + grp * GBYTES;
  char* sB = sA + 2 * 64 * PA;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int RS = p.R * p.S;
  const int ct = lid % p.ctiles; lid /= p.ctiles;
  const int kt = lid % p.ktiles; lid /= p.ktiles;
  const int rs = lid % RS;
  const int split = lid / RS;
  const int k0 = kt * BK_, c0 = ct * BC_;
  const int r = rs / p.S, s = rs - r * p.S;
  const int half = p.pps >> 1;                                      // pps is a multiple of 128
  const int bend = (split * p.pps + p.pps < p.P) ? split * p.pps + p.pps : p.P;
  const int pstart = split * p.pps + grp * half;
  int pend = pstart + half;
  if (pend > bend) pend = bend;
  const int wm = wid >> 1, wn = wid & 1;
  const int chA = tid % CPA, prA = tid / CPA;
  const int chB = tid % CPB, prB = tid / CPB;
  const int OHW = p.OH * p.OW;

  // two stages of untracked loads held in accumulation registers, as in igemm_fwd_kernel: even steps travel through set 0, odd steps
  // through set 1; set S, slot i is a[(S * (NA + NB) + i) * 4 ..+3]
  constexpr int SLOTS = NA + NB;
  acc_reserve<2 * SLOTS * 4>();
  auto gload = [&](int pb, auto set) {
    constexpr int S = decltype(set)::value;
    sfor<NA>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int pp = pb + prA + i * RPA;
      const bf16_t* src = (pp < pend && k0 + chA * 8 < p.K) ? p.dy + (int64_t)pp * p.K + k0 + chA * 8
                                                            : reinterpret_cast<const bf16_t*>(g_conv_zero_page);
      acc_load16<(S * SLOTS + i) * 4>(src);
    });
    sfor<NB>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int pp = pb + prB + i * RPB;
      const int ppc = pp < pend ? pp : pend - 1;                     // pend > pstart >= 0 is not guaranteed: clamp at 0 as well
      const int ppv = ppc < 0 ? 0 : ppc;
      const int n = fdiv(ppv, p.d_ohw), rem = ppv - n * OHW;
      const int oh = fdiv(rem, p.d_ow), ow = rem - oh * p.OW;
      const int ih = oh * p.sh - p.ph + r * p.dh, iw = ow * p.sw - p.pw + s * p.dw_;
      const bool ok = pp < pend && c0 + chB * 8 < p.C && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
      const bf16_t* src = ok ? p.x + ((int64_t)(n * p.H + ih) * p.W + iw) * p.C + c0 + chB * 8
                             : reinterpret_cast<const bf16_t*>(g_conv_zero_page);
      acc_load16<(S * SLOTS + NA + i) * 4>(src);
    });
  };
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  const bool want_db = p.db != nullptr && rs == 0;
  const int db_mode = p.db_mode;
  auto cs_add = [&](uint32_t w, int j) {                            // one register = two 16-bit elements
    if (F16) {
      cs[j] += (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu));
      cs[j + 1] += (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
    } else {
      cs[j] += __builtin_bit_cast(float, w << 16);
      cs[j + 1] += __builtin_bit_cast(float, w & 0xFFFF0000u);
    }
  };
  const uint32_t ldsA = lds_addr_of(sA), ldsB = lds_addr_of(sB);
  // wait for set S (leaving the newer set in flight), add its rows to the bias-gradient column sums when this workgroup owns them
  // (rows past the range were loaded from the zero page), move it to LDS
  auto retire_write = [&](int t, int buf, auto set, bool newer) {
    constexpr int S = decltype(set)::value;
    if (newer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc_reserve<2 * SLOTS * 4>();
    if (want_db) {
      if (db_mode == 1) {
        if ((t % p.ctiles) == ct)
          sfor<NA * 4>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            cs_add(AccSlot<(S * SLOTS + q / 4) * 4>::read(q & 3), (q & 3) * 2);
          });
      } else if ((t % p.ktiles) == kt) {
        sfor<NB * 4>([&](auto ic) {
          constexpr int q = decltype(ic)::value;
          cs_add(AccSlot<(S * SLOTS + NA + q / 4) * 4>::read(q & 3), (q & 3) * 2);
        });
      }
    }
    sfor<NA>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int pl = prA + i * RPA;
      acc_lds_write16<(S * SLOTS + i) * 4>(ldsA + buf * (64 * PA) + pl * PA + ((chA ^ tr_swz<PA>(pl)) << 4));
    });
    sfor<NB>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int pl = prB + i * RPB;
      acc_lds_write16<(S * SLOTS + NA + i) * 4>(ldsB + buf * (64 * PB) + pl * PB + ((chB ^ tr_swz<PB>(pl)) << 4));
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  constexpr std::integral_constant<int, 0> SET0{};
  constexpr std::integral_constant<int, 1> SET1{};

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](int cur) {
    const char* cA = sA + cur * (64 * PA);
    const char* cB = sB + cur * (64 * PB);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = tr_frag<PA>(cA, wm * (BK_ / 2) + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = tr_frag<PB>(cB, wn * (BC_ / 2) + j * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = wg_mfma<F16>(af[i], bfr[j], acc[i][j]);
    }
  };
  // the step count is workgroup-uniform (both groups run the same barriers); a group whose range is shorter loads zeros
  const int nsteps = (half + 63) / 64;
  gload(pstart, SET0);
  if (nsteps > 1) gload(pstart + 64, SET1);
  retire_write(0, 0, SET0, nsteps > 1);
  __syncthreads();
  for (int t = 0; t < nsteps; t += 2) {
    if (t + 2 < nsteps) gload(pstart + (t + 2) * 64, SET0);
    compute(0);
    if (t + 1 < nsteps) retire_write(t + 1, 1, SET1, t + 2 < nsteps);
    __syncthreads();
    if (t + 1 < nsteps) {
      if (t + 3 < nsteps) gload(pstart + (t + 3) * 64, SET1);
      compute(1);
      if (t + 2 < nsteps) retire_write(t + 2, 0, SET0, t + 3 < nsteps);
      __syncthreads();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // group 1 hands its partial tile to group 0 through LDS (the operand buffers are free now: BK_*BC_*4 <= 2*GBYTES)
  float* xch = reinterpret_cast<float*>(smem);
  float* cred = reinterpret_cast<float*>(smem + BK_ * BC_ * 4);      // [2 groups][RPA rows][BK_] (or [RPB][BC_]) behind the exchange tile
  static_assert(BK_ * BC_ * 4 + 2 * RPA * BK_ * 4 <= 2 * GBYTES && BK_ * BC_ * 4 + 2 * RPB * BC_ * 4 <= 2 * GBYTES,
                "column-sum scratch must fit behind the exchange tile");
  if (want_db) {
    if (db_mode == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) cred[(grp * RPA + prA) * BK_ + chA * 8 + j] = cs[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) cred[(grp * RPB + prB) * BC_ + chB * 8 + j] = cs[j];
    }
  }
  if (grp == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[(((i * TN + j) * 16 + e) * 4 + wid) * 64 + lane] = acc[i][j][e];
  }
  __syncthreads();
  if (want_db && grp == 0) {
    if (db_mode == 1) {
      if (tid < BK_ && k0 + tid < p.K) {
        float a = 0.f;
        for (int r2 = 0; r2 < 2 * RPA; ++r2) a += cred[r2 * BK_ + tid];
        atomicAdd(p.db + k0 + tid, a);
      }
    } else if (tid < BC_ && c0 + tid < p.C) {
      float a = 0.f;
      for (int r2 = 0; r2 < 2 * RPB; ++r2) a += cred[r2 * BC_ + tid];
      atomicAdd(p.db + c0 + tid, a);
    }
  }
  if (grp == 1) return;
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = k0 + wm * (BK_ / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int c = c0 + wn * (BC_ / 2) + j * 32 + l31;
        const float v = acc[i][j][e] + xch[(((i * TN + j) * 16 + e) * 4 + wid) * 64 + lane];
        if (k < p.K && c < p.C) {
          float* dst = p.dw + ((int64_t)k * RS + rs) * p.C + c;
          if (p.use_atomic) atomicAdd(dst, v);
          else *dst = v;
        }
      }
}

// fill kernel instead of hipMemsetAsync: memset NODES of captured HIP graphs were observed to misbehave on replay once a
// second graph exists in the process (see softmax_xent.hip / tools/repro_two_captured_nets.py)
__global__ void __launch_bounds__(256) zero_f32x4_kernel(float* __restrict__ p, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    reinterpret_cast<f32x4*>(p)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = 0.f;
}
__global__ void __launch_bounds__(256) zero_f32_scalar_kernel(float* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
}

template <int BK_, int BC_, int F16>
static int launch_wgrad(WGradP& p, int target_blocks, int prezeroed, hipStream_t st) {
  p.ktiles = (p.K + BK_ - 1) / BK_;
  p.ctiles = (p.C + BC_ - 1) / BC_;
  p.d_ohw = make_fastdiv((uint32_t)(p.OH * p.OW));
  p.d_ow = make_fastdiv((uint32_t)p.OW);
  const int tiles = p.ktiles * p.ctiles * p.R * p.S;
  int splits;
  if (target_blocks > 0) {
    splits = (target_blocks + tiles - 1) / tiles;
  } else {
    // auto: fill the chip exactly once. A workgroup needs 512 * (BK_ + BC_) bytes of the 160 KB LDS, so one (two for the
    // 64x64 tile) is resident per CU; a partial second round costs more than the shorter split saves (tools/wgrad_sweep.py)
    const int capacity = DL4J_NUM_CU * ((160 * 1024) / (2 * 2 * 64 * (BK_ + BC_) * 2));
    splits = capacity / tiles;
  }
  int maxs = (p.P + 511) / 512;
  if (splits > maxs) splits = maxs;
  if (splits < 1) splits = 1;
  int pps = (p.P + splits - 1) / splits;
  pps = (pps + 127) / 128 * 128;             // two 64-pixel-aligned halves, one per wave group
  splits = (p.P + pps - 1) / pps;
  p.splits = splits; p.pps = pps;
  p.use_atomic = splits > 1;
  if (p.use_atomic && !prezeroed) {
    const int64_t n = (int64_t)p.K * p.R * p.S * p.C;
    if ((reinterpret_cast<uintptr_t>(p.dw) & 15) == 0)
      hipLaunchKernelGGL(zero_f32x4_kernel, dim3(grid_for(n >> 2, 256)), dim3(256), 0, st, p.dw, n);
    else    // gradient view at an odd offset of the flat vector
      hipLaunchKernelGGL(zero_f32_scalar_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, p.dw, n);
  }
  const int smem = 2 * 2 * 64 * (BK_ + BC_) * 2;      // two groups x double buffer
  static_assert(BK_ * BC_ * 4 <= 2 * 2 * 64 * (BK_ + BC_) * 2, "exchange tile must fit the operand buffers");
  static DeviceOnceLds lds_attr;
  lds_attr.ensure(reinterpret_cast<const void*>(&igemm_wgrad_kernel<BK_, BC_, F16>), smem);
  p.main_blocks = tiles * splits;
  hipLaunchKernelGGL((igemm_wgrad_kernel<BK_, BC_, F16>), dim3(p.main_blocks + p.tail_blocks), dim3(512), smem, st, p);
  DL4J_RET_LAST();
}

// geom = [N,H,W,C, OH,OW,K, R,S, sh,sw, ph,pw, dh,dw]; dw_out fp32 [K][R*S][C]
// prezeroed: the caller guarantees dw_out is all zeros (flat gradient vector cleared once per iteration, or the
// self-cleaning scratch of the 3x3 path), so the per-layer memset in front of the split-K atomics is skipped.
template <int F16>
static int dispatch_wgrad(WGradP& p, int target_blocks, int prezeroed, hipStream_t st) {
  const bool k128 = p.K >= 128 && (p.K % 128 == 0 || p.K > 256), c128 = p.C >= 128 && (p.C % 128 == 0 || p.C > 256);
  // (the IEEE-half 128x128 instantiation is not built: the register allocator used a0..a3 as operand temporaries there, which the hidden
  // loaders own - tools/check_async_loads.py)
  if constexpr (!F16) {
    if (k128 && c128) return launch_wgrad<128, 128, F16>(p, target_blocks, prezeroed, st);
  }
  if (k128) return launch_wgrad<128, 64, F16>(p, target_blocks, prezeroed, st);
  if (c128) return launch_wgrad<64, 128, F16>(p, target_blocks, prezeroed, st);
  return launch_wgrad<64, 64, F16>(p, target_blocks, prezeroed, st);
}

static thread_local float* g_wgrad_db = nullptr;
static thread_local int g_wgrad_db_mode = 1;
struct WGradTail {
  const float* partial; int gy, C; float* sums; float* out0; float* out1;
};
static int wgrad_entry(const void* dy, const void* x, float* dw_out, const int* g, int target_blocks, int prezeroed, int f16,
                       hipStream_t st, const WGradTail* tail = nullptr) {
  WGradP p;
  p.db = g_wgrad_db; p.db_mode = g_wgrad_db_mode;
  p.tail_blocks = 0; p.tail_gy = 0; p.tail_C = 0; p.tail_partial = nullptr; p.tail_sums = p.tail_out0 = p.tail_out1 = nullptr;
  if (tail != nullptr) {
    if (!tail->partial || !tail->sums || tail->gy <= 0 || tail->C <= 0) return -2;
    p.tail_blocks = 0; p.tail_gy = tail->gy; p.tail_C = tail->C;
    p.tail_partial = tail->partial; p.tail_sums = tail->sums; p.tail_out0 = tail->out0; p.tail_out1 = tail->out1;
  }
  p.dy = (const bf16_t*)dy; p.x = (const bf16_t*)x; p.dw = dw_out;
  p.N = g[0]; p.H = g[1]; p.W = g[2]; p.C = g[3]; p.OH = g[4]; p.OW = g[5]; p.K = g[6]; p.R = g[7]; p.S = g[8];
  p.sh = g[9]; p.sw = g[10]; p.ph = g[11]; p.pw = g[12]; p.dh = g[13]; p.dw_ = g[14];
  const int64_t P = (int64_t)p.N * p.OH * p.OW;
  if (p.C % 8 != 0 || p.K % 8 != 0 || P <= 0 || P > 0x7fffffff) return -2;     // 16-byte chunks inside a pixel's channel run
  p.P = (int)P;
  return f16 ? dispatch_wgrad<1>(p, target_blocks, prezeroed, st) : dispatch_wgrad<0>(p, target_blocks, prezeroed, st);
}

DL4J_API int dl4j_igemm_wgrad(const void* dy, const void* x, float* dw_out, const int* g, int target_blocks, int prezeroed,
                              hipStream_t st) {
  return wgrad_entry(dy, x, dw_out, g, target_blocks, prezeroed, 0, st);
}
// same with a dtype code: 0 bf16, 1 fp16
DL4J_API int dl4j_igemm_wgrad_dt(const void* dy, const void* x, float* dw_out, const int* g, int target_blocks, int prezeroed,
                                 int dtype, hipStream_t st) {
  if (dtype != 0 && dtype != 1) return -2;
  return wgrad_entry(dy, x, dw_out, g, target_blocks, prezeroed, dtype, st);
}

// dl4j_igemm_wgrad_dt for a 1x1 problem that also accumulates the column sums of dy into db [K] (fp32, atomics: zeroed or accumulating)
// mode 1: sums of the dy operand -> db[K]; mode 2: sums of the x operand -> db[C] (stride 1, no padding: every pixel row is read once)
DL4J_API int dl4j_igemm_wgrad_db(const void* dy, const void* x, float* dw_out, const int* g, int target_blocks, int prezeroed,
                                 int dtype, float* db, int mode, hipStream_t st) {
  if ((dtype != 0 && dtype != 1) || db == nullptr || g[7] != 1 || g[8] != 1 || (mode != 1 && mode != 2)) return -2;
  if (mode == 2 && (g[9] != 1 || g[10] != 1 || g[11] != 0 || g[12] != 0)) return -2;
  g_wgrad_db = db; g_wgrad_db_mode = mode;
  const int rc = wgrad_entry(dy, x, dw_out, g, target_blocks, prezeroed, dtype, st);
  g_wgrad_db = nullptr; g_wgrad_db_mode = 1;
  return rc;
}

// both riders at once: the bias-gradient column sums (db, mode as above; null = none) and a piggy-backed partial-row reduction
// (partial null = none): what the dense layers of a transformer block use (bias gradient of the layer itself + the pending
// gain / bias gradient sums of a layer norm that ran backward earlier)
DL4J_API int dl4j_igemm_wgrad_riders(const void* dy, const void* x, float* dw_out, const int* g, int target_blocks, int prezeroed,
                                     int dtype, float* db, int mode, const float* partial, int gy, int C, float* sums, float* out0,
                                     float* out1, hipStream_t st) {
  if (dtype != 0 && dtype != 1) return -2;
  if (db != nullptr) {
    if (g[7] != 1 || g[8] != 1 || (mode != 1 && mode != 2)) return -2;
    if (mode == 2 && (g[9] != 1 || g[10] != 1 || g[11] != 0 || g[12] != 0)) return -2;
  }
  WGradTail t{partial, gy, C, sums, out0, out1};
  g_wgrad_db = db; g_wgrad_db_mode = db != nullptr ? mode : 1;
  const int rc = wgrad_entry(dy, x, dw_out, g, target_blocks, prezeroed, dtype, st, partial != nullptr ? &t : nullptr);
  g_wgrad_db = nullptr; g_wgrad_db_mode = 1;
  return rc;
}

// dl4j_igemm_wgrad_dt + the piggy-backed partial-row reduction (WGradP::tail_*): partial [gy][2][C] -> sums [2C], and the dbeta (out0)
// / dgamma (out1) gradient views when given
DL4J_API int dl4j_igemm_wgrad_tail(const void* dy, const void* x, float* dw_out, const int* g, int target_blocks, int prezeroed,
                                   int dtype, const float* partial, int gy, int C, float* sums, float* out0, float* out1, hipStream_t st) {
  if (dtype != 0 && dtype != 1) return -2;
  WGradTail t{partial, gy, C, sums, out0, out1};
  return wgrad_entry(dy, x, dw_out, g, target_blocks, prezeroed, dtype, st, &t);
}

// fp32 [K][RS][C] -> [K][C][RS]  (OHWI -> DL4J OIHW gradient view)
// clear_src: zero every source element after reading it (each is read exactly once), leaving the scratch ready for the next
// layer's split-K atomics without a memset
__global__ void __launch_bounds__(256) ohwi_to_oihw_kernel(float* __restrict__ src, float* __restrict__ dst,
                                                           int64_t total, int RS, int C, int clear_src) {
  for (int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x; d < total; d += (int64_t)gridDim.x * 256) {
    const int64_t k = d / ((int64_t)C * RS);
    const int64_t rem = d - k * (int64_t)C * RS;
    const int c = (int)(rem / RS), rs = (int)(rem - (int64_t)c * RS);
    const int64_t si = (k * RS + rs) * C + c;
    dst[d] = src[si];
    if (clear_src) src[si] = 0.f;
  }
}

DL4J_API int dl4j_ohwi_to_oihw(float* src, float* dst, int K, int RS, int C, int clear_src, hipStream_t st) {
  const int64_t total = (int64_t)K * RS * C;
  hipLaunchKernelGGL(ohwi_to_oihw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, src, dst, total, RS, C, clear_src);
  DL4J_RET_LAST();
}
