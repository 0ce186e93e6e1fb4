// Round-2 convolution / GEMM entry points on the LDS-DMA core (gemm_core.h): implicit-GEMM forward and data gradient for 1-D / 2-D /
// 3-D convolutions and transposed convolutions in bf16, fp16 and fp32, plus the plain / batched NT GEMM used by dense layers,
// recurrent gates, attention projections and SameDiff matmul.
//
// One kernel covers every case through a TAP TABLE: output row m = (n, od, oh, ow) of a logical output grid reads input pixel
// (od*sd - pd + tap_d[k], oh*sh - ph + tap_h[k], ow*sw - pw + tap_w[k]) for tap k, whose filter slice starts at element tap_off[k]
// of a filter row; rows land at ((n*ODs + od*ozs + oz0)*OHs + oh*oys + oy0)*OWs + ow*oxs + ox0 of the output tensor.
//   * forward conv, any stride / padding / dilation:  taps = filter taps, s = stride, o*s = 1.
//   * data gradient (= transposed conv forward), ANY stride: one launch per residue class (a, b, c) of the input pixel index modulo
//     the stride; inside a class the valid taps are a fixed subset and the gather is a stride-1 correlation over dy, written with
//     pixel stride = conv stride (libnd4j computes gradI as tensorDot + col2im scatter, helpers/cuda/convolutions_conv2dBP.cu:83-113).
//     A class with no valid tap (1x1 stride-2 conv) has zero k-tiles and just writes its zeros: no separate fill pass.
//   * C and K need not be multiples of 64: 16-byte chunks past the end of a row come from a zero page, stores are masked.
// Addresses are 64-bit per staged row, so tensors beyond 2^31 elements work (288 GB HBM).
#include "gemm_kernels.h"

using namespace gemmcore;

#define CONV_MAX_TAPS 128

struct ConvP {
  const void* x; const void* w;
  EpiP epi;
  int N, D, H, W, C;            // input tensor (NDHWC)
  int OD, OH, OW, K;            // logical output grid of this launch, output channels
  int sd, sh, sw, pd, ph, pw;   // base input coordinate of an output row = o*s - p
  int ntaps, CT;                // CT = k-tiles per tap = ceil(C / BK)
  int64_t wrow;                 // filter row stride (elements)
  int ODs, OHs, OWs;            // physical output tensor spatial dims
  int ozs, oys, oxs, oz0, oy0, ox0;
  int M, mtiles, ntiles, gm;
  short tap_d[CONV_MAX_TAPS], tap_h[CONV_MAX_TAPS], tap_w[CONV_MAX_TAPS];
  int tap_off[CONV_MAX_TAPS];
};

template <class C> struct ConvSrc {
  const char* ap[C::PA];
  int bd[C::PA], bh[C::PA], bw[C::PA];
  const char* bp[C::PB];
  int kchunk;
  int CT, Cc, D, H, W;
  const ConvP* p;
  struct Tile { int dd, dh, dw, c0; int64_t aoff, boff; };
  // wave-uniform scalars of k-tile t: its tap (one kernarg table lookup) and channel offset
  __device__ __forceinline__ Tile tile(int t) const {
    Tile ts;
    const int tap = (CT == 1) ? t : t / CT;
    ts.c0 = (t - tap * CT) * C::BK;
    ts.dd = p->tap_d[tap]; ts.dh = p->tap_h[tap]; ts.dw = p->tap_w[tap];
    ts.aoff = ((((int64_t)ts.dd * H + ts.dh) * W + ts.dw) * Cc + ts.c0) * C::ESZ;
    ts.boff = ((int64_t)p->tap_off[tap] + ts.c0) * C::ESZ;
    return ts;
  }
  __device__ __forceinline__ const void* a(int i, const Tile& ts) const {
    const int id = bd[i] + ts.dd, ih = bh[i] + ts.dh, iw = bw[i] + ts.dw;
    // bitwise, not short-circuit: a branch per staged row would split the DMA issue stream
    const bool ok = (int)((unsigned)id < (unsigned)D) & (int)((unsigned)ih < (unsigned)H) & (int)((unsigned)iw < (unsigned)W) &
                    (int)(ts.c0 + kchunk < Cc);
    return ok ? (const void*)(ap[i] + ts.aoff) : (const void*)g_zero_page;
  }
  __device__ __forceinline__ const void* b(int i, const Tile& ts) const {
    return ts.c0 + kchunk < Cc ? (const void*)(bp[i] + ts.boff) : (const void*)g_zero_page;
  }
};

template <class C, int OT>
__global__ void __launch_bounds__(C::NT) conv_igemm_kernel(const ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int mt, nt;
  tile_coords(lid, p.mtiles, p.ntiles, p.gm, mt, nt);
  const int m0 = mt * C::BM, n0 = nt * C::BN;
  const int OHW = p.OH * p.OW, ODHW = p.OD * OHW;

  ConvSrc<C> src;
  src.p = &p;
  src.CT = p.CT; src.Cc = p.C; src.D = p.D; src.H = p.H; src.W = p.W;
  src.kchunk = stage_chunk(stage_row<C>(0)) * C::CH;
#pragma unroll
  for (int i = 0; i < C::PA; ++i) {
    const int m = m0 + stage_row<C>(i);
    if (m < p.M) {
      const int n = m / ODHW; int rem = m - n * ODHW;
      const int od = rem / OHW; rem -= od * OHW;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      src.bd[i] = od * p.sd - p.pd; src.bh[i] = oh * p.sh - p.ph; src.bw[i] = ow * p.sw - p.pw;
      src.ap[i] = (const char*)p.x + (((((int64_t)n * p.D + src.bd[i]) * p.H + src.bh[i]) * p.W + src.bw[i]) * p.C + src.kchunk) * C::ESZ;
    } else {
      src.bd[i] = src.bh[i] = src.bw[i] = -(1 << 24);     // every tap out of range: the row is fed zeros and never stored
      src.ap[i] = (const char*)p.x;
    }
  }
#pragma unroll
  for (int i = 0; i < C::PB; ++i) {
    int n = n0 + stage_row<C>(i);
    if (n > p.K - 1) n = p.K - 1;
    src.bp[i] = (const char*)p.w + ((int64_t)n * p.wrow + src.kchunk) * C::ESZ;
  }
  typename C::acc_t acc[C::TM][C::TN];
  mainloop<C>(smem, src, p.ntaps * p.CT, acc);

  epilogue<C, OT>(smem, acc, mt, n0, p.K, p.M - m0, p.epi, [&](int r) -> int64_t {
    const int m = m0 + r;
    if (m >= p.M) return (int64_t)-1;
    const int n = m / ODHW; int rem = m - n * ODHW;
    const int od = rem / OHW; rem -= od * OHW;
    const int oh = rem / p.OW, ow = rem - oh * p.OW;
    return ((((int64_t)n * p.ODs + od * p.ozs + p.oz0) * p.OHs + oh * p.oys + p.oy0) * p.OWs + ow * p.oxs + p.ox0) * p.K + n0;
  });
}

template <class C, int OT>
static int launch_conv(ConvP& p, hipStream_t st) {
  p.mtiles = (p.M + C::BM - 1) / C::BM;
  p.ntiles = (p.K + C::BN - 1) / C::BN;
  p.CT = (p.C + C::BK - 1) / C::BK;
  if (p.gm <= 0) p.gm = 8;
  constexpr int LDS = C::lds_bytes(Out<OT>::SZ);
  static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  static DeviceOnceLds lds_attr;
  lds_attr.ensure(reinterpret_cast<const void*>(&conv_igemm_kernel<C, OT>), LDS);
  hipLaunchKernelGGL((conv_igemm_kernel<C, OT>), dim3(p.mtiles * p.ntiles), dim3(C::NT), LDS, st, p);
  return (int)hipGetLastError();
}

// tile ids shared by conv and GEMM: {BM, BN}
static const int kTiles[][2] = {{128, 128}, {128, 64}, {64, 64}, {64, 128}, {256, 256}};

template <int DT, int OT>
static int dispatch_conv(ConvP& p, int tile, hipStream_t st) {
  // a single k-tile (1x1 conv with C <= BK): one LDS stage, so more workgroups are resident to overlap their load and store phases
  if (p.ntaps * ((p.C + (DT == DT_F32 ? 32 : 64) - 1) / (DT == DT_F32 ? 32 : 64)) == 1) {
    switch (tile) {
      case 0: return launch_conv<Cfg<128, 128, 2, 2, 1, 16, DT>, OT>(p, st);
      case 1: return launch_conv<Cfg<128, 64, 2, 2, 1, 16, DT>, OT>(p, st);
      case 2: return launch_conv<Cfg<64, 64, 2, 2, 1, 16, DT>, OT>(p, st);
      case 3: return launch_conv<Cfg<64, 128, 2, 2, 1, 16, DT>, OT>(p, st);
      default: return -2;
    }
  }
  switch (tile) {
    case 0: return launch_conv<Cfg<128, 128, 2, 2, 2, 16, DT>, OT>(p, st);
    case 1: return launch_conv<Cfg<128, 64, 2, 2, 2, 16, DT>, OT>(p, st);
    case 2: return launch_conv<Cfg<64, 64, 2, 2, 2, 16, DT>, OT>(p, st);
    case 3: return launch_conv<Cfg<64, 128, 2, 2, 2, 16, DT>, OT>(p, st);
    default: return -2;
  }
}

// desc (int64): [0] N D H W C  [5] OD OH OW K  [9] sd sh sw  [12] pd ph pw  [15] ntaps  [16] wrow  [17] ODs OHs OWs
//               [20] ozs oys oxs  [23] oz0 oy0 ox0  [26] tile  [27] dtype (0 bf16, 1 fp16, 2 fp32)  [28] accum  [29] act
//               [30 ..] ntaps x (tap_d, tap_h, tap_w, tap_off)
DL4J_API int dl4j_conv_mfma(const void* x, const void* w, void* y, const float* bias, float* stats, const float* stat_shift,
                            const int64_t* d, int* mtiles_out, hipStream_t st) {
  ConvP p;
  p.x = x; p.w = w;
  p.epi.out = y; p.epi.residual = nullptr; p.epi.bias = bias; p.epi.stats = stats; p.epi.stat_shift = stat_shift;
  p.epi.alpha = 1.f; p.epi.act = (int)d[29]; p.epi.accum = (int)d[28];
  p.N = (int)d[0]; p.D = (int)d[1]; p.H = (int)d[2]; p.W = (int)d[3]; p.C = (int)d[4];
  p.OD = (int)d[5]; p.OH = (int)d[6]; p.OW = (int)d[7]; p.K = (int)d[8];
  p.sd = (int)d[9]; p.sh = (int)d[10]; p.sw = (int)d[11]; p.pd = (int)d[12]; p.ph = (int)d[13]; p.pw = (int)d[14];
  p.ntaps = (int)d[15]; p.wrow = d[16];
  p.ODs = (int)d[17]; p.OHs = (int)d[18]; p.OWs = (int)d[19];
  p.ozs = (int)d[20]; p.oys = (int)d[21]; p.oxs = (int)d[22]; p.oz0 = (int)d[23]; p.oy0 = (int)d[24]; p.ox0 = (int)d[25];
  p.gm = 0;
  const int tile = (int)d[26], dt = (int)d[27];
  if (p.ntaps < 0 || p.ntaps > CONV_MAX_TAPS) return -2;
  const int ch = dt == DT_F32 ? 4 : 8;
  // host-side shape checks: every 16-byte chunk the kernel may touch lies inside a pixel's channel run
  if (p.C % ch != 0 || p.wrow % ch != 0 || p.N <= 0 || p.K <= 0) return -2;
  for (int k = 0; k < p.ntaps; ++k) {
    p.tap_d[k] = (short)d[30 + 4 * k]; p.tap_h[k] = (short)d[31 + 4 * k]; p.tap_w[k] = (short)d[32 + 4 * k];
    p.tap_off[k] = (int)d[33 + 4 * k];
    if (p.tap_off[k] % ch != 0 || p.tap_off[k] + p.C > p.wrow) return -2;
  }
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  if (M <= 0 || M > 0x7fffffff) return -2;
  p.M = (int)M;
  int rc;
  if (dt == DT_BF16) rc = dispatch_conv<DT_BF16, OT_BF16>(p, tile, st);
  else if (dt == DT_F16) rc = dispatch_conv<DT_F16, OT_F16>(p, tile, st);
  else if (dt == DT_F32) rc = dispatch_conv<DT_F32, OT_F32>(p, tile, st);
  else return -2;
  if (mtiles_out) *mtiles_out = p.mtiles;
  return rc;
}

DL4J_API int dl4j_mfma_tile_dims(int tile, int* bm, int* bn) {
  if (tile < 0 || tile > 4) return -2;
  *bm = kTiles[tile][0]; *bn = kTiles[tile][1];
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------ GEMM
template <int DT, int OT>
static int dispatch_gemm(GemmP& p, int tile, int batch, hipStream_t st) {
  if (p.K <= (DT == DT_F32 ? 32 : 64)) {
    switch (tile) {
      case 0: return launch_gemm<Cfg<128, 128, 2, 2, 1, 16, DT>, OT>(p, batch, st);
      case 1: return launch_gemm<Cfg<128, 64, 2, 2, 1, 16, DT>, OT>(p, batch, st);
      case 2: return launch_gemm<Cfg<64, 64, 2, 2, 1, 16, DT>, OT>(p, batch, st);
      case 3: return launch_gemm<Cfg<64, 128, 2, 2, 1, 16, DT>, OT>(p, batch, st);
      default: break;
    }
  }
  switch (tile) {
    case 0: return launch_gemm<Cfg<128, 128, 2, 2, 2, 16, DT>, OT>(p, batch, st);
    case 1: return launch_gemm<Cfg<128, 64, 2, 2, 2, 16, DT>, OT>(p, batch, st);
    case 2: return launch_gemm<Cfg<64, 64, 2, 2, 2, 16, DT>, OT>(p, batch, st);
    case 3: return launch_gemm<Cfg<64, 128, 2, 2, 2, 16, DT>, OT>(p, batch, st);
    case 4:
      if constexpr (DT == DT_BF16 && OT == OT_BF16) return launch_gemm<Cfg<256, 256, 2, 4, 2, 16, DT>, OT, 1>(p, batch, st);
      else return -2;
    case 10:  // 256x256 ping-pong, half-tile staging with counted waits (1.5 k-tiles of LDS-DMA in flight)
      if constexpr (DT == DT_BF16 && OT == OT_BF16) return launch_gemm<Cfg<256, 256, 2, 4, 2, 16, DT>, OT, 2>(p, batch, st);
      else return -2;
    // deeper pipelines (counted vmcnt waits keep 1 - 2 k-tiles in flight across the barrier): 5..8 = tiles 0..3 with 3 stages, 9 = 64x64 with 4
    case 5: return launch_gemm<Cfg<128, 128, 2, 2, 3, 16, DT>, OT>(p, batch, st);
    case 6: return launch_gemm<Cfg<128, 64, 2, 2, 3, 16, DT>, OT>(p, batch, st);
    case 7: return launch_gemm<Cfg<64, 64, 2, 2, 3, 16, DT>, OT>(p, batch, st);
    case 8: return launch_gemm<Cfg<64, 128, 2, 2, 3, 16, DT>, OT>(p, batch, st);
    case 9: return launch_gemm<Cfg<64, 64, 2, 2, 4, 16, DT>, OT>(p, batch, st);
    // 128 x 96 (four waves stacked in M, each 32 x 96): N = 768 / 3072 are multiples of 96, and 4096 x 768 is exactly 256 workgroups
    case 11: return launch_gemm<Cfg<128, 96, 4, 1, 2, 16, DT>, OT>(p, batch, st);
    case 12: return launch_gemm<Cfg<96, 128, 1, 4, 2, 16, DT>, OT>(p, batch, st);
    default: return -2;
  }
}

// C[b] = act(alpha * A[b] . B[b]^T + bias) (+ residual[b]);  dims (int64): M N K lda ldb ldc sA sB sC batch tile dtype out_f32 act
static int gemm_entry(const void* A, const void* B, void* Cout, const float* bias, const void* residual, float alpha, float* stats,
                      const float* stat_shift, int accum, const int64_t* d, hipStream_t st);

DL4J_API int dl4j_gemm_nt(const void* A, const void* B, void* Cout, const float* bias, const void* residual, float alpha,
                          const int64_t* d, hipStream_t st) {
  return gemm_entry(A, B, Cout, bias, residual, alpha, nullptr, nullptr, 0, d, st);
}

// + per-M-tile column statistics of the rounded output ([mtiles][2][N], optionally of out - stat_shift) and in-place accumulation:
// the 1x1 / stride 1 convolution (x channels-last IS an [N*H*W, C] matrix) with its BatchNorm-statistics epilogue
DL4J_API int dl4j_gemm_nt_ex(const void* A, const void* B, void* Cout, const float* bias, const void* residual, float alpha,
                             float* stats, const float* stat_shift, int accum, const int64_t* d, hipStream_t st) {
  return gemm_entry(A, B, Cout, bias, residual, alpha, stats, stat_shift, accum, d, st);
}

static int gemm_entry(const void* A, const void* B, void* Cout, const float* bias, const void* residual, float alpha, float* stats,
                      const float* stat_shift, int accum, const int64_t* d, hipStream_t st) {
  GemmP p;
  p.A = A; p.B = B;
  p.epi.out = Cout; p.epi.residual = residual; p.epi.bias = bias; p.epi.stats = stats; p.epi.stat_shift = stat_shift;
  p.epi.alpha = alpha; p.epi.act = (int)d[13]; p.epi.accum = accum;
  p.M = (int)d[0]; p.N = (int)d[1]; p.K = (int)d[2];
  p.lda = d[3]; p.ldb = d[4]; p.ldc = d[5]; p.sA = d[6]; p.sB = d[7]; p.sC = d[8];
  p.gm = 0;
  const int batch = (int)d[9], tile = (int)d[10], dt = (int)d[11], of32 = (int)d[12];
  const int ch = dt == DT_F32 ? 4 : 8;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || batch <= 0 || batch > 65535) return -2;
  if (p.K % ch || p.lda % ch || p.ldb % ch || p.sA % ch || p.sB % ch) return -2;       // 16-byte aligned operand chunks
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return -2;
  if (dt == DT_BF16) return of32 ? dispatch_gemm<DT_BF16, OT_F32>(p, tile, batch, st) : dispatch_gemm<DT_BF16, OT_BF16>(p, tile, batch, st);
  if (dt == DT_F16) return of32 ? dispatch_gemm<DT_F16, OT_F32>(p, tile, batch, st) : dispatch_gemm<DT_F16, OT_F16>(p, tile, batch, st);
  if (dt == DT_F32) return dispatch_gemm<DT_F32, OT_F32>(p, tile, batch, st);
  return -2;
}
