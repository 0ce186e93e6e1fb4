// Direct MFMA convolution for 3-channel network stems (7x7 stride 2, 64 filters: the ImageNet stem of the zoo ResNet-50)
// on gfx950: forward with fused BatchNorm-statistics epilogue, and weight gradient. bf16 NHWC activations, fp32 accumulate.
//
// Reference: the stem is an ordinary ConvolutionLayer in zoo/model/ResNet50.java (conv2d = im2col + GEMM in libnd4j,
// convolutions_conv2d.cu:78-100, or the cuDNN helper). The generic implicit-GEMM kernels of conv_igemm.hip need C % 64 == 0;
// with C = 3 the reduction has only 147 terms and the op is bound by the 411 MB output / output-gradient tensor, so this
// kernel is built around data movement instead:
//  * a workgroup walks DOWN the output rows of one image and keeps the 7 live input rows in an 8-slot LDS ring — every
//    new output row stages only 2 new input rows (each input byte is read from memory ~once);
//  * input pixels are expanded 3 -> 4 channels while they are staged, so the patch of an output pixel is, per filter row,
//    32 contiguous bf16 (8 taps x 4 channels, zero weights on the pads) starting at a 16-byte aligned address: a patch
//    fragment of the 32x32x16 MFMA is ONE ds_read_b128 (forward), and for the weight gradient the [pixel][tap] operand
//    with the pixels on the k slots is two transposing ds_read_b64_tr_b16 reads of the same rows (pixel pitch 16 bytes);
//  * forward: the filters (A operand, 64 x 224 padded) are loop invariant in registers; the output tile goes through LDS
//    so the global stores are full 16-byte lines and the per-channel sum / sum of squares of the following BatchNorm are
//    accumulated from the same LDS tile (one partial row per workgroup, finished by bn_finalize_partials);
//  * weight gradient: dW^T tile [64 x 7 x 32] stays in the accumulators over all rows of the workgroup and is added
//    to a small fp32 scratch with atomics once per workgroup; a finalize kernel unpads into the OIHW gradient view and
//    re-zeroes the scratch.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

struct StemP {
  const bf16_t* x;      // [N][H][W][3]
  const bf16_t* w;      // [64][KH][8][4] zero padded (forward)
  const float* bias;    // [64] or null
  bf16_t* y;            // [N][OH][OW][64] (forward)
  float* stats;         // [nblocks][2][64] or null
  const bf16_t* dy;     // [N][OH][OW][64] (wgrad)
  float* dwp;           // [64][KH][32] fp32 scratch (wgrad, atomics)
  int N, H, W, OH, OW, pad, rpb, nrb;
  // FUSED weight gradient: dy is never materialised - it is the input gradient of the BatchNorm -> ReLU -> MaxPool(3x3 / 2) node
  // behind this convolution (bn_pool_fused.hip pass 2), recomputed per output row from the pooled gradient:
  //   dy[n,h,w,c] = gs[c] * dz + y[n,h,w,c] * k1[c] + k0[c],  dz = sum of pooled_dy over the covering windows whose argmax is (h, w)
  const bf16_t* bp_y;        // [N][OH][OW][64] the convolution's own (raw) output = BatchNorm input
  const bf16_t* bp_dy;       // [N][PH][PW][64] gradient of the pooled output
  const uint8_t* bp_idx;     // [N][PH][PW][64] window-local argmax tap (kh * 3 + kw), 255 = none
  const float* bp_coef;      // [3][64]: gs, k1, k0
  int PH, PW, ppad;
};

__device__ __forceinline__ int swz128(int pixel) { return ((pixel >> 1) & 1) << 2; }

// [64 pixels][64 channels] bf16 tile, pitch 128 B, chunks XOR-swizzled with swz128: operand with rows = channels cb..cb+31
// and k slots = pixels ks*16..+15
__device__ __forceinline__ bf16x8_t tr_frag128(const char* tile, int cb, int ks, int lane) {
  const int g = lane >> 4, h = g >> 1;
  const int q = (lane & 15) >> 2, pp = lane & 3;
  const int ch = cb + 16 * (g & 1) + 4 * pp;
  const int p0 = ks * 16 + 8 * h + q, p1 = p0 + 4;
  const int a0 = p0 * 128 + (((ch >> 3) ^ swz128(p0)) << 4) + (ch & 7) * 2;
  const int a1 = p1 * 128 + (((ch >> 3) ^ swz128(p1)) << 4) + (ch & 7) * 2;
  typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
  const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + a0));
  const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + a1));
  s16x8 r;
  r[0] = v0[0]; r[1] = v0[1]; r[2] = v0[2]; r[3] = v0[3];
  r[4] = v1[0]; r[5] = v1[1]; r[6] = v1[2]; r[7] = v1[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

// un-swizzled rows of `pitch` bytes per pixel (rows overlap: pitch 16 B, 64 B wide): operand with cols = the 32 taps
// starting at the row base and k slots = pixels ks*16..+15
__device__ __forceinline__ bf16x8_t tr_frag_lin(const char* base, int pitch, int ks, int lane) {
  const int g = lane >> 4, h = g >> 1;
  const int q = (lane & 15) >> 2, pp = lane & 3;
  const int chb = (16 * (g & 1) + 4 * pp) * 2;
  const int p0 = ks * 16 + 8 * h + q;
  typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
  const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(base + p0 * pitch + chb));
  const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(base + (p0 + 4) * pitch + chb));
  s16x8 r;
  r[0] = v0[0]; r[1] = v0[1]; r[2] = v0[2]; r[3] = v0[3];
  r[4] = v1[0]; r[5] = v1[1]; r[6] = v1[2]; r[7] = v1[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

template <int STR>
struct Ring {
  static constexpr int NPX = 127 * STR + 8;                   // input pixels under one 128-pixel output segment
  static constexpr int PITCH = ((NPX * 8 + 15) / 16) * 16;    // bytes per ring row (4 channels x bf16 per pixel)
};

// stage input row ih (zeros outside the image) into its ring slot, expanding 3 -> 4 channels (channel 3 stays zero)
template <int STR>
__device__ __forceinline__ void stage_row(char* ring, const StemP& p, int n, int ih, int tid) {
  char* row = ring + ((ih + 64) & 7) * Ring<STR>::PITCH;
  const bool rv = ih >= 0 && ih < p.H;
  const bf16_t* src = p.x + ((int64_t)n * p.H + (rv ? ih : 0)) * p.W * 3;
  for (int e = tid; e < Ring<STR>::NPX * 3; e += 256) {
    const int px = e / 3, c = e - px * 3;
    const int iw = px - p.pad;
    bf16_t v = 0;
    if (rv && iw >= 0 && iw < p.W) v = src[iw * 3 + c];
    *reinterpret_cast<bf16_t*>(row + (px * 4 + c) * 2) = v;
  }
}

// the same staging split in two so the global loads of the NEXT output row are in flight during the MFMA / epilogue
// phases of the current one (a workgroup is otherwise latency bound: load -> LDS -> barrier -> compute, twice per row)
template <int STR>
struct RowRegs {
  static constexpr int NL = (Ring<STR>::NPX * 3 + 255) / 256;
  bf16_t v[NL];
};
template <int STR>
__device__ __forceinline__ void prefetch_row(RowRegs<STR>& r, const StemP& p, int n, int ih, int tid) {
  const bool rv = ih >= 0 && ih < p.H;
  const bf16_t* src = p.x + ((int64_t)n * p.H + (rv ? ih : 0)) * p.W * 3;
#pragma unroll
  for (int i = 0; i < RowRegs<STR>::NL; ++i) {
    const int e = tid + i * 256;
    const int px = e / 3, c = e - px * 3;
    const int iw = px - p.pad;
    r.v[i] = (rv && e < Ring<STR>::NPX * 3 && iw >= 0 && iw < p.W) ? src[iw * 3 + c] : (bf16_t)0;
  }
}
template <int STR>
__device__ __forceinline__ void commit_row(char* ring, const RowRegs<STR>& r, int ih, int tid) {
  char* row = ring + ((ih + 64) & 7) * Ring<STR>::PITCH;
#pragma unroll
  for (int i = 0; i < RowRegs<STR>::NL; ++i) {
    const int e = tid + i * 256;
    const int px = e / 3, c = e - px * 3;
    if (e < Ring<STR>::NPX * 3) *reinterpret_cast<bf16_t*>(row + (px * 4 + c) * 2) = r.v[i];
  }
}

template <int KH, int KW, int STR>
__global__ void __launch_bounds__(256) conv_c3_fwd_kernel(const StemP p) {
  constexpr int PITCH = Ring<STR>::PITCH;
  constexpr int KS = KH * 2;               // 32 padded taps per filter row = two k16 steps
  constexpr int YP = 136;                  // output tile pitch (bytes): 64 channels + 8 B pad
  static_assert(128 * YP >= 32 * 2 * 64 * 4, "statistics exchange must fit the output tile");
  __shared__ __attribute__((aligned(16))) char ring[8 * PITCH];
  __shared__ __attribute__((aligned(16))) char sY[128 * YP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int n = blockIdx.x / p.nrb, rb = blockIdx.x - n * p.nrb;
  const int oh_a = rb * p.rpb;
  const int oh_b = oh_a + p.rpb < p.OH ? oh_a + p.rpb : p.OH;

  for (int i = tid; i < 8 * PITCH / 16; i += 256) reinterpret_cast<s16x8*>(ring)[i] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};

  bf16x8_t wf[2][KS];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[mb][ks] = *reinterpret_cast<const bf16x8_t*>(p.w + (int64_t)(mb * 32 + l31) * (KH * 32) + ks * 16 + lh * 8);

  __syncthreads();
  {
    const int ih0 = oh_a * STR - p.pad;
    for (int kh = 0; kh < KH; ++kh) stage_row<STR>(ring, p, n, ih0 + kh, tid);
  }
  const int px = wid * 32 + l31;
  const int ck = tid & 7;                  // this thread's 8-channel chunk in the store / statistics phase
  float st_s[8], st_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
  RowRegs<STR> nxt[STR];

  for (int oh = oh_a; oh < oh_b; ++oh) {
    __syncthreads();                       // ring rows staged; previous row's output tile fully consumed
    const int ih0 = oh * STR - p.pad;
    const bool more = oh + 1 < oh_b;
    if (more)
#pragma unroll
      for (int j = 0; j < STR; ++j) prefetch_row<STR>(nxt[j], p, n, ih0 + KH + j, tid);
    f32x16 acc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int g = 2 * ks + lh, kh = g >> 2, sub = g & 3;
      const char* row = ring + ((ih0 + kh + 64) & 7) * PITCH;
      const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(row + px * (STR * 8) + sub * 16);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], pf, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], pf, acc[1], 0, 0, 0);
    }
    // accumulator: row = channel (e&3) + 8*(e>>2) + 4*lh (+32*mb), column = this lane's pixel
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int ch0 = mb * 32 + 8 * g4 + 4 * lh;
        s16x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (short)f2bf(acc[mb][g4 * 4 + j] + (p.bias ? p.bias[ch0 + j] : 0.f));
        *reinterpret_cast<s16x4*>(sY + px * YP + ch0 * 2) = t;
      }
    __syncthreads();
    // full-line stores; the BatchNorm statistics are accumulated from the same registers (bf16-rounded values)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pxl = (tid >> 3) + i * 32;
      if (pxl < p.OW) {
        const s16x4 a = *reinterpret_cast<const s16x4*>(sY + pxl * YP + ck * 16);
        const s16x4 b = *reinterpret_cast<const s16x4*>(sY + pxl * YP + ck * 16 + 8);
        s16x8 v;
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        *reinterpret_cast<s16x8*>(p.y + (((int64_t)n * p.OH + oh) * p.OW + pxl) * 64 + ck * 8) = v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = bf2f((bf16_t)v[j]);
          st_s[j] += f;
          st_q[j] += f * f;
        }
      }
    }
    if (more)                              // the ring rows of this output row are dead (all waves are past the MFMA loop)
#pragma unroll
      for (int j = 0; j < STR; ++j) commit_row<STR>(ring, nxt[j], ih0 + KH + j, tid);
  }

  if (p.stats != nullptr) {
    // 32 threads share a channel chunk: reduce over them through LDS (the output tile is free now)
    __syncthreads();
    float* red = reinterpret_cast<float*>(sY);             // [32 pixel groups][2][64]
    const int pg = tid >> 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(pg * 2 + 0) * 64 + ck * 8 + j] = st_s[j];
      red[(pg * 2 + 1) * 64 + ck * 8 + j] = st_q[j];
    }
    __syncthreads();
    if (tid < 128) {
      float a = 0.f;
      for (int g = 0; g < 32; ++g) a += red[g * 128 + tid];
      p.stats[(int64_t)blockIdx.x * 128 + tid] = a;
    }
  }
}

template <int KH, int KW, int STR, bool FUSED = false>
__global__ void __launch_bounds__(256) conv_c3_wgrad_kernel(const StemP p) {
  constexpr int PITCH = Ring<STR>::PITCH;
  __shared__ __attribute__((aligned(16))) char ring[8 * PITCH];
  __shared__ __attribute__((aligned(16))) char sDY[2 * 64 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int n = blockIdx.x / p.nrb, rb = blockIdx.x - n * p.nrb;
  const int oh_a = rb * p.rpb;
  const int oh_b = oh_a + p.rpb < p.OH ? oh_a + p.rpb : p.OH;
  const int mb = wid & 1;                                  // 32-filter block of this wave
  const int kh0 = (wid >> 1) ? 4 : 0;                      // filter rows of this wave: {0..3} or {4..KH-1}
  const int nkh = (wid >> 1) ? KH - 4 : 4;

  for (int i = tid; i < 8 * PITCH / 16; i += 256) reinterpret_cast<s16x8*>(ring)[i] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();
  {
    const int ih0 = oh_a * STR - p.pad;
    for (int kh = 0; kh < KH; ++kh) stage_row<STR>(ring, p, n, ih0 + kh, tid);
  }
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  RowRegs<STR> nxt[STR];
  s16x8 dyr[4];
  // FUSED: raw operands of the recomputation, requested a row ahead like dy itself (the arithmetic runs after the MFMA loop)
  constexpr int FI = FUSED ? 4 : 1;
  s16x8 fy[FI], fd[FI][4];
  uint2 fi[FI][4];
  f32x4 cg0, cg1, ck10, ck11, ck00, ck01;
  if constexpr (FUSED) {
    const float* cf = p.bp_coef + (tid & 7) * 8;
    cg0 = *reinterpret_cast<const f32x4*>(cf); cg1 = *reinterpret_cast<const f32x4*>(cf + 4);
    ck10 = *reinterpret_cast<const f32x4*>(cf + 64); ck11 = *reinterpret_cast<const f32x4*>(cf + 68);
    ck00 = *reinterpret_cast<const f32x4*>(cf + 128); ck01 = *reinterpret_cast<const f32x4*>(cf + 132);
  }
  // window (a, b) of position (h, w): pooled row (h + ppad) / 2 - a, column (w + ppad) / 2 - b; covers the position iff the tap
  // kh = h + ppad - 2 * prow is in [0, 3) and the row exists (3x3 / 2: at most 2 x 2 windows)
  auto load_dy = [&](int oh) {
    if constexpr (FUSED) {
      const int hp = oh + p.ppad;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pxl = (tid >> 3) + i * 32;
        const int wp = pxl + p.ppad;
        fy[i] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (pxl < p.OW) fy[i] = *reinterpret_cast<const s16x8*>(p.bp_y + (((int64_t)n * p.OH + oh) * p.OW + pxl) * 64 + (tid & 7) * 8);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            int pr = (hp >> 1) - a, pc = (wp >> 1) - b;
            pr = pr < 0 ? 0 : (pr > p.PH - 1 ? p.PH - 1 : pr);       // clamped: the tap test in finish_dy rejects what does not cover
            pc = pc < 0 ? 0 : (pc > p.PW - 1 ? p.PW - 1 : pc);
            const int64_t o = (((int64_t)n * p.PH + pr) * p.PW + pc) * 64 + (tid & 7) * 8;
            fd[i][a * 2 + b] = *reinterpret_cast<const s16x8*>(p.bp_dy + o);
            fi[i][a * 2 + b] = *reinterpret_cast<const uint2*>(p.bp_idx + o);
          }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pxl = (tid >> 3) + i * 32;
        s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (pxl < p.OW) v = *reinterpret_cast<const s16x8*>(p.dy + (((int64_t)n * p.OH + oh) * p.OW + pxl) * 64 + (tid & 7) * 8);
        dyr[i] = v;
      }
    }
  };
  auto finish_dy = [&](int oh) {
    if constexpr (FUSED) {
      const int hp = oh + p.ppad;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pxl = (tid >> 3) + i * 32;
        const int wp = pxl + p.ppad;
        float dz[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dz[j] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int pr = (hp >> 1) - a, pc = (wp >> 1) - b;
            const int kh = hp - 2 * pr, kw = wp - 2 * pc;
            const bool cover = pr >= 0 && pr < p.PH && pc >= 0 && pc < p.PW && kh < 3 && kw < 3;
            const unsigned code = cover ? (unsigned)(kh * 3 + kw) : 0xFEu;      // 0xFE never occurs in idx (taps 0..8, 255 = none)
            const s16x8 d = fd[i][a * 2 + b];
            const uint2 am = fi[i][a * 2 + b];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              dz[j] += ((am.x >> (8 * j)) & 0xFFu) == code ? bf2f((bf16_t)d[j]) : 0.f;
              dz[4 + j] += ((am.y >> (8 * j)) & 0xFFu) == code ? bf2f((bf16_t)d[4 + j]) : 0.f;
            }
          }
        s16x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = (short)f2bf(cg0[j] * dz[j] + bf2f((bf16_t)fy[i][j]) * ck10[j] + ck00[j]);
          v[4 + j] = (short)f2bf(cg1[j] * dz[4 + j] + bf2f((bf16_t)fy[i][4 + j]) * ck11[j] + ck01[j]);
        }
        dyr[i] = pxl < p.OW ? v : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  };
  load_dy(oh_a);
  for (int oh = oh_a; oh < oh_b; ++oh) {
    const int ih0 = oh * STR - p.pad;
    const bool more = oh + 1 < oh_b;
    finish_dy(oh);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pxl = (tid >> 3) + i * 32, pl = pxl & 63;
      *reinterpret_cast<s16x8*>(sDY + (pxl >> 6) * 8192 + pl * 128 + (((tid & 7) ^ swz128(pl)) << 4)) = dyr[i];
    }
    __syncthreads();
    if (more) {                            // next row's operands travel while this row is multiplied
      load_dy(oh + 1);
#pragma unroll
      for (int j = 0; j < STR; ++j) prefetch_row<STR>(nxt[j], p, n, ih0 + KH + j, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8_t af = tr_frag128(sDY + (ks >> 2) * 8192, mb * 32, ks & 3, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nkh) {
          const char* row = ring + ((ih0 + kh0 + j + 64) & 7) * PITCH;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, tr_frag_lin(row, STR * 8, ks, lane), acc[j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (more)
#pragma unroll
      for (int j = 0; j < STR; ++j) commit_row<STR>(ring, nxt[j], ih0 + KH + j, tid);
  }

#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < nkh)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        atomicAdd(p.dwp + ((int64_t)k * KH + kh0 + j) * 32 + l31, acc[j][e]);
      }
}

// dw_out[k][c][kh][kw] (fp32, OIHW) <- scratch[k][kh][kw*4 + c]; the scratch is re-zeroed for the next iteration
__global__ void __launch_bounds__(256) conv_c3_wgrad_finalize_kernel(float* __restrict__ dwp, float* __restrict__ dw_out, int KH,
                                                                     int KW, int accumulate) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 64 * KH * 32) return;
  const int t = idx & 31, kh = (idx >> 5) % KH, k = (idx >> 5) / KH;
  const int kw = t >> 2, c = t & 3;
  const float v = dwp[idx];
  dwp[idx] = 0.f;
  if (kw < KW && c < 3) {
    float* d = dw_out + (((int64_t)k * 3 + c) * KH + kh) * KW + kw;
    *d = accumulate ? *d + v : v;
  }
}

static int stem_check(int N, int H, int W, int OH, int OW, int KH, int KW, int stride, int pad, int rpb) {
  if (KH != 7 || KW != 7 || stride != 2) return -2;
  if (N <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || OW > 128 || pad < 0 || pad > 7 || rpb <= 0) return -3;
  if (OH != (H + 2 * pad - KH) / stride + 1 || OW != (W + 2 * pad - KW) / stride + 1) return -4;
  if ((int64_t)N * H * W * 3 > 0x7fffffffLL || (int64_t)N * OH * OW * 64 > 0x7fffffffLL * 4) return -5;
  return 0;
}

}  // namespace

// x bf16 [N,H,W,3]; wpad bf16 [64][KH][8][4]; y bf16 [N,OH,OW,64]; stats fp32 [N*ceil(OH/rpb)][2][64] or null
DL4J_API int dl4j_conv_c3_fwd(const void* x, const void* wpad, const float* bias, void* y, float* stats, int N, int H, int W, int OH,
                              int OW, int KH, int KW, int stride, int pad, int rpb, hipStream_t st) {
  const int rc = stem_check(N, H, W, OH, OW, KH, KW, stride, pad, rpb);
  if (rc) return rc;
  StemP p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)wpad; p.bias = bias; p.y = (bf16_t*)y; p.stats = stats; p.dy = nullptr; p.dwp = nullptr;
  p.N = N; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.pad = pad; p.rpb = rpb; p.nrb = (OH + rpb - 1) / rpb;
  hipLaunchKernelGGL((conv_c3_fwd_kernel<7, 7, 2>), dim3((unsigned)(N * p.nrb)), dim3(256), 0, st, p);
  DL4J_RET_LAST();
}

// Weight gradient with the gradient tensor recomputed in the loader (StemP::bp_*): the stem convolution followed by the fused
// BatchNorm -> ReLU -> MaxPool(3x3, stride 2, pad ppad) node. y = the convolution's output [N,OH,OW,64]; pooled_dy / idx [N,PH,PW,64];
// coef fp32 [3][64] (16-byte aligned) as written by dl4j_bn_pool_bwd_reduce.
DL4J_API int dl4j_conv_c3_wgrad_bnpool(const void* y, const void* pooled_dy, const uint8_t* idx, const float* coef, int PH, int PW, int ppad,
                                       const void* x, float* scratch, float* dw_out, int N, int H, int W, int OH, int OW, int KH, int KW,
                                       int stride, int pad, int rpb, int accumulate, hipStream_t st) {
  const int rc = stem_check(N, H, W, OH, OW, KH, KW, stride, pad, rpb);
  if (rc) return rc;
  if (!y || !pooled_dy || !idx || !coef || PH <= 0 || PW <= 0 || ppad < 0 || ppad > 2) return -3;
  if (PH != (OH + 2 * ppad - 3) / 2 + 1 || PW != (OW + 2 * ppad - 3) / 2 + 1) return -4;
  if ((((uintptr_t)y | (uintptr_t)pooled_dy | (uintptr_t)coef) & 15) || ((uintptr_t)idx & 7)) return -2;
  StemP p;
  p.x = (const bf16_t*)x; p.w = nullptr; p.bias = nullptr; p.y = nullptr; p.stats = nullptr; p.dy = nullptr; p.dwp = scratch;
  p.N = N; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.pad = pad; p.rpb = rpb; p.nrb = (OH + rpb - 1) / rpb;
  p.bp_y = (const bf16_t*)y; p.bp_dy = (const bf16_t*)pooled_dy; p.bp_idx = idx; p.bp_coef = coef; p.PH = PH; p.PW = PW; p.ppad = ppad;
  hipLaunchKernelGGL((conv_c3_wgrad_kernel<7, 7, 2, true>), dim3((unsigned)(N * p.nrb)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(conv_c3_wgrad_finalize_kernel, dim3((64 * KH * 32 + 255) / 256), dim3(256), 0, st, scratch, dw_out, KH, KW,
                     accumulate);
  DL4J_RET_LAST();
}

// dy bf16 [N,OH,OW,64]; scratch fp32 [64*KH*32] all zeros on entry (left zeroed on exit); dw_out fp32 [64,3,KH,KW]
DL4J_API int dl4j_conv_c3_wgrad(const void* dy, const void* x, float* scratch, float* dw_out, int N, int H, int W, int OH, int OW, int KH,
                                int KW, int stride, int pad, int rpb, int accumulate, hipStream_t st) {
  const int rc = stem_check(N, H, W, OH, OW, KH, KW, stride, pad, rpb);
  if (rc) return rc;
  StemP p;
  p.x = (const bf16_t*)x; p.w = nullptr; p.bias = nullptr; p.y = nullptr; p.stats = nullptr; p.dy = (const bf16_t*)dy; p.dwp = scratch;
  p.N = N; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.pad = pad; p.rpb = rpb; p.nrb = (OH + rpb - 1) / rpb;
  hipLaunchKernelGGL((conv_c3_wgrad_kernel<7, 7, 2>), dim3((unsigned)(N * p.nrb)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(conv_c3_wgrad_finalize_kernel, dim3((64 * KH * 32 + 255) / 256), dim3(256), 0, st, scratch, dw_out, KH, KW,
                     accumulate);
  DL4J_RET_LAST();
}
