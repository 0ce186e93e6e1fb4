// fp32 weight gradient of a 2-D convolution on the exact-fp32 matrix instruction (v_mfma_f32_16x16x4_f32):
//   dW[k][r,s][c] += sum over output pixels p of dy[p][k] * x[pixel(p, r, s)][c]        NHWC fp32 activations, fp32 accumulation.
// DL4J's legal network dtypes are DOUBLE | FLOAT | HALF (NeuralNetConfiguration.java:1149-1154); the bf16 kernels of conv_igemm.hip
// are an extension, this is the FLOAT path (the reference computes gradW as im2col + cublasSgemm, convolutions_conv2dBP.cu:83-100).
// Both operands have the reduction index (the pixel) as their slow memory index, so tiles are staged [pixel][channel] in LDS and the
// MFMA operands (one float per lane: row = lane & 15, k = lane >> 4) are plain ds_read_b32 of 16 consecutive channels per pixel;
// the row pitch is padded by 16 floats so the two pixel rows a 32-lane group touches fall into different bank halves.
// Split over pixels across workgroups (fills the chip for the small late-stage layers), fp32 atomics into the pre-zeroed gradient.
#include "common.h"

struct WG32P {
  const float* dy; const float* x; float* dw;
  int N, H, W, C, OH, OW, K, R, S, sh, sw, ph, pw, dh, dw_;
  int P, ktiles, ctiles, splits, pps, use_atomic;
};

#define WG32_PIX 32
#define WG32_PITCH 80     // 64 channels + 16 floats padding

__global__ void __launch_bounds__(256) wgrad_f32_kernel(const WG32P p) {
  __shared__ __attribute__((aligned(16))) float sA[2][WG32_PIX][WG32_PITCH];
  __shared__ __attribute__((aligned(16))) float sB[2][WG32_PIX][WG32_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  int lid = blockIdx.x;
  const int RS = p.R * p.S;
  const int ct = lid % p.ctiles; lid /= p.ctiles;
  const int kt = lid % p.ktiles; lid /= p.ktiles;
  const int rs = lid % RS;
  const int split = lid / RS;
  const int k0 = kt * 64, c0 = ct * 64;
  const int r = rs / p.S, s = rs - r * p.S;
  const int pstart = split * p.pps;
  const int pend = pstart + p.pps < p.P ? pstart + p.pps : p.P;
  const int OHW = p.OH * p.OW;
  // staging: 32 pixels x 64 channels = 512 float4 per operand, 2 per thread: pixel = tid / 16 (+16), channel chunk = tid % 16
  const int lp = tid >> 4, lc = (tid & 15) * 4;
  f32x4 ra[2], rb[2];
  auto gload = [&](int pb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pp = pb + lp + 16 * i;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (pp < pend) {
        if (k0 + lc < p.K) va = *reinterpret_cast<const f32x4*>(p.dy + (int64_t)pp * p.K + k0 + lc);
        const int n = pp / OHW, rem = pp - n * OHW;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const int ih = oh * p.sh - p.ph + r * p.dh, iw = ow * p.sw - p.pw + s * p.dw_;
        if (((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W) && c0 + lc < p.C)
          vb = *reinterpret_cast<const f32x4*>(p.x + ((int64_t)(n * p.H + ih) * p.W + iw) * p.C + c0 + lc);
      }
      ra[i] = va; rb[i] = vb;
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4*>(&sA[buf][lp + 16 * i][lc]) = ra[i];
      *reinterpret_cast<f32x4*>(&sB[buf][lp + 16 * i][lc]) = rb[i];
    }
  };
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = (pend - pstart + WG32_PIX - 1) / WG32_PIX;
  const int l15 = lane & 15, lg = lane >> 4;
  if (nsteps > 0) {
    gload(pstart);
    swrite(0);
  }
  __syncthreads();
  for (int t = 0; t < nsteps; ++t) {
    const int cur = t & 1;
    if (t + 1 < nsteps) gload(pstart + (t + 1) * WG32_PIX);
#pragma unroll
    for (int kk = 0; kk < WG32_PIX / 4; ++kk) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = sA[cur][kk * 4 + lg][wm * 32 + i * 16 + l15];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = sB[cur][kk * 4 + lg][wn * 32 + j * 16 + l15];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nsteps) swrite(cur ^ 1);
    __syncthreads();
  }
  // C/D layout of the 16x16 block: col = lane & 15 (operand B row = input channel), row = (lane >> 4) * 4 + e (operand A row = k)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + wm * 32 + i * 16 + lg * 4 + e;
        const int c = c0 + wn * 32 + j * 16 + l15;
        if (k < p.K && c < p.C) {
          float* dst = p.dw + ((int64_t)k * RS + rs) * p.C + c;
          if (p.use_atomic) atomicAdd(dst, acc[i][j][e]);
          else *dst = acc[i][j][e];
        }
      }
}

__global__ void __launch_bounds__(256) wg32_zero_kernel(float* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
}

// geom = [N,H,W,C, OH,OW,K, R,S, sh,sw, ph,pw, dh,dw]; dw_out fp32 [K][R*S][C]
DL4J_API int dl4j_wgrad_f32(const float* dy, const float* x, float* dw_out, const int* g, int prezeroed, hipStream_t st) {
  WG32P p;
  p.dy = dy; p.x = x; p.dw = dw_out;
  p.N = g[0]; p.H = g[1]; p.W = g[2]; p.C = g[3]; p.OH = g[4]; p.OW = g[5]; p.K = g[6]; p.R = g[7]; p.S = g[8];
  p.sh = g[9]; p.sw = g[10]; p.ph = g[11]; p.pw = g[12]; p.dh = g[13]; p.dw_ = g[14];
  const int64_t P = (int64_t)p.N * p.OH * p.OW;
  if (p.C % 4 != 0 || p.K % 4 != 0 || P <= 0 || P > 0x7fffffff) return -2;       // 16-byte chunks inside a pixel's channel run
  if ((reinterpret_cast<uintptr_t>(dy) & 15) || (reinterpret_cast<uintptr_t>(x) & 15)) return -2;
  p.P = (int)P;
  p.ktiles = (p.K + 63) / 64; p.ctiles = (p.C + 63) / 64;
  const int tiles = p.ktiles * p.ctiles * p.R * p.S;
  // 40 KB of LDS and 256 threads per workgroup: 3-4 resident per CU; one resident round over the chip
  int splits = (DL4J_NUM_CU * 3) / tiles;
  const int maxs = (p.P + 255) / 256;
  if (splits > maxs) splits = maxs;
  if (splits < 1) splits = 1;
  int pps = (p.P + splits - 1) / splits;
  pps = (pps + WG32_PIX - 1) / WG32_PIX * WG32_PIX;
  splits = (p.P + pps - 1) / pps;
  p.splits = splits; p.pps = pps; p.use_atomic = splits > 1;
  if (p.use_atomic && !prezeroed) {
    const int64_t n = (int64_t)p.K * p.R * p.S * p.C;
    hipLaunchKernelGGL(wg32_zero_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, dw_out, n);
  }
  hipLaunchKernelGGL(wgrad_f32_kernel, dim3(tiles * splits), dim3(256), 0, st, p);
  DL4J_RET_LAST();
}
