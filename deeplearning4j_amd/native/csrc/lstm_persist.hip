// Persistent LSTM sequence kernels for gfx950: ONE launch runs all T time steps of a layer (bf16 activations / weights, fp32 cell
// math). The recurrent weights stay resident in LDS for the whole sequence, the cell state stays in registers, and the workgroups
// hand the hidden state to each other through memory with one grid barrier per step.
//
// Reference: the reference has no such kernel — its GPU path is cuDNN's fused RNN (platform/cudnn/lstmLayer.cu:439) and its own code
// is a per-step loop of mmul + ~20 elementwise ops (helpers/impl/lstmLayer.cpp:200-420, LSTMHelpers.java:211-369). The per-step
// native loop of rnn_seq.hip (one small GEMM + one cell kernel per step) is launch/latency bound at char-RNN sizes: a [64 x 512] x
// [512 x 2048] step fills 32 workgroups for ~10 us. Here a step costs one hidden-state broadcast (64 KB through L2), 32 MFMAs per
// wave and one barrier.
//
// Forward decomposition (n hidden units, mb <= 128 rows, DL4J gate order a | f | o | g):
//   * workgroup w owns hidden units [8w, 8w+8) = 32 gate columns {g*n + 8w + k}; their recurrent weights (32 rows of RW4 [4n, n],
//     K-contiguous) are loaded into LDS once;
//   * every step: all 256 threads copy h_{t-1} [mb, n] into LDS (coalesced 16-byte loads), wave v multiplies rows [32v, 32v+32):
//     gates^T = Wslice (A operand, M = 32 gate columns) x h^T (B operand, N = 32 batch rows), v_mfma_f32_32x32x16_bf16 over K = n.
//     In the 32x32 accumulator layout lane l then holds, for batch row l%32, ALL FOUR gates (register group i = gate) of units
//     4*(l/32)+j: the cell update runs in registers with no transpose, c_t never leaves the lane, and h_t leaves as one 8-byte store;
//   * h_t slices are written with agent-scope (write-through) stores; after s_waitcnt + workgroup barrier one thread adds to the
//     grid counter and spins (with s_sleep) until all workgroups of the step have arrived; an agent-scope acquire fence then drops
//     stale L2 lines before the next step's loads (8 XCDs with private L2s). The spin is bounded: on timeout the kernel sets an
//     abort flag that every workgroup observes, so the grid always drains.
// Requirements checked on the host: n % 8 == 0, n / 8 <= 256 workgroups (all co-resident on 256 CUs), mb <= 128, LDS fits.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __expf(-2.f * fabsf(x));
  const float t = (1.f - e) / (1.f + e);
  return x < 0.f ? -t : t;
}

struct LstmFwdP {
  const bf16_t* xw;        // [T, mb, 4n] input projection + bias
  const bf16_t* rw4;       // [4n, ld_rw] recurrent weights, row = gate column
  int64_t ld_rw;
  const bf16_t* h0;        // [mb, n]
  const bf16_t* c0;        // [mb, n]
  const float* pf;         // peepholes [n] or null (all three or none)
  const float* po;
  const float* pg;
  const float* maskT;      // [T, mb] or null
  bf16_t* H;               // [T, mb, n]
  bf16_t* Cs;              // [T, mb, n]
  bf16_t* SV;              // [T, mb, 6n] or null
  unsigned* bar;           // [0] arrival counter, [1] abort flag; zeroed before the launch
  int T, mb, n, reverse;
};

#define SPIN_LIMIT (1 << 22)
#define LL_BATCH 16           // polled 8-byte words in flight per thread

// one grid barrier; returns false when the kernel must abort. All threads of the block call it.
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1, spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
      if (++spins > SPIN_LIMIT) {
        __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    *s_flag = ok;
  }
  __syncthreads();
  const bool ok = *s_flag != 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok;
}

__device__ __forceinline__ void st_agent_u64(void* p, uint64_t v) {
  __hip_atomic_store((uint64_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t ld_agent_u64(const void* p) {
  return __hip_atomic_load((const uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// "Data is its own flag" hand-off (the idea of NCCL's LL protocol without the flag word): the exchange buffer is filled with a
// bit pattern that valid data never takes (four bf16 NaNs with all bits set), producers publish 8-byte words with single-copy
// atomic write-through stores, consumers poll the words they need until none of them is the sentinel. One store + one load
// latency per step instead of store-ack, counter add, counter poll and an L2 invalidate.
#define LL_SENTINEL 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ uint64_t ll_clean(uint64_t w) { return w == LL_SENTINEL ? 0x7FC07FC07FC07FC0ull : w; }

__device__ __forceinline__ uint64_t pack4(float a, float b, float c, float d) {
  return (uint64_t)f2bf(a) | ((uint64_t)f2bf(b) << 16) | ((uint64_t)f2bf(c) << 32) | ((uint64_t)f2bf(d) << 48);
}
__device__ __forceinline__ void unpack4(uint64_t v, float o[4]) {
  o[0] = bf2f((bf16_t)(v & 0xffff)); o[1] = bf2f((bf16_t)((v >> 16) & 0xffff));
  o[2] = bf2f((bf16_t)((v >> 32) & 0xffff)); o[3] = bf2f((bf16_t)(v >> 48));
}

#define FWD_THREADS 512
__global__ void __launch_bounds__(FWD_THREADS) lstm_persist_fwd_kernel(LstmFwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_flag;
  const int n = p.n, mb = p.mb;
  const int ldw = n + 8;                                // padded LDS row length (elements): rows start 16 bytes apart mod 128
  bf16_t* Wl = (bf16_t*)smem;                           // [32][ldw]
  bf16_t* hl = Wl + 32 * ldw;                           // [mbp][ldw], mbp = mb rounded up to 32
  const int mbp = (mb + 31) & ~31;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 8;
  const int nch = n >> 3;                               // 16-byte chunks per row

  // recurrent weight slice -> LDS (once); row gc = g*8 + k is gate column g*n + u0 + k
  for (int i = tid; i < 32 * nch; i += FWD_THREADS) {
    const int gc = i / nch, ch = i - gc * nch;
    const int64_t col = (int64_t)(gc >> 3) * n + u0 + (gc & 7);
    *(uint4*)(Wl + gc * ldw + ch * 8) = *(const uint4*)(p.rw4 + col * p.ld_rw + ch * 8);
  }
  // rows beyond mb stay zero for the whole kernel
  for (int i = tid; i < (mbp - mb) * nch; i += FWD_THREADS) {
    const int r = mb + i / nch, ch = i % nch;
    *(uint4*)(hl + r * ldw + ch * 8) = make_uint4(0, 0, 0, 0);
  }

  const int row = wave * 32 + (lane & 31);              // batch row of this lane (valid when < mb and wave tile in range)
  const int kq = lane >> 5;                             // 0 / 1: which 8-wide K chunk of a 16-wide MFMA step, and unit quad
  const int uq = u0 + kq * 4;                           // first of this lane's 4 hidden units
  const bool tile_on = wave * 32 < mb;
  const bool row_on = tile_on && row < mb;
  float c_state[4] = {0.f, 0.f, 0.f, 0.f};
  if (row_on) unpack4(*(const uint64_t*)(p.c0 + (int64_t)row * n + uq), c_state);
  float wpf[4] = {0, 0, 0, 0}, wpo[4] = {0, 0, 0, 0}, wpg[4] = {0, 0, 0, 0};
  if (p.pf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { wpf[j] = p.pf[uq + j]; wpo[j] = p.po[uq + j]; wpg[j] = p.pg[uq + j]; }
  }

  const bf16_t* hprev = p.h0;
  for (int s = 0; s < p.T; ++s) {
    const int t = (p.reverse & 1) ? p.T - 1 - s : s;
    // (1) issue this lane's input-projection loads early: 4 gates x 4 units
    uint64_t xwv[4] = {0, 0, 0, 0};
    float mk = 1.f;
    if (row_on) {
      const bf16_t* xr = p.xw + ((int64_t)t * mb + row) * 4 * n + uq;
#pragma unroll
      for (int g = 0; g < 4; ++g) xwv[g] = *(const uint64_t*)(xr + (int64_t)g * n);
      if (p.maskT) mk = p.maskT[(int64_t)t * mb + row];
    }
    // (2) h_{t-1} -> LDS. Step 0 reads h0 (complete before the launch); later steps poll H[t_prev], which the launch pre-filled
    //     with the sentinel and which the owning workgroups overwrite word by word.
    int dead = 0;
    if (s == 0 || (p.reverse & 2)) {                      // bit 1 of `reverse`: timing experiment, no hand-off (every step reads h0)
      if (s == 0 || !(p.reverse & 8))
      for (int i = tid; i < mb * nch; i += FWD_THREADS) {
        const int r = i / nch, ch = i - r * nch;
        *(uint4*)(hl + r * ldw + ch * 8) = *(const uint4*)(hprev + (int64_t)r * n + ch * 8);
      }
    } else {
      const int wpr = n >> 2;                             // 8-byte words per row
      const int words = mb * wpr;
      for (int base = 0; base < words; base += FWD_THREADS * LL_BATCH) {
        uint64_t v[LL_BATCH];
#pragma unroll
        for (int i = 0; i < LL_BATCH; ++i) {
          const int w = base + i * FWD_THREADS + tid;
          v[i] = w < words ? ld_agent_u64(hprev + (int64_t)w * 4) : 0ull;
        }
#pragma unroll
        for (int i = 0; i < LL_BATCH; ++i) {
          const int w = base + i * FWD_THREADS + tid;
          if (w >= words) continue;
          int spins = 0;
          while (v[i] == LL_SENTINEL) {
            __builtin_amdgcn_s_sleep(1);
            v[i] = ld_agent_u64(hprev + (int64_t)w * 4);
            if (++spins > SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(p.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
              __hip_atomic_store(p.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              v[i] = 0ull;                                 // abort: results are void, the host sees the flag
              dead = 1;
              break;
            }
          }
          const int r = w / wpr, c4 = w - r * wpr;
          *(uint64_t*)(hl + r * ldw + c4 * 4) = v[i];
        }
      }
    }
    if (__syncthreads_or(dead)) return;                   // a timed-out poll ends the whole workgroup; the others follow via the flag
    // (3) gates^T tile = Wslice x h^T over K = n
    if (tile_on) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const bf16_t* ap = Wl + (lane & 31) * ldw + kq * 8;
      const bf16_t* bp = hl + (wave * 32 + (lane & 31)) * ldw + kq * 8;
      if (!(p.reverse & 16))
#pragma unroll 4
      for (int k = 0; k < n; k += 16) {
        const bf16x8_t a = *(const bf16x8_t*)(ap + k);
        const bf16x8_t b = *(const bf16x8_t*)(bp + k);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
      // (4) cell update in registers: acc[4*g + j] = pre-activation of gate g, unit uq + j, batch row `row`
      if (row_on) {
        float xa[4], xf[4], xo[4], xg[4];
        unpack4(xwv[0], xa); unpack4(xwv[1], xf); unpack4(xwv[2], xo); unpack4(xwv[3], xg);
        float hn[4], cn[4], va[4], vf[4], vo[4], vg[4], vac[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float cp = c_state[j];
          const float za = acc[j] + xa[j];
          float zf = acc[4 + j] + xf[j], zo = acc[8 + j] + xo[j], zg = acc[12 + j] + xg[j];
          if (p.pf) { zf += cp * wpf[j]; zg += cp * wpg[j]; }
          const float a = tanh_fast(za), f = sigm(zf), g = sigm(zg);
          const float c = f * cp + g * a;
          if (p.pf) zo += c * wpo[j];
          const float o = sigm(zo), ac = tanh_fast(c);
          va[j] = a; vf[j] = f; vo[j] = o; vg[j] = g; vac[j] = ac;
          cn[j] = c;
          hn[j] = o * ac * mk;
          c_state[j] = c * mk;
        }
        const int64_t o_idx = ((int64_t)t * mb + row) * n + uq;
        st_agent_u64(p.H + o_idx, ll_clean(pack4(hn[0], hn[1], hn[2], hn[3])));
        if (!(p.reverse & 4)) *(uint64_t*)(p.Cs + o_idx) = pack4(c_state[0], c_state[1], c_state[2], c_state[3]);
        if (p.SV && !(p.reverse & 4)) {
          bf16_t* sv = p.SV + ((int64_t)t * mb + row) * 6 * n + uq;
          *(uint64_t*)(sv) = pack4(va[0], va[1], va[2], va[3]);
          *(uint64_t*)(sv + n) = pack4(vf[0], vf[1], vf[2], vf[3]);
          *(uint64_t*)(sv + 2 * (int64_t)n) = pack4(vo[0], vo[1], vo[2], vo[3]);
          *(uint64_t*)(sv + 3 * (int64_t)n) = pack4(vg[0], vg[1], vg[2], vg[3]);
          *(uint64_t*)(sv + 4 * (int64_t)n) = pack4(cn[0], cn[1], cn[2], cn[3]);
          *(uint64_t*)(sv + 5 * (int64_t)n) = pack4(vac[0], vac[1], vac[2], vac[3]);
        }
      }
    }
    if (!(p.reverse & 2)) hprev = p.H + (int64_t)t * mb * n;
    // (5) no grid barrier: the next step polls the data. Only the LDS copy of h is shared inside the workgroup.
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------------------------------ backward
// Same ownership (workgroup w = hidden units [8w, 8w+8)). Per step, walking the sequence backwards:
//   dh_t   = E_t + dG_{t+1} . RW4^T           (recurrent part: K = 4n, streamed from the dG rows every workgroup wrote last step)
//   dG_t, dc_{t-1} = cell_backward(dh_t, dc_t, saved gate activations)          (registers; dc never leaves the lane)
// The recurrent product uses v_mfma_f32_16x16x32_bf16 with A = this workgroup's 8 rows of RW4^T [n, 4n] from LDS (rows 8..15 of the
// MFMA tile repeat them) and B = 16 batch rows of dG_{t+1} loaded straight from global memory (each workgroup has to see all 4n gate
// gradients, 256 KB per step at n = 512, mb = 64: that L2 stream is the cost of the step). In the 16x16 accumulator layout lane l
// holds batch row l%16 and units 4*(l/16)+j, so lanes 0..31 of every wave own (row, 4 units) exactly as in the forward kernel.
struct LstmBwdP {
  const bf16_t* E;         // [T, mb, n] dL/dh_t from the layers above
  const bf16_t* rw4t;      // [n, ld_rwt] row j = the 4n gate weights leaving hidden unit j
  int64_t ld_rwt;
  const bf16_t* SV;        // [T, mb, 6n] saved a, f, o, g, c, tanh(c)
  const bf16_t* Cs;        // [T, mb, n]
  const bf16_t* c0;        // [mb, n]
  const float* pf;
  const float* po;
  const float* pg;
  const float* maskT;      // [T, mb] or null
  bf16_t* dG;              // [T, mb, 4n]
  float* dpeep;            // [3, n] accumulated with atomics, or null
  unsigned* bar;
  int T, mb, n, reverse;
  const bf16_t* dG_next;   // gate gradients of the step AFTER the last one of this launch (stepwise launches), or null
  const float* dc_in;      // [mb, n] dL/dc flowing into the last step of this launch, or null (= 0)
  float* dc_out;           // [mb, n] dL/dc leaving the first step of this launch, or null
};

#define BWD_TILES 2         // 16-row batch tiles per wave: mb <= 4 waves * 2 * 16 = 128

#define BWD_THREADS 1024    // 16 waves: wave = (K quarter, batch tile). All of a step's dG loads are in flight at once.
__global__ void __launch_bounds__(BWD_THREADS) lstm_persist_bwd_kernel(LstmBwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_flag;
  const int n = p.n, mb = p.mb, K = 4 * p.n;
  const int ldw = K + 32;                                 // row stride = 16 banks mod 64
  bf16_t* Wb = (bf16_t*)smem;                             // [8][ldw]
  f32x4* red = (f32x4*)(Wb + 8 * ldw);                    // [BWD_TILES][3][4 tiles][64 lanes] split-K partials of K quarters 1..3
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile0 = wave & 3, kq4 = wave >> 2;            // batch tile (of 16 rows) and K quarter of this wave
  const int u0 = blockIdx.x * 8;
  const int kch = K >> 3;
  for (int i = tid; i < 8 * kch; i += BWD_THREADS) {
    const int r = i / kch, ch = i - r * kch;
    *(uint4*)(Wb + r * ldw + ch * 8) = *(const uint4*)(p.rw4t + (int64_t)(u0 + r) * p.ld_rwt + ch * 8);
  }
  __syncthreads();

  const int bcol = lane & 15, q = lane >> 4;
  const bool unit_lane = q < 2 && kq4 == 0;               // lanes 32..63 hold the duplicate MFMA rows; K quarter 0 owns the cell math
  const int uq = u0 + (q & 1) * 4;
  float dc_state[BWD_TILES][4];
  float apf[4] = {0, 0, 0, 0}, apo[4] = {0, 0, 0, 0}, apg[4] = {0, 0, 0, 0};
  float wpf[4] = {0, 0, 0, 0}, wpo[4] = {0, 0, 0, 0}, wpg[4] = {0, 0, 0, 0};
#pragma unroll
  for (int tt = 0; tt < BWD_TILES; ++tt) {
    const int row = (tile0 + 4 * tt) * 16 + bcol;
#pragma unroll
    for (int j = 0; j < 4; ++j) dc_state[tt][j] = (p.dc_in && unit_lane && row < mb) ? p.dc_in[(int64_t)row * n + uq + j] : 0.f;
  }
  if (p.pf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { wpf[j] = p.pf[uq + j]; wpo[j] = p.po[uq + j]; wpg[j] = p.pg[uq + j]; }
  }
  const int Kq = K >> 2;                                  // columns per K quarter (= n)
  const bf16_t* ap = Wb + (bcol & 7) * ldw + kq4 * Kq + q * 8;

  for (int s = p.T - 1; s >= 0; --s) {
    const int t = p.reverse ? p.T - 1 - s : s;
    const int t_prev = p.reverse ? t + 1 : t - 1;
    const int t_next = p.reverse ? t - 1 : t + 1;         // the step handled in the previous iteration of this loop
    const bf16_t* nextp = s < p.T - 1 ? p.dG + (int64_t)t_next * mb * K : p.dG_next;
    f32x4 acc[BWD_TILES];
    uint64_t ev[BWD_TILES], sva[BWD_TILES], svf[BWD_TILES], svo[BWD_TILES], svg[BWD_TILES], svc[BWD_TILES], svac[BWD_TILES], cpv[BWD_TILES];
    float mk[BWD_TILES];
#pragma unroll
    for (int tt = 0; tt < BWD_TILES; ++tt) {
      const int tile = tile0 + 4 * tt;
      acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      mk[tt] = 1.f;
      ev[tt] = sva[tt] = svf[tt] = svo[tt] = svg[tt] = svc[tt] = svac[tt] = cpv[tt] = 0ull;
      if (tile * 16 >= mb) continue;
      const int row = tile * 16 + bcol;
      const bool row_ok = row < mb;
      // saved activations / eps / previous cell state of this lane's (row, 4 units): issued before the recurrent product
      if (unit_lane && row_ok) {
        const int64_t hidx = ((int64_t)t * mb + row) * n + uq;
        ev[tt] = *(const uint64_t*)(p.E + hidx);
        const bf16_t* sv = p.SV + ((int64_t)t * mb + row) * 6 * n + uq;
        sva[tt] = *(const uint64_t*)(sv); svf[tt] = *(const uint64_t*)(sv + n); svo[tt] = *(const uint64_t*)(sv + 2 * (int64_t)n);
        svg[tt] = *(const uint64_t*)(sv + 3 * (int64_t)n); svc[tt] = *(const uint64_t*)(sv + 4 * (int64_t)n);
        svac[tt] = *(const uint64_t*)(sv + 5 * (int64_t)n);
        cpv[tt] = s == 0 ? *(const uint64_t*)(p.c0 + (int64_t)row * n + uq) : *(const uint64_t*)(p.Cs + ((int64_t)t_prev * mb + row) * n + uq);
        if (p.maskT) mk[tt] = p.maskT[(int64_t)t * mb + row];
      }
      if (nextp != nullptr) {
        const bf16_t* bp = nextp + (int64_t)(row_ok ? row : 0) * K + kq4 * Kq + q * 8;
        // no divergent branch around MFMAs (they do not honour EXEC): rows beyond mb read row 0 and are zeroed by a select
        const bf16x8_t zero = {};
#pragma unroll 16
        for (int k = 0; k < Kq; k += 32) {
          const bf16x8_t a = *(const bf16x8_t*)(ap + k);
          bf16x8_t b = *(const bf16x8_t*)(bp + k);
          b = row_ok ? b : zero;
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[tt], 0, 0, 0);
        }
        if (kq4 > 0) red[((tt * 3 + (kq4 - 1)) * 4 + tile0) * 64 + lane] = acc[tt];
      }
    }
    if (nextp != nullptr) __syncthreads();                // split-K partials of quarters 1..3 are in LDS
#pragma unroll
    for (int tt = 0; tt < BWD_TILES; ++tt) {
      const int tile = tile0 + 4 * tt;
      if (tile * 16 >= mb) continue;
      const int row = tile * 16 + bcol;
      if (unit_lane && row < mb) {
        if (nextp != nullptr) {
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            const f32x4 part = red[((tt * 3 + kk) * 4 + tile0) * 64 + lane];
            acc[tt] += part;
          }
        }
        float e4[4], a4[4], f4[4], o4[4], g4[4], c4[4], ac4[4], cp4[4];
        unpack4(ev[tt], e4); unpack4(sva[tt], a4); unpack4(svf[tt], f4); unpack4(svo[tt], o4); unpack4(svg[tt], g4); unpack4(svc[tt], c4);
        unpack4(svac[tt], ac4); unpack4(cpv[tt], cp4);
        float dza[4], dzf[4], dzo[4], dzg[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d_h = (e4[j] + acc[tt][j]) * mk[tt];
          const float d_cn = dc_state[tt][j] * mk[tt];
          const float a = a4[j], f = f4[j], o = o4[j], g = g4[j], ac = ac4[j], cp = cp4[j];
          dzo[j] = d_h * ac * o * (1.f - o);
          float dc = d_cn + d_h * o * (1.f - ac * ac);
          if (p.pf) dc += dzo[j] * wpo[j];
          dza[j] = dc * g * (1.f - a * a);
          dzf[j] = dc * cp * f * (1.f - f);
          dzg[j] = dc * a * g * (1.f - g);
          float dcp = dc * f;
          if (p.pf) {
            dcp += dzf[j] * wpf[j] + dzg[j] * wpg[j];
            apf[j] += dzf[j] * cp; apo[j] += dzo[j] * c4[j]; apg[j] += dzg[j] * cp;
          }
          dc_state[tt][j] = dcp;
        }
        bf16_t* dg = p.dG + ((int64_t)t * mb + row) * K + uq;
        st_agent_u64(dg, pack4(dza[0], dza[1], dza[2], dza[3]));
        st_agent_u64(dg + n, pack4(dzf[0], dzf[1], dzf[2], dzf[3]));
        st_agent_u64(dg + 2 * (int64_t)n, pack4(dzo[0], dzo[1], dzo[2], dzo[3]));
        st_agent_u64(dg + 3 * (int64_t)n, pack4(dzg[0], dzg[1], dzg[2], dzg[3]));
      }
    }
    if (s > 0) {
      if (!grid_barrier(p.bar, (unsigned)(p.T - s) * gridDim.x, &s_flag)) return;
    }
  }
  if (p.dc_out) {
#pragma unroll
    for (int tt = 0; tt < BWD_TILES; ++tt) {
      const int row = (tile0 + 4 * tt) * 16 + bcol;
      if (unit_lane && row < mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) p.dc_out[(int64_t)row * n + uq + j] = dc_state[tt][j];
    }
  }
  // peephole gradients: reduce over the 16 batch columns of the lane group, then one atomic per unit and wave
  if (p.pf && p.dpeep) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float vf = apf[j], vo = apo[j], vg = apg[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { vf += __shfl_xor(vf, o, 64); vo += __shfl_xor(vo, o, 64); vg += __shfl_xor(vg, o, 64); }
      if (unit_lane && bcol == 0) {
        atomicAdd(p.dpeep + uq + j, vf);
        atomicAdd(p.dpeep + n + uq + j, vo);
        atomicAdd(p.dpeep + 2 * n + uq + j, vg);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------ step kernels
// One launch per time step, operands straight from L2 into MFMA registers (no LDS staging, so nothing waits on a fill):
// every wave issues all of its loads at once (input projection, weights, h or dG), which leaves ONE exposed memory latency per
// step. 16 waves = K quarter x batch tile; quarters 1..3 hand their partial accumulators to quarter 0 through LDS, which runs the
// cell math of the persistent kernels. Ordering between steps is the kernel boundary (the cheapest all-to-all hand-off on this
// part: MI355X_MICROARCH price list "boundary" 1.45 us vs "allgather" 3-5 us, confirmed in profiles/r2_lstm_persistent.md).
#define STEP_THREADS 1024
#define STEP_KSTEPS 8        // K-steps of 16 per wave held in registers at once (n <= 512 in one pass)

__global__ void __launch_bounds__(STEP_THREADS) lstm_step_fwd_kernel(LstmFwdP p) {
  __shared__ __attribute__((aligned(16))) f32x16 red[3 * 4 * 64];       // partial accumulators of K quarters 1..3 [3][4 tiles][64 lanes]
  const int n = p.n, mb = p.mb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = wave & 3, kp = wave >> 2;
  const int u0 = blockIdx.x * 8;
  const int row = tile * 32 + (lane & 31);
  const int kq = lane >> 5;
  const int uq = u0 + kq * 4;
  const bool tile_on = tile * 32 < mb;
  const bool row_ok = row < mb;
  const bool cell_lane = tile_on && row_ok && kp == 0;
  const int Kq = n >> 2;
  // (1) everything this wave needs, issued back to back
  uint64_t xwv[4] = {0, 0, 0, 0}, cpv = 0;
  float mk = 1.f;
  if (cell_lane) {
    const bf16_t* xr = p.xw + (int64_t)row * 4 * n + uq;
#pragma unroll
    for (int g = 0; g < 4; ++g) xwv[g] = *(const uint64_t*)(xr + (int64_t)g * n);
    cpv = *(const uint64_t*)(p.c0 + (int64_t)row * n + uq);
    if (p.maskT) mk = p.maskT[row];
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (tile_on) {
    const int gc = lane & 31;                                            // gate column of the slice: g*8 + k
    const bf16_t* ap = p.rw4 + ((int64_t)(gc >> 3) * n + u0 + (gc & 7)) * p.ld_rw + kp * Kq + kq * 8;
    const bf16_t* bp = p.h0 + (int64_t)(row_ok ? row : 0) * n + kp * Kq + kq * 8;
    const bf16x8_t zero = {};
    for (int k0 = 0; k0 < Kq; k0 += 16 * STEP_KSTEPS) {
      bf16x8_t a[STEP_KSTEPS], b[STEP_KSTEPS];
#pragma unroll
      for (int i = 0; i < STEP_KSTEPS; ++i) {
        const int k = k0 + 16 * i;
        a[i] = k < Kq ? *(const bf16x8_t*)(ap + k) : zero;
        b[i] = (k < Kq && row_ok) ? *(const bf16x8_t*)(bp + k) : zero;
      }
#pragma unroll
      for (int i = 0; i < STEP_KSTEPS; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc, 0, 0, 0);
    }
    if (kp > 0) red[((kp - 1) * 4 + tile) * 64 + lane] = acc;
  }
  __syncthreads();
  if (!cell_lane) return;
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const f32x16 part = red[(kk * 4 + tile) * 64 + lane];
    acc += part;
  }
  float c_state[4];
  unpack4(cpv, c_state);
  float wpf[4] = {0, 0, 0, 0}, wpo[4] = {0, 0, 0, 0}, wpg[4] = {0, 0, 0, 0};
  if (p.pf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { wpf[j] = p.pf[uq + j]; wpo[j] = p.po[uq + j]; wpg[j] = p.pg[uq + j]; }
  }
  float xa[4], xf[4], xo[4], xg[4];
  unpack4(xwv[0], xa); unpack4(xwv[1], xf); unpack4(xwv[2], xo); unpack4(xwv[3], xg);
  float hn[4], cn[4], va[4], vf[4], vo[4], vg[4], vac[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float cp = c_state[j];
    const float za = acc[j] + xa[j];
    float zf = acc[4 + j] + xf[j], zo = acc[8 + j] + xo[j], zg = acc[12 + j] + xg[j];
    if (p.pf) { zf += cp * wpf[j]; zg += cp * wpg[j]; }
    const float a = tanh_fast(za), f = sigm(zf), g = sigm(zg);
    const float c = f * cp + g * a;
    if (p.pf) zo += c * wpo[j];
    const float o = sigm(zo), ac = tanh_fast(c);
    va[j] = a; vf[j] = f; vo[j] = o; vg[j] = g; vac[j] = ac;
    cn[j] = c;
    hn[j] = o * ac * mk;
    c_state[j] = c * mk;
  }
  const int64_t o_idx = (int64_t)row * n + uq;
  *(uint64_t*)(p.H + o_idx) = pack4(hn[0], hn[1], hn[2], hn[3]);
  *(uint64_t*)(p.Cs + o_idx) = pack4(c_state[0], c_state[1], c_state[2], c_state[3]);
  if (p.SV) {
    bf16_t* sv = p.SV + (int64_t)row * 6 * n + uq;
    *(uint64_t*)(sv) = pack4(va[0], va[1], va[2], va[3]);
    *(uint64_t*)(sv + n) = pack4(vf[0], vf[1], vf[2], vf[3]);
    *(uint64_t*)(sv + 2 * (int64_t)n) = pack4(vo[0], vo[1], vo[2], vo[3]);
    *(uint64_t*)(sv + 3 * (int64_t)n) = pack4(vg[0], vg[1], vg[2], vg[3]);
    *(uint64_t*)(sv + 4 * (int64_t)n) = pack4(cn[0], cn[1], cn[2], cn[3]);
    *(uint64_t*)(sv + 5 * (int64_t)n) = pack4(vac[0], vac[1], vac[2], vac[3]);
  }
}

// Backward step: p.E / p.SV / p.dG / p.maskT point at THIS step; p.c0 = cell state that fed it; p.dG_next = gate gradients of the
// following step (or null); p.dc_in / p.dc_out = dL/dc between launches (fp32).
__global__ void __launch_bounds__(STEP_THREADS) lstm_step_bwd_kernel(LstmBwdP p) {
  __shared__ __attribute__((aligned(16))) f32x4 red[BWD_TILES * 3 * 4 * 64];
  const int n = p.n, mb = p.mb, K = 4 * p.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile0 = wave & 3, kq4 = wave >> 2;
  const int u0 = blockIdx.x * 8;
  const int bcol = lane & 15, q = lane >> 4;
  const bool unit_lane = q < 2 && kq4 == 0;
  const int uq = u0 + (q & 1) * 4;
  const int Kq = K >> 2;
  const bf16_t* ap = p.rw4t + (int64_t)(u0 + (bcol & 7)) * p.ld_rwt + kq4 * Kq + q * 8;
  f32x4 acc[BWD_TILES];
  uint64_t ev[BWD_TILES], sva[BWD_TILES], svf[BWD_TILES], svo[BWD_TILES], svg[BWD_TILES], svc[BWD_TILES], svac[BWD_TILES], cpv[BWD_TILES];
  float mk[BWD_TILES], dcn[BWD_TILES][4];
#pragma unroll
  for (int tt = 0; tt < BWD_TILES; ++tt) {
    const int tile = tile0 + 4 * tt;
    acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    mk[tt] = 1.f;
    ev[tt] = sva[tt] = svf[tt] = svo[tt] = svg[tt] = svc[tt] = svac[tt] = cpv[tt] = 0ull;
#pragma unroll
    for (int j = 0; j < 4; ++j) dcn[tt][j] = 0.f;
    if (tile * 16 >= mb) continue;
    const int row = tile * 16 + bcol;
    const bool row_ok = row < mb;
    if (unit_lane && row_ok) {
      const int64_t hidx = (int64_t)row * n + uq;
      ev[tt] = *(const uint64_t*)(p.E + hidx);
      const bf16_t* sv = p.SV + (int64_t)row * 6 * n + uq;
      sva[tt] = *(const uint64_t*)(sv); svf[tt] = *(const uint64_t*)(sv + n); svo[tt] = *(const uint64_t*)(sv + 2 * (int64_t)n);
      svg[tt] = *(const uint64_t*)(sv + 3 * (int64_t)n); svc[tt] = *(const uint64_t*)(sv + 4 * (int64_t)n);
      svac[tt] = *(const uint64_t*)(sv + 5 * (int64_t)n);
      cpv[tt] = *(const uint64_t*)(p.c0 + hidx);
      if (p.maskT) mk[tt] = p.maskT[row];
      if (p.dc_in) {
        const f32x4 d = *(const f32x4*)(p.dc_in + hidx);
#pragma unroll
        for (int j = 0; j < 4; ++j) dcn[tt][j] = d[j];
      }
    }
    if (p.dG_next != nullptr) {
      const bf16_t* bp = p.dG_next + (int64_t)(row_ok ? row : 0) * K + kq4 * Kq + q * 8;
      const bf16x8_t zero = {};
#pragma unroll 16
      for (int k = 0; k < Kq; k += 32) {
        const bf16x8_t a = *(const bf16x8_t*)(ap + k);
        bf16x8_t b = *(const bf16x8_t*)(bp + k);
        b = row_ok ? b : zero;
        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[tt], 0, 0, 0);
      }
      if (kq4 > 0) red[((tt * 3 + (kq4 - 1)) * 4 + tile0) * 64 + lane] = acc[tt];
    }
  }
  __syncthreads();
  float apf[4] = {0, 0, 0, 0}, apo[4] = {0, 0, 0, 0}, apg[4] = {0, 0, 0, 0};
  float wpf[4] = {0, 0, 0, 0}, wpo[4] = {0, 0, 0, 0}, wpg[4] = {0, 0, 0, 0};
  if (p.pf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { wpf[j] = p.pf[uq + j]; wpo[j] = p.po[uq + j]; wpg[j] = p.pg[uq + j]; }
  }
#pragma unroll
  for (int tt = 0; tt < BWD_TILES; ++tt) {
    const int tile = tile0 + 4 * tt;
    if (tile * 16 >= mb) continue;
    const int row = tile * 16 + bcol;
    if (!(unit_lane && row < mb)) continue;
    if (p.dG_next != nullptr) {
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const f32x4 part = red[((tt * 3 + kk) * 4 + tile0) * 64 + lane];
        acc[tt] += part;
      }
    }
    float e4[4], a4[4], f4[4], o4[4], g4[4], c4[4], ac4[4], cp4[4];
    unpack4(ev[tt], e4); unpack4(sva[tt], a4); unpack4(svf[tt], f4); unpack4(svo[tt], o4); unpack4(svg[tt], g4); unpack4(svc[tt], c4);
    unpack4(svac[tt], ac4); unpack4(cpv[tt], cp4);
    float dza[4], dzf[4], dzo[4], dzg[4];
    f32x4 dco;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d_h = (e4[j] + acc[tt][j]) * mk[tt];
      const float d_cn = dcn[tt][j] * mk[tt];
      const float a = a4[j], f = f4[j], o = o4[j], g = g4[j], ac = ac4[j], cp = cp4[j];
      dzo[j] = d_h * ac * o * (1.f - o);
      float dc = d_cn + d_h * o * (1.f - ac * ac);
      if (p.pf) dc += dzo[j] * wpo[j];
      dza[j] = dc * g * (1.f - a * a);
      dzf[j] = dc * cp * f * (1.f - f);
      dzg[j] = dc * a * g * (1.f - g);
      float dcp = dc * f;
      if (p.pf) {
        dcp += dzf[j] * wpf[j] + dzg[j] * wpg[j];
        apf[j] += dzf[j] * cp; apo[j] += dzo[j] * c4[j]; apg[j] += dzg[j] * cp;
      }
      dco[j] = dcp;
    }
    const int64_t hidx = (int64_t)row * n + uq;
    if (p.dc_out) *(f32x4*)(p.dc_out + hidx) = dco;
    bf16_t* dg = p.dG + (int64_t)row * K + uq;
    *(uint64_t*)(dg) = pack4(dza[0], dza[1], dza[2], dza[3]);
    *(uint64_t*)(dg + n) = pack4(dzf[0], dzf[1], dzf[2], dzf[3]);
    *(uint64_t*)(dg + 2 * (int64_t)n) = pack4(dzo[0], dzo[1], dzo[2], dzo[3]);
    *(uint64_t*)(dg + 3 * (int64_t)n) = pack4(dzg[0], dzg[1], dzg[2], dzg[3]);
  }
  if (p.pf && p.dpeep) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float vf = apf[j], vo = apo[j], vg = apg[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { vf += __shfl_xor(vf, o, 64); vo += __shfl_xor(vo, o, 64); vg += __shfl_xor(vg, o, 64); }
      if (unit_lane && bcol == 0) {
        atomicAdd(p.dpeep + uq + j, vf);
        atomicAdd(p.dpeep + n + uq + j, vo);
        atomicAdd(p.dpeep + 2 * n + uq + j, vg);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------ diagonal kernels
// A stack of L layers over T steps is a wavefront: (layer l, step t) needs (l, t - 1) and (l - 1, t). All cells of one anti-diagonal
// d = l + t are independent, so ONE launch runs them all (blockIdx.y = layer) and a pass over the stack is T + L - 1 launches instead
// of L * T step launches plus the per-chunk projection GEMMs, stream forks and events of the chunked pipeline. The input projection
// of the upper layers is folded into the step as a second K segment: gates = b + H_below[t] . W + h[t-1] . RW (the first layer's
// projection does not depend on the recurrence and stays one GEMM over the whole sequence). Backward mirrors it:
// dL/dh(l, t) = E_ext + dG(l, t+1) . RW^T + dG(l+1, t) . W(l+1)^T, diagonals processed from (L-1, T-1) down.
#define DIAG_MAXL 4
template <typename T>
struct LstmDiagLayerF {
  const T* xw;        // [T, mb, 4n] projection + bias of the whole sequence (first layer), or null
  const T* rw4;       // [4n, ld_rw]
  int64_t ld_rw;
  const T* w2;        // [4n, ld_w2] input weights, row = gate column (layers without xw)
  int64_t ld_w2;
  const T* x2;        // [T, mb, nIn] activations of the layer below
  const float* bias;       // [4n]
  const T* h0;
  const T* c0;
  const float* pf;
  const float* po;
  const float* pg;
  T* H;
  T* Cs;
  T* SV;
  int nIn;
};
template <typename E>
struct LstmDiagFwdP {
  LstmDiagLayerF<E> lay[DIAG_MAXL];
  const float* maskT;      // [T, mb] or null: masked steps emit h = 0 and carry c = 0 (as the step kernels)
  int nl, d, T, mb, n, reverse;   // reverse: processing step s runs time index T - 1 - s
};

// element-type traits of the diagonal kernels: bf16 (MFMA 32x32x16 / 16x16x32, 16-byte loads = 8 K values per lane) and fp32 (MFMA
// 32x32x2 / 16x16x4: a 16-byte load holds 4 K values per lane and feeds 4 MFMAs; A and B use the same permuted K order)
struct DiagBf16 {
  typedef bf16_t T;
  typedef bf16x8_t vec;
  static constexpr int FSTEP = 16, FLANE = 8, BSTEP = 32, BLANE = 8, FK = 4, BK = STEP_KSTEPS;
  static __device__ __forceinline__ void mfma_f(f32x16& acc, const vec& a, const vec& b) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); }
  static __device__ __forceinline__ void mfma_b(f32x4& acc, const vec& a, const vec& b) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); }
  static __device__ __forceinline__ void ld4(const T* p, float o[4]) { unpack4(*(const uint64_t*)p, o); }
  static __device__ __forceinline__ void st4(T* p, float a, float b, float c, float d) { *(uint64_t*)p = pack4(a, b, c, d); }
};
struct DiagF32 {
  typedef float T;
  typedef f32x4 vec;
  static constexpr int FSTEP = 8, FLANE = 4, BSTEP = 16, BLANE = 4, FK = 8, BK = STEP_KSTEPS;
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

// One workgroup = 8 hidden units (32 gate columns) x 32 batch rows of one (layer, step) cell: one 32x32 MFMA tile whose K range (h[t-1],
// then the layer below's h[t]) is split over the DIAG_WAVES waves in 16-wide k-steps; the partial tiles meet in LDS and wave 0 runs the cell.
// (The 4 batch tiles x 4 K quarters form of lstm_step_fwd_kernel leaves half of its waves idle at mb = 64 and moves twice the bytes
// per workgroup; this kernel is bound by operand bytes per CU like the backward one.)
#define DIAG_WAVES 8         // waves per workgroup = K slices (char-RNN fit: 16 waves 7.64 ms, 8 waves 7.22, 4 waves 7.25)
#define DIAG_LOG2W 3
#define DIAG_THREADS (64 * DIAG_WAVES)
#define DIAG_FK 4            // k-steps per wave held in registers at once (n <= 512 in one pass)
template <typename TR>
__global__ void __launch_bounds__(DIAG_THREADS) lstm_diag_fwd_kernel(LstmDiagFwdP<typename TR::T> p) {
  typedef typename TR::T T;
  typedef typename TR::vec vec;
  // partial tiles as [wave][4 register groups][64 lanes] of 16 bytes: a lane stride of 16 bytes keeps the 128-bit LDS accesses free of
  // bank conflicts (as [wave][lane] of 64 bytes they ran at 6x the conflict-free cycle count, profiles/r3_pmc_sq_lstm.md - though
  // the hand-off is short enough that the step time did not change)
  __shared__ __attribute__((aligned(16))) f32x4 red[(DIAG_WAVES - 1) * 4 * 64];
  const int li = blockIdx.z, sp = p.d - li;                             // sp: processing step
  if (li >= p.nl || sp < 0 || sp >= p.T) return;                        // uniform per workgroup
  const int t = p.reverse ? p.T - 1 - sp : sp, tp = p.reverse ? t + 1 : t - 1;
  const LstmDiagLayerF<T>& q = p.lay[li];
  const int n = p.n, mb = p.mb;
  const int64_t hn = (int64_t)mb * n;
  const T* hprev = sp == 0 ? q.h0 : q.H + tp * hn;
  const T* cprev = sp == 0 ? q.c0 : q.Cs + tp * hn;
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
  const int gc = lane & 31;
  const vec zero = {};
  const int64_t wrow = (int64_t)(gc >> 3) * n + u0 + (gc & 7);
  {
    const int S = n / TR::FSTEP, s0 = (wave * S) >> DIAG_LOG2W, s1 = ((wave + 1) * S) >> DIAG_LOG2W;       // this wave's 16-wide k-steps of segment 1
    const T* ap = q.rw4 + wrow * q.ld_rw + kq * TR::FLANE;
    const T* bp = hprev + (int64_t)(row_ok ? row : 0) * n + kq * TR::FLANE;
    for (int sb = s0; sb < s1; sb += TR::FK) {
      vec a[TR::FK], b[TR::FK];
#pragma unroll
      for (int i = 0; i < TR::FK; ++i) {
        const int k = (sb + i) * TR::FSTEP;
        a[i] = sb + i < s1 ? *(const vec*)(ap + k) : zero;
        b[i] = (sb + i < s1 && row_ok) ? *(const vec*)(bp + k) : zero;
      }
#pragma unroll
      for (int i = 0; i < TR::FK; ++i) TR::mfma_f(acc, a[i], b[i]);
    }
  }
  if (q.xw == nullptr) {
    const int S = q.nIn / TR::FSTEP, s0 = (wave * S) >> DIAG_LOG2W, s1 = ((wave + 1) * S) >> DIAG_LOG2W;
    const T* ap = q.w2 + wrow * q.ld_w2 + kq * TR::FLANE;
    const T* bp = q.x2 + ((int64_t)t * mb + (row_ok ? row : 0)) * q.nIn + kq * TR::FLANE;
    for (int sb = s0; sb < s1; sb += TR::FK) {
      vec a[TR::FK], b[TR::FK];
#pragma unroll
      for (int i = 0; i < TR::FK; ++i) {
        const int k = (sb + i) * TR::FSTEP;
        a[i] = sb + i < s1 ? *(const vec*)(ap + k) : zero;
        b[i] = (sb + i < s1 && row_ok) ? *(const vec*)(bp + k) : zero;
      }
#pragma unroll
      for (int i = 0; i < TR::FK; ++i) TR::mfma_f(acc, a[i], b[i]);
    }
  }
  // wave 0 owns the cell: its operands travel while the other waves hand their partial tiles over
  float mk = 1.f;
  float xg4[4][4], c_state[4] = {0.f, 0.f, 0.f, 0.f};         // gate pre-activations from outside the K loop: xw (first layer) or the bias
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) xg4[g][j] = 0.f;
  if (cell_lane) {
    if (q.xw != nullptr) {
      const T* xr = q.xw + ((int64_t)t * mb + row) * 4 * n + uq;
#pragma unroll
      for (int g = 0; g < 4; ++g) TR::ld4(xr + (int64_t)g * n, xg4[g]);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *(const f32x4*)(q.bias + (int64_t)g * n + uq);
#pragma unroll
        for (int j = 0; j < 4; ++j) xg4[g][j] = b4[j];
      }
    }
    TR::ld4(cprev + (int64_t)row * n + uq, c_state);
    if (p.maskT) mk = p.maskT[(int64_t)t * mb + row];
  }
  if (wave > 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) red[((wave - 1) * 4 + g) * 64 + lane] = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
  }
  __syncthreads();
  if (!cell_lane) return;
#pragma unroll 3        // (fully unrolled the 15 partial tiles are all read ahead: 240 registers, scratch spills)
  for (int w = 0; w < DIAG_WAVES - 1; ++w) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 part = red[(w * 4 + g) * 64 + lane];
      acc[4 * g] += part[0]; acc[4 * g + 1] += part[1]; acc[4 * g + 2] += part[2]; acc[4 * g + 3] += part[3];
    }
  }
  float wpf[4] = {0, 0, 0, 0}, wpo[4] = {0, 0, 0, 0}, wpg[4] = {0, 0, 0, 0};
  if (q.pf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { wpf[j] = q.pf[uq + j]; wpo[j] = q.po[uq + j]; wpg[j] = q.pg[uq + j]; }
  }
  float hn4[4], cn[4], va[4], vf[4], vo[4], vg[4], vac[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float cp = c_state[j];
    const float za = acc[j] + xg4[0][j];
    float zf = acc[4 + j] + xg4[1][j], zo = acc[8 + j] + xg4[2][j], zg = acc[12 + j] + xg4[3][j];
    if (q.pf) { zf += cp * wpf[j]; zg += cp * wpg[j]; }
    const float a = tanh_fast(za), f = sigm(zf), g = sigm(zg);
    const float c = f * cp + g * a;
    if (q.pf) zo += c * wpo[j];
    const float o = sigm(zo), ac = tanh_fast(c);
    va[j] = a; vf[j] = f; vo[j] = o; vg[j] = g; vac[j] = ac;
    cn[j] = c;
    hn4[j] = o * ac * mk;
    c_state[j] = c * mk;
  }
  const int64_t o_idx = (int64_t)t * hn + (int64_t)row * n + uq;
  TR::st4(q.H + o_idx, hn4[0], hn4[1], hn4[2], hn4[3]);
  TR::st4(q.Cs + o_idx, c_state[0], c_state[1], c_state[2], c_state[3]);
  if (q.SV) {
    T* sv = q.SV + ((int64_t)t * mb + row) * 6 * n + uq;
    TR::st4(sv, va[0], va[1], va[2], va[3]);
    TR::st4(sv + n, vf[0], vf[1], vf[2], vf[3]);
    TR::st4(sv + 2 * (int64_t)n, vo[0], vo[1], vo[2], vo[3]);
    TR::st4(sv + 3 * (int64_t)n, vg[0], vg[1], vg[2], vg[3]);
    TR::st4(sv + 4 * (int64_t)n, cn[0], cn[1], cn[2], cn[3]);
    TR::st4(sv + 5 * (int64_t)n, vac[0], vac[1], vac[2], vac[3]);
  }
}

template <typename T>
struct LstmDiagLayerB {
  const T* E;         // [T, mb, n] dL/dh from outside the stack (top layer), or null
  const T* rw4t;      // [n, ld_rwt]
  int64_t ld_rwt;
  const T* wct;       // input weights of the layer ABOVE [n, ld_wct] (row = this layer's unit), or null
  int64_t ld_wct;
  const T* dG_above;  // [T, mb, 4n] gate gradients of the layer above
  const T* SV;
  const T* Cs;
  const T* c0;
  const float* pf;
  const float* po;
  const float* pg;
  T* dG;              // [T, mb, 4n]
  float* dpeep;
  float* dc_ws;            // [2, mb, n]
};
template <typename E>
struct LstmDiagBwdP {
  LstmDiagLayerB<E> lay[DIAG_MAXL];
  const float* maskT;
  int nl, d, T, mb, n, reverse;
};

// One workgroup = 16 hidden units x 16 batch rows of one (layer, step) cell: ONE 16x16 MFMA tile whose K range (4n gate gradients of
// the next step, then 4n of the layer above) is split over the DIAG_WAVES waves; the partial tiles meet in LDS and wave 0 runs the cell
// backward. Sized by what bounds this kernel - operand bytes per CU (every byte comes from L2, a wave-load of 1 KB costs ~60 cycles of
// L1 fill): the 8-unit x all-rows form of lstm_step_bwd_kernel moved 768 KB per workgroup, with half of every MFMA's rows duplicates,
// and took 32 us per diagonal; this one moves 256 KB per workgroup over twice as many workgroups.
template <typename TR>
__global__ void __launch_bounds__(DIAG_THREADS) lstm_diag_bwd_kernel(LstmDiagBwdP<typename TR::T> p) {
  typedef typename TR::T T;
  typedef typename TR::vec vec;
  __shared__ __attribute__((aligned(16))) f32x4 red[(DIAG_WAVES - 1) * 64];
  const int kk_l = blockIdx.z;                                          // 0 = top layer
  const int li = p.nl - 1 - kk_l, sp = p.T - 1 - (p.d - kk_l);          // sp: processing step of the forward pass, highest first
  if (li < 0 || sp < 0 || sp >= p.T) return;
  const int t = p.reverse ? p.T - 1 - sp : sp, tp = p.reverse ? t + 1 : t - 1, tn = p.reverse ? t - 1 : t + 1;
  const LstmDiagLayerB<T>& L = p.lay[li];
  const int n = p.n, mb = p.mb, K = 4 * p.n;
  const int64_t hn = (int64_t)mb * n;
  const T* dG_next = sp == p.T - 1 ? nullptr : L.dG + 4 * tn * hn;
  const T* dG_up = L.dG_above ? L.dG_above + 4 * t * hn : nullptr;
  const float* dc_in = sp == p.T - 1 ? nullptr : L.dc_ws + (int64_t)((p.T - 2 - sp) & 1) * hn;
  float* dc_out = L.dc_ws + (int64_t)((p.T - 1 - sp) & 1) * hn;
  const T* cprev = sp == 0 ? L.c0 : L.Cs + tp * hn;
  const T* SVt = L.SV + 6 * t * hn;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 16, r0 = blockIdx.y * 16;
  const int bcol = lane & 15, q = lane >> 4;
  const int row = r0 + bcol;
  const bool row_ok = row < mb;
  const int uq = u0 + q * 4;
  const int Kw = K >> DIAG_LOG2W;                                                // K range of this wave (a multiple of 32: n % 128 == 0)
  const bool any_k = dG_next != nullptr || dG_up != nullptr;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const vec zero = {};
  const int64_t koff = (int64_t)wave * Kw + q * TR::BLANE;
  // both segments' operands are requested before the first MFMA (Kw / 32 k-steps each: 4 at n = 512)
  if (dG_next != nullptr) {
    const T* ap = L.rw4t + (int64_t)(u0 + bcol) * L.ld_rwt + koff;
    const T* bp = dG_next + (int64_t)(row_ok ? row : 0) * K + koff;
    for (int k0 = 0; k0 < Kw; k0 += TR::BSTEP * TR::BK) {
      vec a[TR::BK], b[TR::BK];
#pragma unroll
      for (int i = 0; i < TR::BK; ++i) {
        const int k = k0 + TR::BSTEP * i;
        a[i] = k < Kw ? *(const vec*)(ap + k) : zero;
        b[i] = (k < Kw && row_ok) ? *(const vec*)(bp + k) : zero;
      }
#pragma unroll
      for (int i = 0; i < TR::BK; ++i) TR::mfma_b(acc, a[i], b[i]);
    }
  }
  if (dG_up != nullptr) {
    const T* ap = L.wct + (int64_t)(u0 + bcol) * L.ld_wct + koff;
    const T* bp = dG_up + (int64_t)(row_ok ? row : 0) * K + koff;
    for (int k0 = 0; k0 < Kw; k0 += TR::BSTEP * TR::BK) {
      vec a[TR::BK], b[TR::BK];
#pragma unroll
      for (int i = 0; i < TR::BK; ++i) {
        const int k = k0 + TR::BSTEP * i;
        a[i] = k < Kw ? *(const vec*)(ap + k) : zero;
        b[i] = (k < Kw && row_ok) ? *(const vec*)(bp + k) : zero;
      }
#pragma unroll
      for (int i = 0; i < TR::BK; ++i) TR::mfma_b(acc, a[i], b[i]);
    }
  }
  // wave 0 owns the cell: its operands travel while the other waves hand their partial tiles over
  float e4[4] = {0, 0, 0, 0}, a4[4] = {0, 0, 0, 0}, f4[4] = {0, 0, 0, 0}, o4[4] = {0, 0, 0, 0}, g4[4] = {0, 0, 0, 0}, c4[4] = {0, 0, 0, 0};
  float ac4[4] = {0, 0, 0, 0}, cp4[4] = {0, 0, 0, 0};
  float dcn[4] = {0.f, 0.f, 0.f, 0.f};
  float mk = 1.f;
  const bool cell_lane = wave == 0 && row_ok;
  if (cell_lane) {
    const int64_t hidx = (int64_t)row * n + uq;
    if (L.E) TR::ld4(L.E + t * hn + hidx, e4);
    const T* sv = SVt + (int64_t)row * 6 * n + uq;
    TR::ld4(sv, a4); TR::ld4(sv + n, f4); TR::ld4(sv + 2 * (int64_t)n, o4); TR::ld4(sv + 3 * (int64_t)n, g4);
    TR::ld4(sv + 4 * (int64_t)n, c4); TR::ld4(sv + 5 * (int64_t)n, ac4);
    TR::ld4(cprev + hidx, cp4);
    if (p.maskT) mk = p.maskT[(int64_t)t * mb + row];
    if (dc_in) {
      const f32x4 d = *(const f32x4*)(dc_in + hidx);
#pragma unroll
      for (int j = 0; j < 4; ++j) dcn[j] = d[j];
    }
  }
  if (any_k && wave > 0) red[(wave - 1) * 64 + lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  if (any_k) {
#pragma unroll
    for (int w = 0; w < DIAG_WAVES - 1; ++w) {
      const f32x4 part = red[w * 64 + lane];
      acc += part;
    }
  }
  float apf[4] = {0, 0, 0, 0}, apo[4] = {0, 0, 0, 0}, apg[4] = {0, 0, 0, 0};
  float wpf[4] = {0, 0, 0, 0}, wpo[4] = {0, 0, 0, 0}, wpg[4] = {0, 0, 0, 0};
  if (L.pf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { wpf[j] = L.pf[uq + j]; wpo[j] = L.po[uq + j]; wpg[j] = L.pg[uq + j]; }
  }
  if (row_ok) {
    float dza[4], dzf[4], dzo[4], dzg[4];
    f32x4 dco;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d_h = (e4[j] + acc[j]) * mk;
      const float d_cn = dcn[j] * mk;
      const float a = a4[j], f = f4[j], o = o4[j], g = g4[j], ac = ac4[j], cp = cp4[j];
      dzo[j] = d_h * ac * o * (1.f - o);
      float dc = d_cn + d_h * o * (1.f - ac * ac);
      if (L.pf) dc += dzo[j] * wpo[j];
      dza[j] = dc * g * (1.f - a * a);
      dzf[j] = dc * cp * f * (1.f - f);
      dzg[j] = dc * a * g * (1.f - g);
      float dcp = dc * f;
      if (L.pf) {
        dcp += dzf[j] * wpf[j] + dzg[j] * wpg[j];
        apf[j] += dzf[j] * cp; apo[j] += dzo[j] * c4[j]; apg[j] += dzg[j] * cp;
      }
      dco[j] = dcp;
    }
    const int64_t hidx = (int64_t)row * n + uq;
    *(f32x4*)(dc_out + hidx) = dco;
    T* dg = L.dG + 4 * t * hn + (int64_t)row * K + uq;
    TR::st4(dg, dza[0], dza[1], dza[2], dza[3]);
    TR::st4(dg + n, dzf[0], dzf[1], dzf[2], dzf[3]);
    TR::st4(dg + 2 * (int64_t)n, dzo[0], dzo[1], dzo[2], dzo[3]);
    TR::st4(dg + 3 * (int64_t)n, dzg[0], dzg[1], dzg[2], dzg[3]);
  }
  if (L.pf && L.dpeep) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float vf = apf[j], vo = apo[j], vg = apg[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { vf += __shfl_xor(vf, o, 64); vo += __shfl_xor(vo, o, 64); vg += __shfl_xor(vg, o, 64); }
      if (bcol == 0) {
        atomicAdd(L.dpeep + uq + j, vf);
        atomicAdd(L.dpeep + n + uq + j, vo);
        atomicAdd(L.dpeep + 2 * n + uq + j, vg);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------ fp32 step kernels
// The same one-launch-per-step decomposition for fp32 networks (the reference's default data type): v_mfma_f32_32x32x2_f32 forward,
// v_mfma_f32_16x16x4_f32 backward. A 16-byte operand load holds 4 consecutive K values per lane and feeds 4 MFMAs; A and B use the same
// (permuted) K order, so no shuffling is needed.
struct LstmFwdF32 {
  const float* xw; const float* rw4; int64_t ld_rw; const float* h0; const float* c0;
  const float* pf; const float* po; const float* pg; const float* mask;
  float* H; float* Cs; float* SV;
  int mb, n;
};

__global__ void __launch_bounds__(STEP_THREADS) lstm_step_fwd_f32_kernel(LstmFwdF32 p) {
  __shared__ __attribute__((aligned(16))) f32x16 red[3 * 4 * 64];
  const int n = p.n, mb = p.mb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = wave & 3, kp = wave >> 2;
  const int u0 = blockIdx.x * 8;
  const int row = tile * 32 + (lane & 31);
  const int kq = lane >> 5;
  const int uq = u0 + kq * 4;
  const bool tile_on = tile * 32 < mb;
  const bool row_ok = row < mb;
  const bool cell_lane = tile_on && row_ok && kp == 0;
  const int Kq = n >> 2;
  f32x4 xwv[4], cpv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 4; ++g) xwv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mk = 1.f;
  if (cell_lane) {
    const float* xr = p.xw + (int64_t)row * 4 * n + uq;
#pragma unroll
    for (int g = 0; g < 4; ++g) xwv[g] = *(const f32x4*)(xr + (int64_t)g * n);
    cpv = *(const f32x4*)(p.c0 + (int64_t)row * n + uq);
    if (p.mask) mk = p.mask[row];
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (tile_on) {
    const int gc = lane & 31;
    const float* ap = p.rw4 + ((int64_t)(gc >> 3) * n + u0 + (gc & 7)) * p.ld_rw + kp * Kq + kq * 4;
    const float* bp = p.h0 + (int64_t)(row_ok ? row : 0) * n + kp * Kq + kq * 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < Kq; k0 += 8 * STEP_KSTEPS) {
      f32x4 a[STEP_KSTEPS], b[STEP_KSTEPS];
#pragma unroll
      for (int i = 0; i < STEP_KSTEPS; ++i) {
        const int k = k0 + 8 * i;
        a[i] = k < Kq ? *(const f32x4*)(ap + k) : zero;
        b[i] = (k < Kq && row_ok) ? *(const f32x4*)(bp + k) : zero;
      }
#pragma unroll
      for (int i = 0; i < STEP_KSTEPS; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][c], b[i][c], acc, 0, 0, 0);
    }
    if (kp > 0) red[((kp - 1) * 4 + tile) * 64 + lane] = acc;
  }
  __syncthreads();
  if (!cell_lane) return;
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const f32x16 part = red[(kk * 4 + tile) * 64 + lane];
    acc += part;
  }
  f32x4 hn, cs, va, vf, vo, vg, vc, vac;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float cp = cpv[j];
    const float za = acc[j] + xwv[0][j];
    float zf = acc[4 + j] + xwv[1][j], zo = acc[8 + j] + xwv[2][j], zg = acc[12 + j] + xwv[3][j];
    if (p.pf) { zf += cp * p.pf[uq + j]; zg += cp * p.pg[uq + j]; }
    const float a = tanhf(za), f = sigm(zf), g = sigm(zg);
    const float c = f * cp + g * a;
    if (p.pf) zo += c * p.po[uq + j];
    const float o = sigm(zo), ac = tanhf(c);
    va[j] = a; vf[j] = f; vo[j] = o; vg[j] = g; vac[j] = ac; vc[j] = c;
    hn[j] = o * ac * mk;
    cs[j] = c * mk;
  }
  const int64_t o_idx = (int64_t)row * n + uq;
  *(f32x4*)(p.H + o_idx) = hn;
  *(f32x4*)(p.Cs + o_idx) = cs;
  if (p.SV) {
    float* sv = p.SV + (int64_t)row * 6 * n + uq;
    *(f32x4*)(sv) = va; *(f32x4*)(sv + n) = vf; *(f32x4*)(sv + 2 * (int64_t)n) = vo; *(f32x4*)(sv + 3 * (int64_t)n) = vg;
    *(f32x4*)(sv + 4 * (int64_t)n) = vc; *(f32x4*)(sv + 5 * (int64_t)n) = vac;
  }
}

struct LstmBwdF32 {
  const float* E; const float* rw4t; int64_t ld_rwt; const float* SV; const float* cprev;
  const float* pf; const float* po; const float* pg; const float* mask;
  float* dG; float* dpeep; const float* dG_next; const float* dc_in; float* dc_out;
  int mb, n;
};

// dynamic LDS: the workgroup's 8 rows of RW4^T [8][4n + 16] fp32
__global__ void __launch_bounds__(STEP_THREADS) lstm_step_bwd_f32_kernel(LstmBwdF32 p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) f32x4 red[BWD_TILES * 3 * 4 * 64];
  const int n = p.n, mb = p.mb, K = 4 * p.n;
  const int ldw = K + 16;                                  // row stride = 16 banks mod 64
  float* Wb = (float*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile0 = wave & 3, kq4 = wave >> 2;
  const int u0 = blockIdx.x * 8;
  const int bcol = lane & 15, q = lane >> 4;
  const bool unit_lane = q < 2 && kq4 == 0;
  const int uq = u0 + (q & 1) * 4;
  const int Kq = K >> 2;
  const bool has_next = p.dG_next != nullptr;
  // this step's saved values first (independent of everything else), then the weight rows into LDS
  f32x4 ev[BWD_TILES], sva[BWD_TILES], svf[BWD_TILES], svo[BWD_TILES], svg[BWD_TILES], svc[BWD_TILES], svac[BWD_TILES], cpv[BWD_TILES], dcn[BWD_TILES];
  float mk[BWD_TILES];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tt = 0; tt < BWD_TILES; ++tt) {
    const int row = (tile0 + 4 * tt) * 16 + bcol;
    ev[tt] = sva[tt] = svf[tt] = svo[tt] = svg[tt] = svc[tt] = svac[tt] = cpv[tt] = dcn[tt] = zero;
    mk[tt] = 1.f;
    if (unit_lane && row < mb) {
      const int64_t hidx = (int64_t)row * n + uq;
      ev[tt] = *(const f32x4*)(p.E + hidx);
      const float* sv = p.SV + (int64_t)row * 6 * n + uq;
      sva[tt] = *(const f32x4*)(sv); svf[tt] = *(const f32x4*)(sv + n); svo[tt] = *(const f32x4*)(sv + 2 * (int64_t)n);
      svg[tt] = *(const f32x4*)(sv + 3 * (int64_t)n); svc[tt] = *(const f32x4*)(sv + 4 * (int64_t)n); svac[tt] = *(const f32x4*)(sv + 5 * (int64_t)n);
      cpv[tt] = *(const f32x4*)(p.cprev + hidx);
      if (p.mask) mk[tt] = p.mask[row];
      if (p.dc_in) dcn[tt] = *(const f32x4*)(p.dc_in + hidx);
    }
  }
  f32x4 acc[BWD_TILES];
#pragma unroll
  for (int tt = 0; tt < BWD_TILES; ++tt) acc[tt] = zero;
  if (has_next) {
    const int kch = K >> 2;
    for (int i = tid; i < 8 * kch; i += STEP_THREADS) {
      const int r = i / kch, ch = i - r * kch;
      *(f32x4*)(Wb + r * ldw + ch * 4) = *(const f32x4*)(p.rw4t + (int64_t)(u0 + r) * p.ld_rwt + ch * 4);
    }
    __syncthreads();
    const float* ap = Wb + (bcol & 7) * ldw + kq4 * Kq + q * 4;
#pragma unroll
    for (int tt = 0; tt < BWD_TILES; ++tt) {
      const int tile = tile0 + 4 * tt;
      if (tile * 16 >= mb) continue;
      const int row = tile * 16 + bcol;
      const bool row_ok = row < mb;
      const float* bp = p.dG_next + (int64_t)(row_ok ? row : 0) * K + kq4 * Kq + q * 4;
#pragma unroll 8
      for (int k = 0; k < Kq; k += 16) {
        const f32x4 a = *(const f32x4*)(ap + k);
        f32x4 b = *(const f32x4*)(bp + k);
        b = row_ok ? b : zero;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[c], acc[tt], 0, 0, 0);
      }
      if (kq4 > 0) red[((tt * 3 + (kq4 - 1)) * 4 + tile0) * 64 + lane] = acc[tt];
    }
    __syncthreads();
  }
  float apf[4] = {0, 0, 0, 0}, apo[4] = {0, 0, 0, 0}, apg[4] = {0, 0, 0, 0};
#pragma unroll
  for (int tt = 0; tt < BWD_TILES; ++tt) {
    const int tile = tile0 + 4 * tt;
    if (tile * 16 >= mb) continue;
    const int row = tile * 16 + bcol;
    if (!(unit_lane && row < mb)) continue;
    if (has_next) {
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const f32x4 part = red[((tt * 3 + kk) * 4 + tile0) * 64 + lane];
        acc[tt] += part;
      }
    }
    f32x4 dza, dzf, dzo, dzg, dco;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d_h = (ev[tt][j] + acc[tt][j]) * mk[tt];
      const float d_cn = dcn[tt][j] * mk[tt];
      const float a = sva[tt][j], f = svf[tt][j], o = svo[tt][j], g = svg[tt][j], ac = svac[tt][j], cp = cpv[tt][j];
      dzo[j] = d_h * ac * o * (1.f - o);
      float dc = d_cn + d_h * o * (1.f - ac * ac);
      if (p.pf) dc += dzo[j] * p.po[uq + j];
      dza[j] = dc * g * (1.f - a * a);
      dzf[j] = dc * cp * f * (1.f - f);
      dzg[j] = dc * a * g * (1.f - g);
      float dcp = dc * f;
      if (p.pf) {
        dcp += dzf[j] * p.pf[uq + j] + dzg[j] * p.pg[uq + j];
        apf[j] += dzf[j] * cp; apo[j] += dzo[j] * svc[tt][j]; apg[j] += dzg[j] * cp;
      }
      dco[j] = dcp;
    }
    const int64_t hidx = (int64_t)row * n + uq;
    if (p.dc_out) *(f32x4*)(p.dc_out + hidx) = dco;
    float* dg = p.dG + (int64_t)row * K + uq;
    *(f32x4*)(dg) = dza; *(f32x4*)(dg + n) = dzf; *(f32x4*)(dg + 2 * (int64_t)n) = dzo; *(f32x4*)(dg + 3 * (int64_t)n) = dzg;
  }
  if (p.pf && p.dpeep) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float vf = apf[j], vo = apo[j], vg = apg[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { vf += __shfl_xor(vf, o, 64); vo += __shfl_xor(vo, o, 64); vg += __shfl_xor(vg, o, 64); }
      if (unit_lane && bcol == 0) {
        atomicAdd(p.dpeep + uq + j, vf);
        atomicAdd(p.dpeep + n + uq + j, vo);
        atomicAdd(p.dpeep + 2 * n + uq + j, vg);
      }
    }
  }
}

}  // namespace

extern "C" int dl4j_fill_zero_bytes(void* p, size_t bytes, hipStream_t st);

__global__ void __launch_bounds__(256) fill_ones_u64_kernel(uint64_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = LL_SENTINEL;
}
static int dl4j_fill_ones_bytes(void* p, size_t bytes, hipStream_t st) {
  hipLaunchKernelGGL(fill_ones_u64_kernel, dim3(grid_for((int64_t)(bytes / 8), 1024)), dim3(256), 0, st, (uint64_t*)p, bytes / 8);
  return (int)hipGetLastError();
}

// LDS bytes of the forward kernel; 0 when the shape is not supported
static int lstm_persist_fwd_lds(int mb, int n) {
  if (n <= 0 || mb <= 0 || n % 8 != 0 || n / 8 > 256 || mb > 128) return 0;
  const int mbp = (mb + 31) & ~31;
  const int64_t bytes = (int64_t)(32 + mbp) * (n + 8) * 2;
  return bytes <= 156 * 1024 ? (int)bytes : 0;     // leave room for the static LDS of the barrier flag
}

static int lstm_persist_bwd_lds(int mb, int n);
DL4J_API int dl4j_lstm_persist_supported(int mb, int n) { return (n % 64 == 0 && lstm_persist_fwd_lds(mb, n) > 0 && lstm_persist_bwd_lds(mb, n) > 0) ? 1 : 0; }

// bar: 2 x uint32 device scratch, zeroed here (stream ordered). Returns -2 for an unsupported shape.
DL4J_API int dl4j_lstm_persist_fwd(const void* xw, const void* rw4, int64_t ld_rw, const void* h0, const void* c0, const float* pf,
                                   const float* po, const float* pg, const float* maskT, void* H, void* Cs, void* SV, unsigned* bar, int T,
                                   int mb, int n, int reverse, hipStream_t st) {
  if (T <= 0) return 0;
  const int lds = lstm_persist_fwd_lds(mb, n);
  if (lds == 0 || ld_rw % 8 != 0 || ((uintptr_t)rw4 & 15) || ((uintptr_t)h0 & 15) || ((uintptr_t)H & 15) || ((uintptr_t)xw & 7)) return -2;
  if ((pf != nullptr) != (po != nullptr) || (pf != nullptr) != (pg != nullptr)) return -2;
  static DeviceOnceLds lds_attr;
  lds_attr.ensure((const void*)lstm_persist_fwd_kernel, lds);
  // a fill kernel, not hipMemsetAsync: memset nodes misbehaved on replay inside captured graphs (see rnn_seq.hip)
  const int zrc = dl4j_fill_zero_bytes(bar, 2 * sizeof(unsigned), st);
  if (zrc) return zrc;
  LstmFwdP p{(const bf16_t*)xw, (const bf16_t*)rw4, ld_rw, (const bf16_t*)h0, (const bf16_t*)c0, pf, po, pg, maskT,
             (bf16_t*)H, (bf16_t*)Cs, (bf16_t*)SV, bar, T, mb, n, reverse};
  // the exchange buffer is H itself: every word must read as "not written yet" before the first workgroup polls it
  const int frc = dl4j_fill_ones_bytes(H, (size_t)T * mb * n * 2, st);
  if (frc) return frc;
  hipLaunchKernelGGL(lstm_persist_fwd_kernel, dim3(n / 8), dim3(FWD_THREADS), lds, st, p);
  DL4J_RET_LAST();
}

// The same forward kernel launched once per time step (T = 1 per launch: no in-kernel hand-off, the kernel boundary orders the
// steps). One fused launch per step instead of GEMM + cell kernel; cheaper than the in-kernel all-gather of h at these sizes
// (measured: profiles/r2_lstm_persistent.md).
// [s_begin, s_end): the steps (in processing order) this call launches; the state of step s_begin - 1 is read from H / Cs, so a sequence
// can be issued in chunks (the multi-layer pipeline of nn/layers/lstm_stack.py interleaves chunks of several layers on several streams).
DL4J_API int dl4j_lstm_steps_fwd_range(const void* xw, const void* rw4, int64_t ld_rw, const void* h0, const void* c0, const float* pf,
                                       const float* po, const float* pg, const float* maskT, void* H, void* Cs, void* SV, unsigned* bar, int T,
                                       int mb, int n, int reverse, int s_begin, int s_end, hipStream_t st) {
  if (T <= 0 || s_begin >= s_end) return 0;
  if (s_begin < 0 || s_end > T) return -2;
  const int lds = lstm_persist_fwd_lds(mb, n);
  if (lds == 0 || ld_rw % 8 != 0 || ((uintptr_t)rw4 & 15) || ((uintptr_t)h0 & 15) || ((uintptr_t)H & 15) || ((uintptr_t)xw & 7)) return -2;
  if ((pf != nullptr) != (po != nullptr) || (pf != nullptr) != (pg != nullptr)) return -2;
  if (n % 64 != 0) return -2;                              // whole 16-wide K-steps per K quarter
  const size_t hn = (size_t)mb * n * 2;
  const char* hp = (const char*)h0;
  const char* cp = (const char*)c0;
  if (s_begin > 0) {
    const int tp = reverse ? T - s_begin : s_begin - 1;
    hp = (const char*)H + tp * hn;
    cp = (const char*)Cs + tp * hn;
  }
  (void)lds;
  for (int s = s_begin; s < s_end; ++s) {
    const int t = reverse ? T - 1 - s : s;
    LstmFwdP p{(const bf16_t*)xw + (size_t)t * mb * 4 * n, (const bf16_t*)rw4, ld_rw, (const bf16_t*)hp, (const bf16_t*)cp, pf, po, pg,
               maskT ? maskT + (size_t)t * mb : nullptr, (bf16_t*)((char*)H + t * hn), (bf16_t*)((char*)Cs + t * hn),
               SV ? (bf16_t*)((char*)SV + (size_t)t * 6 * hn) : nullptr, bar, 1, mb, n, 0};
    hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3(n / 8), dim3(STEP_THREADS), 0, st, p);
    hp = (const char*)H + t * hn;
    cp = (const char*)Cs + t * hn;
  }
  DL4J_RET_LAST();
}

DL4J_API int dl4j_lstm_steps_fwd(const void* xw, const void* rw4, int64_t ld_rw, const void* h0, const void* c0, const float* pf,
                                 const float* po, const float* pg, const float* maskT, void* H, void* Cs, void* SV, unsigned* bar, int T, int mb,
                                 int n, int reverse, hipStream_t st) {
  return dl4j_lstm_steps_fwd_range(xw, rw4, ld_rw, h0, c0, pf, po, pg, maskT, H, Cs, SV, bar, T, mb, n, reverse, 0, T, st);
}

static int lstm_persist_bwd_lds(int mb, int n) {
  if (n <= 0 || mb <= 0 || n % 32 != 0 || n / 8 > 256 || mb > 64 * BWD_TILES) return 0;     // n % 32: whole MFMA K-steps per K quarter
  const int64_t bytes = (int64_t)8 * (4 * n + 32) * 2 + (int64_t)BWD_TILES * 3 * 4 * 64 * 16;
  return bytes <= 156 * 1024 ? (int)bytes : 0;
}

// dpeep [3, n] fp32 must be zeroed by the caller (accumulated); bar: 2 x uint32 scratch zeroed here.
DL4J_API int dl4j_lstm_persist_bwd(const void* E, const void* rw4t, int64_t ld_rwt, const void* SV, const void* Cs, const void* c0,
                                   const float* pf, const float* po, const float* pg, const float* maskT, void* dG, float* dpeep, unsigned* bar,
                                   int T, int mb, int n, int reverse, hipStream_t st) {
  if (T <= 0) return 0;
  const int lds = lstm_persist_bwd_lds(mb, n);
  if (lds == 0 || ld_rwt % 8 != 0 || ((uintptr_t)rw4t & 15) || ((uintptr_t)dG & 15) || ((uintptr_t)E & 7) || ((uintptr_t)SV & 7) ||
      ((uintptr_t)Cs & 7) || ((uintptr_t)c0 & 7))
    return -2;
  if ((pf != nullptr) != (po != nullptr) || (pf != nullptr) != (pg != nullptr)) return -2;
  static DeviceOnceLds lds_attr;
  lds_attr.ensure((const void*)lstm_persist_bwd_kernel, lds);
  // a fill kernel, not hipMemsetAsync: memset nodes misbehaved on replay inside captured graphs (see rnn_seq.hip)
  const int zrc = dl4j_fill_zero_bytes(bar, 2 * sizeof(unsigned), st);
  if (zrc) return zrc;
  LstmBwdP p{(const bf16_t*)E, (const bf16_t*)rw4t, ld_rwt, (const bf16_t*)SV, (const bf16_t*)Cs, (const bf16_t*)c0, pf, po, pg, maskT,
             (bf16_t*)dG, dpeep, bar, T, mb, n, reverse, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(lstm_persist_bwd_kernel, dim3(n / 8), dim3(BWD_THREADS), lds, st, p);
  DL4J_RET_LAST();
}

// The backward kernel launched once per time step (see dl4j_lstm_steps_fwd). dc_ws: [2, mb, n] fp32 scratch for dL/dc between launches.
// dc_last: optional [mb, n] fp32 gradient flowing into the cell state after the last step. The gradient of the initial cell state is
// left in dc_ws[(T - 1) % 2].
// Steps are processed from s = T - 1 down to 0; a call launches s in [s_begin, s_end), highest first. The carry between calls lives in the
// arrays: dL/dh comes from the dG of the step processed before (read from dG), dL/dc from dc_ws[parity of the step].
DL4J_API int dl4j_lstm_steps_bwd_range(const void* E, const void* rw4t, int64_t ld_rwt, const void* SV, const void* Cs, const void* c0,
                                       const float* pf, const float* po, const float* pg, const float* maskT, void* dG, float* dpeep,
                                       unsigned* bar, float* dc_ws, const float* dc_last, int T, int mb, int n, int reverse, int s_begin,
                                       int s_end, hipStream_t st) {
  if (T <= 0 || s_begin >= s_end) return 0;
  if (s_begin < 0 || s_end > T) return -2;
  const int lds = lstm_persist_bwd_lds(mb, n);
  if (lds == 0 || ld_rwt % 8 != 0 || ((uintptr_t)rw4t & 15) || ((uintptr_t)dG & 15) || ((uintptr_t)E & 7) || ((uintptr_t)SV & 7) ||
      ((uintptr_t)Cs & 7) || ((uintptr_t)c0 & 7))
    return -2;
  if ((pf != nullptr) != (po != nullptr) || (pf != nullptr) != (pg != nullptr)) return -2;
  // measured (profiles/r2_lstm_persistent.md): with the 8 weight rows staged in LDS (the persistent kernel run for one step) a backward
  // step takes 14.2 us, with weights read straight from L2 by every wave (lstm_step_bwd_kernel) 18.3 us -> the LDS form is used
  static DeviceOnceLds lds_attr;
  lds_attr.ensure((const void*)lstm_persist_bwd_kernel, lds);
  const size_t hn = (size_t)mb * n;
  const bf16_t* Eb = (const bf16_t*)E;
  const bf16_t* SVb = (const bf16_t*)SV;
  const bf16_t* Csb = (const bf16_t*)Cs;
  bf16_t* dGb = (bf16_t*)dG;
  for (int s = s_end - 1; s >= s_begin; --s) {
    const int t = reverse ? T - 1 - s : s;
    const int t_prev = reverse ? t + 1 : t - 1;
    const int t_next = reverse ? t - 1 : t + 1;       // the step processed before this one
    const float* dc_cur = s == T - 1 ? dc_last : dc_ws + (size_t)((T - 2 - s) & 1) * hn;   // dL/dc into this step (null = zeros)
    float* dc_nxt = dc_ws + (size_t)((T - 1 - s) & 1) * hn;
    const bf16_t* next = s == T - 1 ? nullptr : dGb + 4 * t_next * hn;
    LstmBwdP p{Eb + t * hn, (const bf16_t*)rw4t, ld_rwt, SVb + 6 * t * hn, Csb, s == 0 ? (const bf16_t*)c0 : Csb + t_prev * hn, pf, po, pg,
               maskT ? maskT + (size_t)t * mb : nullptr, dGb + 4 * t * hn, dpeep, bar, 1, mb, n, 0, next, dc_cur, dc_nxt};
    if (getenv("DL4J_AMD_LSTM_STEP_BWD_NOLDS") != nullptr)
      hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3(n / 8), dim3(STEP_THREADS), 0, st, p);
    else
      hipLaunchKernelGGL(lstm_persist_bwd_kernel, dim3(n / 8), dim3(BWD_THREADS), lds, st, p);
  }
  DL4J_RET_LAST();
}

DL4J_API int dl4j_lstm_steps_bwd(const void* E, const void* rw4t, int64_t ld_rwt, const void* SV, const void* Cs, const void* c0,
                                 const float* pf, const float* po, const float* pg, const float* maskT, void* dG, float* dpeep, unsigned* bar,
                                 float* dc_ws, const float* dc_last, int T, int mb, int n, int reverse, hipStream_t st) {
  return dl4j_lstm_steps_bwd_range(E, rw4t, ld_rwt, SV, Cs, c0, pf, po, pg, maskT, dG, dpeep, bar, dc_ws, dc_last, T, mb, n, reverse, 0, T, st);
}

// Diagonal passes over a stack of 1 <= nl <= DIAG_MAXL layers with the same width n (see lstm_diag_fwd_kernel; nl = 1 is a plain
// sequence of step launches in the one-tile-per-workgroup decomposition). Per-layer pointers come as
// arrays of nl entries; xw[0] is the first layer's projection of the whole sequence, w2 / x2 / bias are used by layers 1.. (entry 0 is
// ignored). nIn[l] = width of layer l's input. Every launch covers the cells of one anti-diagonal: T + nl - 1 launches.
DL4J_API int dl4j_lstm_diag_supported(int mb, int n, int nl) {
  return (nl >= 1 && nl <= DIAG_MAXL && n % 128 == 0 && n / 8 <= 256 && mb > 0 && mb <= 128) ? 1 : 0;    // n % 128: whole k-steps per wave
}

template <typename TR>
static int lstm_diag_fwd_t(const void* xw0, const void* const* rw4, const void* const* w2, const void* const* x2, const float* const* bias,
                           const void* const* h0, const void* const* c0, const float* const* pf, const float* const* po,
                           const float* const* pg, void* const* H, void* const* Cs, void* const* SV, const int* nIn, int nl, int T, int mb,
                           int n, const float* maskT, int reverse, hipStream_t st) {
  typedef typename TR::T E;
  LstmDiagFwdP<E> p;
  p.nl = nl; p.T = T; p.mb = mb; p.n = n; p.d = 0; p.maskT = maskT; p.reverse = reverse;
  for (int l = 0; l < nl; ++l) {
    LstmDiagLayerF<E>& q = p.lay[l];
    q.xw = l == 0 ? (const E*)xw0 : nullptr;
    q.rw4 = (const E*)rw4[l]; q.ld_rw = n;
    q.w2 = l == 0 ? nullptr : (const E*)w2[l]; q.ld_w2 = l == 0 ? 0 : nIn[l];
    q.x2 = l == 0 ? nullptr : (const E*)x2[l];
    q.bias = l == 0 ? nullptr : bias[l];
    q.h0 = (const E*)h0[l]; q.c0 = (const E*)c0[l];
    q.pf = pf ? pf[l] : nullptr; q.po = po ? po[l] : nullptr; q.pg = pg ? pg[l] : nullptr;
    q.H = (E*)H[l]; q.Cs = (E*)Cs[l]; q.SV = SV ? (E*)SV[l] : nullptr;
    q.nIn = l == 0 ? 0 : nIn[l];
    if (((uintptr_t)q.rw4 & 15) || ((uintptr_t)q.h0 & 15) || ((uintptr_t)q.H & 15) || ((uintptr_t)q.c0 & 15) || ((uintptr_t)q.Cs & 15) ||
        ((uintptr_t)q.SV & 15))
      return -2;
    if ((q.pf != nullptr) != (q.po != nullptr) || (q.pf != nullptr) != (q.pg != nullptr)) return -2;
    if (l > 0 && (q.w2 == nullptr || q.x2 == nullptr || q.bias == nullptr || q.nIn % 64 != 0 || ((uintptr_t)q.w2 & 15) || ((uintptr_t)q.x2 & 15) ||
                  ((uintptr_t)q.bias & 15)))
      return -2;
  }
  for (int l = nl; l < DIAG_MAXL; ++l) p.lay[l] = p.lay[0];
  for (int d = 0; d < T + nl - 1; ++d) {
    p.d = d;
    hipLaunchKernelGGL((lstm_diag_fwd_kernel<TR>), dim3(n / 8, (mb + 31) / 32, nl), dim3(DIAG_THREADS), 0, st, p);
  }
  DL4J_RET_LAST();
}

// dtype: 0 = bf16, 1 = fp32 (every operand except bias / peepholes / dc_ws, which are fp32 always)
DL4J_API int dl4j_lstm_diag_fwd(const void* xw0, const void* const* rw4, const void* const* w2, const void* const* x2, const float* const* bias,
                                const void* const* h0, const void* const* c0, const float* const* pf, const float* const* po,
                                const float* const* pg, void* const* H, void* const* Cs, void* const* SV, const int* nIn, int nl, int T, int mb,
                                int n, int dtype, const float* maskT, int reverse, hipStream_t st) {
  if (T <= 0) return 0;
  if (reverse && nl != 1) return -2;                       // a reversed stack would need the layer below at the mirrored index
  if (!dl4j_lstm_diag_supported(mb, n, nl) || xw0 == nullptr || ((uintptr_t)xw0 & 15)) return -2;
  if (dtype == 1) return lstm_diag_fwd_t<DiagF32>(xw0, rw4, w2, x2, bias, h0, c0, pf, po, pg, H, Cs, SV, nIn, nl, T, mb, n, maskT, reverse, st);
  if (dtype == 0) return lstm_diag_fwd_t<DiagBf16>(xw0, rw4, w2, x2, bias, h0, c0, pf, po, pg, H, Cs, SV, nIn, nl, T, mb, n, maskT, reverse, st);
  return -2;
}

template <typename TR>
static int lstm_diag_bwd_t(const void* E_top, const void* const* rw4t, const void* const* wct, const void* const* SV, const void* const* Cs,
                           const void* const* c0, const float* const* pf, const float* const* po, const float* const* pg, void* const* dG,
                           float* const* dpeep, float* const* dc_ws, int nl, int T, int mb, int n, const float* maskT, int reverse, hipStream_t st) {
  typedef typename TR::T E;
  LstmDiagBwdP<E> p;
  p.nl = nl; p.T = T; p.mb = mb; p.n = n; p.d = 0; p.maskT = maskT; p.reverse = reverse;
  for (int l = 0; l < nl; ++l) {
    LstmDiagLayerB<E>& q = p.lay[l];
    q.E = l == nl - 1 ? (const E*)E_top : nullptr;
    q.rw4t = (const E*)rw4t[l]; q.ld_rwt = 4 * (int64_t)n;
    q.wct = l == nl - 1 ? nullptr : (const E*)wct[l]; q.ld_wct = 4 * (int64_t)n;
    q.dG_above = l == nl - 1 ? nullptr : (const E*)dG[l + 1];
    q.SV = (const E*)SV[l]; q.Cs = (const E*)Cs[l]; q.c0 = (const E*)c0[l];
    q.pf = pf ? pf[l] : nullptr; q.po = po ? po[l] : nullptr; q.pg = pg ? pg[l] : nullptr;
    q.dG = (E*)dG[l]; q.dpeep = dpeep ? dpeep[l] : nullptr; q.dc_ws = dc_ws[l];
    if (((uintptr_t)q.rw4t & 15) || ((uintptr_t)q.dG & 15) || ((uintptr_t)q.SV & 15) || ((uintptr_t)q.Cs & 15) || ((uintptr_t)q.c0 & 15) ||
        q.dc_ws == nullptr || ((uintptr_t)q.dc_ws & 15))
      return -2;
    if (l < nl - 1 && (q.wct == nullptr || ((uintptr_t)q.wct & 15))) return -2;
    if ((q.pf != nullptr) != (q.po != nullptr) || (q.pf != nullptr) != (q.pg != nullptr)) return -2;
  }
  for (int l = nl; l < DIAG_MAXL; ++l) p.lay[l] = p.lay[0];
  for (int d = 0; d < T + nl - 1; ++d) {
    p.d = d;
    hipLaunchKernelGGL((lstm_diag_bwd_kernel<TR>), dim3(n / 16, (mb + 15) / 16, nl), dim3(DIAG_THREADS), 0, st, p);
  }
  DL4J_RET_LAST();
}

// E_top: [T, mb, n] gradient entering the top layer's activations. wct[l] = input weights of layer l + 1 as [n, 4n] rows (null for the top
// layer); dc_ws[l]: [2, mb, n] fp32 scratch; dpeep[l]: [3, n] fp32 zeroed by the caller, or null. dtype as in dl4j_lstm_diag_fwd.
DL4J_API int dl4j_lstm_diag_bwd(const void* E_top, const void* const* rw4t, const void* const* wct, const void* const* SV, const void* const* Cs,
                                const void* const* c0, const float* const* pf, const float* const* po, const float* const* pg,
                                void* const* dG, float* const* dpeep, float* const* dc_ws, int nl, int T, int mb, int n, int dtype,
                                const float* maskT, int reverse, hipStream_t st) {
  if (T <= 0) return 0;
  if (reverse && nl != 1) return -2;
  if (!dl4j_lstm_diag_supported(mb, n, nl) || E_top == nullptr || ((uintptr_t)E_top & 15)) return -2;
  if (dtype == 1) return lstm_diag_bwd_t<DiagF32>(E_top, rw4t, wct, SV, Cs, c0, pf, po, pg, dG, dpeep, dc_ws, nl, T, mb, n, maskT, reverse, st);
  if (dtype == 0) return lstm_diag_bwd_t<DiagBf16>(E_top, rw4t, wct, SV, Cs, c0, pf, po, pg, dG, dpeep, dc_ws, nl, T, mb, n, maskT, reverse, st);
  return -2;
}

// ---------------------------------------------------------------------------------------------------------------- fp32 entry points
static int lstm_f32_bwd_lds(int mb, int n) {
  if (n <= 0 || mb <= 0 || n % 64 != 0 || n / 8 > 256 || mb > 64 * BWD_TILES || mb > 128) return 0;
  const int64_t bytes = (int64_t)8 * (4 * n + 16) * 4;
  return bytes <= 128 * 1024 ? (int)bytes : 0;               // + 24 KB static split-K buffer
}
DL4J_API int dl4j_lstm_steps_f32_supported(int mb, int n) { return lstm_f32_bwd_lds(mb, n) > 0 ? 1 : 0; }

DL4J_API int dl4j_lstm_steps_fwd_f32_range(const float* xw, const float* rw4, int64_t ld_rw, const float* h0, const float* c0, const float* pf,
                                           const float* po, const float* pg, const float* maskT, float* H, float* Cs, float* SV, int T, int mb,
                                           int n, int reverse, int s_begin, int s_end, hipStream_t st) {
  if (T <= 0 || s_begin >= s_end) return 0;
  if (s_begin < 0 || s_end > T) return -2;
  if (lstm_f32_bwd_lds(mb, n) == 0 || ld_rw % 4 != 0 || ((uintptr_t)rw4 & 15) || ((uintptr_t)h0 & 15) || ((uintptr_t)c0 & 15) ||
      ((uintptr_t)H & 15) || ((uintptr_t)Cs & 15) || ((uintptr_t)SV & 15) || ((uintptr_t)xw & 15))
    return -2;
  if ((pf != nullptr) != (po != nullptr) || (pf != nullptr) != (pg != nullptr)) return -2;
  const size_t hn = (size_t)mb * n;
  const float* hp = h0;
  const float* cp = c0;
  if (s_begin > 0) {
    const int tp = reverse ? T - s_begin : s_begin - 1;
    hp = H + tp * hn;
    cp = Cs + tp * hn;
  }
  for (int s = s_begin; s < s_end; ++s) {
    const int t = reverse ? T - 1 - s : s;
    LstmFwdF32 p{xw + 4 * t * hn, rw4, ld_rw, hp, cp, pf, po, pg, maskT ? maskT + (size_t)t * mb : nullptr, H + t * hn, Cs + t * hn,
                 SV ? SV + 6 * t * hn : nullptr, mb, n};
    hipLaunchKernelGGL(lstm_step_fwd_f32_kernel, dim3(n / 8), dim3(STEP_THREADS), 0, st, p);
    hp = H + t * hn;
    cp = Cs + t * hn;
  }
  DL4J_RET_LAST();
}

DL4J_API int dl4j_lstm_steps_fwd_f32(const float* xw, const float* rw4, int64_t ld_rw, const float* h0, const float* c0, const float* pf,
                                     const float* po, const float* pg, const float* maskT, float* H, float* Cs, float* SV, int T, int mb, int n,
                                     int reverse, hipStream_t st) {
  return dl4j_lstm_steps_fwd_f32_range(xw, rw4, ld_rw, h0, c0, pf, po, pg, maskT, H, Cs, SV, T, mb, n, reverse, 0, T, st);
}

// dc_ws: [2, mb, n] fp32 scratch; dpeep [3, n] accumulated (zeroed by the caller) or null
DL4J_API int dl4j_lstm_steps_bwd_f32_range(const float* E, const float* rw4t, int64_t ld_rwt, const float* SV, const float* Cs, const float* c0,
                                           const float* pf, const float* po, const float* pg, const float* maskT, float* dG, float* dpeep,
                                           float* dc_ws, const float* dc_last, int T, int mb, int n, int reverse, int s_begin, int s_end,
                                           hipStream_t st) {
  if (T <= 0 || s_begin >= s_end) return 0;
  if (s_begin < 0 || s_end > T) return -2;
  const int lds = lstm_f32_bwd_lds(mb, n);
  if (lds == 0 || ld_rwt % 4 != 0 || ((uintptr_t)rw4t & 15) || ((uintptr_t)dG & 15) || ((uintptr_t)E & 15) || ((uintptr_t)SV & 15) ||
      ((uintptr_t)Cs & 15) || ((uintptr_t)c0 & 15) || ((uintptr_t)dc_ws & 15))
    return -2;
  if ((pf != nullptr) != (po != nullptr) || (pf != nullptr) != (pg != nullptr)) return -2;
  static DeviceOnceLds lds_attr;
  lds_attr.ensure((const void*)lstm_step_bwd_f32_kernel, lds);
  const size_t hn = (size_t)mb * n;
  for (int s = s_end - 1; s >= s_begin; --s) {
    const int t = reverse ? T - 1 - s : s;
    const int t_prev = reverse ? t + 1 : t - 1;
    const int t_next = reverse ? t - 1 : t + 1;
    const float* dc_cur = s == T - 1 ? dc_last : dc_ws + (size_t)((T - 2 - s) & 1) * hn;
    float* dc_nxt = dc_ws + (size_t)((T - 1 - s) & 1) * hn;
    const float* next = s == T - 1 ? nullptr : dG + 4 * t_next * hn;
    LstmBwdF32 p{E + t * hn, rw4t, ld_rwt, SV + 6 * t * hn, s == 0 ? c0 : Cs + t_prev * hn, pf, po, pg, maskT ? maskT + (size_t)t * mb : nullptr,
                 dG + 4 * t * hn, dpeep, next, dc_cur, dc_nxt, mb, n};
    hipLaunchKernelGGL(lstm_step_bwd_f32_kernel, dim3(n / 8), dim3(STEP_THREADS), lds, st, p);
  }
  DL4J_RET_LAST();
}

DL4J_API int dl4j_lstm_steps_bwd_f32(const float* E, const float* rw4t, int64_t ld_rwt, const float* SV, const float* Cs, const float* c0,
                                     const float* pf, const float* po, const float* pg, const float* maskT, float* dG, float* dpeep, float* dc_ws,
                                     const float* dc_last, int T, int mb, int n, int reverse, hipStream_t st) {
  return dl4j_lstm_steps_bwd_f32_range(E, rw4t, ld_rwt, SV, Cs, c0, pf, po, pg, maskT, dG, dpeep, dc_ws, dc_last, T, mb, n, reverse, 0, T, st);
}
