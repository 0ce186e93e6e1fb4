// Fused scaled-dot-product attention forward for gfx950 (flash-attention style: the [Tq, Tk] score matrix is never
// written to memory), bf16 inputs, fp32 accumulation, head dim 64 or 128.
//
// Reference: libnd4j dot_product_attention_v2 (generic/nn/dot_product_attention_v2.cpp:86-181, helpers/impl/
// AttentionHelper.cpp) materialises logits and softmax weights as full [b, h, Tq, Tk] tensors and returns them; SURVEY
// §5.7 asks for a fused kernel instead (the materialised variant stays available in ops/attention.py for callers
// that request the scores).
//
// CDNA4 design (not a port of a warp-level CUDA kernel):
//  * One workgroup = 4 wavefronts = 128 queries of one (batch, head); each wave owns 32 queries and streams over the
//    keys in tiles of 64.
//  * The scores are computed TRANSPOSED, S^T = K Q^T, with v_mfma_f32_32x32x16_bf16: in the 32x32 accumulator layout a
//    lane then holds 16 keys of ONE query, so the online-softmax max / sum are in-register reductions plus a single
//    64-lane cross-half shuffle — no shared-memory tree, no per-row butterfly.
//  * The K rows are fed in a bit-swapped order (bits 2 and 3 of the MFMA row index) so that, after exponentiation, a
//    lane's registers ARE a valid B operand of the second MFMA (O^T = V^T P^T): P never leaves the register file.
//  * V stays [key][d] in LDS and the V^T operand comes from ds_read_b64_tr_b16 transposing loads.
//  * Q fragments are loop invariant and live in registers (loaded straight from global memory).
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

struct AttnP {
  const bf16_t* q;      // [B*H][Tq][HD]
  const bf16_t* k;      // [B*H][Tk][HD]
  const bf16_t* v;      // [B*H][Tk][HD]
  const float* kmask;   // [B][Tk] 1 = attend, 0 = masked (optional)
  bf16_t* o;            // [B*H][Tq][HD]
  float* lse;           // [B*H][Tq] natural-log-sum-exp of the scaled logits (optional; +inf for fully masked rows)
  int H, Tq, Tk;
  int nbh, nqb;         // number of (batch, head) pairs / of 128-query blocks
  float scale_log2e;    // softmax scale * log2(e)
  int causal;           // 1: key j masked for query i when j > i
  int bthd;             // layout of q / k / v / o: 0 = [B][H][T][HD], 1 = [B][T][H][HD] (a linear op's [B*T, H*HD] output seen per head)
};

// first element of (batch, head) bh and the row (token) stride for either layout
__device__ __forceinline__ int64_t head_base(int bthd, int bh, int H, int T, int HD) {
  const int b = bh / H, h = bh - b * H;
  return bthd ? ((int64_t)b * T * H + h) * HD : (int64_t)bh * T * HD;
}
__device__ __forceinline__ int64_t row_stride(int bthd, int H, int HD) { return bthd ? (int64_t)H * HD : (int64_t)HD; }

template <int PITCH>
__device__ __forceinline__ int tr_swz(int row) {
  return PITCH == 256 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2);
}

// [64 keys][HD] tile (row pitch PITCH bytes, 16-byte chunks XOR-swizzled with tr_swz): operand whose 32 rows/cols are the
// d indices cb..cb+31 and whose 16 k slots are the keys ks*16 .. ks*16+15 (standard slot order), via two transposing reads.
template <int PITCH>
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int cb, int ks, int lane) {
  const int g = lane >> 4, h = g >> 1;
  const int qq = (lane & 15) >> 2, pp = lane & 3;
  const int ch = cb + 16 * (g & 1) + 4 * pp;
  const int p0 = ks * 16 + 8 * h + qq;
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

__device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

template <int HD>
__global__ void __launch_bounds__(256) sdpa_fwd_kernel(const AttnP p) {
  constexpr int PITCH = HD * 2;          // bytes per K / V row in LDS
  constexpr int KS = HD / 16;            // MFMA k-steps over the head dim (S^T = K Q^T)
  constexpr int DB = HD / 32;            // 32-row blocks of O^T
  constexpr int CPR = HD / 8;            // 16-byte chunks per row
  constexpr int RPP = 256 / CPR;         // rows loaded per pass
  constexpr int NPASS = 64 / RPP;
  __shared__ __attribute__((aligned(16))) char sK[64 * PITCH];
  __shared__ __attribute__((aligned(16))) char sV[64 * PITCH];
  __shared__ float sM[64];

  // XCD-aware mapping: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2. All query blocks of
  // one (batch, head) re-read the same K / V, so they are placed on the SAME XCD (id % 8 fixed per bh) and run back to back;
  // for causal attention the longest (last) query blocks are issued first to shorten the tail.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / p.nqb) * 8 + xcd;
  if (bh >= p.nbh) return;
  const int qb = slot - (slot / p.nqb) * p.nqb;
  const int b = bh / p.H;
  const int q0 = (p.causal ? p.nqb - 1 - qb : qb) * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int qi = q0 + wid * 32 + l31;                    // this lane's query (two lanes per query: lh = 0 / 1)
  const bf16_t* Q = p.q + head_base(p.bthd, bh, p.H, p.Tq, HD);
  const int64_t RS = row_stride(p.bthd, p.H, HD);
  const bf16_t* K = p.k + head_base(p.bthd, bh, p.H, p.Tk, HD);
  const bf16_t* V = p.v + head_base(p.bthd, bh, p.H, p.Tk, HD);

  // Q^T B-operand fragments: column = query, k slots = d
  bf16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    s16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qi < p.Tq) t = *reinterpret_cast<const s16x8*>(Q + (int64_t)qi * RS + ks * 16 + lh * 8);
    qf[ks] = __builtin_bit_cast(bf16x8_t, t);
  }

  f32x16 oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int chunk = tid % CPR, lrow = tid / CPR;
  int kend = p.Tk;
  if (p.causal) {                                          // keys beyond the last query of this block are never needed
    const int last = q0 + 127 < p.Tq - 1 ? q0 + 127 : p.Tq - 1;
    kend = last + 1 < p.Tk ? last + 1 : p.Tk;
  }
  const int ntiles = (kend + 63) / 64;

  // (a register prefetch of the next K / V tile was measured SLOWER: 654 vs 543 us at T=2048, D=128 — the extra 32 VGPRs
  // cost more occupancy than the overlap gains; several resident workgroups per CU already hide the load latency)
  for (int kt = 0; kt < ntiles; ++kt) {
    const int key0 = kt * 64;
    __syncthreads();                                       // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = lrow + i * RPP;
      const int key = key0 + row;
      s16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
      if (key < p.Tk) {
        kv = *reinterpret_cast<const s16x8*>(K + (int64_t)key * RS + chunk * 8);
        vv = *reinterpret_cast<const s16x8*>(V + (int64_t)key * RS + chunk * 8);
      }
      *reinterpret_cast<s16x8*>(sK + row * PITCH + ((chunk ^ ((row >> 1) & 7)) << 4)) = kv;
      *reinterpret_cast<s16x8*>(sV + row * PITCH + ((chunk ^ tr_swz<PITCH>(row)) << 4)) = vv;
    }
    if (tid < 64) {
      const int key = key0 + tid;
      float mv = (key < p.Tk) ? 0.f : -INFINITY;
      if (p.kmask != nullptr && key < p.Tk && p.kmask[(int64_t)b * p.Tk + key] == 0.f) mv = -INFINITY;
      sM[tid] = mv;
    }
    __syncthreads();
    if (p.causal && key0 > q0 + wid * 32 + 31) continue;   // every key of this tile is in the future of this wave's queries

    // ---- S^T = K Q^T for the 64 keys of the tile (two 32-key blocks), rows fed in swap23 order
    f32x16 sacc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[mb][e] = 0.f;
      const int row = mb * 32 + swap23(l31);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c = ks * 2 + lh;
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + row * PITCH + ((c ^ ((row >> 1) & 7)) << 4));
        sacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[mb], 0, 0, 0);
      }
    }

    // ---- online softmax over this lane's 32 keys (+ the other half-wave's 32 through one shuffle)
    float mx = -INFINITY;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kl = mb * 32 + 16 * (e >> 3) + 8 * lh + (e & 7);     // key (tile-local) held by this accumulator slot
        float s = sacc[mb][e] * p.scale_log2e + sM[kl];
        if (p.causal && key0 + kl > qi) s = -INFINITY;
        sacc[mb][e] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_safe);                          // m_run = -inf -> 0
    float psum = 0.f;
    bf16x8_t pf[4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        s16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = exp2f(sacc[mb][hf * 8 + j] - m_safe);
          psum += pv;
          t[j] = (short)f2bf(pv);
        }
        pf[mb * 2 + hf] = __builtin_bit_cast(bf16x8_t, t);
      }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;

    // ---- O^T += V^T P^T : A = transposed V fragment (rows = d), B = P registers (cols = query, slots = keys)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const bf16x8_t vf = tr_frag<PITCH>(sV, d * 32, s, lane);
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], oacc[d], 0, 0, 0);
      }
  }

  if (qi >= p.Tq) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  bf16_t* O = p.o + head_base(p.bthd, bh, p.H, p.Tq, HD) + (int64_t)qi * RS;
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      s16x4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (short)f2bf(oacc[d][g * 4 + j] * inv);
      *reinterpret_cast<s16x4*>(O + d * 32 + 8 * g + 4 * lh) = t;
    }
  if (p.lse != nullptr && lh == 0)
    p.lse[(int64_t)bh * p.Tq + qi] = l_run > 0.f ? (m_run + log2f(l_run)) * 0.6931471805599453f : INFINITY;
}


// =====================================================================================================================
// Backward. Two deterministic passes (no atomics), both recompute the probabilities from the saved log-sum-exp:
//   pass Q  (one workgroup = 128 queries, streams over key tiles):   dQ   = dS K
//   pass KV (one workgroup = 128 keys,    streams over query tiles): dV   = P^T dO,  dK = dS^T Q
// with P = exp(S*scale - lse), dP = dO V^T, dS = P o (dP - delta) * scale, delta_i = <dO_i, O_i>.
// Both passes keep the forward kernel's register trick: the first two GEMMs of a tile are computed with the streamed
// operand's rows fed in swap23 order, so P and dS (32 x 32 per wave, one owner row/column per lane) are, after the bf16
// pack, directly the B operands of the following GEMMs; the A operands of those (K^T, Q^T, dO^T) come from transposing
// LDS reads of a second, tr-swizzled copy of the streamed tile. The owner-side operands (Q/dO or K/V of the lane's own
// row) are loop invariant and live in registers.
struct AttnBwdP {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  const float* kmask;
  const bf16_t* dout;
  const float* lse;     // [B*H][Tq] natural log
  const float* delta;   // [B*H][Tq]
  bf16_t* dq;
  bf16_t* dk;
  bf16_t* dv;
  int H, Tq, Tk;
  int nbh, nblk;
  float scale, scale_log2e;
  int causal;
  int bthd;             // same layout flag as AttnP, shared by q / k / v / dout / dq / dk / dv
};

// delta[row] = sum_d dO[row, d] * O[row, d]; HD/8 lanes per row, 16 bytes per lane
template <int HD>
__global__ void __launch_bounds__(256) sdpa_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                         float* __restrict__ delta, int64_t rows, int H, int Tq, int bthd) {
  constexpr int LPR = HD / 8;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / LPR;
  const int c = (int)(t - row * LPR);
  float s = 0.f;
  if (row < rows) {
    const int bh = (int)(row / Tq);
    const int64_t off = head_base(bthd, bh, H, Tq, HD) + (row - (int64_t)bh * Tq) * row_stride(bthd, H, HD);
    const s16x8 a = *reinterpret_cast<const s16x8*>(o + off + c * 8);
    const s16x8 b = *reinterpret_cast<const s16x8*>(dout + off + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += bf2f((bf16_t)a[j]) * bf2f((bf16_t)b[j]);
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (row < rows && c == 0) delta[row] = s;
}

__device__ __forceinline__ bf16x8_t row_frag(const char* tile, int pitch, int row, int c) {
  return *reinterpret_cast<const bf16x8_t*>(tile + row * pitch + ((c ^ ((row >> 1) & 7)) << 4));
}

// ---- pass Q: dQ for 128 queries per workgroup, key tiles of 64
template <int HD>
__global__ void __launch_bounds__(256) sdpa_bwd_dq_kernel(const AttnBwdP p) {
  constexpr int PITCH = HD * 2;
  constexpr int KS = HD / 16;
  constexpr int DB = HD / 32;
  constexpr int CPR = HD / 8;
  constexpr int RPP = 256 / CPR;
  constexpr int NPASS = 64 / RPP;
  __shared__ __attribute__((aligned(16))) char sK[64 * PITCH];      // row-fragment layout
  __shared__ __attribute__((aligned(16))) char sKt[64 * PITCH];     // transposing-read layout
  __shared__ __attribute__((aligned(16))) char sV[64 * PITCH];      // row-fragment layout
  __shared__ float sM[64];

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / p.nblk) * 8 + xcd;
  if (bh >= p.nbh) return;
  const int qb = slot - (slot / p.nblk) * p.nblk;
  const int b = bh / p.H;
  const int q0 = (p.causal ? p.nblk - 1 - qb : qb) * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int qi = q0 + wid * 32 + l31;
  const bf16_t* Q = p.q + head_base(p.bthd, bh, p.H, p.Tq, HD);
  const int64_t RS = row_stride(p.bthd, p.H, HD);
  const bf16_t* DO = p.dout + head_base(p.bthd, bh, p.H, p.Tq, HD);
  const bf16_t* K = p.k + head_base(p.bthd, bh, p.H, p.Tk, HD);
  const bf16_t* V = p.v + head_base(p.bthd, bh, p.H, p.Tk, HD);

  bf16x8_t qf[KS], dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    s16x8 t = {0, 0, 0, 0, 0, 0, 0, 0}, u = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qi < p.Tq) {
      t = *reinterpret_cast<const s16x8*>(Q + (int64_t)qi * RS + ks * 16 + lh * 8);
      u = *reinterpret_cast<const s16x8*>(DO + (int64_t)qi * RS + ks * 16 + lh * 8);
    }
    qf[ks] = __builtin_bit_cast(bf16x8_t, t);
    dof[ks] = __builtin_bit_cast(bf16x8_t, u);
  }
  const float lse2 = qi < p.Tq ? p.lse[(int64_t)bh * p.Tq + qi] * 1.4426950408889634f : INFINITY;
  const float dlt = qi < p.Tq ? p.delta[(int64_t)bh * p.Tq + qi] : 0.f;

  f32x16 dqacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqacc[d][e] = 0.f;

  const int chunk = tid % CPR, lrow = tid / CPR;
  int kend = p.Tk;
  if (p.causal) {
    const int last = q0 + 127 < p.Tq - 1 ? q0 + 127 : p.Tq - 1;
    kend = last + 1 < p.Tk ? last + 1 : p.Tk;
  }
  const int ntiles = (kend + 63) / 64;

  for (int kt = 0; kt < ntiles; ++kt) {
    const int key0 = kt * 64;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = lrow + i * RPP;
      const int key = key0 + row;
      s16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
      if (key < p.Tk) {
        kv = *reinterpret_cast<const s16x8*>(K + (int64_t)key * RS + chunk * 8);
        vv = *reinterpret_cast<const s16x8*>(V + (int64_t)key * RS + chunk * 8);
      }
      *reinterpret_cast<s16x8*>(sK + row * PITCH + ((chunk ^ ((row >> 1) & 7)) << 4)) = kv;
      *reinterpret_cast<s16x8*>(sKt + row * PITCH + ((chunk ^ tr_swz<PITCH>(row)) << 4)) = kv;
      *reinterpret_cast<s16x8*>(sV + row * PITCH + ((chunk ^ ((row >> 1) & 7)) << 4)) = vv;
    }
    if (tid < 64) {
      const int key = key0 + tid;
      float mv = (key < p.Tk) ? 0.f : -INFINITY;
      if (p.kmask != nullptr && key < p.Tk && p.kmask[(int64_t)b * p.Tk + key] == 0.f) mv = -INFINITY;
      sM[tid] = mv;
    }
    __syncthreads();
    if (p.causal && key0 > q0 + wid * 32 + 31) continue;

    f32x16 sacc[2], dpacc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { sacc[mb][e] = 0.f; dpacc[mb][e] = 0.f; }
      const int row = mb * 32 + swap23(l31);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(sK, PITCH, row, ks * 2 + lh), qf[ks], sacc[mb], 0, 0, 0);
        dpacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(sV, PITCH, row, ks * 2 + lh), dof[ks], dpacc[mb], 0, 0, 0);
      }
    }

    bf16x8_t dsf[4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        s16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = hf * 8 + j;
          const int kl = mb * 32 + 16 * hf + 8 * lh + j;
          float sl = sacc[mb][e] * p.scale_log2e + sM[kl] - lse2;
          if (p.causal && key0 + kl > qi) sl = -INFINITY;
          const float pv = exp2f(sl);
          t[j] = (short)f2bf(pv * (dpacc[mb][e] - dlt) * p.scale);
        }
        dsf[mb * 2 + hf] = __builtin_bit_cast(bf16x8_t, t);
      }

    // dQ^T += K^T dS^T
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int d = 0; d < DB; ++d)
        dqacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<PITCH>(sKt, d * 32, s, lane), dsf[s], dqacc[d], 0, 0, 0);
  }

  if (qi >= p.Tq) return;
  bf16_t* DQ = p.dq + head_base(p.bthd, bh, p.H, p.Tq, HD) + (int64_t)qi * RS;
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      s16x4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (short)f2bf(dqacc[d][g * 4 + j]);
      *reinterpret_cast<s16x4*>(DQ + d * 32 + 8 * g + 4 * lh) = t;
    }
}

// ---- pass KV: dK, dV for 128 keys per workgroup, query tiles of QT = NMB * 32 (64 for HD 64, 32 for HD 128: 4 LDS tiles)
template <int HD, int NMB>
__global__ void __launch_bounds__(256) sdpa_bwd_dkv_kernel(const AttnBwdP p) {
  constexpr int PITCH = HD * 2;
  constexpr int KS = HD / 16;
  constexpr int DB = HD / 32;
  constexpr int CPR = HD / 8;
  constexpr int RPP = 256 / CPR;
  constexpr int QT = NMB * 32;
  constexpr int NPASS = QT / RPP;
  static_assert(NPASS >= 1, "tile smaller than one load pass");
  __shared__ __attribute__((aligned(16))) char sQ[QT * PITCH];
  __shared__ __attribute__((aligned(16))) char sQt[QT * PITCH];
  __shared__ __attribute__((aligned(16))) char sO[QT * PITCH];
  __shared__ __attribute__((aligned(16))) char sOt[QT * PITCH];
  __shared__ float sL[QT];
  __shared__ float sD[QT];

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / p.nblk) * 8 + xcd;
  if (bh >= p.nbh) return;
  const int kb = slot - (slot / p.nblk) * p.nblk;
  const int b = bh / p.H;
  const int k0 = kb * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int kj = k0 + wid * 32 + l31;                      // this lane's key
  const bf16_t* Q = p.q + head_base(p.bthd, bh, p.H, p.Tq, HD);
  const int64_t RS = row_stride(p.bthd, p.H, HD);
  const bf16_t* DO = p.dout + head_base(p.bthd, bh, p.H, p.Tq, HD);
  const bf16_t* K = p.k + head_base(p.bthd, bh, p.H, p.Tk, HD);
  const bf16_t* V = p.v + head_base(p.bthd, bh, p.H, p.Tk, HD);

  bf16x8_t kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    s16x8 t = {0, 0, 0, 0, 0, 0, 0, 0}, u = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kj < p.Tk) {
      t = *reinterpret_cast<const s16x8*>(K + (int64_t)kj * RS + ks * 16 + lh * 8);
      u = *reinterpret_cast<const s16x8*>(V + (int64_t)kj * RS + ks * 16 + lh * 8);
    }
    kf[ks] = __builtin_bit_cast(bf16x8_t, t);
    vf[ks] = __builtin_bit_cast(bf16x8_t, u);
  }
  float maskk = kj < p.Tk ? 0.f : -INFINITY;
  if (p.kmask != nullptr && kj < p.Tk && p.kmask[(int64_t)b * p.Tk + kj] == 0.f) maskk = -INFINITY;

  f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dkacc[d][e] = 0.f; dvacc[d][e] = 0.f; }

  const int chunk = tid % CPR, lrow = tid / CPR;
  const int ntiles = (p.Tq + QT - 1) / QT;
  const int t0 = p.causal ? k0 / QT : 0;                   // queries before the block's first key never see these keys

  for (int qt = t0; qt < ntiles; ++qt) {
    const int qs = qt * QT;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = lrow + i * RPP;
      const int qr = qs + row;
      s16x8 qv = {0, 0, 0, 0, 0, 0, 0, 0}, ov = {0, 0, 0, 0, 0, 0, 0, 0};
      if (qr < p.Tq) {
        qv = *reinterpret_cast<const s16x8*>(Q + (int64_t)qr * RS + chunk * 8);
        ov = *reinterpret_cast<const s16x8*>(DO + (int64_t)qr * RS + chunk * 8);
      }
      const int ro = row * PITCH + ((chunk ^ ((row >> 1) & 7)) << 4), to = row * PITCH + ((chunk ^ tr_swz<PITCH>(row)) << 4);
      *reinterpret_cast<s16x8*>(sQ + ro) = qv;
      *reinterpret_cast<s16x8*>(sQt + to) = qv;
      *reinterpret_cast<s16x8*>(sO + ro) = ov;
      *reinterpret_cast<s16x8*>(sOt + to) = ov;
    }
    if (tid < QT) {
      const int qr = qs + tid;
      sL[tid] = qr < p.Tq ? p.lse[(int64_t)bh * p.Tq + qr] * 1.4426950408889634f : INFINITY;
      sD[tid] = qr < p.Tq ? p.delta[(int64_t)bh * p.Tq + qr] : 0.f;
    }
    __syncthreads();
    if (p.causal && qs + QT - 1 < k0 + wid * 32) continue;   // every query of this tile precedes this wave's keys

    f32x16 sacc[NMB], dpacc[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { sacc[mb][e] = 0.f; dpacc[mb][e] = 0.f; }
      const int row = mb * 32 + swap23(l31);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(sQ, PITCH, row, ks * 2 + lh), kf[ks], sacc[mb], 0, 0, 0);
        dpacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(sO, PITCH, row, ks * 2 + lh), vf[ks], dpacc[mb], 0, 0, 0);
      }
    }

    bf16x8_t pf[NMB * 2], dsf[NMB * 2];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        s16x8 tp, td;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = hf * 8 + j;
          const int ql = mb * 32 + 16 * hf + 8 * lh + j;      // tile-local query of this accumulator slot
          float sl = sacc[mb][e] * p.scale_log2e + maskk - sL[ql];
          if (p.causal && kj > qs + ql) sl = -INFINITY;
          const float pv = exp2f(sl);
          tp[j] = (short)f2bf(pv);
          td[j] = (short)f2bf(pv * (dpacc[mb][e] - sD[ql]) * p.scale);
        }
        pf[mb * 2 + hf] = __builtin_bit_cast(bf16x8_t, tp);
        dsf[mb * 2 + hf] = __builtin_bit_cast(bf16x8_t, td);
      }

    // dV^T += dO^T P,  dK^T += Q^T dS   (rows = d, columns = this lane's key, k slots = the tile's queries)
#pragma unroll
    for (int s = 0; s < NMB * 2; ++s)
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        dvacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<PITCH>(sOt, d * 32, s, lane), pf[s], dvacc[d], 0, 0, 0);
        dkacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<PITCH>(sQt, d * 32, s, lane), dsf[s], dkacc[d], 0, 0, 0);
      }
  }

  if (kj >= p.Tk) return;
  bf16_t* DK = p.dk + head_base(p.bthd, bh, p.H, p.Tk, HD) + (int64_t)kj * RS;
  bf16_t* DV = p.dv + head_base(p.bthd, bh, p.H, p.Tk, HD) + (int64_t)kj * RS;
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      s16x4 tk, tv;
#pragma unroll
      for (int j = 0; j < 4; ++j) { tk[j] = (short)f2bf(dkacc[d][g * 4 + j]); tv[j] = (short)f2bf(dvacc[d][g * 4 + j]); }
      *reinterpret_cast<s16x4*>(DK + d * 32 + 8 * g + 4 * lh) = tk;
      *reinterpret_cast<s16x4*>(DV + d * 32 + 8 * g + 4 * lh) = tv;
    }
}

}  // namespace

// q,k,v,o: bf16 [B*H, T, HD] contiguous; kmask: fp32 [B, Tk] or null; lse: fp32 [B*H, Tq] or null. HD in {64, 128}.
// bthd = 1: q / k / v / o are [B][T][H][HD] (see AttnP).
DL4J_API int dl4j_sdpa_fwd_l(const void* q, const void* k, const void* v, const float* kmask, void* o, float* lse, int B, int H, int Tq,
                             int Tk, int HD, float scale, int causal, int bthd, hipStream_t st) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return 0;
  if (HD != 64 && HD != 128) return -2;
  AttnP p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.kmask = kmask; p.o = (bf16_t*)o; p.lse = lse;
  p.H = H; p.Tq = Tq; p.Tk = Tk; p.scale_log2e = scale * 1.4426950408889634f; p.causal = causal; p.bthd = bthd ? 1 : 0;
  p.nbh = B * H; p.nqb = (Tq + 127) / 128;
  const dim3 grid((unsigned)(((p.nbh + 7) / 8) * 8 * p.nqb));
  if (HD == 64) hipLaunchKernelGGL(sdpa_fwd_kernel<64>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(sdpa_fwd_kernel<128>, grid, dim3(256), 0, st, p);
  DL4J_RET_LAST();
}

// Backward of dl4j_sdpa_fwd. o / lse are the forward outputs, dout = dL/do; delta_ws: fp32 [B*H*Tq] scratch.
// dq [B*H, Tq, HD], dk / dv [B*H, Tk, HD] are fully written.
DL4J_API int dl4j_sdpa_bwd_l(const void* q, const void* k, const void* v, const float* kmask, const void* o, const float* lse,
                             const void* dout, void* dq, void* dk, void* dv, float* delta_ws, int B, int H, int Tq, int Tk, int HD,
                             float scale, int causal, int bthd, hipStream_t st) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return 0;
  if (HD != 64 && HD != 128) return -2;
  if (!lse || !delta_ws) return -3;
  AttnBwdP p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.kmask = kmask; p.dout = (const bf16_t*)dout;
  p.lse = lse; p.delta = delta_ws; p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
  p.H = H; p.Tq = Tq; p.Tk = Tk; p.nbh = B * H; p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f; p.causal = causal; p.bthd = bthd ? 1 : 0;
  const int64_t rows = (int64_t)B * H * Tq;
  const int64_t dthreads = rows * (HD / 8);
  const unsigned dgrid = (unsigned)((dthreads + 255) / 256);
  const int bh8 = ((p.nbh + 7) / 8) * 8;
  AttnBwdP pq = p, pk = p;
  pq.nblk = (Tq + 127) / 128;
  pk.nblk = (Tk + 127) / 128;
  if (HD == 64) {
    hipLaunchKernelGGL(sdpa_delta_kernel<64>, dim3(dgrid), dim3(256), 0, st, (const bf16_t*)o, (const bf16_t*)dout, delta_ws, rows, H, Tq, bthd);
    hipLaunchKernelGGL(sdpa_bwd_dq_kernel<64>, dim3((unsigned)(bh8 * pq.nblk)), dim3(256), 0, st, pq);
    hipLaunchKernelGGL((sdpa_bwd_dkv_kernel<64, 2>), dim3((unsigned)(bh8 * pk.nblk)), dim3(256), 0, st, pk);
  } else {
    hipLaunchKernelGGL(sdpa_delta_kernel<128>, dim3(dgrid), dim3(256), 0, st, (const bf16_t*)o, (const bf16_t*)dout, delta_ws, rows, H, Tq, bthd);
    hipLaunchKernelGGL(sdpa_bwd_dq_kernel<128>, dim3((unsigned)(bh8 * pq.nblk)), dim3(256), 0, st, pq);
    hipLaunchKernelGGL((sdpa_bwd_dkv_kernel<128, 1>), dim3((unsigned)(bh8 * pk.nblk)), dim3(256), 0, st, pk);
  }
  DL4J_RET_LAST();
}

DL4J_API int dl4j_sdpa_fwd(const void* q, const void* k, const void* v, const float* kmask, void* o, float* lse, int B, int H, int Tq,
                           int Tk, int HD, float scale, int causal, hipStream_t st) {
  return dl4j_sdpa_fwd_l(q, k, v, kmask, o, lse, B, H, Tq, Tk, HD, scale, causal, 0, st);
}

DL4J_API int dl4j_sdpa_bwd(const void* q, const void* k, const void* v, const float* kmask, const void* o, const float* lse,
                           const void* dout, void* dq, void* dk, void* dv, float* delta_ws, int B, int H, int Tq, int Tk, int HD,
                           float scale, int causal, hipStream_t st) {
  return dl4j_sdpa_bwd_l(q, k, v, kmask, o, lse, dout, dq, dk, dv, delta_ws, B, H, Tq, Tk, HD, scale, causal, 0, st);
}
