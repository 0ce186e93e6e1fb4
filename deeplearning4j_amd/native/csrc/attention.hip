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
};

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
  const bf16_t* Q = p.q + (int64_t)bh * p.Tq * HD;
  const bf16_t* K = p.k + (int64_t)bh * p.Tk * HD;
  const bf16_t* V = p.v + (int64_t)bh * p.Tk * HD;

  // Q^T B-operand fragments: column = query, k slots = d
  bf16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    s16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qi < p.Tq) t = *reinterpret_cast<const s16x8*>(Q + (int64_t)qi * HD + ks * 16 + lh * 8);
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
        kv = *reinterpret_cast<const s16x8*>(K + (int64_t)key * HD + chunk * 8);
        vv = *reinterpret_cast<const s16x8*>(V + (int64_t)key * HD + chunk * 8);
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
  bf16_t* O = p.o + ((int64_t)bh * p.Tq + qi) * HD;
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

}  // namespace

// q,k,v,o: bf16 [B*H, T, HD] contiguous; kmask: fp32 [B, Tk] or null; lse: fp32 [B*H, Tq] or null. HD in {64, 128}.
DL4J_API int dl4j_sdpa_fwd(const void* q, const void* k, const void* v, const float* kmask, void* o, float* lse, int B, int H, int Tq,
                           int Tk, int HD, float scale, int causal, hipStream_t st) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return 0;
  if (HD != 64 && HD != 128) return -2;
  AttnP p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.kmask = kmask; p.o = (bf16_t*)o; p.lse = lse;
  p.H = H; p.Tq = Tq; p.Tk = Tk; p.scale_log2e = scale * 1.4426950408889634f; p.causal = causal;
  p.nbh = B * H; p.nqb = (Tq + 127) / 128;
  const dim3 grid((unsigned)(((p.nbh + 7) / 8) * 8 * p.nqb));
  if (HD == 64) hipLaunchKernelGGL(sdpa_fwd_kernel<64>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(sdpa_fwd_kernel<128>, grid, dim3(256), 0, st, p);
  DL4J_RET_LAST();
}
