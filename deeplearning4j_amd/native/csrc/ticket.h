// "Last block done" hand-off between the workgroups of ONE launch, so a reduction over per-block partial results finishes
// inside the kernel that produced them instead of in a second, latency-bound launch (the reference reduces BatchNorm
// statistics in separate kernels, helpers/cuda/batchnorm.cu; on MI355X each dependent launch costs ~5-7 us of drain + ramp).
//
// Protocol (CDNA3/4 memory model, 8 XCDs with private, mutually non-coherent L2s):
//   * partial results are written with agent-scope atomic stores (sc1 write-through: performed at the device coherence
//     point, never left dirty in the XCD's L2) and read back with agent-scope atomic loads (sc1: not served from a stale L2);
//   * a block waits for its own stores to complete (s_waitcnt vmcnt(0): stores are counted by vmcnt on gfx9), passes a
//     workgroup barrier (so every thread's stores are complete), then ONE thread takes a ticket with an agent-scope atomic
//     add. The block whose ticket is expected-1 knows that every other block's partials are complete and visible;
//   * that block resets the counter, so the same counter array serves every later launch on the stream (self-cleaning,
//     replay-safe inside captured graphs). Every block arrives exactly once: no spinning, no co-residency requirement;
//   * the last block then executes ONE agent-scope acquire fence (buffer_inv sc1: drops possibly stale lines of this XCD's
//     caches) and reads the partials with ordinary, pipelined loads. Reading them with per-element atomic loads serialised
//     ~32 memory round trips per thread: 50 us per BatchNorm backward (measured, profiles/r2_ticket_finalize.md).
#pragma once
#include "common.h"

namespace ticket {

__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Split form, so independent work (the output-tile stores of a GEMM epilogue) can be issued while the ticket is in flight:
//   publish():   all threads, after their partial stores — waits for them, then a workgroup barrier
//   take():      thread 0 only — returns the ticket (the value is not needed until resolve())
//   resolve():   all threads — true in every thread of the block that arrived last; that block has acquired the partials
__device__ __forceinline__ void publish() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
__device__ __forceinline__ unsigned take(unsigned* counter) {
  return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool resolve(unsigned* counter, unsigned ticket, unsigned expected, int* s_flag) {
  if (threadIdx.x == 0) {
    const bool last = (ticket + 1u == expected);
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = last ? 1 : 0;
  }
  __syncthreads();
  const bool last = *s_flag != 0;
  if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return last;
}
// Called by ALL threads of the block (contains barriers). s_flag: one int of LDS. True in every thread of the last block.
__device__ __forceinline__ bool arrive_last(unsigned* counter, unsigned expected, int* s_flag) {
  publish();
  unsigned t = 0;
  if (threadIdx.x == 0) t = take(counter);
  return resolve(counter, t, expected, s_flag);
}
// sum of `rows` floats at p[0], p[stride], ... with four independent accumulators (ordinary loads: call after resolve())
__device__ __forceinline__ float sum_rows(const float* p, int rows, int64_t stride) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = 0;
  for (; r + 3 < rows; r += 4) {
    a0 += p[(int64_t)r * stride]; a1 += p[(int64_t)(r + 1) * stride];
    a2 += p[(int64_t)(r + 2) * stride]; a3 += p[(int64_t)(r + 3) * stride];
  }
  for (; r < rows; ++r) a0 += p[(int64_t)r * stride];
  return (a0 + a1) + (a2 + a3);
}

// BatchNorm finalize of one channel from its (sum, sum of squares): shared by the in-kernel and the stand-alone finalize.
struct BnFin {
  const float *gamma, *beta, *conv_bias, *gmean, *gvar;
  float *mean, *invstd, *scale, *shift, *dgmean, *dgvar;
  float* gpart;          // [ngroups][2][C] level-1 results
  unsigned* cnt;         // [ntiles] level-2 counters, then [ntiles][ngroups] level-1 counters; null = finalize not fused
  float invM, eps, decay;
  int stat_mode;         // 0 none, 1 variance parameter, 2 log10(stdev) parameter
  int G;                 // blocks per level-1 group
};

__device__ __forceinline__ void bn_finalize_channel(const BnFin& f, int c, float su, float sq) {
  const float m = su * f.invM;
  float var = sq * f.invM - m * m;
  var = var < 0.f ? 0.f : var;
  const float is = rsqrtf(var + f.eps);
  const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
  f.mean[c] = m; f.invstd[c] = is;
  const float sc = g * is;
  f.scale[c] = sc; f.shift[c] = b - m * sc;
  if (f.stat_mode != 0) {
    const float bm = m + (f.conv_bias ? f.conv_bias[c] : 0.f);
    f.dgmean[c] = (f.gmean[c] - bm) * (1.f - f.decay);
    if (f.stat_mode == 1) {
      f.dgvar[c] = (f.gvar[c] - var) * (1.f - f.decay);
    } else {
      const float vt = powf(10.f, 2.f * f.gvar[c]);
      const float vnew = f.decay * vt + (1.f - f.decay) * var;
      f.dgvar[c] = logf(vt / vnew) * (0.5f / 2.302585092994046f);
    }
  }
}

}  // namespace ticket
