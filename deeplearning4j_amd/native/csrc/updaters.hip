// Segmented fused updater over the flat parameter / gradient / updater-state vectors.
//
// The reference fragments the flat vector into >100 UpdaterBlocks for ResNet-50 and issues, per
// block and per iteration, a divide-by-minibatch op, 1-2 regularisation Axpy ops, one fused
// updater op (+ stream sync) and a final params -= update step function
// (deeplearning4j-nn/.../nn/updater/BaseMultiLayerUpdater.java:83-141,329-382, UpdaterBlock.java:131-211,
// libnd4j/include/ops/declarable/helpers/cuda/updater*.cu). Here the whole chain is ONE launch:
// a device-resident segment table carries per-block hyper-parameters and a block table maps each
// workgroup to (segment, chunk). Formulas follow SURVEY Appendix B.5 / the reference kernels.
// The iteration counter lives in device memory so the launch is hipGraph-replayable.
#include "common.h"

struct __attribute__((packed, aligned(8))) UpdSeg {
  int64_t off;        // offset into params/grads
  int64_t len;
  int64_t state_off;  // offset into the flat updater state
  int32_t kind;       // 0 NoOp 1 Sgd 2 Nesterovs 3 Adam 4 RmsProp 5 AdaGrad 6 AdaDelta 7 AdaMax 8 Nadam 9 AMSGrad 10 AdaBelief
  int32_t flags;      // bit0: divide grad by minibatch; bit1: weight decay scaled by lr; bit2: emit update only
  float lr, a, b, eps;  // a/b: momentum | beta1,beta2 | rmsDecay | rho
  float l1, l2, wd, pad;
};
static_assert(sizeof(UpdSeg) == 64, "UpdSeg layout is part of the ctypes ABI");

#define UPD_CHUNK 2048

// element type of the flat params / grads / state vectors: fp32 masters, or a bf16 / fp16 network without fp32 masters
// (DL4J dataType(HALF): NeuralNetConfiguration.java:1149-1154). Math is always fp32.
struct PF32 { typedef float T; static __device__ __forceinline__ float ld(const T* p, int64_t i) { return p[i]; }
              static __device__ __forceinline__ void st(T* p, int64_t i, float v) { p[i] = v; } };
struct PBF16 { typedef bf16_t T; static __device__ __forceinline__ float ld(const T* p, int64_t i) { return bf2f(p[i]); }
               static __device__ __forceinline__ void st(T* p, int64_t i, float v) { p[i] = f2bf(v); } };
struct PF16 { typedef _Float16 T; static __device__ __forceinline__ float ld(const T* p, int64_t i) { return (float)p[i]; }
              static __device__ __forceinline__ void st(T* p, int64_t i, float v) { p[i] = (_Float16)v; } };

// Device-resident learning-rate schedule: lr_sched[(iter - sched_base) * nseg + segment] for the next sched_window iterations,
// pre-evaluated on the host (ISchedule.valueAt) and refilled between replays when the window runs out — so a network with a
// schedule keeps its captured training step (the reference re-reads the schedule on the host every iteration,
// BaseMultiLayerUpdater.java:329-382). NaN entries mean "segment has no learning rate": the table value is kept.
struct LrSched {
  const float* table;
  const int64_t* base;
  int window, nseg;
};

template <class P>
__global__ void __launch_bounds__(256) segmented_updater_kernel(
    typename P::T* __restrict__ params, typename P::T* __restrict__ grads, typename P::T* __restrict__ state,
    bf16_t* __restrict__ shadow, const UpdSeg* __restrict__ segs, const int32_t* __restrict__ blk_seg,
    const int32_t* __restrict__ blk_chunk, const int64_t* __restrict__ iter_dev, float inv_batch,
    int write_update, LrSched sched) {
  UpdSeg s = segs[blk_seg[blockIdx.x]];
  if (sched.table != nullptr) {
    int64_t k = iter_dev[0] - sched.base[0];
    k = k < 0 ? 0 : (k >= sched.window ? sched.window - 1 : k);
    const float lr = sched.table[k * sched.nseg + blk_seg[blockIdx.x]];
    if (lr == lr) s.lr = lr;
  }
  const int64_t start = (int64_t)blk_chunk[blockIdx.x] * UPD_CHUNK;
  const int64_t end = (start + UPD_CHUNK < s.len) ? start + UPD_CHUNK : s.len;
  const float t = (float)(iter_dev[0] + 1);
  const bool div = s.flags & 1, wd_lr = s.flags & 2, upd_only = s.flags & 4;
  const int64_t n = s.len;
  typename P::T* stp = state + s.state_off;
  // per-launch scalars
  float alpha_t = s.lr;
  const float eps = (s.eps == 0.f && s.kind >= 3) ? 1e-7f : s.eps;
  if (s.kind == 3 || s.kind == 9 || s.kind == 10) {
    const float b1t = powf(s.a, t), b2t = powf(s.b, t);
    alpha_t = s.lr * sqrtf(1.f - b2t) / (1.f - b1t);
    if (isnan(alpha_t) || alpha_t == 0.f || isinf(alpha_t)) alpha_t = eps;
  } else if (s.kind == 7) {
    alpha_t = s.lr / (1.f - powf(s.a, t));
    if (isnan(alpha_t) || alpha_t == 0.f || isinf(alpha_t)) alpha_t = eps;
  } else if (s.kind == 8) {
    alpha_t = 1.f / (1.f - powf(s.a, t));
  }
  for (int64_t i = start + threadIdx.x; i < end; i += 256) {
    const int64_t gi = s.off + i;
    float g = P::ld(grads, gi);
    if (div) g *= inv_batch;
    const float w = upd_only ? 0.f : P::ld(params, gi);
    if (s.l1 != 0.f) g += s.l1 * (w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f));
    if (s.l2 != 0.f) g += s.l2 * w;
    float u;
    switch (s.kind) {
      case 0: u = g; break;
      case 1: u = s.lr * g; break;
      case 2: {  // Nesterovs
        const float vprev = s.a * P::ld(stp, i);
        const float v = vprev - s.lr * g;
        P::st(stp, i, v);
        u = vprev - (s.a + 1.f) * v;
      } break;
      case 3: {  // Adam, state [m | v]
        const float m = s.a * P::ld(stp, i) + (1.f - s.a) * g;
        const float v = s.b * P::ld(stp, n + i) + (1.f - s.b) * g * g;
        P::st(stp, i, m); P::st(stp, n + i, v);
        u = m * alpha_t / (sqrtf(v) + eps);
      } break;
      case 4: {  // RmsProp
        const float r = s.a * P::ld(stp, i) + (1.f - s.a) * g * g;
        P::st(stp, i, r);
        u = s.lr * g / (sqrtf(r) + eps);
      } break;
      case 5: {  // AdaGrad
        const float h = P::ld(stp, i) + g * g;
        P::st(stp, i, h);
        u = s.lr * g / (sqrtf(h) + eps);
      } break;
      case 6: {  // AdaDelta, state [msg | msdx]
        const float msg = s.a * P::ld(stp, i) + (1.f - s.a) * g * g;
        const float msdx0 = P::ld(stp, n + i);
        u = g * sqrtf(msdx0 + eps) / sqrtf(msg + eps);
        P::st(stp, i, msg); P::st(stp, n + i, s.a * msdx0 + (1.f - s.a) * u * u);
      } break;
      case 7: {  // AdaMax, state [m | u]
        const float m = s.a * P::ld(stp, i) + (1.f - s.a) * g;
        const float ui = fmaxf(s.b * P::ld(stp, n + i), fabsf(g)) + 1e-32f;
        P::st(stp, i, m); P::st(stp, n + i, ui);
        u = m * alpha_t / ui;
      } break;
      case 8: {  // Nadam, state [m | v]
        const float m = s.a * P::ld(stp, i) + (1.f - s.a) * g;
        const float v = s.b * P::ld(stp, n + i) + (1.f - s.b) * g * g;
        P::st(stp, i, m); P::st(stp, n + i, v);
        u = s.lr * ((s.a * m + (1.f - s.a) * g) * alpha_t) / (sqrtf(v) + eps);
      } break;
      case 9: {  // AMSGrad, state [m | v | vHat]
        const float m = s.a * P::ld(stp, i) + (1.f - s.a) * g;
        const float v = s.b * P::ld(stp, n + i) + (1.f - s.b) * g * g;
        const float h = fmaxf(P::ld(stp, 2 * n + i), v);
        P::st(stp, i, m); P::st(stp, n + i, v); P::st(stp, 2 * n + i, h);
        u = alpha_t * m / (sqrtf(h) + eps);
      } break;
      default: {  // AdaBelief, state [m | s]
        const float m = s.a * P::ld(stp, i) + (1.f - s.a) * g;
        const float d = g - m;
        const float sv = s.b * P::ld(stp, n + i) + (1.f - s.b) * d * d + eps;
        P::st(stp, i, m); P::st(stp, n + i, sv);
        u = alpha_t * m / (sqrtf(sv) + eps);
      } break;
    }
    if (s.wd != 0.f) u += s.wd * (wd_lr ? s.lr : 1.f) * w;
    if (upd_only) {
      P::st(grads, gi, u);
    } else {
      const float wn = w - u;
      P::st(params, gi, wn);
      if (shadow != nullptr) shadow[gi] = f2bf(wn);
      if (write_update) P::st(grads, gi, u);
    }
  }
}

__global__ void incr_iter_kernel(int64_t* it) {
  if (threadIdx.x == 0 && blockIdx.x == 0) it[0] += 1;
}

// dtype: 0 fp32, 1 bf16, 2 fp16 (params, grads and state share it). lr_sched / sched_base / window / nseg: see LrSched (null = none).
DL4J_API int dl4j_segmented_update_ex(void* params, void* grads, void* state, void* shadow_bf16,
                                      const void* segs, const int32_t* blk_seg, const int32_t* blk_chunk,
                                      int num_blocks, int64_t* iter_dev, float inv_batch, int write_update,
                                      int increment_iter, int dtype, const float* lr_sched, const int64_t* sched_base,
                                      int sched_window, int nseg, hipStream_t st) {
  LrSched sc{lr_sched, sched_base, sched_window, nseg};
  if (lr_sched == nullptr || sched_base == nullptr || sched_window <= 0) sc.table = nullptr;
  if (num_blocks > 0) {
    if (dtype == 0)
      hipLaunchKernelGGL(segmented_updater_kernel<PF32>, dim3(num_blocks), dim3(256), 0, st, (float*)params, (float*)grads,
                         (float*)state, (bf16_t*)shadow_bf16, (const UpdSeg*)segs, blk_seg, blk_chunk, iter_dev, inv_batch,
                         write_update, sc);
    else if (dtype == 1)
      hipLaunchKernelGGL(segmented_updater_kernel<PBF16>, dim3(num_blocks), dim3(256), 0, st, (bf16_t*)params, (bf16_t*)grads,
                         (bf16_t*)state, (bf16_t*)nullptr, (const UpdSeg*)segs, blk_seg, blk_chunk, iter_dev, inv_batch,
                         write_update, sc);
    else if (dtype == 2)
      hipLaunchKernelGGL(segmented_updater_kernel<PF16>, dim3(num_blocks), dim3(256), 0, st, (_Float16*)params,
                         (_Float16*)grads, (_Float16*)state, (bf16_t*)nullptr, (const UpdSeg*)segs, blk_seg, blk_chunk, iter_dev,
                         inv_batch, write_update, sc);
    else
      return -2;
  }
  if (increment_iter) hipLaunchKernelGGL(incr_iter_kernel, dim3(1), dim3(64), 0, st, iter_dev);
  DL4J_RET_LAST();
}
DL4J_API int dl4j_segmented_update(float* params, float* grads, float* state, void* shadow_bf16,
                                   const void* segs, const int32_t* blk_seg, const int32_t* blk_chunk,
                                   int num_blocks, int64_t* iter_dev, float inv_batch, int write_update,
                                   int increment_iter, hipStream_t st) {
  return dl4j_segmented_update_ex(params, grads, state, shadow_bf16, segs, blk_seg, blk_chunk, num_blocks, iter_dev, inv_batch,
                                  write_update, increment_iter, 0, nullptr, nullptr, 0, 0, st);
}

// ---------------------------------------------------------------------------------------
// Gradient normalization (GradientNormalization.java; applied in BaseMultiLayerUpdater.preApply:405-453) over a table of
// ranges of the flat gradient vector — one range per layer (…PerLayer modes) or per parameter tensor (…PerParamType):
//   mode 1 renormalize: g /= ||g||_2      mode 2 clip element-wise to [-thr, thr]      mode 3 clip L2: g *= thr/||g|| if ||g|| > thr
// Two launches for the whole network (sum of squares per range, then apply), no host round trip: graph-capturable.
// ---------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(8))) GnSeg {
  int64_t off, len;
  int32_t mode;
  float thr;
};
static_assert(sizeof(GnSeg) == 24, "GnSeg layout is part of the ctypes ABI");

template <class P>
__global__ void __launch_bounds__(256) gn_sumsq_kernel(const typename P::T* __restrict__ grads, const GnSeg* __restrict__ segs,
                                                       const int32_t* __restrict__ blk_seg, const int32_t* __restrict__ blk_chunk,
                                                       float* __restrict__ acc) {
  __shared__ float red[16];
  const int si = blk_seg[blockIdx.x];
  const GnSeg s = segs[si];
  if (s.mode == 2) return;
  const int64_t start = (int64_t)blk_chunk[blockIdx.x] * UPD_CHUNK;
  const int64_t end = (start + UPD_CHUNK < s.len) ? start + UPD_CHUNK : s.len;
  float a = 0.f;
  for (int64_t i = start + threadIdx.x; i < end; i += 256) { const float g = P::ld(grads, s.off + i); a += g * g; }
  a = block_sum(a, red);
  if (threadIdx.x == 0) atomicAdd(acc + si, a);
}

template <class P>
__global__ void __launch_bounds__(256) gn_apply_kernel(typename P::T* __restrict__ grads, const GnSeg* __restrict__ segs,
                                                       const int32_t* __restrict__ blk_seg, const int32_t* __restrict__ blk_chunk,
                                                       const float* __restrict__ acc) {
  const int si = blk_seg[blockIdx.x];
  const GnSeg s = segs[si];
  const int64_t start = (int64_t)blk_chunk[blockIdx.x] * UPD_CHUNK;
  const int64_t end = (start + UPD_CHUNK < s.len) ? start + UPD_CHUNK : s.len;
  float scale = 1.f;
  if (s.mode == 1) {
    scale = 1.f / fmaxf(sqrtf(acc[si]), 1e-12f);
  } else if (s.mode == 3) {
    const float nrm = sqrtf(acc[si]);
    if (!(nrm > s.thr)) return;
    scale = s.thr / fmaxf(nrm, 1e-12f);
  }
  for (int64_t i = start + threadIdx.x; i < end; i += 256) {
    const float g = P::ld(grads, s.off + i);
    P::st(grads, s.off + i, s.mode == 2 ? fminf(fmaxf(g, -s.thr), s.thr) : g * scale);
  }
}

DL4J_API int dl4j_grad_normalize(void* grads, const void* segs, int nseg, const int32_t* blk_seg, const int32_t* blk_chunk,
                                 int num_blocks, float* acc /* nseg floats */, int dtype, hipStream_t st) {
  if (num_blocks <= 0 || nseg <= 0) return 0;
  if (hipMemsetAsync(acc, 0, sizeof(float) * (size_t)nseg, st) != hipSuccess) return -1;
  const dim3 g(num_blocks), b(256);
  const GnSeg* sg = (const GnSeg*)segs;
  if (dtype == 0) {
    hipLaunchKernelGGL(gn_sumsq_kernel<PF32>, g, b, 0, st, (const float*)grads, sg, blk_seg, blk_chunk, acc);
    hipLaunchKernelGGL(gn_apply_kernel<PF32>, g, b, 0, st, (float*)grads, sg, blk_seg, blk_chunk, acc);
  } else if (dtype == 1) {
    hipLaunchKernelGGL(gn_sumsq_kernel<PBF16>, g, b, 0, st, (const bf16_t*)grads, sg, blk_seg, blk_chunk, acc);
    hipLaunchKernelGGL(gn_apply_kernel<PBF16>, g, b, 0, st, (bf16_t*)grads, sg, blk_seg, blk_chunk, acc);
  } else if (dtype == 2) {
    hipLaunchKernelGGL(gn_sumsq_kernel<PF16>, g, b, 0, st, (const _Float16*)grads, sg, blk_seg, blk_chunk, acc);
    hipLaunchKernelGGL(gn_apply_kernel<PF16>, g, b, 0, st, (_Float16*)grads, sg, blk_seg, blk_chunk, acc);
  } else {
    return -2;
  }
  DL4J_RET_LAST();
}

// ---------------------------------------------------------------------------------------
// Layout-aware cast table: refresh compute-dtype (bf16) weight replicas from the fp32 master
// vector in one launch. Each entry copies a [O, I, K] (DL4J OIHW, K = kH*kW) block to
// [O, K, I] (channels-last filter layout used by the implicit-GEMM kernels) or straight.
// ---------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(8))) CastSeg {
  int64_t src_off;   // into fp32 master vector
  int64_t dst_off;   // into bf16 arena
  int64_t len;
  int32_t I, K;      // mode 0, K > 1: permute [O][I][K] -> [O][K][I]
  int32_t O, mode;   // mode 1: [O][I][K] -> [I][K flipped][O]  (filters for the dgrad implicit GEMM)
};
static_assert(sizeof(CastSeg) == 40, "CastSeg layout is part of the ctypes ABI");

#define CAST_LDS_FLOATS 13312
// destination element types of the compute replicas: bf16 (raw bits), fp16, fp32 (layout change only: the OHWI / flipped filter
// images of an fp32 network)
struct CastBf16 { typedef bf16_t T; static __device__ __forceinline__ T cv(float f) { return f2bf(f); } };
struct CastF16 { typedef _Float16 T; static __device__ __forceinline__ T cv(float f) { return (_Float16)f; } };
struct CastF32 { typedef float T; static __device__ __forceinline__ T cv(float f) { return f; } };

template <class CV>
__global__ void __launch_bounds__(256) cast_table_kernel(const float* __restrict__ src, typename CV::T* __restrict__ dst,
                                                         const CastSeg* __restrict__ segs,
                                                         const int32_t* __restrict__ blk_seg,
                                                         const int32_t* __restrict__ blk_chunk) {
  const CastSeg s = segs[blk_seg[blockIdx.x]];
  __shared__ __attribute__((aligned(16))) float lds[CAST_LDS_FLOATS];
  if (s.mode == 1) {
    // dest [i][k'][o] (k' = K-1-k) <- src [o][i][k], as 64 x 64 transpose tiles (the table holds one block per TILE for these
    // entries): 256-byte source runs in, 64 consecutive o out, the flip of k resolved on the way out. Destination chunks of a
    // fixed length (the first version) gathered 36-byte source runs for the 3x3 filters.
    float* tile = lds;                                   // 64 x 65 floats of the staging array
    const uint32_t O_ = (uint32_t)s.O, IK_ = (uint32_t)s.I * (uint32_t)s.K, K_ = (uint32_t)s.K;
    const uint32_t tiles_q = (IK_ + 63) / 64;
    const uint32_t t = (uint32_t)blk_chunk[blockIdx.x];
    const uint32_t o0 = (t / tiles_q) * 64, q0 = (t % tiles_q) * 64;
    const float* sp_ = src + s.src_off;
    typename CV::T* dp_ = dst + s.dst_off;
    for (uint32_t idx = threadIdx.x; idx < 4096; idx += 256) {
      const uint32_t ol = idx >> 6, j = idx & 63;
      if (o0 + ol < O_ && q0 + j < IK_) tile[ol * 65 + j] = sp_[(o0 + ol) * IK_ + q0 + j];
    }
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < 4096; idx += 256) {
      const uint32_t ql = idx >> 6, ol = idx & 63;
      const uint32_t q = q0 + ql;
      if (o0 + ol < O_ && q < IK_) {
        const uint32_t i = q / K_, k = q - i * K_;
        dp_[(i * K_ + (K_ - 1 - k)) * O_ + o0 + ol] = CV::cv(tile[ol * 65 + ql]);
      }
    }
    return;
  }
  const int64_t start = (int64_t)blk_chunk[blockIdx.x] * UPD_CHUNK;
  const int64_t end = (start + UPD_CHUNK < s.len) ? start + UPD_CHUNK : s.len;
  // one entry is one weight tensor (< 2^31 elements, checked on the host): 32-bit index math. The permuting modes stage
  // the source through LDS: a destination chunk is contiguous and its sources form a few contiguous runs, so both the
  // global reads and the global writes are coalesced (the first version gathered 4 bytes per 128-byte line).
  const uint32_t I = (uint32_t)s.I, K = (uint32_t)s.K, O = (uint32_t)s.O;
  const uint32_t IK = I * K, KO = K * O;
  const float* sp = src + s.src_off;
  typename CV::T* dp = dst + s.dst_off;
  const uint32_t d0 = (uint32_t)start, d1 = (uint32_t)end;
  if (s.mode == 0 && K > 1) {                       // [o][i][k] -> [o][k][i]: the permutation stays inside each o block
    const uint32_t o0 = d0 / IK, o1 = (d1 - 1) / IK;
    const uint32_t base = o0 * IK, n_src = (o1 + 1 - o0) * IK;
    if (n_src <= CAST_LDS_FLOATS) {
      if (((s.src_off + base) & 3) == 0 && (n_src & 3) == 0) {       // 16-byte source loads
        const f32x4* sp4 = reinterpret_cast<const f32x4*>(sp + base);
        for (uint32_t j = threadIdx.x; j < (n_src >> 2); j += 256) *reinterpret_cast<f32x4*>(lds + 4 * j) = sp4[j];
      } else {
        for (uint32_t j = threadIdx.x; j < n_src; j += 256) lds[j] = sp[base + j];
      }
      __syncthreads();
      if ((I & 3) == 0 && (d0 & 3) == 0 && ((d1 - d0) & 3) == 0 && ((s.dst_off + d0) & 3) == 0) {
        // four consecutive destinations = four consecutive i of one (o, k): one 4-element store per thread
        for (uint32_t d = d0 + 4 * threadIdx.x; d < d1; d += 1024) {
          const uint32_t o = d / IK, rem = d - o * IK;
          const uint32_t k = rem / I, i = rem - k * I;
          const float* q = lds + (o * IK + i * K + k - base);
          typename CV::T v4[4] = {CV::cv(q[0]), CV::cv(q[K]), CV::cv(q[2 * K]), CV::cv(q[3 * K])};
          if (sizeof(typename CV::T) == 2) *reinterpret_cast<uint2*>(dp + d) = *reinterpret_cast<const uint2*>(v4);
          else *reinterpret_cast<uint4*>(dp + d) = *reinterpret_cast<const uint4*>(v4);
        }
        return;
      }
      for (uint32_t d = d0 + threadIdx.x; d < d1; d += 256) {
        const uint32_t o = d / IK, rem = d - o * IK;
        const uint32_t k = rem / I, i = rem - k * I;
        dp[d] = CV::cv(lds[o * IK + i * K + k - base]);
      }
      return;
    }
  } else if (s.mode == 1) {                          // dest [i][k'][o], k' = K-1-k  <-  src [o][i][k]
    const uint32_t r0 = d0 / O, r1 = (d1 - 1) / O;   // destination rows r = i*K + k'
    const uint32_t q0 = (r0 / K) * K, q1 = (r1 / K + 1) * K;   // covering range of source (i, k) indices
    const uint32_t qlen = q1 - q0, pitch = qlen | 1;
    if (O * pitch <= CAST_LDS_FLOATS) {
      for (uint32_t idx = threadIdx.x; idx < O * qlen; idx += 256) {
        const uint32_t o = idx / qlen, j = idx - o * qlen;
        lds[o * pitch + j] = sp[o * IK + q0 + j];
      }
      __syncthreads();
      if ((O & 3) == 0 && (d0 & 3) == 0 && ((d1 - d0) & 3) == 0 && ((s.dst_off + d0) & 3) == 0) {
        // four consecutive destinations = four consecutive o of one (i, k'): one 4-element store per thread (the odd LDS pitch
        // keeps the four reads on different banks)
        for (uint32_t d = d0 + 4 * threadIdx.x; d < d1; d += 1024) {
          const uint32_t r = d / O, o = d - r * O;
          const uint32_t i = r / K, kf = r - i * K;
          const float* q = lds + (o * pitch + i * K + (K - 1 - kf) - q0);
          typename CV::T v4[4] = {CV::cv(q[0]), CV::cv(q[pitch]), CV::cv(q[2 * pitch]), CV::cv(q[3 * pitch])};
          if (sizeof(typename CV::T) == 2) *reinterpret_cast<uint2*>(dp + d) = *reinterpret_cast<const uint2*>(v4);
          else *reinterpret_cast<uint4*>(dp + d) = *reinterpret_cast<const uint4*>(v4);
        }
        return;
      }
      for (uint32_t d = d0 + threadIdx.x; d < d1; d += 256) {
        const uint32_t r = d / O, o = d - r * O;
        const uint32_t i = r / K, kf = r - i * K;
        dp[d] = CV::cv(lds[o * pitch + i * K + (K - 1 - kf) - q0]);
      }
      return;
    }
  }
  if (s.mode == 0 && K == 1 && ((s.src_off + d0) & 3) == 0 && ((s.dst_off + d0) & 3) == 0) {
    // straight cast (1x1 filters, dense weights, biases): 16-byte loads, 4-element stores, scalar tail
    const uint32_t n4 = (d1 - d0) >> 2;
    const f32x4* sp4 = reinterpret_cast<const f32x4*>(sp + d0);
    for (uint32_t j = threadIdx.x; j < n4; j += 256) {
      const f32x4 v = sp4[j];
      typename CV::T v4[4] = {CV::cv(v[0]), CV::cv(v[1]), CV::cv(v[2]), CV::cv(v[3])};
      if (sizeof(typename CV::T) == 2) *reinterpret_cast<uint2*>(dp + d0 + 4 * j) = *reinterpret_cast<const uint2*>(v4);
      else *reinterpret_cast<uint4*>(dp + d0 + 4 * j) = *reinterpret_cast<const uint4*>(v4);
    }
    for (uint32_t d = d0 + 4 * n4 + threadIdx.x; d < d1; d += 256) dp[d] = CV::cv(sp[d]);
    return;
  }
  for (uint32_t d = d0 + threadIdx.x; d < d1; d += 256) {        // straight copies and oversize fallbacks
    uint32_t si = d;
    if (s.mode == 1) {
      const uint32_t i = d / KO, rem = d - i * KO;
      const uint32_t kf = rem / O, o = rem - kf * O;
      si = o * IK + i * K + (K - 1 - kf);
    } else if (K > 1) {
      const uint32_t o = d / IK, rem = d - o * IK;
      const uint32_t k = rem / I, i = rem - k * I;
      si = o * IK + i * K + k;
    }
    dp[d] = CV::cv(sp[si]);
  }
}

// dtype of the replicas: 0 fp32, 1 bf16, 2 fp16
DL4J_API int dl4j_cast_table_dt(const float* src, void* dst, const void* segs, const int32_t* blk_seg,
                                const int32_t* blk_chunk, int num_blocks, int dtype, hipStream_t st) {
  if (num_blocks > 0) {
    if (dtype == 1)
      hipLaunchKernelGGL(cast_table_kernel<CastBf16>, dim3(num_blocks), dim3(256), 0, st, src, (bf16_t*)dst,
                         (const CastSeg*)segs, blk_seg, blk_chunk);
    else if (dtype == 2)
      hipLaunchKernelGGL(cast_table_kernel<CastF16>, dim3(num_blocks), dim3(256), 0, st, src, (_Float16*)dst,
                         (const CastSeg*)segs, blk_seg, blk_chunk);
    else if (dtype == 0)
      hipLaunchKernelGGL(cast_table_kernel<CastF32>, dim3(num_blocks), dim3(256), 0, st, src, (float*)dst,
                         (const CastSeg*)segs, blk_seg, blk_chunk);
    else
      return -2;
  }
  DL4J_RET_LAST();
}
DL4J_API int dl4j_cast_table(const float* src, void* dst_bf16, const void* segs, const int32_t* blk_seg,
                             const int32_t* blk_chunk, int num_blocks, hipStream_t st) {
  return dl4j_cast_table_dt(src, dst_bf16, segs, blk_seg, blk_chunk, num_blocks, 1, st);
}

// Sum of squares over the flat gradient (global L2 norm for gradient clipping), two-stage.
__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float* __restrict__ x, int64_t n,
                                                            float* __restrict__ partial) {
  __shared__ float red[16];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += x[i] * x[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
__global__ void __launch_bounds__(256) sum_final_kernel(const float* __restrict__ partial, int n,
                                                        float* __restrict__ out) {
  __shared__ float red[16];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) out[0] = a;
}
DL4J_API int dl4j_sumsq(const float* x, int64_t n, float* partial /* >=1024 floats */, float* out, hipStream_t st) {
  int g = grid_for(n, 256 * 8);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(g), dim3(256), 0, st, x, n, partial);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, st, partial, g, out);
  DL4J_RET_LAST();
}
