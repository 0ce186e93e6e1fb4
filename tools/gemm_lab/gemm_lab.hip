// Stand-alone bench + correctness harness for the LDS-DMA GEMM core (no torch): builds here, runs on the GPU box.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deeplearning4j_amd/native/csrc tools/gemm_lab/gemm_lab.hip -o tools/gemm_lab/gemm_lab
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "gemm_kernels.h"
using namespace gemmcore;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void ref_gemm_kernel(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K) {
  const int n = blockIdx.x * 16 + threadIdx.x, m = blockIdx.y * 16 + threadIdx.y;
  if (m >= M || n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += bf2f(A[(int64_t)m * K + k]) * bf2f(B[(int64_t)n * K + k]);
  C[(int64_t)m * N + n] = s;
}
__global__ void fill_kernel(bf16_t* p, int64_t n, uint32_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = f2bf(((h & 0xffff) / 32768.0f) - 1.0f);
  }
}
__global__ void cmp_kernel(const bf16_t* C, const float* R, int64_t n, float* maxerr, float* maxref) {
  float e = 0.f, r = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    e = fmaxf(e, fabsf(bf2f(C[i]) - R[i])); r = fmaxf(r, fabsf(R[i]));
  }
  atomicMax((int*)maxerr, __float_as_int(e)); atomicMax((int*)maxref, __float_as_int(r));
}

struct Variant { const char* name; int (*launch)(GemmP&, int, hipStream_t); int bm, bn, gm; };
#define V(bm, bn, wm, wn, st, mf, gm) {#bm "x" #bn "_w" #wm "x" #wn "_s" #st "_mf" #mf "_g" #gm, &launch_gemm<Cfg<bm, bn, wm, wn, st, mf>, OT_BF16>, bm, bn, gm}
static Variant variants[] = {
  {"256x256_pingpong_g4", &launch_gemm<Cfg<256, 256, 2, 4, 2, 16>, OT_BF16, true>, 256, 256, 4},
  {"256x256_pingpong_g8", &launch_gemm<Cfg<256, 256, 2, 4, 2, 16>, OT_BF16, true>, 256, 256, 8},
  V(256, 256, 2, 4, 2, 32, 1), V(256, 256, 2, 4, 2, 32, 4), V(256, 256, 2, 4, 2, 16, 4), V(256, 256, 2, 4, 2, 16, 8),
  V(256, 128, 4, 2, 3, 32, 4), V(256, 128, 4, 2, 3, 16, 4), V(256, 128, 4, 2, 2, 16, 4), V(128, 256, 2, 4, 3, 16, 4),
  V(128, 128, 2, 2, 2, 32, 4), V(128, 128, 2, 2, 2, 16, 4), V(128, 128, 2, 2, 2, 16, 8),
  V(128, 64, 2, 2, 3, 32, 4), V(128, 64, 2, 2, 3, 16, 4), V(128, 64, 2, 2, 2, 16, 4), V(128, 64, 2, 2, 2, 16, 8), V(128, 64, 2, 2, 1, 16, 8),
  V(64, 64, 2, 2, 4, 16, 4), V(64, 64, 2, 2, 2, 16, 8), V(64, 64, 2, 2, 1, 16, 8), V(64, 128, 2, 2, 2, 16, 8),
};

int main(int argc, char** argv) {
  struct Shape { int M, N, K; };
  std::vector<Shape> shapes = {{4096, 768, 768}, {4096, 3072, 768}, {4096, 768, 3072}, {8192, 4096, 4096}, {16384, 1024, 1024},
                               {65536, 256, 64}, {200704, 256, 64}, {1024, 1024, 1024}, {12544, 256, 1024}, {50176, 128, 512},
                               {1000, 520, 264}};
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const char* vfilter = argc > 2 ? argv[2] : "";          // substring of the variant name ("" = all)
  if (argc > 5) shapes = {{atoi(argv[3]), atoi(argv[4]), atoi(argv[5])}};
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* d_err; CK(hipMalloc(&d_err, 8));
  for (auto s : shapes) {
    const int64_t na = (int64_t)s.M * s.K, nb = (int64_t)s.N * s.K, nc = (int64_t)s.M * s.N;
    bf16_t *A, *B, *C; float* R;
    CK(hipMalloc(&A, na * 2)); CK(hipMalloc(&B, nb * 2)); CK(hipMalloc(&C, nc * 2)); CK(hipMalloc(&R, nc * 4));
    fill_kernel<<<2048, 256, 0, st>>>(A, na, 1u); fill_kernel<<<2048, 256, 0, st>>>(B, nb, 77u);
    const bool check = (double)s.M * s.N * s.K < 3e11;
    if (check) ref_gemm_kernel<<<dim3((s.N + 15) / 16, (s.M + 15) / 16), dim3(16, 16), 0, st>>>(A, B, R, s.M, s.N, s.K);
    CK(hipStreamSynchronize(st));
    printf("shape M=%d N=%d K=%d  (%.2f GFLOP)\n", s.M, s.N, s.K, 2.0 * s.M * s.N * s.K / 1e9);
    for (auto& v : variants) {
      if (vfilter[0] && !strstr(v.name, vfilter)) continue;
      if (v.launch == nullptr) continue;
      if ((s.K + 63) / 64 > 1 && strstr(v.name, "_s1_")) continue;     // single-stage build is for one k-tile only
      GemmP p{}; p.gm = v.gm; p.A = A; p.B = B; p.epi.out = C; p.epi.alpha = 1.f; p.M = s.M; p.N = s.N; p.K = s.K; p.lda = s.K; p.ldb = s.K; p.ldc = s.N;
      CK(hipMemsetAsync(C, 0, nc * 2, st));
      int rc = v.launch(p, 1, st);
      if (rc) { printf("  %-18s launch rc=%d\n", v.name, rc); continue; }
      float err[2] = {0, 0};
      if (check) {
        CK(hipMemsetAsync(d_err, 0, 8, st));
        cmp_kernel<<<1024, 256, 0, st>>>(C, R, nc, d_err, d_err + 1);
        CK(hipMemcpyAsync(err, d_err, 8, hipMemcpyDeviceToHost, st));
      }
      CK(hipStreamSynchronize(st));
      for (int i = 0; i < 3; ++i) v.launch(p, 1, st);
      std::vector<float> ts;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) v.launch(p, 1, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1000.f / iters);
      }
      std::sort(ts.begin(), ts.end());
      const double tf = 2.0 * s.M * s.N * s.K / ts[0] / 1e6;
      printf("  %-18s %8.1f us  %7.0f TF/s  tiles=%d  relerr=%.2e %s\n", v.name, ts[0], tf, p.mtiles * p.ntiles,
             check ? err[0] / fmaxf(err[1], 1e-20f) : -1.f, (check && err[0] > 0.02f * err[1]) ? "MISMATCH" : "");
      fflush(stdout);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(R));
  }
  return 0;
}
