#!/usr/bin/env python
"""Evidence for the two-stream fp32 vendor-GEMM dead-lock of round 2, collected WITHOUT provoking it: everything below runs on ONE
stream. Run under `rocprofv3 --kernel-trace`; tools/library_gemm_report.py turns the trace into profiles/r3_library_gemm_streams.md.

1. the fp32 GEMMs of the bidirectional LSTM case (2 x GravesBidirectionalLSTM(512) on [64, 77, 50]: input projections
   [3200, 77] x [77, 2048] and [3200, 1024] x [1024, 2048], their weight-gradient and input-gradient forms) through torch, i.e. the
   vendor library, so the trace names the solutions (tile, stream-K / split-K fields, workgroup counts, scratch / workspace use);
2. host-side facts about how torch binds the library to streams: BLAS handle per stream, workspace configuration."""
import json
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])


def main():
    d = torch.device("cuda", 0)
    facts = {"torch": torch.__version__, "hip": torch.version.hip,
             "preferred_blas_library": str(torch.backends.cuda.preferred_blas_library()),
             "env": {k: os.environ.get(k) for k in ("CUBLAS_WORKSPACE_CONFIG", "HIPBLASLT_WORKSPACE_SIZE", "ROCBLAS_STREAM_ORDER_ALLOC",
                                                    "TORCH_BLAS_PREFER_HIPBLASLT", "GPU_MAX_HW_QUEUES", "ROCBLAS_LAYER", "HIPBLASLT_LOG_LEVEL")},
             "cus": torch.cuda.get_device_properties(0).multi_processor_count}
    # one handle per (thread, device) in torch; the stream is set on it per call. Query only - no GEMM is issued on the second stream.
    try:
        h0 = torch.cuda.current_blas_handle()
        with torch.cuda.stream(torch.cuda.Stream()):
            h1 = torch.cuda.current_blas_handle()
        facts["blas_handle_main_stream"], facts["blas_handle_side_stream"] = int(h0), int(h1)
        facts["same_handle_on_both_streams"] = int(h0) == int(h1)
    except Exception as e:
        facts["blas_handle_error"] = repr(e)
    T, B, V, H = 50, 64, 77, 512
    shapes = {"proj_l0 addmm [3200,77]x[77,2048]": (T * B, V, 4 * H), "proj_l1 addmm [3200,1024]x[1024,2048]": (T * B, 2 * H, 4 * H)}
    mem0 = torch.cuda.memory_allocated()
    for name, (M, K, N), dt in [(n, s, t) for t in (torch.float32, torch.bfloat16) for n, s in shapes.items()]:
        x = torch.randn(M, K, device=d, dtype=dt)
        w = torch.randn(K, N, device=d, dtype=dt)
        b = torch.randn(N, device=d, dtype=dt)
        dz = torch.randn(M, N, device=d, dtype=dt)
        for _ in range(3):
            torch.addmm(b, x, w)                # forward projection (NN)
            torch.mm(x.t(), dz)                 # weight gradient   (TN)
            torch.mm(dz, w.t())                 # input gradient    (NT)
        torch.cuda.synchronize()
    # recurrent-step GEMM of the older loop form: [64, 512] x [512, 2048]
    h = torch.randn(B, H, device=d)
    rw = torch.randn(H, 4 * H, device=d)
    for _ in range(3):
        torch.mm(h, rw)
    torch.cuda.synchronize()
    facts["reserved_bytes_after"] = torch.cuda.memory_reserved()
    facts["allocated_growth_bytes"] = torch.cuda.memory_allocated() - mem0
    print(json.dumps(facts))


if __name__ == "__main__":
    main()
