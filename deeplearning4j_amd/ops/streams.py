"""Side HIP stream for weight-gradient kernels.

In backward, a convolution's wgrad and dgrad both consume dz and are independent of each other; only dgrad is on the
critical path (its result feeds the previous layer), while dW is not read before the gradient all-reduce / the updater.
The network's ``_backward`` therefore opens a scope in which wgrad kernels are issued on a second stream: they overlap the
next layers' BatchNorm backward chain (bandwidth- and latency-bound: profiles/r1_hbm_traffic.md) with MFMA work. Inside a
HIP-graph capture the fork/join becomes a parallel branch of the graph. The scope joins the side stream on exit, and
the data-parallel hooks join it before a gradient bucket is handed to RCCL or a capture segment is cut.

Tensors read on the side stream are kept referenced until the join, so the caching allocator (also the graph-private
pool during capture) cannot hand their memory to a later main-stream allocation while the side stream still reads it.
Measured on the ResNet-50 headline step (profiles/r1_perf_notes.md): 10.50 ms with the side stream vs 10.25 ms without —
the wgrad grids are already sized to fill all 256 CUs (split-K chosen from the CU capacity), so a concurrent BatchNorm
kernel only takes CUs and HBM bandwidth away from them and the 52 fork/join edges add dependency latency to the graph.
The mechanism therefore stays OFF by default; ``DL4J_AMD_WGRAD_STREAM=1`` enables it (useful for small-batch training
where single kernels do not fill the chip).
"""
from __future__ import annotations

import os

import torch

_side = None
_pending = False
_held = []
_depth = 0
_ENABLED = os.environ.get("DL4J_AMD_WGRAD_STREAM", "0") == "1"


def set_enabled(flag: bool):
    global _ENABLED
    join()
    _ENABLED = bool(flag)


def active() -> bool:
    return _ENABLED and _depth > 0


def fork(*tensors):
    """Make the side stream wait for everything issued so far on the current stream; returns the side stream."""
    global _side, _pending
    if _side is None:
        _side = torch.cuda.Stream()
    _side.wait_stream(torch.cuda.current_stream())
    _held.extend(tensors)
    _pending = True
    return _side


def join():
    """Current stream waits for the side stream's work (no-op when nothing is pending)."""
    global _pending
    if _pending:
        torch.cuda.current_stream().wait_stream(_side)
        _held.clear()
        _pending = False


# Dense layers (ops/linear.py FusedDenseFn.backward): dW = x^T . dy goes straight into the flat gradient vector and nothing reads it before
# the updater, while dx = dy . W^T is on the critical path. At transformer sizes neither GEMM fills the chip (BERT-base, 4096 tokens:
# 192 tiles of 128 x 128 for dx, 36 output tiles x split-K for dW), so the two run side by side. Both are in-tree kernels that never wait
# on another workgroup (see nn/layers/lstm_stack.py for why library GEMMs are kept off side streams).
# Measured on the BERT-base step (tools/bench_bert.py, batch 32 x 128, captured step): 11.19 ms with the side stream, 11.08 ms without:
# the split-K dW grid already covers the chip, so like the convolution case this stays OFF by default (DL4J_AMD_DENSE_WGRAD_STREAM=1).
_DENSE_ENABLED = os.environ.get("DL4J_AMD_DENSE_WGRAD_STREAM", "0") == "1"
_dense_depth = 0


def dense_active() -> bool:
    return _DENSE_ENABLED and _dense_depth > 0


class dense_scope:
    """Backward region in which dense weight gradients may run on the side stream; joins on exit."""

    def __enter__(self):
        global _dense_depth
        _dense_depth += 1
        return self

    def __exit__(self, *exc):
        global _dense_depth
        _dense_depth -= 1
        if _dense_depth == 0:
            join()
        return False


class backward_scope:
    def __enter__(self):
        global _depth
        _depth += 1
        return self

    def __exit__(self, *exc):
        global _depth
        _depth -= 1
        if _depth == 0:
            join()
        return False
