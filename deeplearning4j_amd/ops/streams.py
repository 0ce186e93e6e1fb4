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


# ---------------------------------------------------------------------------------------------------------------------------------
# Library GEMMs and side streams
#
# Round 2 found that two fp32 vendor GEMMs issued at the same time from two HIP streams can dead-lock the device (bidirectional LSTM,
# input projections on two streams; profiles/r2_lstm_persistent.md, cause analysis in profiles/r3_library_gemm_streams.md). The
# reference serialises its vendor GEMMs for the same reason: one mutex per device around every cuBLAS call
# (libnd4j/include/helpers/cuda_off/MmulHelper.cu:261). Here the rule "vendor GEMMs run on one stream at a time" is enforced in code:
#
#   * ``library_gemm_guard("assert")``  - entered by every side-stream region of this package: reaching a vendor GEMM entry point with
#     device operands inside it raises LibraryGemmOffPrimaryStream (the side streams only ever carry native/csrc kernels, none of which
#     waits on another workgroup).
#   * ``library_gemm_guard("route")``   - entered by the ParallelInference workers (each owns a stream): mm / matmul / addmm / linear
#     on 2-D operands go to the in-tree MFMA GEMM (ops/gemm.py, bf16 / fp16 / fp32); whatever cannot (batched forms, unaligned rows)
#     runs under the per-device token below, so two vendor GEMMs never overlap on the device even though the streams do.
#   * ``library_gemm_token(device)``    - lock + event chain: the holder's stream first waits for the event recorded after the previous
#     holder's work, and records its own on release. Other kernels of the two streams still overlap.
#
# The guard is a TorchFunctionMode (thread-local, costs one set lookup per torch call inside the scope, nothing outside it);
# DL4J_AMD_STREAM_GUARD=0 turns the side-stream assertions off.
import threading

from torch.overrides import TorchFunctionMode

GUARD_ENABLED = os.environ.get("DL4J_AMD_STREAM_GUARD", "1") != "0"


class LibraryGemmOffPrimaryStream(RuntimeError):
    pass


_LIB_FUNCS = None


def _library_gemm_functions():
    global _LIB_FUNCS
    if _LIB_FUNCS is None:
        F, T = torch.nn.functional, torch.Tensor
        names = [torch.mm, torch.matmul, torch.addmm, torch.bmm, torch.baddbmm, torch.addbmm, torch.mv, torch.addmv, torch.einsum,
                 torch.tensordot, torch.linalg.matmul, torch.linalg.multi_dot, torch.chain_matmul, torch.inner, torch.outer, torch.addr,
                 F.linear, F.bilinear, T.mm, T.matmul, T.__matmul__, T.__rmatmul__, T.addmm, T.addmm_, T.bmm, T.baddbmm, T.baddbmm_,
                 T.addbmm, T.addbmm_, T.mv, T.addmv, T.addmv_, T.inner, T.outer, T.addr, T.addr_]
        _LIB_FUNCS = frozenset(names)
    return _LIB_FUNCS


def _device_float_operand(args, kwargs, device_types):
    for a in tuple(args) + tuple(kwargs.values()):
        if isinstance(a, torch.Tensor):
            if a.device.type in device_types and a.is_floating_point():
                return a
        elif isinstance(a, (list, tuple)):
            for b in a:
                if isinstance(b, torch.Tensor) and b.device.type in device_types and b.is_floating_point():
                    return b
    return None


class _Token:
    def __init__(self):
        self.lock = threading.RLock()
        self.event = None
        self.stream = None
        self.depth = 0


_TOKENS = {}
_TOKENS_LOCK = threading.Lock()


class library_gemm_token:
    """Device-wide order for vendor GEMMs issued from different streams / host threads (see above)."""

    def __init__(self, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.index = dev.index if dev.index is not None else torch.cuda.current_device()
        with _TOKENS_LOCK:
            self.tok = _TOKENS.setdefault(self.index, _Token())

    def __enter__(self):
        t = self.tok
        t.lock.acquire()
        t.depth += 1
        if t.depth == 1:
            cur = torch.cuda.current_stream(self.index)
            if t.event is not None and t.stream != cur and not torch.cuda.is_current_stream_capturing():
                cur.wait_event(t.event)
        return self

    def __exit__(self, *exc):
        t = self.tok
        t.depth -= 1
        if t.depth == 0 and not torch.cuda.is_current_stream_capturing():
            cur = torch.cuda.current_stream(self.index)
            if t.event is None:
                t.event = torch.cuda.Event()
            t.event.record(cur)
            t.stream = cur
        t.lock.release()
        return False


def _as_nt(b):
    """[K, N] right operand -> its [N, K] K-contiguous image (free for a transposed view, one copy kernel otherwise)."""
    bt = b.t()
    return bt if bt.stride(1) == 1 else bt.contiguous()


def _route_in_tree(func, args, kwargs):
    """(handled, result): 2-D mm / matmul / addmm / linear on the in-tree GEMM when its operand rules hold."""
    from . import gemm as G
    F, T = torch.nn.functional, torch.Tensor
    if torch.is_grad_enabled() and any(isinstance(a, torch.Tensor) and a.requires_grad for a in args):
        return False, None
    try:
        if func in (torch.mm, torch.matmul, T.mm, T.matmul, T.__matmul__, torch.linalg.matmul) and len(args) == 2 and not kwargs:
            a, b = args
            if a.dim() == 2 and b.dim() == 2:
                bt = _as_nt(b)
                a = a if a.stride(1) == 1 else a.contiguous()
                if G.supported(a, bt):
                    return True, G.gemm_nt(a, bt)
        elif func is F.linear:
            x, w = args[0], args[1]
            bias = args[2] if len(args) > 2 else kwargs.get("bias")
            if w.dim() == 2 and x.dim() >= 2:
                x2 = x.reshape(-1, x.shape[-1])
                if G.supported(x2, w):
                    y = G.gemm_nt(x2, w, bias=bias)
                    return True, y.view(*x.shape[:-1], w.shape[0])
        elif func in (torch.addmm, T.addmm) and len(args) == 3 and not kwargs:
            c, a, b = args
            if a.dim() == 2 and b.dim() == 2 and c.dim() == 1 and c.shape[0] == b.shape[1]:
                bt = _as_nt(b)
                a = a if a.stride(1) == 1 else a.contiguous()
                if G.supported(a, bt):
                    return True, G.gemm_nt(a, bt, bias=c)
    except (AssertionError, RuntimeError):
        pass
    return False, None


class library_gemm_guard(TorchFunctionMode):
    """mode 'assert': a vendor GEMM with device operands raises; mode 'route': in-tree GEMM where possible, device token otherwise.
    ``counts`` = {'routed', 'serialized', 'captured'}: 'captured' counts vendor GEMMs recorded into a HIP graph (a replay cannot be
    ordered from inside; the caller replays such a graph under ``library_gemm_token``)."""

    def __init__(self, mode="assert", where="side stream", device_types=("cuda",)):
        super().__init__()
        assert mode in ("assert", "route")
        self.mode, self.where, self.device_types = mode, where, tuple(device_types)
        self.counts = {"routed": 0, "serialized": 0, "captured": 0}
        self.saw_capture = False

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _library_gemm_functions():
            t = _device_float_operand(args, kwargs, self.device_types)
            if t is not None:
                if self.mode == "assert":
                    raise LibraryGemmOffPrimaryStream(
                        "%s reached inside a %s region: vendor GEMMs run on the primary stream only (two of them on two streams can "
                        "dead-lock the device); issue it before the fork / after the join or use ops/gemm.py" %
                        (getattr(func, "__name__", str(func)), self.where))
                if t.is_cuda:
                    ok, y = _route_in_tree(func, args, kwargs)
                    if ok:
                        self.counts["routed"] += 1
                        return y
                    if torch.cuda.is_current_stream_capturing():
                        self.counts["captured"] += 1
                        self.saw_capture = True
                        return func(*args, **kwargs)
                    self.counts["serialized"] += 1
                    with library_gemm_token(t.device):
                        return func(*args, **kwargs)
        return func(*args, **kwargs)


class _stream_and_guard:
    """``with side_region(stream):`` = torch.cuda.stream(stream) + the assert-mode guard."""

    def __init__(self, stream, where):
        self._s = torch.cuda.stream(stream)
        self._g = library_gemm_guard("assert", where) if GUARD_ENABLED else None

    def __enter__(self):
        self._s.__enter__()
        if self._g is not None:
            self._g.__enter__()
        return self

    def __exit__(self, *exc):
        if self._g is not None:
            self._g.__exit__(*exc)
        return self._s.__exit__(*exc)


def side_region(stream, where="side stream"):
    return _stream_and_guard(stream, where)
