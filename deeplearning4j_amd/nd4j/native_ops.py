"""ctypes view of the native op-execution ABI (native/csrc/rt_opctx.cpp) under the names of the reference's NativeOps interface
(nd4j-api nativeblas/NativeOps.java: createGraphContext, setGraphContextInputArrays, …, execCustomOp2, calculateOutputShapes2 —
libnd4j/include/legacy/NativeOps.h:1488-1508, 1610-1646). This is the surface a JavaCPP preset would bind; Python is just one
client of it. ``exec_custom_op`` shows the calling sequence of CudaExecutioner.exec(CustomOp) (:1801-1859): build shapeInfo for every
array, fill an OpaqueContext, call execCustomOp2 with the op HASH.

Ops that have no native executor yet reach the Python implementations through ``registerCustomOpFallback`` — installed by
:func:`install_python_fallback` — so all 525 ops are callable through the one ABI while the native table grows.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from ..native import lib as L

_FLAG = {torch.bfloat16: 2048, torch.float16: 4096, torch.float32: 8192, torch.float64: 16384, torch.int32: 131072, torch.int64: 262144,
         torch.bool: 524288}
_FROM_FLAG = {v: k for k, v in _FLAG.items()}
_SIG = False
P64 = C.POINTER(C.c_int64)


def lib():
    global _SIG
    rt = L.rt()
    if not _SIG:
        vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
        VPP = C.POINTER(C.c_void_p)
        sig = {
            "createGraphContext": (vp, [i32]), "deleteGraphContext": (None, [vp]), "ctxPurge": (None, [vp]),
            "ctxAllowHelpers": (None, [vp, C.c_bool]), "ctxSetExecutionMode": (None, [vp, i32]), "markGraphContextInplace": (None, [vp, C.c_bool]),
            "setGraphContextCudaContext": (None, [vp, vp, vp, vp]),
            "setGraphContextInputArrays": (None, [vp, i32, VPP, VPP, VPP, VPP]),
            "setGraphContextOutputArrays": (None, [vp, i32, VPP, VPP, VPP, VPP]),
            "setGraphContextTArguments": (None, [vp, C.POINTER(C.c_double), i32]),
            "setGraphContextIArguments": (None, [vp, P64, i32]),
            "setGraphContextBArguments": (None, [vp, C.POINTER(C.c_bool), i32]),
            "setGraphContextDArguments": (None, [vp, C.POINTER(C.c_int), i32]),
            "ctxNumInputs": (i32, [vp]), "ctxNumOutputs": (i32, [vp]), "ctxInputBuffer": (vp, [vp, i32, i32]),
            "ctxInputShapeInfo": (P64, [vp, i32]), "ctxOutputBuffer": (vp, [vp, i32, i32]), "ctxOutputShapeInfo": (P64, [vp, i32]),
            "ctxNumTArgs": (i32, [vp]), "ctxNumIArgs": (i32, [vp]), "ctxNumBArgs": (i32, [vp]), "ctxTArg": (C.c_double, [vp, i32]),
            "ctxIArg": (i64, [vp, i32]), "ctxBArg": (i32, [vp, i32]),
            "registerCustomOpFallback": (None, [vp, vp]), "nativeOpImplemented": (i32, [i64]), "nativeOpCount": (i32, []),
            "execCustomOp2": (i32, [VPP, i64, vp]), "nativeOpList": (C.c_char_p, []),
            "calculateOutputShapes2": (vp, [VPP, i64, VPP, VPP, i32, C.POINTER(C.c_double), i32, P64, i32, C.POINTER(C.c_bool), i32,
                                            C.POINTER(C.c_int), i32]),
            "getShapeListSize": (i64, [vp]), "getShape": (P64, [vp, i64]), "deleteShapeList": (None, [vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(rt, name)
            fn.restype, fn.argtypes = res, args
        _SIG = True
    return rt


def shape_info(t: torch.Tensor):
    """shapeInfo descriptor (LongType[2*rank+4]) of a c-contiguous tensor, data type in the extras field."""
    rank = t.dim()
    out = (C.c_int64 * (2 * rank + 4))()
    shp = (C.c_int64 * max(rank, 1))(*[int(s) for s in t.shape])
    L.rt().nd4j_shape_buffer(rank, shp, b"c", C.c_int64(_FLAG[t.dtype]), out)
    return out


def _ptr_array(vals):
    return (C.c_void_p * max(len(vals), 1))(*[C.c_void_p(v) if v else None for v in vals])


def _describe(si):
    rank = int(si[0])
    shape = [int(si[1 + k]) for k in range(rank)]
    extras = int(si[1 + 2 * rank])
    dt = next((d for f, d in _FROM_FLAG.items() if extras & f), torch.float32)
    return shape, dt


def calculate_output_shapes(name: str, inputs: Sequence[torch.Tensor], iArgs=(), tArgs=(), bArgs=(), dArgs=()):
    """[(shape, dtype)] from the NATIVE shape function of ``name`` (None if the op has none)."""
    rt = lib()
    sis = [shape_info(t) for t in inputs]
    ips = _ptr_array([C.addressof(s) for s in sis])
    t = (C.c_double * max(len(tArgs), 1))(*[float(v) for v in tArgs])
    i = (C.c_int64 * max(len(iArgs), 1))(*[int(v) for v in iArgs])
    b = (C.c_bool * max(len(bArgs), 1))(*[bool(v) for v in bArgs])
    d = (C.c_int * max(len(dArgs), 1))(*[int(v) for v in dArgs])
    lst = rt.calculateOutputShapes2(None, C.c_int64(rt.ops_hash(name.encode())), None, ips, len(sis), t, len(tArgs), i, len(iArgs), b,
                                    len(bArgs), d, len(dArgs))
    if not lst:
        return None
    try:
        return [_describe(rt.getShape(lst, k)) for k in range(rt.getShapeListSize(lst))]
    finally:
        rt.deleteShapeList(lst)


def exec_custom_op(name: str, inputs: Sequence[torch.Tensor], outputs: Optional[Sequence[torch.Tensor]] = None, iArgs=(), tArgs=(),
                   bArgs=(), stream=None, dArgs=()) -> List[torch.Tensor]:
    """Run ``name`` through execCustomOp2. Host tensors travel as (buffer, shapeInfo), GPU tensors as special buffers. Without
    ``outputs`` they are allocated from the native shape function."""
    rt = lib()
    inputs = [t.contiguous() for t in inputs]
    if outputs is None:
        shapes = calculate_output_shapes(name, inputs, iArgs, tArgs, bArgs, dArgs)
        if shapes is None:
            raise NotImplementedError("op '%s' has no native shape function: pass preallocated outputs" % name)
        outputs = [torch.empty(s, dtype=dt, device=inputs[0].device) for s, dt in shapes]
    ctx = rt.createGraphContext(0)
    keep = []
    try:
        for arrs, setter in ((inputs, rt.setGraphContextInputArrays), (outputs, rt.setGraphContextOutputArrays)):
            sis = [shape_info(t) for t in arrs]
            keep.append(sis)
            host = _ptr_array([0 if t.is_cuda else t.data_ptr() for t in arrs])
            dev = _ptr_array([t.data_ptr() if t.is_cuda else 0 for t in arrs])
            sip = _ptr_array([C.addressof(s) for s in sis])
            keep += [host, dev, sip]
            setter(ctx, len(arrs), host, sip, dev, sip)
        if tArgs:
            rt.setGraphContextTArguments(ctx, (C.c_double * len(tArgs))(*[float(v) for v in tArgs]), len(tArgs))
        if iArgs:
            rt.setGraphContextIArguments(ctx, (C.c_int64 * len(iArgs))(*[int(v) for v in iArgs]), len(iArgs))
        if bArgs:
            rt.setGraphContextBArguments(ctx, (C.c_bool * len(bArgs))(*[bool(v) for v in bArgs]), len(bArgs))
        if dArgs:
            rt.setGraphContextDArguments(ctx, (C.c_int * len(dArgs))(*[int(v) for v in dArgs]), len(dArgs))
        if inputs and inputs[0].is_cuda:
            s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
            rt.setGraphContextCudaContext(ctx, C.c_void_p(s), None, None)
        rc = rt.execCustomOp2(None, C.c_int64(rt.ops_hash(name.encode())), ctx)
        if rc != 0:
            raise RuntimeError("execCustomOp2(%s) failed with status %d: %s" % (name, rc, (rt.nd4j_last_error_message() or b"").decode()))
        return list(outputs)
    finally:
        rt.deleteGraphContext(ctx)


def native_op_names() -> List[str]:
    """Names of the ops that execCustomOp2 runs in C++ / HIP without any fallback."""
    return [n for n in (lib().nativeOpList() or b"").decode().split("\n") if n]


def list_custom_ops() -> dict:
    """getAllCustomOps-style listing (NativeOps.h getAllCustomOps): every registered op name -> "native" (C++ executor and shape
    function behind execCustomOp2 / calculateOutputShapes2) or "fallback" (served by the Python op library through
    registerCustomOpFallback)."""
    from .ops import custom
    custom._load_impls()
    native = set(native_op_names())
    return {n: ("native" if n in native else "fallback") for n in sorted(set(custom._IMPL) | native)}


# ---------------------------------------------------------------------------------------------- fallback into the Python op library
_FALLBACK_CB = None
_NP = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64, torch.bool: np.bool_,
       torch.float16: np.float16}


def _wrap_host(ptr, shape, dt):
    n = int(np.prod(shape)) if shape else 1
    if dt == torch.bfloat16:
        buf = (C.c_uint16 * n).from_address(ptr)
        return torch.frombuffer(buf, dtype=torch.bfloat16).view(shape)
    buf = (C.c_char * (n * np.dtype(_NP[dt]).itemsize)).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=_NP[dt]).reshape(shape))


def install_python_fallback():
    """Ops without a native executor are served by the registered Python implementations, reading / writing the context's HOST
    buffers in place (device buffers need an owning tensor and stay with the native table)."""
    global _FALLBACK_CB
    rt = lib()
    from .ops import custom

    @C.CFUNCTYPE(C.c_int, C.c_int64, C.c_void_p, C.c_void_p)
    def cb(h, ctx, _user):
        try:
            name = custom.name_by_hash(h)
            if name is None:
                return 14
            ins = []
            for k in range(rt.ctxNumInputs(ctx)):
                shape, dt = _describe(rt.ctxInputShapeInfo(ctx, k))
                p = rt.ctxInputBuffer(ctx, k, 0)
                if not p:
                    return 15
                ins.append(_wrap_host(p, shape, dt))
            t = [rt.ctxTArg(ctx, k) for k in range(rt.ctxNumTArgs(ctx))]
            i = [rt.ctxIArg(ctx, k) for k in range(rt.ctxNumIArgs(ctx))]
            b = [bool(rt.ctxBArg(ctx, k)) for k in range(rt.ctxNumBArgs(ctx))]
            outs = custom.exec_op(name, ins, iArgs=i, tArgs=t, bArgs=b)
            for k in range(min(rt.ctxNumOutputs(ctx), len(outs))):
                shape, dt = _describe(rt.ctxOutputShapeInfo(ctx, k))
                dst = _wrap_host(rt.ctxOutputBuffer(ctx, k, 0), shape, dt)
                dst.copy_(outs[k].reshape(shape).to(dt))
            return 0
        except Exception:      # noqa: BLE001 — errors cross the ABI as status codes
            return 16
    _FALLBACK_CB = cb
    rt.registerCustomOpFallback(C.cast(cb, C.c_void_p), None)


def uninstall_fallback():
    global _FALLBACK_CB
    lib().registerCustomOpFallback(None, None)
    _FALLBACK_CB = None
