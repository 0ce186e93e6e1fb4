"""ctypes bindings for the in-tree native libraries.

``hip()`` returns the HIP kernel library (``libdl4j_hip.so``); it is only loadable where the
HIP runtime is present. On a GPU box a missing library is a hard error (never a silent
fallback to eager PyTorch): callers use :func:`use_hip` to decide and :func:`hip` raises.
``rt()`` returns the host C++ runtime (``libnd4jrocm_rt.so``) which has no GPU dependency.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
_HIP = None
_RT = None
_HIP_ERR = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64
c_float = ctypes.c_float


def _load(path: Path):
    if not path.exists():
        from . import build as _b
        _b.build()
    return ctypes.CDLL(str(path), mode=ctypes.RTLD_GLOBAL)


def hip():
    """The HIP kernel library. Raises if it cannot be loaded."""
    global _HIP, _HIP_ERR
    if _HIP is None:
        try:
            _HIP = _load(HERE / "libdl4j_hip.so")
            _HIP.dl4j_bn_ws_floats.restype = ctypes.c_int64
        except Exception as e:  # pragma: no cover - depends on box
            _HIP_ERR = e
            raise RuntimeError(
                "deeplearning4j_amd: libdl4j_hip.so could not be loaded (%s). Build it with "
                "`python -m deeplearning4j_amd.native.build` — GPU ops never fall back silently." % (e,))
    return _HIP


def rt():
    """The host C++ runtime library."""
    global _RT
    if _RT is None:
        _RT = _load(HERE / "libnd4jrocm_rt.so")
        _declare_rt(_RT)
    return _RT


def _declare_rt(lib):
    # signatures are declared lazily by the modules that use them
    pass


def gpu_available() -> bool:
    return torch.cuda.is_available()


def use_hip(*tensors) -> bool:
    """True when every tensor lives on the GPU (then the HIP path is mandatory)."""
    ok = False
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            return False
        ok = True
    return ok and os.environ.get("DL4J_AMD_FORCE_EAGER", "0") != "1"


def stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> c_void_p:
    return c_void_p(0 if t is None else t.data_ptr())


DT_F32, DT_BF16 = 0, 1


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    return -1


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError("HIP kernel launch failed in %s (hipError %d)" % (what, code))
