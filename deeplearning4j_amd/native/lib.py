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
            for _n in ("dl4j_bn_ws_floats", "dl4j_threshold_ws_ints", "dl4j_legacy_reduce_segs", "dl4j_legacy_reduce_partial_len"):
                getattr(_HIP, _n).restype = ctypes.c_int64
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
    C = ctypes
    P64 = C.POINTER(C.c_int64)
    PI = C.POINTER(C.c_int)
    PF = C.POINTER(C.c_float)
    vp, i32, i64, u64, f32, f64, cp = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_char_p
    sig = {
        # error channel
        "nd4j_last_error_code": (i32, []), "nd4j_last_error_message": (cp, []), "nd4j_clear_error": (None, []),
        "nd4j_backtrace_toggle": (None, [i32]), "nd4j_backtrace_enabled": (i32, []), "nd4j_last_error_backtrace": (cp, []),
        "nd4j_error_category": (cp, [i32]), "nd4j_current_backtrace": (i64, [cp, i64]), "nd4j_raise_for_test": (i32, [i32, i64]),
        "nd4j_install_crash_handler": (None, []),
        # shape / TAD
        "nd4j_shape_info_length": (i32, [i32]),
        "nd4j_shape_buffer": (i32, [i32, P64, C.c_char, i64, P64]),
        "nd4j_shape_buffer_ex": (i32, [i32, P64, P64, C.c_char, i64, P64]),
        "nd4j_shape_intern": (vp, [P64]), "nd4j_shape_cache_size": (i64, []),
        "nd4j_shape_length": (i64, [P64]),
        "nd4j_index2coords": (None, [i64, P64, P64]), "nd4j_coords2offset": (i64, [P64, P64]),
        "nd4j_coords2index": (i64, [P64, P64]), "nd4j_index_offset": (i64, [i64, P64]),
        "nd4j_index_offsets": (None, [P64, P64]),
        "nd4j_permute": (i32, [P64, PI, P64]),
        "nd4j_broadcast_shape": (i32, [i32, P64, i32, P64, P64]),
        "nd4j_reshape_nocopy": (i32, [P64, i32, P64, C.c_char, P64]),
        "nd4j_tad_pack": (vp, [P64, PI, i32]), "nd4j_tad_num": (i64, [vp]),
        "nd4j_tad_shape_info": (P64, [vp]), "nd4j_tad_offsets": (P64, [vp]), "nd4j_tad_cache_size": (i64, []),
        "nd4j_ravel_multi_index": (None, [P64, P64, i64, P64, i32, i32]),
        "nd4j_unravel_index": (None, [P64, P64, i64, P64, i32]),
        # workspace / memory
        "ws_create": (vp, [i32, f64, i64]), "ws_destroy": (None, [vp]), "ws_bind": (i32, [vp, i32, u64, i64]),
        "ws_scope_in": (None, [vp]), "ws_rewind": (None, [vp, i32]), "ws_scope_out": (None, [vp, P64, P64]), "ws_alloc": (u64, [vp, i32, i64]),
        "ws_stats": (None, [vp, i32, P64]), "ws_owns": (i32, [vp, i32, u64]),
        "mem_counter_add": (i64, [i32, i32, i64]), "mem_counter_get": (i64, [i32, i32]),
        "mem_counter_set_limit": (None, [i32, i32, i64]), "mem_counter_get_limit": (i64, [i32, i32]),
        "mem_counter_validate": (i32, [i32, i32, i64]),
        # codecs
        "nd4j_threshold_count_f32": (i64, [vp, i64, f32]),
        "nd4j_threshold_encode_f32": (i64, [vp, i64, f32, vp, i64]),
        "nd4j_threshold_decode_f32": (i32, [vp, vp, i64]),
        "nd4j_bitmap_encode_f32": (i64, [vp, i64, f32, vp]),
        "nd4j_bitmap_decode_f32": (i32, [vp, vp, i64]),
        "nd4j_estimate_threshold_f32": (f32, [vp, i64, i64, u64]),
        # buffers / npy
        "nd4j_bswap_copy": (i32, [vp, vp, i64, i32]),
        "nd4j_npy_save": (i32, [cp, vp, P64, i32, cp, i32, i32]),
        "nd4j_npy_header": (i64, [cp, P64, PI, cp, PI]),
        "nd4j_npy_read_data": (i32, [cp, i64, vp, i64]),
        # rng
        "rng_create": (vp, [u64, u64]), "rng_destroy": (None, [vp]), "rng_set_states": (None, [vp, u64, u64]),
        "rng_root_state": (u64, [vp]), "rng_node_state": (u64, [vp]), "rng_offset": (u64, [vp]),
        "rng_rewind": (None, [vp, u64]), "rng_relative_int": (C.c_uint32, [vp, u64]),
        "rng_relative_long": (u64, [vp, u64]), "rng_relative_float": (f32, [vp, u64]),
        "rng_relative_double": (f64, [vp, u64]),
        "rng_fill_uniform_f32": (None, [vp, i64, u64, u64, u64, f32, f32]),
        "rng_fill_gaussian_f32": (None, [vp, i64, u64, u64, u64, f32, f32]),
        "rng_fill_bernoulli_f32": (None, [vp, i64, u64, u64, u64, f32]),
        # environment / launch dims / op trace
        "launch_dims_get": (None, [cp, PI]), "launch_dims_set": (None, [cp, i32, i32, i32]),
        "launch_dims_count": (i32, []), "launch_dims_name": (cp, [i32]),
        "op_trace_toggle": (None, [i32]), "op_trace_enabled": (i32, []), "op_trace_record": (None, [cp, f64]),
        "op_trace_count": (i32, []), "op_trace_purge": (None, []), "op_trace_list": (i64, [cp, i64]),
        "env_set_max_primary_memory": (None, [i64]), "env_get_max_primary_memory": (i64, []),
        "env_set_max_special_memory": (None, [i64]), "env_get_max_special_memory": (i64, []),
        "env_set_max_device_memory": (None, [i64]), "env_get_max_device_memory": (i64, []),
        # op registry
        "ops_hash": (i64, [cp]), "ops_count": (i32, []), "ops_describe": (i32, [cp, PI, P64]),
        "ops_name_by_hash": (cp, [i64]), "ops_name_at": (cp, [i32]), "ops_kind_at": (cp, [i32]),
        "ops_group_at": (cp, [i32]), "ops_register": (i32, [cp, i32, i32, i32, i32, i32]), "ops_all_custom": (cp, []),
        # datavec io
        "dv_csv_parse": (vp, [cp, C.c_char, i32]), "dv_svmlight_parse": (vp, [cp, i32, i32]),
        "dv_rows": (i64, [vp]), "dv_cols": (i64, [vp]), "dv_data": (PF, [vp]), "dv_copy": (None, [vp, vp]),
        "dv_free": (None, [vp]),
    }
    for flag in ("debug", "verbose", "profiling", "leak_detect", "helpers_allowed", "precision_boost",
                 "check_input_change", "tad_threshold", "elementwise_threshold", "max_threads", "max_master_threads",
                 "default_float_dtype", "func_trace"):
        sig["env_set_" + flag] = (None, [i32])
        sig["env_get_" + flag] = (i32, [])
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def rt_check(code, what="native runtime"):
    """Non-zero status of a runtime entry point -> the matching exception class of nd4j/exceptions.py (all are RuntimeErrors)."""
    if code != 0:
        from ..nd4j.exceptions import raise_native
        raise_native(code, what)


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


DT_F32, DT_BF16, DT_F64, DT_I32, DT_I64 = 0, 1, 2, 3, 4
_DT_ALL = {torch.float32: 0, torch.bfloat16: 1, torch.float64: 2, torch.int32: 3, torch.int64: 4}


def dtype_code_all(t: torch.Tensor) -> int:
    """Type code of the generic (legacy-op / random) kernels; -1 if the dtype has no HIP instantiation."""
    return _DT_ALL.get(t.dtype, -1)


DT_F16 = 2     # in the fp32 / bf16 / fp16 code space of the NN kernels (batchnorm, conv, gemm); the generic kernels use _DT_ALL


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float16:
        return DT_F16
    return -1


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError("HIP kernel launch failed in %s (hipError %d)" % (what, code))
