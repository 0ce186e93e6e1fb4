"""Environment flags, launch-dims registry, op profiler.

Reference: sd::Environment (libnd4j system/Environment.h:41-163) mirrored to Java as
Nd4j.getEnvironment(); ND4JSystemProperties / ND4JEnvironmentVars (nd4j-common); LaunchDims registry
(execution/cuda/LaunchDims.cu) ; OpProfiler / ProfilerConfig (nd4j-api linalg/profiler/**).
State lives in the native runtime (rt_env.cpp) so C++ and Python see the same flags.
"""
from __future__ import annotations

import ctypes
import os
import time
from collections import defaultdict
from contextlib import contextmanager

import torch

from ..native import lib as L


class ND4JEnvironmentVars:
    BACKEND_PRIORITY_GPU = "BACKEND_PRIORITY_GPU"
    ND4J_CUDA_FORCE_SINGLE_GPU = "ND4J_CUDA_FORCE_SINGLE_GPU"
    ND4J_CUDA_MIN_BLOCK_SIZE = "ND4J_CUDA_MIN_BLOCK_SIZE"
    ND4J_CUDA_MAX_BLOCK_SIZE = "ND4J_CUDA_MAX_BLOCK_SIZE"
    ND4J_CUDA_MAX_DEVICE_CACHE = "ND4J_CUDA_MAX_DEVICE_CACHE"
    ND4J_CUDA_USE_PREALLOCATION = "ND4J_CUDA_USE_PREALLOCATION"
    OMP_NUM_THREADS = "OMP_NUM_THREADS"
    ND4J_RESOURCES_CACHE_DIR = "ND4J_RESOURCES_CACHE_DIR"
    ND4J_SKIP_BLAS_THREADS = "ND4J_SKIP_BLAS_THREADS"
    DL4J_AMD_CONV = "DL4J_AMD_CONV"
    DL4J_AMD_FORCE_EAGER = "DL4J_AMD_FORCE_EAGER"


class ND4JSystemProperties:
    DTYPE = "dtype"
    LOG_INITIALIZATION = "org.nd4j.log.initialization"
    ND4J_IGNORE_AVX = "org.nd4j.ignoreavx"
    VERSION_CHECK_PROPERTY = "org.nd4j.versioncheck"
    WORKSPACE_DEBUG = "org.nd4j.workspace.debug"
    RESOURCES_CACHE_DIR = "org.nd4j.test.resources.cache.dir"


_FLAGS = ("debug", "verbose", "profiling", "leak_detect", "helpers_allowed", "precision_boost", "check_input_change",
          "tad_threshold", "elementwise_threshold", "max_threads", "max_master_threads", "default_float_dtype", "func_trace")


class Environment:
    _inst = None

    def __init__(self):
        self._rt = L.rt()
        n = os.environ.get(ND4JEnvironmentVars.OMP_NUM_THREADS)
        if n and n.isdigit():
            self._rt.env_set_max_threads(int(n))

    @classmethod
    def getInstance(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def _get(self, name):
        return getattr(self._rt, "env_get_" + name)()

    def _set(self, name, v):
        getattr(self._rt, "env_set_" + name)(int(v))

    def isDebug(self): return bool(self._get("debug"))
    def setDebug(self, b): self._set("debug", b)
    def isVerbose(self): return bool(self._get("verbose"))
    def setVerbose(self, b): self._set("verbose", b)
    def isProfiling(self): return bool(self._get("profiling"))
    def setProfiling(self, b): self._set("profiling", b)
    def isDetectingLeaks(self): return bool(self._get("leak_detect"))
    def setLeaksDetector(self, b): self._set("leak_detect", b)
    def helpersAllowed(self): return bool(self._get("helpers_allowed"))
    def allowHelpers(self, b): self._set("helpers_allowed", b)
    def precisionBoostAllowed(self): return bool(self._get("precision_boost"))
    def allowPrecisionBoost(self, b): self._set("precision_boost", b)
    def isCheckInputChange(self): return bool(self._get("check_input_change"))
    def setCheckInputChange(self, b): self._set("check_input_change", b)
    def tadThreshold(self): return self._get("tad_threshold")
    def setTadThreshold(self, v): self._set("tad_threshold", v)
    def elementwiseThreshold(self): return self._get("elementwise_threshold")
    def setElementwiseThreshold(self, v): self._set("elementwise_threshold", v)
    def maxThreads(self): return self._get("max_threads")
    def setMaxThreads(self, v): self._set("max_threads", v)
    def maxMasterThreads(self): return self._get("max_master_threads")
    def setMaxMasterThreads(self, v): self._set("max_master_threads", v)
    def isFuncTracePrintAllocate(self): return bool(self._get("func_trace"))
    def setFuncTracePrintAllocate(self, b): self._set("func_trace", b)
    def defaultFloatDataType(self): return self._get("default_float_dtype")
    def setDefaultFloatDataType(self, v): self._set("default_float_dtype", v)

    def setMaxPrimaryMemory(self, v): self._rt.env_set_max_primary_memory(int(v))
    def maxPrimaryMemory(self): return self._rt.env_get_max_primary_memory()
    def setMaxSpecialMemory(self, v): self._rt.env_set_max_special_memory(int(v))
    def maxSpecialMemory(self): return self._rt.env_get_max_special_memory()
    def setMaxDeviceMemory(self, v): self._rt.env_set_max_device_memory(int(v))
    def maxDeviceMemory(self): return self._rt.env_get_max_device_memory()

    def isCPU(self):
        return not torch.cuda.is_available()

    def blasMajorVersion(self):
        return int((torch.version.hip or "0").split(".")[0])

    def getDeviceCounter(self, device=0):
        return self._rt.mem_counter_get(device, 1)

    # -- launch dims
    def launchDims(self, name):
        out = (ctypes.c_int * 3)()
        self._rt.launch_dims_get(name.encode(), out)
        return tuple(out)

    def setLaunchDims(self, name, grid, block, smem=0):
        """Registry entry + the override the HIP kernel families read before launching (native/csrc/launch_dims.hip): the
        pairwise / transform / reduce legacy kernels, BatchNorm and pooling cap their grid (and, where the kernel is written for a
        variable block, their block size) to these values."""
        self._rt.launch_dims_set(name.encode(), grid, block, smem)
        from ..native import lib as _L
        if _L.gpu_available():
            _L.hip().dl4j_launch_dims_set(name.encode(), int(grid), int(block), int(smem))

    def kernelLaunchDims(self, name):
        """(grid override, block override, workgroups of the family's most recent launch) as seen by the kernels, or None when the
        name is not one of this library's kernel families."""
        import ctypes
        from ..native import lib as _L
        out = (ctypes.c_int * 3)()
        if not _L.gpu_available() or _L.hip().dl4j_launch_dims_get(name.encode(), out) != 0:
            return None
        return tuple(out)

    def launchDimNames(self):
        return [self._rt.launch_dims_name(i).decode() for i in range(self._rt.launch_dims_count())]


class ProfilerConfig:
    def __init__(self, checkForNAN=False, checkForINF=False, stackTrace=False, checkElapsedTime=False, checkWorkspaces=False,
                 checkLocality=False, nativeStatistics=False, notOptimalArguments=False, notOptimalTAD=False):
        self.checkForNAN, self.checkForINF, self.stackTrace = checkForNAN, checkForINF, stackTrace
        self.checkElapsedTime, self.checkWorkspaces, self.checkLocality = checkElapsedTime, checkWorkspaces, checkLocality
        self.nativeStatistics, self.notOptimalArguments, self.notOptimalTAD = nativeStatistics, notOptimalArguments, notOptimalTAD


class OpProfiler:
    """Per-op invocation counts / wall time, NaN/Inf panic checks, stride penalties
    (reference: linalg/profiler/OpProfiler.java). Timing synchronises the stream only when
    checkElapsedTime is on, so the default path stays asynchronous."""
    _inst = None

    def __init__(self):
        self.config = ProfilerConfig()
        self.reset()

    @classmethod
    def getInstance(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def reset(self):
        self.invocations = defaultdict(int)
        self.timeNs = defaultdict(int)
        self.penalties = defaultdict(int)
        self.total = 0

    def setConfig(self, cfg):
        self.config = cfg
        L.rt().op_trace_toggle(int(cfg.nativeStatistics))

    @contextmanager
    def op(self, name, *tensors):
        cfg = self.config
        if cfg.notOptimalArguments:
            for t in tensors:
                if isinstance(t, torch.Tensor) and t.dim() > 1 and not t.is_contiguous():
                    self.penalties["STRIDED_ACCESS:" + name] += 1
        t0 = None
        if cfg.checkElapsedTime:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t0 = time.perf_counter_ns()
        yield
        self.invocations[name] += 1
        self.total += 1
        if t0 is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dt = time.perf_counter_ns() - t0
            self.timeNs[name] += dt
            L.rt().op_trace_record(name.encode(), dt / 1e6)

    def checkOutputs(self, name, *outs):
        cfg = self.config
        for o in outs:
            if isinstance(o, torch.Tensor) and o.is_floating_point() and o.numel():
                if cfg.checkForNAN and bool(torch.isnan(o).any()):
                    raise FloatingPointError("NaN in output of op " + name)
                if cfg.checkForINF and bool(torch.isinf(o).any()):
                    raise FloatingPointError("Inf in output of op " + name)

    def getInvocationsCount(self, name=None):
        return self.total if name is None else self.invocations[name]

    def printOutDashboard(self):
        rows = ["%-40s %10s %14s" % ("op", "calls", "total ms")]
        for k in sorted(self.invocations, key=lambda k: -self.timeNs[k]):
            rows.append("%-40s %10d %14.3f" % (k, self.invocations[k], self.timeNs[k] / 1e6))
        for k, v in self.penalties.items():
            rows.append("penalty %-32s %10d" % (k, v))
        s = "\n".join(rows)
        print(s)
        return s

    def nativeTraces(self):
        rt = L.rt()
        n = rt.op_trace_list(None, 0)
        buf = ctypes.create_string_buffer(int(n))
        rt.op_trace_list(buf, n)
        return [tuple(l.split("\t")) for l in buf.value.decode().splitlines()]
