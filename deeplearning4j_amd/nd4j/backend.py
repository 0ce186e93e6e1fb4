"""Backend SPI, device-local arrays, event logging / unified profiler, deallocation service, launch validation, array inspection.

Reference: Nd4jBackend / Nd4j.initWithBackend (nd4j-api linalg/factory/{Nd4jBackend,Nd4j}.java:5210-5345, backend
properties file role->class map), DeviceLocalNDArray (linalg/util/DeviceLocalNDArray.java:62-181),
EventLogger / UnifiedProfiler (linalg/profiler/**), Deallocatable / DeallocatorService (linalg/api/memory/deallocation),
DeviceValidator (libnd4j execution/cuda/DeviceValidator.cu), inspectArray (NativeOps.h:1582), DL4JSystemProperties
(resources org/deeplearning4j/config/DL4JSystemProperties.java).
"""
from __future__ import annotations

import ctypes
import os
import threading
import time
import weakref
from collections import deque
from typing import Callable, Dict, List, Optional

import torch

from ..native import lib as L


# ----------------------------------------------------------------------------- backend SPI
class Nd4jBackend:
    BACKEND_PRIORITY_CPU, BACKEND_PRIORITY_GPU = 0, 100
    _registry: List["Nd4jBackend"] = []
    _current: Optional["Nd4jBackend"] = None

    name = "abstract"

    def getPriority(self) -> int:
        raise NotImplementedError

    def isAvailable(self) -> bool:
        raise NotImplementedError

    def canRun(self) -> bool:
        return self.isAvailable()

    def allowsOrder(self) -> bool:
        return True

    def getEnvironment(self):
        from .environment import Environment
        return Environment.getInstance()

    def getProperties(self) -> Dict[str, str]:
        """role -> implementation (the backend .properties file of the reference)."""
        raise NotImplementedError

    def device(self) -> torch.device:
        raise NotImplementedError

    def logBackendInit(self):
        print("Loaded [%s] backend" % type(self).__name__)

    @classmethod
    def register(cls, backend: "Nd4jBackend"):
        cls._registry.append(backend)
        cls._current = None

    @classmethod
    def load(cls) -> "Nd4jBackend":
        """Highest-priority backend that can run (ServiceLoader + priority sort in the reference)."""
        if cls._current is None:
            cands = [b for b in cls._registry if b.canRun()]
            if not cands:
                raise RuntimeError("No Nd4jBackend can run here")
            cls._current = sorted(cands, key=lambda b: -b.getPriority())[0]
            if os.environ.get("ND4J_LOG_INITIALIZATION", "false").lower() == "true":
                cls._current.logBackendInit()
        return cls._current


class RocmBackend(Nd4jBackend):
    """MI355X backend: HIP kernels (libdl4j_hip.so) + ATen/rocBLAS/hipBLASLt/MIOpen for library ops, RCCL for collectives."""
    name = "rocm-gfx950"

    def getPriority(self):
        return int(os.environ.get("BACKEND_PRIORITY_GPU", self.BACKEND_PRIORITY_GPU))

    def isAvailable(self):
        return torch.cuda.is_available() and os.environ.get("ND4J_FORCE_CPU", "0") != "1"

    def device(self):
        return torch.device("cuda", torch.cuda.current_device())

    def getProperties(self):
        return {"real.class.double": "deeplearning4j_amd.nd4j.ndarray.INDArray", "opexec": "deeplearning4j_amd.nd4j.ops.custom.OpExecutioner",
                "legacy.opexec": "deeplearning4j_amd.nd4j.ops.legacy.LegacyExecutioner", "ndarrayfactory.class": "deeplearning4j_amd.nd4j.factory.Nd4j",
                "affinitymanager": "deeplearning4j_amd.nd4j.factory.AffinityManager", "memorymanager": "torch.cuda caching allocator (HIP)",
                "workspacemanager": "deeplearning4j_amd.nd4j.workspace.Nd4jWorkspaceManager", "rng": "deeplearning4j_amd.nd4j.rng.Random (Philox4x32-10)",
                "blas.ops": "hipBLASLt / rocBLAS via ATen", "collectives": "RCCL (torch.distributed backend nccl)",
                "native.kernels": "deeplearning4j_amd/native/libdl4j_hip.so", "native.runtime": "deeplearning4j_amd/native/libnd4jrocm_rt.so",
                "dtype": "float", "alloc": "javacpp-free: ctypes C ABI", "ndarray.order": "c"}


class CpuBackend(Nd4jBackend):
    """Host fallback (tests, gradient checks in fp64, gloo collectives)."""
    name = "cpu"

    def getPriority(self):
        return int(os.environ.get("BACKEND_PRIORITY_CPU", self.BACKEND_PRIORITY_CPU))

    def isAvailable(self):
        return True

    def device(self):
        return torch.device("cpu")

    def getProperties(self):
        p = RocmBackend().getProperties()
        p.update({"memorymanager": "host allocator", "blas.ops": "ATen CPU BLAS", "collectives": "gloo", "native.kernels": "-"})
        return p


Nd4jBackend.register(RocmBackend())
Nd4jBackend.register(CpuBackend())


class DL4JSystemProperties:
    """DL4J-level configuration keys (environment-variable spelling in parentheses is what this runtime reads)."""
    DL4J_TEMP_DIR_PROPERTY = "org.deeplearning4j.tempdir"                      # (DL4J_TEMP_DIR)
    DL4J_RESOURCES_DIR_PROPERTY = "org.deeplearning4j.resources.directory"     # (DL4J_RESOURCES_DIR)
    DL4J_RESOURCES_BASE_URL_PROPERTY = "org.deeplearning4j.resources.baseurl"
    CRASH_DUMP_ENABLED_PROPERTY = "org.deeplearning4j.crash.reporting.enabled"  # (DL4J_CRASH_REPORTING_ENABLED)
    CRASH_DUMP_OUTPUT_DIRECTORY_PROPERTY = "org.deeplearning4j.crash.reporting.directory"  # (DL4J_CRASH_REPORTING_DIR)
    UI_SERVER_PORT_PROPERTY = "org.deeplearning4j.ui.port"                      # (DL4J_UI_PORT)
    DISABLE_HELPER_PROPERTY = "org.eclipse.deeplearning4j.helpers.disable"      # (SD_FORBID_HELPERS)
    HELPER_DISABLE_DEFAULT_VALUE = "true"           # as the reference (DL4JSystemProperties.java:115): HelperUtils objects are opt-in
    NLP_UIMA_TOKENIZER_CACHE = "org.deeplearning4j.nlp.uima.tokenizer.cache"
    _ENV = {DL4J_TEMP_DIR_PROPERTY: "DL4J_TEMP_DIR", DL4J_RESOURCES_DIR_PROPERTY: "DL4J_RESOURCES_DIR",
            CRASH_DUMP_ENABLED_PROPERTY: "DL4J_CRASH_REPORTING_ENABLED", CRASH_DUMP_OUTPUT_DIRECTORY_PROPERTY: "DL4J_CRASH_REPORTING_DIR",
            UI_SERVER_PORT_PROPERTY: "DL4J_UI_PORT", DISABLE_HELPER_PROPERTY: "SD_FORBID_HELPERS"}

    @staticmethod
    def get(prop, default=None):
        return os.environ.get(DL4JSystemProperties._ENV.get(prop, prop), default)


# ----------------------------------------------------------------------------- device-local arrays
class DeviceLocal:
    """One value per device (linalg/util/DeviceLocal.java)."""

    def __init__(self, delayedMode=True):
        self.delayedMode = delayedMode
        self._vals: Dict[int, object] = {}
        self._lock = threading.RLock()

    @staticmethod
    def _n():
        return max(torch.cuda.device_count(), 1) if torch.cuda.is_available() else 1

    @staticmethod
    def _cur():
        return torch.cuda.current_device() if torch.cuda.is_available() else 0

    def get(self, deviceId=None):
        with self._lock:
            return self._vals.get(self._cur() if deviceId is None else deviceId)

    def set(self, deviceId_or_value, value=None):
        with self._lock:
            if value is None:
                self._vals[self._cur()] = deviceId_or_value
            else:
                self._vals[int(deviceId_or_value)] = value

    def clear(self):
        with self._lock:
            self._vals.clear()


class DeviceLocalNDArray(DeviceLocal):
    """Per-device replicas of an array. ``broadcast`` pushes device-to-device over xGMI; in delayed mode a replica is
    materialised on the first ``get()`` from that device."""

    def __init__(self, array=None, delayedMode=True):
        super().__init__(delayedMode)
        self._src = None
        self._stale = set()
        if array is not None:
            self.broadcast(array)

    @staticmethod
    def _t(a):
        return getattr(a, "_t", a)

    def _wrap(self, t):
        from .ndarray import INDArray
        return INDArray(t)

    def _replicate(self, dev):
        src = self._src
        if not torch.cuda.is_available() or src.device == torch.device("cuda", dev):
            return src.clone()          # own copy: update() writes replicas in place and must not touch the caller's array
        return src.to(torch.device("cuda", dev), non_blocking=True)      # peer-to-peer copy when the source is on another GPU

    def broadcast(self, array):
        with self._lock:
            self._src = self._t(array).detach()
            self._vals.clear()
            devs = range(self._n())
            if self.delayedMode:
                self._stale = set(devs)
            else:
                for d in devs:
                    self._vals[d] = self._replicate(d)
                self._stale = set()

    def update(self, array):
        """Replace the value everywhere (in place where a replica already exists)."""
        with self._lock:
            new = self._t(array).detach()
            self._src = new
            for d, t in list(self._vals.items()):
                if t is not new and t.shape == new.shape:
                    t.copy_(new, non_blocking=True)
                else:
                    self._vals[d] = self._replicate(d)
            self._stale = set(range(self._n())) - set(self._vals)

    def get(self, deviceId=None):
        with self._lock:
            d = self._cur() if deviceId is None else int(deviceId)
            if d not in self._vals:
                if self._src is None:
                    return None
                self._vals[d] = self._replicate(d)
                self._stale.discard(d)
            return self._wrap(self._vals[d])


# ----------------------------------------------------------------------------- event logger / profiler / deallocation
class EventType:
    ALLOCATION, DEALLOCATION = "ALLOCATION", "DEALLOCATION"


class ObjectAllocationType:
    OP_CONTEXT, DATA_BUFFER, WORKSPACE = "OP_CONTEXT", "DATA_BUFFER", "WORKSPACE"


class LogEvent:
    __slots__ = ("eventType", "objectAllocationType", "associatedWorkspace", "threadName", "eventTimeMs", "bytes", "dataType", "objectId")

    def __init__(self, eventType, objectAllocationType, nbytes=0, workspace=None, dataType=None, objectId=None):
        self.eventType, self.objectAllocationType, self.bytes = eventType, objectAllocationType, int(nbytes)
        self.associatedWorkspace, self.dataType, self.objectId = workspace, dataType, objectId
        self.threadName = threading.current_thread().name
        self.eventTimeMs = int(time.time() * 1000)

    def __repr__(self):
        return "%d,%s,%s,%s,%s,%d" % (self.eventTimeMs, self.eventType, self.objectAllocationType, self.associatedWorkspace, self.threadName, self.bytes)


class EventLogger:
    """Allocation / deallocation event stream (off unless enabled: ND4J_EVENT_LOG_ENABLED=true or setEnabled)."""
    _inst = None

    def __init__(self):
        self.enabled = os.environ.get("ND4J_EVENT_LOG_ENABLED", "false").lower() == "true"
        self.listeners: List[Callable[[LogEvent], None]] = []
        self.events = deque(maxlen=int(os.environ.get("ND4J_EVENT_LOG_CAPACITY", "100000")))
        self._lock = threading.Lock()
        self.formatTimeAsDate = False

    @classmethod
    def getInstance(cls):
        if cls._inst is None:
            cls._inst = EventLogger()
        return cls._inst

    def isEnabled(self):
        return self.enabled

    def setEnabled(self, b):
        self.enabled = bool(b)

    def addEventLogListener(self, fn):
        self.listeners.append(fn)

    def removeEventLogListener(self, fn):
        if fn in self.listeners:
            self.listeners.remove(fn)

    def log(self, event: LogEvent):
        if not self.enabled:
            return
        with self._lock:
            self.events.append(event)
        for l in list(self.listeners):
            l(event)

    def logAllocation(self, kind, nbytes, workspace=None, dataType=None, objectId=None):
        self.log(LogEvent(EventType.ALLOCATION, kind, nbytes, workspace, dataType, objectId))

    def logDeallocation(self, kind, nbytes, workspace=None, dataType=None, objectId=None):
        self.log(LogEvent(EventType.DEALLOCATION, kind, nbytes, workspace, dataType, objectId))

    def aggregate(self):
        out: Dict[tuple, List[int]] = {}
        for e in list(self.events):
            a = out.setdefault((e.eventType, e.objectAllocationType), [0, 0])
            a[0] += 1
            a[1] += e.bytes
        return out


class DeallocatorService:
    """Frees native resources when their owner is garbage collected (ReferenceQueue threads in the reference; weakref
    finalizers here) and reports to the EventLogger. ``Deallocatable`` = any object + a zero-argument release callable."""
    _inst = None

    def __init__(self):
        self._n = 0
        self._live: Dict[int, weakref.finalize] = {}
        self._lock = threading.Lock()
        self.blockDeallocator = False            # DeallocationExtension: hold deallocations during a test

    @classmethod
    def getInstance(cls):
        if cls._inst is None:
            cls._inst = DeallocatorService()
        return cls._inst

    def nextValue(self):
        with self._lock:
            self._n += 1
            return self._n

    def pickObject(self, owner, release: Callable[[], None], nbytes=0, kind=ObjectAllocationType.DATA_BUFFER):
        uid = self.nextValue()
        EventLogger.getInstance().logAllocation(kind, nbytes, objectId=uid)

        def fin():
            if self.blockDeallocator:
                self._pending.append((release, kind, nbytes, uid))
                return
            self._run(release, kind, nbytes, uid)
        f = weakref.finalize(owner, fin)
        with self._lock:
            self._live[uid] = f
        return uid

    _pending: List[tuple] = []

    def _run(self, release, kind, nbytes, uid):
        try:
            release()
        finally:
            EventLogger.getInstance().logDeallocation(kind, nbytes, objectId=uid)
            with self._lock:
                self._live.pop(uid, None)

    def toggleDeallocationBlock(self, block):
        self.blockDeallocator = bool(block)
        if not block:
            pend, DeallocatorService._pending = DeallocatorService._pending, []
            for release, kind, nbytes, uid in pend:
                self._run(release, kind, nbytes, uid)

    def numLive(self):
        with self._lock:
            return sum(1 for f in self._live.values() if f.alive)


class UnifiedProfiler:
    """One switch for op statistics (OpProfiler), the allocation event stream (EventLogger), memory snapshots and — on the
    GPU — HIP kernel timelines through torch.profiler (what rocprofv3 shows, inside the process)."""
    _inst = None

    def __init__(self):
        self.running = False
        self.snapshots: List[dict] = []
        self._tp = None

    @classmethod
    def getInstance(cls):
        if cls._inst is None:
            cls._inst = UnifiedProfiler()
        return cls._inst

    def start(self, kernels=False):
        from .environment import OpProfiler, ProfilerConfig
        OpProfiler.getInstance().reset()
        OpProfiler.getInstance().setConfig(ProfilerConfig(checkElapsedTime=True, nativeStatistics=True))
        EventLogger.getInstance().setEnabled(True)
        self.running = True
        self.snapshot("start")
        if kernels and torch.cuda.is_available():
            from torch.profiler import ProfilerActivity, profile
            self._tp = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
            self._tp.__enter__()

    def snapshot(self, tag=""):
        s = {"tag": tag, "timeMs": int(time.time() * 1000)}
        if torch.cuda.is_available():
            s.update(deviceAllocated=torch.cuda.memory_allocated(), deviceReserved=torch.cuda.memory_reserved(),
                     devicePeak=torch.cuda.max_memory_allocated())
        try:
            import psutil
            s["hostRss"] = psutil.Process().memory_info().rss
        except Exception:
            pass
        self.snapshots.append(s)
        return s

    def stop(self):
        from .environment import OpProfiler, ProfilerConfig
        self.snapshot("stop")
        EventLogger.getInstance().setEnabled(False)
        OpProfiler.getInstance().setConfig(ProfilerConfig())
        self.running = False
        kernels = None
        if self._tp is not None:
            self._tp.__exit__(None, None, None)
            kernels = self._tp.key_averages().table(sort_by="cuda_time_total", row_limit=25)
            self._tp = None
        return {"ops": OpProfiler.getInstance().getInvocationsCount(), "events": EventLogger.getInstance().aggregate(),
                "memory": list(self.snapshots), "kernels": kernels}


# ----------------------------------------------------------------------------- launch validation / array inspection
_declared = False


def _rt():
    global _declared
    lib = L.rt()
    if not _declared:
        P64 = ctypes.POINTER(ctypes.c_int64)
        lib.device_validate_launch.restype = ctypes.c_int
        lib.device_validate_launch.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int64, ctypes.c_int, P64]
        lib.device_limits.restype = None
        lib.device_limits.argtypes = [P64]
        _declared = True
    return lib


class DeviceValidator:
    BLOCK, LDS, GRID, VGPR, PARTIAL_WAVE = 1, 2, 4, 8, 16

    @staticmethod
    def limits():
        out = (ctypes.c_int64 * 8)()
        _rt().device_limits(out)
        return dict(zip(("waveSize", "maxThreadsPerBlock", "ldsPerCu", "vgprsPerLane", "simdsPerCu", "maxWavesPerSimd", "numCu", "numXcd"), out))

    @staticmethod
    def validate(grid, block, smemBytes=0, vgprsPerThread=0):
        g = tuple(grid) + (1,) * (3 - len(tuple(grid))) if not isinstance(grid, int) else (grid, 1, 1)
        b = tuple(block) + (1,) * (3 - len(tuple(block))) if not isinstance(block, int) else (block, 1, 1)
        out = (ctypes.c_int64 * 6)()
        mask = _rt().device_validate_launch(int(g[0]), int(g[1]), int(g[2]), int(b[0]), int(b[1]), int(b[2]), int(smemBytes),
                                            int(vgprsPerThread), out)
        return {"valid": (mask & ~DeviceValidator.PARTIAL_WAVE) == 0, "mask": mask, "wavesPerBlock": out[0], "blocksPerCu": out[1],
                "wavesPerSimd": out[2], "occupancyPercent": out[3], "blocksToFillChip": out[4], "ldsLimited": bool(out[5])}


class DebugInfo:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __repr__(self):
        return "DebugInfo(%s)" % ", ".join("%s=%s" % kv for kv in self.__dict__.items())


def inspectArray(arr) -> DebugInfo:
    """Native-style array statistics in ONE device pass + one small transfer (NativeOps inspectArray / DebugHelper::retrieveDebugStatistics)."""
    t = getattr(arr, "_t", arr).detach()
    if t.numel() == 0:
        return DebugInfo(min=0.0, max=0.0, mean=0.0, stdDev=0.0, zeroCount=0, positiveCount=0, negativeCount=0, nanCount=0, infCount=0)
    f = t.double() if not t.is_complex() else t.abs().double()
    finite = torch.isfinite(f)
    clean = torch.where(finite, f, torch.zeros_like(f))
    n_fin = finite.sum().clamp_min(1)
    mean = clean.sum() / n_fin
    var = (torch.where(finite, (f - mean) ** 2, torch.zeros_like(f))).sum() / n_fin
    big = torch.finfo(torch.float64).max
    stats = torch.stack([torch.where(finite, f, torch.full_like(f, big)).min(), torch.where(finite, f, torch.full_like(f, -big)).max(),
                         mean, var.sqrt(), (f == 0).sum().double(), (f > 0).sum().double(), (f < 0).sum().double(),
                         torch.isnan(f).sum().double(), torch.isinf(f).sum().double()]).cpu().tolist()
    return DebugInfo(min=stats[0], max=stats[1], mean=stats[2], stdDev=stats[3], zeroCount=int(stats[4]), positiveCount=int(stats[5]),
                     negativeCount=int(stats[6]), nanCount=int(stats[7]), infCount=int(stats[8]))
