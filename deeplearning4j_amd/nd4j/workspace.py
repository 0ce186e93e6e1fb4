"""Scoped arena workspaces (reference: nd4j-api linalg/api/memory/** — MemoryWorkspace,
WorkspaceConfiguration, Nd4jWorkspace.alloc :348-459 / initializeWorkspace :466-525,
MemoryWorkspaceManager; CUDA impl nd4j-cuda CudaWorkspace.java; ADR 0024).

MI355X design: the arena bookkeeping (bump pointer, learning policy, spill accounting, cycles,
generations) is native (rt_workspace.cpp); the arena *memory* is a single uint8 slab taken from the
torch caching allocator (device) or pinned host memory, so workspaces and ordinary tensors share one
allocator and 288 GB of HBM is never fragmented between two pools. ``alloc`` returns a typed view
of the slab; spilled requests become ordinary allocations that die at scope exit.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Dict, List, Optional

import torch

from ..native import lib as L


class AllocationPolicy:
    STRICT, OVERALLOCATE = "STRICT", "OVERALLOCATE"


class LearningPolicy:
    NONE, FIRST_LOOP, OVER_TIME = "NONE", "FIRST_LOOP", "OVER_TIME"


class SpillPolicy:
    EXTERNAL, REALLOCATE, FAIL = "EXTERNAL", "REALLOCATE", "FAIL"


class ResetPolicy:
    BLOCK_LEFT, ENDOFBUFFER_REACHED = "BLOCK_LEFT", "ENDOFBUFFER_REACHED"


class MirroringPolicy:
    FULL, HOST_ONLY = "FULL", "HOST_ONLY"


class LocationPolicy:
    RAM, MMAP, DEVICE_POOL = "RAM", "MMAP", "DEVICE_POOL"        # DEVICE_POOL: private pool of the HIP caching allocator


class MemoryKind:
    HOST, DEVICE = 0, 1


class WorkspaceConfiguration:
    def __init__(self, initialSize=0, maxSize=0, minSize=0, overallocationLimit=0.0, cyclesBeforeInitialization=0,
                 policyAllocation=AllocationPolicy.OVERALLOCATE, policyLearning=LearningPolicy.FIRST_LOOP,
                 policySpill=SpillPolicy.EXTERNAL, policyReset=ResetPolicy.BLOCK_LEFT, policyMirroring=MirroringPolicy.FULL,
                 policyLocation=LocationPolicy.RAM, tempFilePath=None):
        self.initialSize, self.maxSize, self.minSize = int(initialSize), int(maxSize), int(minSize)
        self.overallocationLimit = float(overallocationLimit)
        self.cyclesBeforeInitialization = int(cyclesBeforeInitialization)
        self.policyAllocation, self.policyLearning, self.policySpill = policyAllocation, policyLearning, policySpill
        self.policyReset, self.policyMirroring, self.policyLocation = policyReset, policyMirroring, policyLocation
        self.tempFilePath = tempFilePath

    class Builder:
        def __init__(self):
            self._kw = {}

        def __getattr__(self, name):
            def setter(v):
                self._kw[name] = v
                return self
            return setter

        def build(self):
            return WorkspaceConfiguration(**self._kw)

    @staticmethod
    def builder():
        return WorkspaceConfiguration.Builder()


_LEARN = {LearningPolicy.NONE: 0, LearningPolicy.FIRST_LOOP: 1, LearningPolicy.OVER_TIME: 2}
_TLS = threading.local()


def _stack() -> List["MemoryWorkspace"]:
    if not hasattr(_TLS, "stack"):
        _TLS.stack = []
    return _TLS.stack


class MemoryWorkspace:
    def __init__(self, conf: WorkspaceConfiguration, wsId: str, device=None):
        self.conf, self.id = conf, wsId
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self._rt = L.rt()
        over = conf.overallocationLimit if conf.policyAllocation == AllocationPolicy.OVERALLOCATE else 0.0
        self._h = self._rt.ws_create(_LEARN[conf.policyLearning], over, conf.cyclesBeforeInitialization)
        self._slab: Dict[int, Optional[torch.Tensor]] = {0: None, 1: None}
        self._spilled: List[torch.Tensor] = []
        self._open = 0
        self._borrowed = False
        if conf.initialSize > 0:
            self._bind(MemoryKind.DEVICE if self.device.type == "cuda" else MemoryKind.HOST, conf.initialSize)

    def __del__(self):
        try:
            self._rt.ws_destroy(self._h)
        except Exception:
            pass

    # -- arena management
    def _bind(self, kind, nbytes):
        if self.conf.maxSize > 0:
            nbytes = min(nbytes, self.conf.maxSize)
        nbytes = max(nbytes, self.conf.minSize)
        nbytes = (nbytes + 255) // 256 * 256
        if kind == MemoryKind.DEVICE:
            raw = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        else:
            raw = torch.empty(nbytes + 256, dtype=torch.uint8)
            if torch.cuda.is_available():
                raw = raw.pin_memory()
        skew = (-raw.data_ptr()) % 256              # 256-byte aligned arena base (host malloc gives 64)
        slab = raw[skew:skew + nbytes]
        self._slab[kind] = slab
        L.rt_check(self._rt.ws_bind(self._h, kind, ctypes.c_uint64(slab.data_ptr()), nbytes), "ws_bind")

    def _stats(self, kind):
        out = (ctypes.c_int64 * 9)()
        self._rt.ws_stats(self._h, kind, out)
        keys = ("capacity", "offset", "peakCycle", "peakAll", "spilledCycle", "spilledTotal", "cycles", "generation", "depth")
        return dict(zip(keys, out))

    def _kind(self, device=None):
        d = self.device if device is None else torch.device(device)
        return MemoryKind.DEVICE if d.type == "cuda" else MemoryKind.HOST

    # -- scope protocol
    def notifyScopeEntered(self):
        self._rt.ws_scope_in(self._h)
        self._open += 1
        _stack().append(self)
        return self

    def notifyScopeLeft(self):
        wh, wd = ctypes.c_int64(0), ctypes.c_int64(0)
        self._rt.ws_scope_out(self._h, ctypes.byref(wh), ctypes.byref(wd))
        self._open -= 1
        st = _stack()
        if st and st[-1] is self:
            st.pop()
        if self._open == 0:
            self._spilled.clear()       # spilled allocations are purged at BLOCK_LEFT
            if wh.value > 0:
                self._bind(MemoryKind.HOST, wh.value)
            if wd.value > 0:
                self._bind(MemoryKind.DEVICE, wd.value)
        return self

    def notifyScopeBorrowed(self):
        self._borrowed = True
        _stack().append(self)
        return self

    __enter__ = notifyScopeEntered

    def __exit__(self, *exc):
        self.notifyScopeLeft()
        return False

    close = notifyScopeLeft

    def isScopeActive(self):
        return self._open > 0

    def tagOutOfScopeUse(self):
        return _ScopeOut()

    # -- allocation
    def alloc(self, shape, dtype=torch.float32, initialize=False, device=None) -> torch.Tensor:
        kind = self._kind(device)
        n = 1
        for s in shape:
            n *= int(s)
        es = torch.empty(0, dtype=dtype).element_size()
        nbytes = max(n * es, 1)
        if self.conf.policyReset == ResetPolicy.ENDOFBUFFER_REACHED and self._slab[kind] is not None:
            # cyclic workspace: wrap around to the start of the arena (AsyncDataSetIterator's buffer mode)
            st = self._stats(kind)
            need = (nbytes + 255) // 256 * 256
            if st["offset"] + need > st["capacity"] and need <= st["capacity"]:
                self._rt.ws_rewind(self._h, kind)
        addr = self._rt.ws_alloc(self._h, kind, nbytes)
        if addr == 0:
            if self.conf.policySpill == SpillPolicy.FAIL:
                raise MemoryError("workspace '%s': %d bytes do not fit (capacity %d) and spill policy is FAIL" % (
                    self.id, nbytes, self._stats(kind)["capacity"]))
            dev = self.device if kind == MemoryKind.DEVICE else torch.device("cpu")
            t = torch.zeros(shape, dtype=dtype, device=dev) if initialize else torch.empty(shape, dtype=dtype, device=dev)
            self._spilled.append(t)
            return t
        slab = self._slab[kind]
        off = addr - slab.data_ptr()
        t = slab[off:off + n * es].view(dtype).reshape(list(shape))
        if initialize:
            t.zero_()
        return t

    def create(self, *shape, dtype=torch.float32, order="c"):
        from .ndarray import INDArray
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (list, tuple)) else shape
        a = INDArray(self.alloc(shape, dtype, True), order)
        a._ws = self.id
        return a

    # -- introspection (MemoryWorkspace getters)
    def getId(self):
        return self.id

    def getCurrentSize(self, kind=None):
        return self._stats(self._kind() if kind is None else kind)["capacity"]

    def getCurrentOffset(self, kind=None):
        return self._stats(self._kind() if kind is None else kind)["offset"]

    def getSpilledSize(self, kind=None):
        return self._stats(self._kind() if kind is None else kind)["spilledTotal"]

    def getMaxCycleAllocations(self, kind=None):
        return self._stats(self._kind() if kind is None else kind)["peakAll"]

    def getThisCycleAllocations(self, kind=None):
        return self._stats(self._kind() if kind is None else kind)["peakCycle"]

    def getGenerationId(self):
        return self._stats(0)["generation"]

    def getCyclesCount(self):
        return self._stats(0)["cycles"]

    def destroyWorkspace(self, extended=True):
        self._slab = {0: None, 1: None}
        self._spilled.clear()
        self._rt.ws_bind(self._h, 0, ctypes.c_uint64(0), 0)
        self._rt.ws_bind(self._h, 1, ctypes.c_uint64(0), 0)

    def getWorkspaceConfiguration(self):
        return self.conf


class _ScopeOut:
    """scopeOutOfWorkspaces(): temporarily detach the thread from every workspace."""

    def __enter__(self):
        self._saved = list(_stack())
        _stack().clear()
        return self

    def __exit__(self, *exc):
        _stack().extend(self._saved)
        return False


class DevicePoolWorkspace(MemoryWorkspace):
    """Device workspace = a PRIVATE POOL of the HIP caching allocator (torch.cuda.MemPool). While its scope is open every device
    allocation of the thread — the outputs the op wrappers and HIP launchers allocate themselves, not only explicit ``alloc`` calls
    — is served from the pool; blocks released inside the scope go back to the pool, never to the global allocator, so the next
    cycle reuses exactly the same memory. This is the reference's cyclic workspace (one arena per purpose, reused every iteration,
    nothing freed between iterations: ADR 0024, MultiLayerNetwork.java:1688-1705) expressed with the allocator the kernels already
    use; the 'learning' of the arena size is the pool keeping its high-water mark. Arrays that must outlive the scope simply stay
    referenced: their blocks remain owned by the pool until released (the 'leverage / detach' step needs no copy)."""

    def __init__(self, conf: WorkspaceConfiguration, wsId: str, device=None):
        super().__init__(conf, wsId, device)
        self._pool = None
        self._ctx = []
        self._cycles = 0

    @staticmethod
    def supported(device) -> bool:
        return (torch.device(device).type == "cuda" and hasattr(torch.cuda, "MemPool") and hasattr(torch.cuda, "use_mem_pool")
                and not torch.cuda.is_current_stream_capturing())

    def notifyScopeEntered(self):
        if self._pool is None:
            self._pool = torch.cuda.MemPool()
        dev = self.device if self.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        cm = torch.cuda.use_mem_pool(self._pool, device=dev)
        cm.__enter__()
        self._ctx.append(cm)
        self._open += 1
        _stack().append(self)
        return self

    def notifyScopeLeft(self):
        cm = self._ctx.pop()
        cm.__exit__(None, None, None)
        self._open -= 1
        st = _stack()
        if st and st[-1] is self:
            st.pop()
        if self._open == 0:
            self._cycles += 1
        return self

    __enter__ = notifyScopeEntered
    close = notifyScopeLeft

    def alloc(self, shape, dtype=torch.float32, initialize=False, device=None) -> torch.Tensor:
        if not self.isScopeActive():
            raise RuntimeError("workspace '%s' is not open" % self.id)
        return (torch.zeros if initialize else torch.empty)(tuple(shape), dtype=dtype, device=self.device)

    def _segments(self):
        return self._pool.snapshot() if self._pool is not None else []

    def getCurrentSize(self, kind=None):
        """Bytes reserved by the pool (its high-water mark: nothing is returned between cycles)."""
        return int(sum(seg["total_size"] for seg in self._segments()))

    def getCurrentOffset(self, kind=None):
        """Bytes currently handed out."""
        return int(sum(seg["allocated_size"] for seg in self._segments()))

    def getCyclesCount(self):
        return self._cycles

    def destroyWorkspace(self, extended=True):
        self._pool = None


class DummyWorkspace(MemoryWorkspace):
    def __init__(self):
        super().__init__(WorkspaceConfiguration(policyLearning=LearningPolicy.NONE), "DUMMY")

    def alloc(self, shape, dtype=torch.float32, initialize=False, device=None):
        dev = self.device if device is None else device
        return torch.zeros(shape, dtype=dtype, device=dev) if initialize else torch.empty(shape, dtype=dtype, device=dev)


class MemoryWorkspaceManager:
    """Thread-local named workspaces (reference: BasicWorkspaceManager / CudaWorkspaceManager)."""

    def __init__(self):
        self._default = WorkspaceConfiguration(initialSize=0, overallocationLimit=0.3, policyLearning=LearningPolicy.FIRST_LOOP)

    def _map(self) -> Dict[str, MemoryWorkspace]:
        if not hasattr(_TLS, "ws"):
            _TLS.ws = {}
        return _TLS.ws

    def setDefaultWorkspaceConfiguration(self, conf):
        self._default = conf

    @staticmethod
    def _make(conf, wsId, device):
        if conf.policyLocation == LocationPolicy.DEVICE_POOL and device is not None and DevicePoolWorkspace.supported(device):
            return DevicePoolWorkspace(conf, wsId, device)
        return MemoryWorkspace(conf, wsId, device)

    def createNewWorkspace(self, conf=None, wsId=None, device=None):
        ws = self._make(conf or self._default, wsId or "WS_%d" % len(self._map()), device)
        self._map()[ws.id] = ws
        return ws

    def getWorkspaceForCurrentThread(self, conf=None, wsId="DEFAULT", device=None):
        m = self._map()
        if wsId not in m:
            m[wsId] = self._make(conf or self._default, wsId, device)
        return m[wsId]

    def getAndActivateWorkspace(self, conf=None, wsId="DEFAULT", device=None):
        return self.getWorkspaceForCurrentThread(conf, wsId, device).notifyScopeEntered()

    def checkIfWorkspaceExists(self, wsId):
        return wsId in self._map()

    def checkIfWorkspaceExistsAndActive(self, wsId):
        return wsId in self._map() and self._map()[wsId].isScopeActive()

    def getCurrentWorkspace(self):
        st = _stack()
        return st[-1] if st else None

    def anyWorkspaceActiveForCurrentThread(self):
        return bool(_stack())

    def getAllWorkspacesIdsForCurrentThread(self):
        return list(self._map().keys())

    def getAllWorkspacesForCurrentThread(self):
        return list(self._map().values())

    def destroyWorkspace(self, ws=None):
        ws = ws or self.getCurrentWorkspace()
        if ws is not None:
            ws.destroyWorkspace()
            self._map().pop(ws.id, None)

    def destroyAllWorkspacesForCurrentThread(self):
        for ws in list(self._map().values()):
            ws.destroyWorkspace()
        self._map().clear()
        _stack().clear()

    def scopeOutOfWorkspaces(self):
        return _ScopeOut()

    def printAllocationStatisticsForCurrentThread(self):
        lines = []
        for ws in self._map().values():
            for kind, nm in ((0, "host"), (1, "device")):
                s = ws._stats(kind)
                lines.append("%-28s %-6s size=%d peak=%d spilled=%d cycles=%d" % (ws.id, nm, s["capacity"], s["peakAll"],
                                                                                   s["spilledTotal"], s["cycles"]))
        out = "\n".join(lines)
        print(out)
        return out


_MGR = MemoryWorkspaceManager()


def getWorkspaceManager() -> MemoryWorkspaceManager:
    return _MGR


# ----------------------------------------------------------------------------- DL4J LayerWorkspaceMgr
class ArrayType:
    """What an array is used for inside a network (reference: deeplearning4j-nn nn/workspace/ArrayType.java)."""
    INPUT, ACTIVATIONS, ACTIVATION_GRAD, FF_WORKING_MEM, BP_WORKING_MEM, RNN_FF_LOOP_WORKING_MEM, RNN_BP_LOOP_WORKING_MEM, \
        UPDATER_WORKING_MEM, FF_CACHE = ("INPUT", "ACTIVATIONS", "ACTIVATION_GRAD", "FF_WORKING_MEM", "BP_WORKING_MEM",
                                         "RNN_FF_LOOP_WORKING_MEM", "RNN_BP_LOOP_WORKING_MEM", "UPDATER_WORKING_MEM", "FF_CACHE")
    ALL = [INPUT, ACTIVATIONS, ACTIVATION_GRAD, FF_WORKING_MEM, BP_WORKING_MEM, RNN_FF_LOOP_WORKING_MEM, RNN_BP_LOOP_WORKING_MEM,
           UPDATER_WORKING_MEM, FF_CACHE]


class _EnteredScope:
    """`try (MemoryWorkspace ws = mgr.notifyScopeEntered(type))` in Python: the scope is already entered, leaving the
    ``with`` block closes it exactly once."""

    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        return self.ws

    def __exit__(self, *exc):
        self.ws.notifyScopeLeft()
        return False

    def close(self):
        self.ws.notifyScopeLeft()

    def __getattr__(self, k):
        return getattr(self.ws, k)


class LayerWorkspaceMgr:
    """Maps ArrayType -> named workspace (reference: nn/workspace/LayerWorkspaceMgr.java; names WS_ALL_LAYERS_ACT,
    WS_LAYER_WORKING_MEM, WS_RNN_LOOP_WORKING_MEM set up at MultiLayerNetwork.java:1688-1705).

    ``forNetwork`` is what MultiLayerNetwork / ComputationGraph enter around every eager iteration and inference call on the GPU:
    activations, gradients and scratch come from private allocator pools (DevicePoolWorkspace) that are reused cycle after cycle.
    Captured (HIP-graph) steps already live in the graph's own static pool and skip the manager."""
    WS_ALL_LAYERS_ACT, WS_LAYER_WORKING_MEM, WS_LAYER_ACT_1, WS_LAYER_ACT_2, WS_RNN_LOOP_WORKING_MEM, WS_OUTPUT_MEM = (
        "WS_ALL_LAYERS_ACT", "WS_LAYER_WORKING_MEM", "WS_LAYER_ACT_1", "WS_LAYER_ACT_2", "WS_RNN_LOOP_WORKING_MEM", "WS_OUTPUT_MEM")

    class Builder:
        def __init__(self):
            self._m = LayerWorkspaceMgr()

        def with_(self, arrayType, wsName, conf=None):
            self._m._names[arrayType] = wsName
            self._m._confs[wsName] = conf or WorkspaceConfiguration()
            self._m._scoped_out.discard(arrayType)
            return self

        def noWorkspaceFor(self, arrayType):
            self._m._scoped_out.add(arrayType)
            self._m._names.pop(arrayType, None)
            return self

        def defaultNoWorkspace(self):
            for t in ArrayType.ALL:
                if t not in self._m._names:
                    self._m._scoped_out.add(t)
            return self

        def defaultWorkspace(self, wsName, conf=None):
            for t in ArrayType.ALL:
                if t not in self._m._names and t not in self._m._scoped_out:
                    self.with_(t, wsName, conf)
            return self

        def build(self):
            return self._m

    def __init__(self):
        self._names: Dict[str, str] = {}
        self._confs: Dict[str, WorkspaceConfiguration] = {}
        self._scoped_out = set()
        self._helper_ws: Dict[str, object] = {}
        self.device = None

    @staticmethod
    def builder():
        return LayerWorkspaceMgr.Builder()

    @staticmethod
    def noWorkspaces():
        return LayerWorkspaceMgr.Builder().defaultNoWorkspace().build()

    @staticmethod
    def forNetwork(training: bool, device=None):
        """The manager MultiLayerNetwork.java:1688-1705 builds: activations (+ their gradients when training) in WS_ALL_LAYERS_ACT,
        per-layer / updater scratch in WS_LAYER_WORKING_MEM, recurrent loop scratch in WS_RNN_LOOP_WORKING_MEM; device pools."""
        c = WorkspaceConfiguration(policyLocation=LocationPolicy.DEVICE_POOL)
        b = LayerWorkspaceMgr.Builder()
        A = ArrayType
        b.with_(A.INPUT, LayerWorkspaceMgr.WS_ALL_LAYERS_ACT, c).with_(A.ACTIVATIONS, LayerWorkspaceMgr.WS_ALL_LAYERS_ACT, c)
        b.with_(A.FF_WORKING_MEM, LayerWorkspaceMgr.WS_LAYER_WORKING_MEM, c)
        b.with_(A.RNN_FF_LOOP_WORKING_MEM, LayerWorkspaceMgr.WS_RNN_LOOP_WORKING_MEM, c)
        if training:
            b.with_(A.ACTIVATION_GRAD, LayerWorkspaceMgr.WS_ALL_LAYERS_ACT, c)
            b.with_(A.BP_WORKING_MEM, LayerWorkspaceMgr.WS_LAYER_WORKING_MEM, c)
            b.with_(A.RNN_BP_LOOP_WORKING_MEM, LayerWorkspaceMgr.WS_RNN_LOOP_WORKING_MEM, c)
            b.with_(A.UPDATER_WORKING_MEM, LayerWorkspaceMgr.WS_LAYER_WORKING_MEM, c)
        m = b.defaultNoWorkspace().build()
        m.device = device
        return m

    noWorkspacesImmutable = noWorkspaces

    def hasConfiguration(self, arrayType):
        return arrayType in self._names or arrayType in self._scoped_out

    def isScopedOut(self, arrayType):
        return arrayType in self._scoped_out

    def getWorkspaceName(self, arrayType):
        self._check(arrayType)
        return self._names.get(arrayType)

    def getConfiguration(self, arrayType):
        return self._confs.get(self._names.get(arrayType))

    def setScopedOutFor(self, arrayType):
        self._scoped_out.add(arrayType)
        self._names.pop(arrayType, None)

    def setWorkspace(self, arrayType, wsName, conf=None):
        self._names[arrayType] = wsName
        self._confs[wsName] = conf or WorkspaceConfiguration()
        self._scoped_out.discard(arrayType)

    def _check(self, arrayType):
        if not self.hasConfiguration(arrayType):
            raise ValueError("No workspace configuration has been provided for array type %s" % arrayType)

    def _ws(self, arrayType) -> "MemoryWorkspace":
        return getWorkspaceManager().getWorkspaceForCurrentThread(self._confs[self._names[arrayType]], self._names[arrayType], self.device)

    def notifyScopeEntered(self, arrayType):
        self._check(arrayType)
        if arrayType in self._scoped_out:
            return getWorkspaceManager().scopeOutOfWorkspaces()
        return _EnteredScope(self._ws(arrayType).notifyScopeEntered())

    def notifyScopeBorrowed(self, arrayType):
        self._check(arrayType)
        if arrayType in self._scoped_out:
            return getWorkspaceManager().scopeOutOfWorkspaces()
        return self._ws(arrayType).notifyScopeBorrowed()

    def isWorkspaceOpen(self, arrayType):
        self._check(arrayType)
        return arrayType not in self._scoped_out and getWorkspaceManager().checkIfWorkspaceExistsAndActive(self._names[arrayType])

    def assertOpen(self, arrayType, msg=""):
        if arrayType not in self._scoped_out and not self.isWorkspaceOpen(arrayType):
            raise RuntimeError("Assertion failed: expected workspace for array type %s to be open: %s" % (arrayType, msg))

    def assertNotOpen(self, arrayType, msg=""):
        if arrayType not in self._scoped_out and self.isWorkspaceOpen(arrayType):
            raise RuntimeError("Assertion failed: expected workspace for array type %s to not be open: %s" % (arrayType, msg))

    def create(self, arrayType, dtype, *shape, order="c"):
        t = self.createUninitialized(arrayType, dtype, *shape)
        return t.zero_()

    def createUninitialized(self, arrayType, dtype, *shape, order="c"):
        self._check(arrayType)
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (list, tuple)) else tuple(shape)
        if arrayType in self._scoped_out or not self.isWorkspaceOpen(arrayType):
            return torch.empty(shape, dtype=dtype, device=self.device)
        return self._ws(arrayType).alloc(shape, dtype, False, self.device)

    def leverageTo(self, arrayType, array):
        """Move an array into the workspace for ``arrayType`` (detached copy when scoped out)."""
        self._check(arrayType)
        t = getattr(array, "_t", array)
        if arrayType in self._scoped_out or not self.isWorkspaceOpen(arrayType):
            return t.clone()
        out = self._ws(arrayType).alloc(tuple(t.shape), t.dtype, False, t.device)
        return out.copy_(t)

    def dup(self, arrayType, array):
        return self.leverageTo(arrayType, array)

    def castTo(self, arrayType, dtype, array, dupIfCorrectType=True):
        t = getattr(array, "_t", array)
        if t.dtype == dtype:
            return self.dup(arrayType, t) if dupIfCorrectType else t
        out = self.createUninitialized(arrayType, dtype, *t.shape)
        return out.copy_(t)

    def setHelperWorkspace(self, key, value):
        self._helper_ws[key] = value

    def getHelperWorkspace(self, key):
        return self._helper_ws.get(key)
