"""Session-level objects of the SameDiff runtime under the names the reference uses.

The reference executes a graph through ``AbstractSession`` subclasses (nd4j-api
org/nd4j/autodiff/samediff/internal/AbstractSession.java, InferenceSession.java, TrainingSession.java), hands array
lifetimes to a ``SessionMemMgr`` (internal/memory/ArrayCacheMemoryMgr.java, NoOpMemoryMgr.java) and tracks readiness with
a ``DependencyTracker`` (internal/DependencyTracker.java). Here the dataflow interpreter lives in
:meth:`SameDiff._run_graph`; an :class:`InferenceSession` plugs an :class:`ArrayLifetime` into it: a DependencyTracker records
which ops still have to read each produced array, the array goes to the session's memory manager the moment the last reader has
run, and same-shape transform / pairwise ops take their output buffer from that manager (ArrayCacheMemoryMgr: recycled).
"""
from __future__ import annotations

from collections import defaultdict, deque
from typing import Dict, Hashable, Iterable, List, Optional, Sequence

import torch


# ---------------------------------------------------------------------------------------------- memory managers
class SessionMemMgr:
    """Allocate / release interface (reference: internal/SessionMemMgr.java)."""

    def allocate(self, detached: bool, dataType: torch.dtype, *shape, device=None) -> torch.Tensor:
        raise NotImplementedError

    def release(self, array: torch.Tensor) -> None:
        raise NotImplementedError

    def ulike(self, arr: torch.Tensor) -> torch.Tensor:
        return self.allocate(False, arr.dtype, *arr.shape, device=arr.device)

    def dup(self, arr: torch.Tensor) -> torch.Tensor:
        out = self.ulike(arr)
        out.copy_(arr)
        return out

    def close(self) -> None:
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class NoOpMemoryMgr(SessionMemMgr):
    """Every allocation is fresh, release does nothing (reference: internal/memory/NoOpMemoryMgr.java)."""

    def allocate(self, detached, dataType, *shape, device=None):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
        return torch.empty(shape, dtype=dataType, device=device)

    def release(self, array):
        pass


class ArrayCacheMemoryMgr(SessionMemMgr):
    """Released arrays are kept, keyed by (device, dtype, element count), and reused for later allocations of the same
    size (reference: internal/memory/ArrayCacheMemoryMgr.java: cache bounded by a fraction of device memory, LRU eviction
    beyond it). ``maxCacheBytes`` bounds the cache; eviction is oldest-first."""

    def __init__(self, maxMemFrac: float = 0.25, maxCacheBytes: Optional[int] = None, smallArrayThreshold: int = 1024):
        if maxCacheBytes is None:
            total = torch.cuda.get_device_properties(0).total_memory if torch.cuda.is_available() else (4 << 30)
            maxCacheBytes = int(total * maxMemFrac)
        self.maxCacheBytes = int(maxCacheBytes)
        self.smallArrayThreshold = smallArrayThreshold
        self._cache: Dict[tuple, List[torch.Tensor]] = defaultdict(list)
        self._lru: deque = deque()
        self.currentCacheSize = 0
        self.hits = self.misses = 0

    @staticmethod
    def _key(device, dtype, numel):
        return (str(torch.device(device) if device is not None else torch.device("cpu")), dtype, int(numel))

    def allocate(self, detached, dataType, *shape, device=None):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        n = 1
        for s in shape:
            n *= int(s)
        k = self._key(device, dataType, n)
        pool = self._cache.get(k)
        if pool:
            t = pool.pop()
            self._lru.remove(id(t))
            self.currentCacheSize -= t.numel() * t.element_size()
            self.hits += 1
            return t.view(shape)
        self.misses += 1
        return torch.empty(shape, dtype=dataType, device=device)

    def release(self, array):
        if array is None or array._base is not None and array._base.numel() != array.numel():
            return                                  # a strict view of a larger buffer is not ours to recycle
        flat = array.detach().reshape(-1)
        nbytes = flat.numel() * flat.element_size()
        if nbytes > self.maxCacheBytes:
            return
        while self.currentCacheSize + nbytes > self.maxCacheBytes and self._lru:
            self._evict_oldest()
        k = self._key(flat.device, flat.dtype, flat.numel())
        self._cache[k].append(flat)
        self._lru.append(id(flat))
        self.currentCacheSize += nbytes

    def _evict_oldest(self):
        victim = self._lru.popleft()
        for k, pool in self._cache.items():
            for i, t in enumerate(pool):
                if id(t) == victim:
                    self.currentCacheSize -= t.numel() * t.element_size()
                    del pool[i]
                    return

    def close(self):
        self._cache.clear()
        self._lru.clear()
        self.currentCacheSize = 0


# ---------------------------------------------------------------------------------------------- dependency tracking
class DependencyTracker:
    """``addDependency(y, x)``: y needs x. ``markSatisfied(x, True)`` may make dependees "newly all-satisfied"; they are
    drained with :meth:`getNewAllSatisfied`. OR-dependencies (either of a pair) as in the reference
    (internal/AbstractDependencyTracker.java:addOrDependency)."""

    def __init__(self):
        self._deps: Dict[Hashable, set] = defaultdict(set)
        self._or_deps: Dict[Hashable, List[tuple]] = defaultdict(list)
        self._rev: Dict[Hashable, set] = defaultdict(set)
        self._satisfied: set = set()
        self._queue: deque = deque()
        self._queued: set = set()
        self._reported: set = set()

    def clear(self):
        self.__init__()

    def isEmpty(self):
        return not self._deps and not self._or_deps and not self._queue

    def isSatisfied(self, x):
        return x in self._satisfied

    def hasDependency(self, y):
        return bool(self._deps.get(y)) or bool(self._or_deps.get(y))

    def getDependencies(self, y):
        return set(self._deps.get(y, ())), list(self._or_deps.get(y, ()))

    def addDependency(self, y, x):
        self._deps[y].add(x)
        self._rev[x].add(y)
        self._check(y)

    def addOrDependency(self, y, x1, x2):
        self._or_deps[y].append((x1, x2))
        self._rev[x1].add(y)
        self._rev[x2].add(y)
        self._check(y)

    def removeDependency(self, y, x):
        self._deps.get(y, set()).discard(x)
        self._or_deps[y] = [p for p in self._or_deps.get(y, []) if x not in p]
        self._rev.get(x, set()).discard(y)
        self._check(y)

    def _all_satisfied(self, y):
        return (all(d in self._satisfied for d in self._deps.get(y, ()))
                and all(a in self._satisfied or b in self._satisfied for a, b in self._or_deps.get(y, ())))

    def _check(self, y):
        ok = self._all_satisfied(y) and self.hasDependency(y)
        if ok and y not in self._queued and y not in self._reported:
            self._queue.append(y)
            self._queued.add(y)
        elif not ok:
            self._reported.discard(y)             # becomes reportable again once re-satisfied
            if y in self._queued:
                self._queued.discard(y)
                self._queue = deque(q for q in self._queue if q != y)

    def markSatisfied(self, x, satisfied: bool = True):
        if satisfied:
            self._satisfied.add(x)
        else:
            self._satisfied.discard(x)
        for y in list(self._rev.get(x, ())):
            self._check(y)

    def hasNewAllSatisfied(self):
        return bool(self._queue)

    def getNewAllSatisfied(self):
        y = self._queue.popleft()
        self._queued.discard(y)
        self._reported.add(y)
        return y

    def getNewAllSatisfiedList(self):
        out = list(self._queue)
        self._queue.clear()
        self._queued.clear()
        self._reported.update(out)
        return out


class HashDependencyTracker(DependencyTracker):
    """Keys compared by ``==``/hash (the default here); reference: internal/DependencyTracker.java vs
    IdentityDependencyTracker.java."""


class IdentityDependencyTracker(DependencyTracker):
    """Keys compared by identity: values are wrapped so that equal-but-distinct tensors are distinct dependencies."""

    class _Id:
        __slots__ = ("o",)

        def __init__(self, o):
            self.o = o

        def __hash__(self):
            return id(self.o)

        def __eq__(self, other):
            return isinstance(other, IdentityDependencyTracker._Id) and other.o is self.o

    def _w(self, x):
        return x if isinstance(x, IdentityDependencyTracker._Id) else IdentityDependencyTracker._Id(x)

    def addDependency(self, y, x):
        super().addDependency(self._w(y), self._w(x))

    def addOrDependency(self, y, x1, x2):
        super().addOrDependency(self._w(y), self._w(x1), self._w(x2))

    def markSatisfied(self, x, satisfied=True):
        super().markSatisfied(self._w(x), satisfied)

    def isSatisfied(self, x):
        return super().isSatisfied(self._w(x))

    def hasDependency(self, y):
        return super().hasDependency(self._w(y))

    def getNewAllSatisfied(self):
        return super().getNewAllSatisfied().o

    def getNewAllSatisfiedList(self):
        return [w.o for w in super().getNewAllSatisfiedList()]


class ArrayLifetime:
    """Per-execution array lifetimes of an inference session (reference: InferenceSession.arrayUseTracker + mmgr.release in
    internal/InferenceSession.java:96-140, 340-380). Arrays that share storage (views from reshape / permute / identity) are released
    only when the last alias is dead."""

    def __init__(self, mmgr: SessionMemMgr):
        self.mmgr = mmgr
        self.tracker = DependencyTracker()
        self.released = 0
        self.peakLive = 0

    def begin(self, keep):
        self.tracker.clear()
        self._keep = set(keep)
        self._arrays: Dict[str, torch.Tensor] = {}
        self._aliases: Dict[int, int] = defaultdict(int)
        self.released = 0
        self.peakLive = 0

    @staticmethod
    def _sid(t):
        return t.untyped_storage().data_ptr() if t.numel() else id(t)

    def produced(self, name, arr, consumers):
        if not isinstance(arr, torch.Tensor) or name in self._keep:
            if isinstance(arr, torch.Tensor):
                self._aliases[self._sid(arr)] += 1                      # an alias that never dies: pins the storage
            return
        self._arrays[name] = arr
        self._aliases[self._sid(arr)] += 1
        self.peakLive = max(self.peakLive, len(self._arrays))
        for c in consumers:
            self.tracker.addDependency(("arr", name), ("op", c))
        if not consumers:
            self.tracker.addDependency(("arr", name), ("op", None))
            self.tracker.markSatisfied(("op", None))

    def pin(self, arr):
        """Storage owned by someone else (variables, constants, placeholders): never recycled."""
        if isinstance(arr, torch.Tensor):
            self._aliases[self._sid(arr)] += 1

    def executed(self, opName) -> List[str]:
        self.tracker.markSatisfied(("op", opName))
        dead = []
        while self.tracker.hasNewAllSatisfied():
            _, name = self.tracker.getNewAllSatisfied()
            arr = self._arrays.pop(name, None)
            if arr is None:
                continue
            dead.append(name)
            sid = self._sid(arr)
            self._aliases[sid] -= 1
            if self._aliases[sid] <= 0:
                del self._aliases[sid]
                self.mmgr.release(arr)
                self.released += 1
        return dead


# ---------------------------------------------------------------------------------------------- sessions
class AbstractSession:
    """One graph + one executor (reference: internal/AbstractSession.java:output())."""

    def __init__(self, sameDiff, mmgr: Optional[SessionMemMgr] = None):
        self.sameDiff = sameDiff
        self.mmgr = mmgr or NoOpMemoryMgr()

    def _names(self, variables):
        from .samediff import SDVariable
        if variables is None:
            return self.sameDiff.outputs()
        if isinstance(variables, (str, SDVariable)):
            variables = [variables]
        return [v.name if isinstance(v, SDVariable) else v for v in variables]

    def subgraphOps(self, variables: Sequence) -> List[str]:
        """Names of the ops that have to run for ``variables`` (reference: AbstractSession.initSubgraph)."""
        need = self.sameDiff._required_ops(self._names(variables))
        return [n for n in self.sameDiff.ops if n in need]

    def output(self, variables=None, placeholderValues=None, listeners=(), at=None, training=False):
        sd = self.sameDiff
        ph = sd._prep_placeholders(placeholderValues)
        return sd._run_graph(ph, self._names(variables), training, tuple(sd.listeners) + tuple(listeners), at)


class InferenceSession(AbstractSession):
    """Forward execution without autograd recording (reference: internal/InferenceSession.java). Outputs of the previous
    call that the caller handed back with :meth:`release` are recycled through the memory manager."""

    def __init__(self, sameDiff, mmgr: Optional[SessionMemMgr] = None):
        super().__init__(sameDiff, mmgr or ArrayCacheMemoryMgr(maxCacheBytes=256 << 20))

    def output(self, variables=None, placeholderValues=None, listeners=(), at=None, training=False):
        from .listeners import At, Operation
        at = at or At(0, 0, Operation.INFERENCE)
        sd = self.sameDiff
        self.lifetime = ArrayLifetime(self.mmgr)
        with torch.no_grad():
            ph = sd._prep_placeholders(placeholderValues)
            return sd._run_graph(ph, self._names(variables), False, tuple(sd.listeners) + tuple(listeners), at, lifetime=self.lifetime)

    def release(self, outputs: Dict[str, torch.Tensor]):
        for t in outputs.values():
            self.mmgr.release(t)


class TrainingSession(InferenceSession):
    """One training iteration: forward, backward and the updater, with listener callbacks (reference:
    internal/TrainingSession.java:trainingIteration). Returns {loss variable: value} of the iteration."""

    def __init__(self, sameDiff):
        AbstractSession.__init__(self, sameDiff, NoOpMemoryMgr())

    def trainingIteration(self, config=None, placeholders=None, paramsToTrain: Optional[Iterable[str]] = None, updaters=None,
                          batch=None, lossVariables: Optional[Sequence[str]] = None, listeners=(), at=None):
        from .listeners import At, Operation
        sd = self.sameDiff
        if config is not None and sd.trainingConfig is not config:
            sd.setTrainingConfig(config)
        tc = sd.trainingConfig
        if tc is None:
            raise ValueError("no TrainingConfig set")
        if lossVariables:
            sd.setLossVariables(*lossVariables)
        trainable = [n for n in sd.trainableVariables() if paramsToTrain is None or n in set(paramsToTrain)]
        sd._ensure_updaters(trainable)
        at = at or At(tc.epochCount, tc.iterationCount, Operation.TRAINING)
        ls = list(sd.listeners) + list(listeners)
        ph = placeholders if placeholders is not None else sd._placeholders_from_dataset(batch)
        losses, grads, _ = sd._loss_and_grads(ph, trainable, (), ls, at)
        sd._apply_updates(grads, at)
        sd._last_grads.update(grads)
        tc.iterationCount += 1
        return dict(zip(sd.lossVariables, [float(v) for v in losses]))


# ---------------------------------------------------------------------------------------------- scopes / control flow
class NameScope:
    """``with NameScope(sd, "block1"):`` — variables and ops created inside get the prefix ``block1/`` (reference:
    samediff/NameScope.java, SameDiff.withNameScope)."""

    def __init__(self, sameDiff, name: str):
        self.sameDiff, self.name = sameDiff, name

    def getName(self):
        return self.name

    def __enter__(self):
        self.sameDiff._scope.append(self.name)
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if self.sameDiff._scope and self.sameDiff._scope[-1] == self.name:
            self.sameDiff._scope.pop()

    def __str__(self):
        return "NameScope(%s)" % self.name


class ControlFlow:
    """Static builders for conditionals and loops in TF-frame form (Switch/Merge, Enter/Exit/NextIteration/LoopCond);
    reference: samediff/internal/… + SameDiff.ifCond / whileLoop, which these delegate to."""

    @staticmethod
    def ifCond(sameDiff, cond, trueBody, falseBody, name=None):
        return sameDiff.ifCond(cond, trueBody, falseBody, name=name)

    @staticmethod
    def whileLoop(sameDiff, loopVars, cond, body, name=None):
        return sameDiff.whileLoop(loopVars, cond, body, name=name)


class ArgumentInterceptor:
    """Callable applied to every SDVariable argument of ops created while it is installed with
    ``sd.addArgumentInterceptor(i)`` / removed with ``sd.removeArgumentInterceptor()`` (reference:
    samediff/ArgumentInterceptor.java; SameDiff.ifCond uses one to route outer values through Switch)."""

    def intercept(self, argument):
        return argument

    def __call__(self, argument):
        return self.intercept(argument)
