"""python4j analogue (reference: python4j/python4j-core PythonExecutioner, PythonGIL, PythonContextManager,
PythonVariables / PythonTypes, PythonJob; python4j-numpy NumpyArray INDArray<->numpy zero-copy).

In the reference this embeds CPython in the JVM so pipelines can run user Python. Here the host language already IS
Python, so the component reduces to its contract: run user code in named, isolated variable contexts under one
global interpreter lock, with typed variable marshalling, and exchange tensors with numpy without copying.
"""
from __future__ import annotations

import threading
from typing import Any, Dict, List, Optional

import numpy as np
import torch


class PythonType:
    def __init__(self, name, pytypes, adapter=None):
        self.name, self.pytypes, self.adapter = name, pytypes, adapter

    def accepts(self, v):
        return isinstance(v, self.pytypes)

    def toJava(self, v):           # name kept from the reference: python object -> framework value
        return self.adapter(v) if self.adapter else v

    def __repr__(self):
        return "PythonType(%s)" % self.name


def _to_ndarray(v):
    from .nd4j.ndarray import INDArray
    if isinstance(v, INDArray):
        return v
    if isinstance(v, np.ndarray):
        return INDArray(torch.from_numpy(v))          # shares memory with the numpy array
    return INDArray(torch.as_tensor(v))


class PythonTypes:
    INT = PythonType("int", (int, np.integer), int)
    FLOAT = PythonType("float", (float, np.floating, int), float)
    STR = PythonType("str", (str,), str)
    BOOL = PythonType("bool", (bool, np.bool_), bool)
    BYTES = PythonType("bytes", (bytes, bytearray, memoryview), bytes)
    LIST = PythonType("list", (list, tuple), list)
    DICT = PythonType("dict", (dict,), dict)
    NDARRAY = PythonType("numpy.ndarray", (np.ndarray, torch.Tensor, object), _to_ndarray)
    _ALL = None

    @staticmethod
    def get(name):
        for t in (PythonTypes.INT, PythonTypes.FLOAT, PythonTypes.STR, PythonTypes.BOOL, PythonTypes.BYTES, PythonTypes.LIST,
                  PythonTypes.DICT, PythonTypes.NDARRAY):
            if t.name == name:
                return t
        raise KeyError("Unknown python type: %s" % name)

    @staticmethod
    def getPythonTypeForObject(v):
        from .nd4j.ndarray import INDArray
        if isinstance(v, (bool, np.bool_)):
            return PythonTypes.BOOL
        if isinstance(v, (np.ndarray, torch.Tensor, INDArray)):
            return PythonTypes.NDARRAY
        for t in (PythonTypes.INT, PythonTypes.FLOAT, PythonTypes.STR, PythonTypes.BYTES, PythonTypes.LIST, PythonTypes.DICT):
            if t.accepts(v):
                return t
        raise TypeError("No python4j type for %s" % type(v).__name__)


class PythonVariable:
    def __init__(self, name, type_: PythonType, value=None):
        self.name, self.type, self.value = name, type_, value

    def getName(self):
        return self.name

    def getType(self):
        return self.type

    def getValue(self):
        return self.value

    def setValue(self, v):
        self.value = v


class PythonVariables(list):
    def add(self, name, type_, value=None):
        self.append(PythonVariable(name, type_, value))
        return self

    def get(self, name):
        for v in self:
            if v.name == name:
                return v
        raise KeyError(name)


class NumpyArray:
    """INDArray <-> numpy without copying (host tensors) — python4j-numpy NumpyArray."""

    @staticmethod
    def toNumpy(arr) -> np.ndarray:
        t = getattr(arr, "_t", arr)
        t = t.detach()
        if t.is_cuda:
            t = t.cpu()                                # device tensors need one D2H copy
        if t.dtype == torch.bfloat16:
            t = t.float()
        return t.numpy()

    @staticmethod
    def fromNumpy(a: np.ndarray):
        return _to_ndarray(a)


class PythonGIL:
    """Process-wide re-entrant lock guarding user-code execution (`try (PythonGIL gil = PythonGIL.lock())`)."""
    _lock = threading.RLock()
    _owner_depth = threading.local()

    def __enter__(self):
        PythonGIL._lock.acquire()
        PythonGIL._owner_depth.d = getattr(PythonGIL._owner_depth, "d", 0) + 1
        return self

    def __exit__(self, *exc):
        PythonGIL._owner_depth.d -= 1
        PythonGIL._lock.release()
        return False

    @staticmethod
    def lock():
        return PythonGIL()

    @staticmethod
    def locked():
        return getattr(PythonGIL._owner_depth, "d", 0) > 0

    @staticmethod
    def assertThreadSafe():
        if not PythonGIL.locked():
            raise RuntimeError("Attempt to use Python outside of a PythonGIL block")


class PythonContextManager:
    """Named, isolated global-variable contexts."""
    _contexts: Dict[str, dict] = {"main": {"__name__": "__python4j_main__"}}
    _current = threading.local()

    @staticmethod
    def getCurrentContext():
        return getattr(PythonContextManager._current, "name", "main")

    @staticmethod
    def setContext(name):
        PythonContextManager._contexts.setdefault(name, {"__name__": "__python4j_%s__" % name})
        PythonContextManager._current.name = name

    @staticmethod
    def setMainContext():
        PythonContextManager.setContext("main")

    @staticmethod
    def deleteContext(name):
        if name == "main":
            raise ValueError("Can not delete main context")
        PythonContextManager._contexts.pop(name, None)
        if PythonContextManager.getCurrentContext() == name:
            PythonContextManager.setMainContext()

    @staticmethod
    def deleteNonMainContexts():
        for n in [k for k in PythonContextManager._contexts if k != "main"]:
            PythonContextManager.deleteContext(n)

    @staticmethod
    def getContexts():
        return list(PythonContextManager._contexts)

    @staticmethod
    def globals_():
        return PythonContextManager._contexts[PythonContextManager.getCurrentContext()]

    @staticmethod
    def reset():
        g = PythonContextManager.globals_()
        name = g.get("__name__")
        g.clear()
        g["__name__"] = name


class PythonExecutioner:
    @staticmethod
    def setVariable(name, value):
        PythonGIL.assertThreadSafe()
        PythonContextManager.globals_()[name] = value

    @staticmethod
    def setVariables(variables: List[PythonVariable]):
        for v in variables or []:
            PythonExecutioner.setVariable(v.name, v.value)

    @staticmethod
    def getVariable(name, type_: Optional[PythonType] = None):
        PythonGIL.assertThreadSafe()
        g = PythonContextManager.globals_()
        if name not in g:
            raise KeyError("Variable not found: %s" % name)
        v = g[name]
        t = type_ or PythonTypes.getPythonTypeForObject(v)
        return PythonVariable(name, t, t.toJava(v))

    @staticmethod
    def getVariables(variables: List[PythonVariable]):
        for v in variables or []:
            v.setValue(PythonExecutioner.getVariable(v.name, v.type).value)

    @staticmethod
    def simpleExec(code: str):
        PythonGIL.assertThreadSafe()
        g = PythonContextManager.globals_()
        exec(compile(code, "<python4j>", "exec"), g, g)

    @staticmethod
    def exec(code: str, inputs: Optional[List[PythonVariable]] = None, outputs: Optional[List[PythonVariable]] = None):
        PythonExecutioner.setVariables(inputs)
        PythonExecutioner.simpleExec(code)
        PythonExecutioner.getVariables(outputs)

    @staticmethod
    def execAndReturnAllVariables(code: str, inputs: Optional[List[PythonVariable]] = None) -> PythonVariables:
        PythonExecutioner.setVariables(inputs)
        PythonExecutioner.simpleExec(code)
        out = PythonVariables()
        for k, v in PythonContextManager.globals_().items():
            if k.startswith("__") or callable(v) or type(v).__name__ == "module":
                continue
            try:
                t = PythonTypes.getPythonTypeForObject(v)
            except TypeError:
                continue
            out.add(k, t, t.toJava(v))
        return out

    @staticmethod
    def getAllVariables() -> PythonVariables:
        return PythonExecutioner.execAndReturnAllVariables("")


class PythonJob:
    """A reusable unit of user code with optional ``setup()`` / ``run(**inputs)`` functions, bound to its own context."""

    def __init__(self, name: str, code: str, setupRunMode: bool = False):
        self.name, self.code, self.setupRunMode = name, code, setupRunMode
        self.context = "__job_" + name
        self._setup_done = False

    def _enter(self):
        self._prev = PythonContextManager.getCurrentContext()
        PythonContextManager.setContext(self.context)

    def _leave(self):
        PythonContextManager.setContext(self._prev)

    def setup(self):
        with PythonGIL.lock():
            self._enter()
            try:
                PythonContextManager.reset()
                PythonExecutioner.simpleExec(self.code)
                if self.setupRunMode and "setup" in PythonContextManager.globals_():
                    PythonContextManager.globals_()["setup"]()
                self._setup_done = True
            finally:
                self._leave()

    def exec(self, inputs: Optional[List[PythonVariable]] = None, outputs: Optional[List[PythonVariable]] = None):
        with PythonGIL.lock():
            self._enter()
            try:
                if self.setupRunMode:
                    if not self._setup_done:
                        self._leave()
                        self.setup()
                        self._enter()
                    res = PythonContextManager.globals_()["run"](**{v.name: v.value for v in (inputs or [])})
                    for o in outputs or []:
                        o.setValue(o.type.toJava(res[o.name]))
                else:
                    PythonExecutioner.exec(self.code, inputs, outputs)
            finally:
                self._leave()

    def clearState(self):
        PythonContextManager.deleteContext(self.context)
        self._setup_done = False
