"""The framework-neutral face of model import, under the names of the reference's samediff-import modules.

Reference: nd4j/samediff-import/samediff-import-api (ImportGraph.kt, OpMappingRegistry.kt, process/MappingProcess.kt) with
the -tensorflow and -onnx front ends. The actual importers here are :class:`TFGraphMapper` (import_tf.py) and
:class:`OnnxImporter` (import_onnx.py): protobuf is decoded by the wire-format reader in protowire.py and every framework
op maps to one rule function ``rule(ctx, node) -> SDVariable | [SDVariable]``. This module puts the two rule tables
behind one registry object, gives a rule a describable wrapper (MappingProcess) and offers one entry point that detects
the file format.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

from . import import_onnx as _onnx
from . import import_tf as _tf
from . import serde as _serde


class MappingProcess:
    """One import rule: which framework op it consumes, which graph op(s) it emits, and the callable doing it."""

    def __init__(self, inputFramework: str, inputFrameworkOpName: str, rule: Callable, opName: Optional[str] = None):
        self._fw, self._in, self._rule = inputFramework, inputFrameworkOpName, rule
        self._op = opName or getattr(rule, "__name__", "rule").lstrip("_")

    def inputFramework(self):
        return self._fw

    def inputFrameworkOpName(self):
        return self._in

    def opName(self):
        return self._op

    def applyProcess(self, ctx, node):
        return self._rule(ctx, node)

    __call__ = applyProcess

    def __repr__(self):
        return "MappingProcess(%s:%s -> %s)" % (self._fw, self._in, self._op)


class OpMappingRegistry:
    """Rule table of one input framework ("tensorflow" or "onnx")."""

    _TABLES = {"tensorflow": _tf._RULES, "onnx": _onnx._RULES}

    def __init__(self, inputFrameworkName: str):
        key = inputFrameworkName.lower()
        if key not in self._TABLES:
            raise ValueError("unknown framework '%s' (tensorflow | onnx)" % inputFrameworkName)
        self.inputFrameworkName = key
        self._rules: Dict[str, Callable] = self._TABLES[key]

    def inputFrameworkOpNames(self) -> List[str]:
        return sorted(self._rules)

    def hasMappingOpProcess(self, inputFrameworkOpName: str) -> bool:
        return inputFrameworkOpName in self._rules

    def lookupOpMappingProcess(self, inputFrameworkOpName: str) -> MappingProcess:
        if inputFrameworkOpName not in self._rules:
            raise KeyError("no mapping process for %s op '%s'" % (self.inputFrameworkName, inputFrameworkOpName))
        r = self._rules[inputFrameworkOpName]
        return r if isinstance(r, MappingProcess) else MappingProcess(self.inputFrameworkName, inputFrameworkOpName, r)

    def registerMappingProcess(self, inputFrameworkOpName: str, processOrRule) -> None:
        self._rules[inputFrameworkOpName] = processOrRule

    def opMappingProcesses(self) -> Dict[str, MappingProcess]:
        return {k: self.lookupOpMappingProcess(k) for k in self._rules}


def _sniff(buf: bytes) -> str:
    """GraphDef: top-level field 1 (node, length-delimited) -> first byte 0x0A. ModelProto: field 1 is ir_version (varint)
    -> first byte 0x08. SameDiff FlatBuffers: a uoffset_t root pointer, then usually a vtable — neither tag byte first
    followed by a sane protobuf stream."""
    if not buf:
        raise ValueError("empty model file")
    if buf[0] == 0x08:
        return "onnx"
    if buf[0] == 0x0A:
        return "tensorflow"
    return "flatbuffers"


class ImportGraph:
    """``ImportGraph().importGraph(path)``: TensorFlow frozen GraphDef (.pb), ONNX ModelProto (.onnx) or a SameDiff
    FlatBuffers file (.fb), chosen by extension and else by the first bytes."""

    def importGraph(self, path_or_bytes, framework: Optional[str] = None, overrides=None, device=None, **kw):
        if isinstance(path_or_bytes, (bytes, bytearray)):
            buf, name = bytes(path_or_bytes), ""
        else:
            name = str(path_or_bytes).lower()
            with open(str(path_or_bytes), "rb") as f:
                buf = f.read()
        if framework is None:
            framework = ("onnx" if name.endswith(".onnx") else "tensorflow" if name.endswith((".pb", ".pbtxt"))
                         else "flatbuffers" if name.endswith((".fb", ".sdz")) else _sniff(buf))
        framework = framework.lower()
        if framework == "onnx":
            return _onnx.OnnxImporter.importGraph(buf, overrides=overrides, device=device, **kw)
        if framework in ("tensorflow", "tf"):
            return _tf.TFGraphMapper.importGraph(buf, overrides=overrides, device=device, **kw)
        if framework == "flatbuffers":
            return _serde.from_flatbuffers(buf, device=device, **kw)
        raise ValueError("unknown framework '%s'" % framework)


class FlatBuffersMapper:
    """Graph <-> FlatBuffers bytes (reference: nd4j-api org/nd4j/autodiff/samediff/serde/FlatBuffersMapper.java; the
    schema is libnd4j/include/graph/scheme/*.fbs, written by the builder in flatbuf.py)."""

    @staticmethod
    def asFlatBuffers(sd, saveUpdaterState: bool = True) -> bytes:
        return _serde.to_flatbuffers(sd, saveUpdaterState)

    @staticmethod
    def fromFlatBuffers(buf: bytes, loadUpdaterState: bool = True, device=None):
        return _serde.from_flatbuffers(buf, loadUpdaterState, device)

    @staticmethod
    def asFlatArray(t):
        return _serde.flat_array(t)
