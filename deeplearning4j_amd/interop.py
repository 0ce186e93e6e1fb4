"""Foreign-model runners (reference: nd4j-onnxruntime OnnxRuntimeRunner, nd4j-tensorflow GraphRunner,
nd4j-tensorflow-lite TensorFlowLiteRunner, nd4j-tvm TvmRunner — each wraps a foreign C runtime through JavaCPP presets
and exchanges INDArray <-> native tensors).

MI355X design: a foreign graph is not handed to a second runtime (which would run on the CPU or need its own ROCm
build); it is imported into SameDiff and executed by THIS framework's kernels, so ONNX / TensorFlow models get the
same device residency, HIP-graph-friendly execution and op coverage as native models. If the foreign runtime happens
to be importable (onnxruntime is not in this image) ``useForeignRuntime=True`` can select it for cross-checking.
TensorFlow Lite files are decoded and executed operator by operator on device tensors (tflite.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch


def _unwrap(v):
    return getattr(v, "_t", v)


class _ImportedGraphRunner:
    def __init__(self, sd, inputNames: Optional[Sequence[str]], outputNames: Optional[Sequence[str]]):
        self.sd = sd
        self.inputNames = list(inputNames) if inputNames else list(sd.inputs())
        self.outputNames = list(outputNames) if outputNames else list(sd.outputs())

    def exec(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        missing = [n for n in self.inputNames if n not in inputs]
        if missing:
            raise ValueError("Missing input(s) %s; graph inputs are %s" % (missing, self.inputNames))
        return self.sd.output({k: _unwrap(v) for k, v in inputs.items()}, self.outputNames)

    run = exec

    def getInputNames(self) -> List[str]:
        return list(self.inputNames)

    def getOutputNames(self) -> List[str]:
        return list(self.outputNames)

    def close(self):
        self.sd = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False


class OnnxRuntimeRunner(_ImportedGraphRunner):
    """``OnnxRuntimeRunner.builder().modelUri(path).build().exec(inputs)``."""

    class Builder:
        def __init__(self):
            self._uri, self._outs, self._foreign, self._device = None, None, False, None

        def modelUri(self, uri):
            self._uri = str(uri); return self

        def outputNames(self, names):
            self._outs = list(names); return self

        def useForeignRuntime(self, b):
            self._foreign = bool(b); return self

        def device(self, d):
            self._device = d; return self

        def build(self):
            return OnnxRuntimeRunner(self._uri, self._outs, self._foreign, self._device)

    @staticmethod
    def builder():
        return OnnxRuntimeRunner.Builder()

    def __init__(self, modelUri, outputNames=None, useForeignRuntime=False, device=None):
        self._session = None
        if useForeignRuntime:
            try:
                import onnxruntime as ort
            except ImportError as e:
                raise RuntimeError("onnxruntime is not installed in this environment; use the built-in importer "
                                   "(useForeignRuntime=False)") from e
            self._session = ort.InferenceSession(modelUri)
            self.inputNames = [i.name for i in self._session.get_inputs()]
            self.outputNames = list(outputNames) if outputNames else [o.name for o in self._session.get_outputs()]
            self.sd = None
            return
        from .autodiff.import_onnx import OnnxImporter
        super().__init__(OnnxImporter.importGraph(modelUri, device=device), None, outputNames)

    def exec(self, inputs):
        if self._session is not None:
            res = self._session.run(self.outputNames, {k: _unwrap(v).cpu().numpy() for k, v in inputs.items()})
            return {n: torch.from_numpy(r) for n, r in zip(self.outputNames, res)}
        return super().exec(inputs)

    run = exec


class GraphRunner(_ImportedGraphRunner):
    """TensorFlow frozen-graph runner (``GraphRunner.builder().graphPath(..).inputNames(..).outputNames(..).build()``)."""

    class Builder:
        def __init__(self):
            self._path = self._bytes = self._ins = self._outs = self._device = None

        def graphPath(self, p):
            self._path = str(p); return self

        def graphBytes(self, b):
            self._bytes = bytes(b); return self

        def inputNames(self, n):
            self._ins = list(n); return self

        def outputNames(self, n):
            self._outs = list(n); return self

        def device(self, d):
            self._device = d; return self

        def build(self):
            return GraphRunner(self._bytes if self._bytes is not None else self._path, self._ins, self._outs, self._device)

    @staticmethod
    def builder():
        return GraphRunner.Builder()

    def __init__(self, graph, inputNames=None, outputNames=None, device=None):
        from .autodiff.import_tf import TFGraphMapper
        super().__init__(TFGraphMapper.importGraph(graph, device=device), inputNames, outputNames)


from .tflite import TensorFlowLiteRunner  # noqa: E402,F401  (.tflite FlatBuffers decoded and executed on device tensors)


class TvmRunner:
    """nd4j-tvm runner: TVM compiled modules are CPU/CUDA artefacts and are not loadable on gfx950; re-import the source model."""

    def __init__(self, modelUri):
        raise NotImplementedError("TVM modules are not portable to gfx950; import the original ONNX / TensorFlow model instead")
