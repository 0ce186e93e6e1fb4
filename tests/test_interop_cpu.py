"""python4j / omnihub / interop-runner analogues (reference: python4j-core tests PythonExecutionerTest, PythonContextManagerTest,
PythonJobTest; nd4j-onnxruntime OnnxRuntimeRunnerTests (add.onnx fixture); omnihub staging)."""
import glob
import os
import threading

import numpy as np
import pytest
import torch

from deeplearning4j_amd.python4j import (NumpyArray, PythonContextManager, PythonExecutioner, PythonGIL, PythonJob, PythonTypes,
                                         PythonVariable, PythonVariables)


def test_python4j_exec_contexts_and_jobs():
    with pytest.raises(RuntimeError):
        PythonExecutioner.simpleExec("x = 1")                      # outside a GIL block
    with PythonGIL.lock():
        ins = PythonVariables().add("x", PythonTypes.INT, 10).add("y", PythonTypes.INT, 20)
        out = [PythonVariable("z", PythonTypes.INT)]
        PythonExecutioner.exec("z = x + y", ins, out)
        assert out[0].getValue() == 30
        a = np.arange(6, dtype=np.float32).reshape(2, 3)
        o = [PythonVariable("b", PythonTypes.NDARRAY)]
        PythonExecutioner.exec("b = a * 2 + 1", [PythonVariable("a", PythonTypes.NDARRAY, a)], o)
        assert torch.equal(o[0].getValue().tensor(), torch.from_numpy(a * 2 + 1))
        allv = PythonExecutioner.execAndReturnAllVariables("s = 'hi'; f = 1.5; l = [1, 2]")
        assert allv.get("s").getValue() == "hi" and allv.get("f").getValue() == 1.5 and allv.get("l").getValue() == [1, 2]
        # contexts are isolated
        PythonContextManager.setContext("c1")
        PythonExecutioner.simpleExec("v = 1")
        PythonContextManager.setContext("c2")
        with pytest.raises(KeyError):
            PythonExecutioner.getVariable("v")
        PythonContextManager.setContext("c1")
        assert PythonExecutioner.getVariable("v").getValue() == 1
        PythonContextManager.setMainContext()
        PythonContextManager.deleteNonMainContexts()
        assert PythonContextManager.getContexts() == ["main"]
    # zero-copy numpy bridge
    arr = np.zeros(4, dtype=np.float32)
    nd = NumpyArray.fromNumpy(arr)
    arr[2] = 7
    assert float(nd.tensor()[2]) == 7 and NumpyArray.toNumpy(nd)[2] == 7
    # jobs keep state between runs, in their own context; the GIL serialises threads
    job = PythonJob("counter", "n = 0\ndef setup():\n    global n\n    n = 100\ndef run(inc):\n    global n\n    n += inc\n    return {'n': n}", True)
    res = []

    def work():
        o = [PythonVariable("n", PythonTypes.INT)]
        job.exec([PythonVariable("inc", PythonTypes.INT, 1)], o)
        res.append(o[0].getValue())
    ts = [threading.Thread(target=work) for _ in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert sorted(res) == list(range(101, 109))
    job.clearState()


def _onnx_add_model():
    from deeplearning4j_amd.autodiff.protowire import enc_msg

    def vi(name):
        shape = enc_msg(*[(1, enc_msg((1, d))) for d in (2, 2)])
        return enc_msg((1, name), (2, enc_msg((1, enc_msg((1, 1), (2, shape))))))
    node = enc_msg((1, "x"), (1, "y"), (2, "z"), (4, "Add"))
    graph = enc_msg((1, node), (2, "g"), (11, vi("x")), (11, vi("y")), (12, vi("z")))
    return enc_msg((1, 7), (7, graph), (8, enc_msg((2, 13))))


def test_onnx_runner_and_omnihub(tmp_path):
    from deeplearning4j_amd.interop import OnnxRuntimeRunner, TvmRunner
    from deeplearning4j_amd.omnihub import OmniHub
    p = tmp_path / "add.onnx"
    p.write_bytes(_onnx_add_model())
    with OnnxRuntimeRunner.builder().modelUri(p).device("cpu").build() as r:
        assert r.getInputNames() == ["x", "y"] and r.getOutputNames() == ["z"]
        x, y = torch.ones(2, 2), torch.full((2, 2), 2.0)
        assert torch.equal(r.exec({"x": x, "y": y})["z"], torch.full((2, 2), 3.0))
        with pytest.raises(ValueError):
            r.exec({"x": x})
    with pytest.raises(RuntimeError):
        OnnxRuntimeRunner(str(p), useForeignRuntime=True)          # onnxruntime is not installed here
    with pytest.raises(NotImplementedError):
        TvmRunner("m.so")
    hub = OmniHub(tmp_path / "hub")
    hub.stage("onnx", "add", p, source="unit-test")
    assert hub.listModels("onnx") == ["onnx/add"]
    sd = hub.load("onnx", "add", device="cpu")                    # converts + caches
    assert (tmp_path / "hub" / "samediff" / "add.fb").exists() and "samediff/add" in hub.listModels()
    sd2 = OmniHub(tmp_path / "hub").load("onnx", "add", device="cpu")   # second hub instance: served from the converted cache
    out = sd2.output({"x": torch.ones(2, 2), "y": torch.ones(2, 2)}, ["z"])["z"]
    assert torch.equal(out, torch.full((2, 2), 2.0)) and sd.outputs() == sd2.outputs()
    with pytest.raises(FileNotFoundError):
        hub.download("tensorflow", "resnet")


def test_tf_graph_runner_on_reference_fixture():
    fix = "/root/reference/nd4j/nd4j-backends/nd4j-tests/src/test/resources/lenet_frozen.pb"
    cands = glob.glob("/root/reference/**/lenet_frozen.pb", recursive=True) if not os.path.exists(fix) else [fix]
    if not cands:
        pytest.skip("reference fixture not present")
    from deeplearning4j_amd.interop import GraphRunner
    r = GraphRunner.builder().graphPath(cands[0]).device("cpu").build()
    assert r.getInputNames() and r.getOutputNames()
