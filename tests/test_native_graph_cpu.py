"""Native (C++) graph runtime: FlatGraph loader, layering, GraphHolder, control-flow executor, profiles.
Mirrors libnd4j tests_cpu GraphTests / FlatBuffersTests / GraphHolderTests; .fb fixtures are the reference's own
data files (libnd4j/tests_cpu/resources), read with our own FlatBuffers parser (nothing from them is executed)."""
import glob
import os

import pytest
import torch

from deeplearning4j_amd.autodiff.samediff import SameDiff
from deeplearning4j_amd.autodiff.native_graph import NativeGraphExecutioner, NativeGraph

FIX = "/root/reference/libnd4j/tests_cpu/resources"


def _mlp():
    torch.manual_seed(0)
    sd = SameDiff.create("cpu")
    x = sd.placeHolder("in", "FLOAT", -1, 4)
    w0, b0 = sd.var("w0", torch.randn(4, 8)), sd.var("b0", torch.randn(8))
    w1 = sd.var("w1", torch.randn(8, 3))
    h = sd.nn.relu(x.mmul(w0) + b0, 0.0)
    sd.nn.softmax(h.mmul(w1), 1, name="probs")
    return sd


def test_register_layers_execute_unregister():
    sd = _mlp()
    g = NativeGraphExecutioner.registerGraph(7, sd, device="cpu")
    assert g.numNodes() == len(sd.ops) and set(sd.variables) <= set(g.variableNames())
    assert g.placeholders() == ["in"]
    layers = g.layers()
    nodes = g.nodes()
    # a node's inputs are produced in strictly earlier layers
    producer = {o: n for n in nodes for o in n.outputs}
    for n in nodes:
        for i in n.inputs:
            if i in producer:
                assert producer[i].layer < n.layer
    assert sum(len(l) for l in layers) == len(nodes)
    assert torch.equal(g.getVariableArray("w0"), sd.getArrForVarName("w0").tensor())
    x = torch.randn(5, 4)
    a = sd.output({"in": x}, "probs")["probs"]
    b = g.execute({"in": x}, ["probs"])["probs"]
    assert torch.equal(a, b)
    # stored-graph API + profile
    b2 = NativeGraphExecutioner.executeStoredGraph(7, {"in": x}, ["probs"], device="cpu")["probs"]
    assert torch.equal(a, b2)
    prof = g.profile()
    assert prof["executions"] >= 1 and any(r[1] == "matmul" and r[2] >= 1 for r in prof["nodes"])
    g.resetProfile()
    assert g.profile()["executions"] == 0
    # updated parameters are visible to the next execution (VariableSpace semantics)
    g.setVariableArray("w1", torch.zeros(8, 3))
    assert torch.allclose(g.execute({"in": x}, ["probs"])["probs"], torch.full((5, 3), 1 / 3))
    NativeGraphExecutioner.unregisterGraph(7)
    with pytest.raises(KeyError):
        NativeGraph(7)


def test_control_flow_matches_python_session():
    sw = SameDiff.create("cpu")
    n = sw.placeHolder("n", "FLOAT")
    i0, acc0 = sw.constant("i0", torch.tensor(0.0)), sw.constant("acc0", torch.tensor(0.0))
    res = sw.whileLoop([i0, acc0], lambda s, v: v[0].lt(n), lambda s, v: [v[0] + 1.0, v[1] + v[0] * v[0]], name="loop")
    names = [res[0].name, res[1].name]
    g = NativeGraphExecutioner.registerGraph(11, sw, device="cpu")
    for k in (5.0, 0.0, 3.0):
        r = g.execute({"n": torch.tensor(k)}, names)
        ref = sw.output({"n": torch.tensor(k)}, names)
        assert float(r[names[0]]) == float(ref[names[0]]) == k
        assert float(r[names[1]]) == float(ref[names[1]]) == sum(i * i for i in range(int(k)))
    NativeGraphExecutioner.unregisterGraph(11)
    sd2 = SameDiff.create("cpu")
    a2 = sd2.placeHolder("a", "FLOAT")
    sd2.ifCond(lambda s: a2.gt(0.0), lambda s, t: s.base.castTo(t, "FLOAT") * 0.0 + a2 * 2.0,
               lambda s, f: s.base.castTo(f, "FLOAT") * 0.0 + a2 - 10.0).rename("o")
    buf = sd2.asFlatBuffers(False)
    assert float(NativeGraphExecutioner.executeFlatGraph(buf, {"a": torch.tensor(3.0)}, ["o"], device="cpu")["o"]) == 6.0
    assert float(NativeGraphExecutioner.executeFlatGraph(buf, {"a": torch.tensor(-3.0)}, ["o"], device="cpu")["o"]) == -13.0
    # a loop that never terminates is reported through the error channel, not a hang
    sl = SameDiff.create("cpu")
    j0 = sl.constant("j0", torch.tensor(0.0))
    out = sl.whileLoop([j0], lambda s, v: v[0].gte(0.0), lambda s, v: [v[0] + 1.0], name="inf")
    gl = NativeGraphExecutioner.registerGraph(12, sl, device="cpu")
    with pytest.raises(RuntimeError, match="did not terminate"):
        gl.execute({}, [out[0].name], maxIterations=50)
    NativeGraphExecutioner.unregisterGraph(12)


@pytest.mark.skipif(not os.path.isdir(FIX), reason="reference fixtures not present")
def test_reference_fb_fixtures_native_executor():
    def run(name, ph, outs):
        buf = open(os.path.join(FIX, name), "rb").read()
        return NativeGraphExecutioner.executeFlatGraph(buf, ph, outs, device="cpu")
    o = run("simplewhile_1.fb", {"input_0": torch.ones(2, 2), "input_1": torch.tensor(10.0)}, ["output"])["output"]
    assert torch.equal(o, torch.full((2, 2), 12.0))
    o = run("while_iter3.fb", {}, ["while/Exit", "while/Exit_1"])
    assert float(o["while/Exit"]) == 3.0 and float(o["while/Exit_1"]) == 3.0
    # every fixture parses + layers natively and agrees with the Python session where both can run it
    agree = 0
    for f in sorted(glob.glob(FIX + "/*.fb")):
        buf = open(f, "rb").read()
        g = NativeGraphExecutioner.registerGraph(99, buf, device="cpu")
        assert g.numNodes() >= 0 and sum(len(l) for l in g.layers()) == g.numNodes()
        sd = SameDiff.load(f, device="cpu")
        if not sd.inputs() and sd.outputs():
            try:
                ref = sd.output({}, sd.outputs())
            except Exception:
                NativeGraphExecutioner.unregisterGraph(99)
                continue
            got = g.execute({}, sd.outputs())
            for k, v in ref.items():
                if v.is_floating_point():
                    assert torch.allclose(got[k], v, equal_nan=True), (f, k)
                else:
                    assert torch.equal(got[k], v), (f, k)
            agree += 1
        NativeGraphExecutioner.unregisterGraph(99)
    assert agree >= 5
