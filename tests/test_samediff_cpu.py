"""SameDiff: graph building, sessions, gradients, training, control flow, listeners, FlatBuffers serde
(mirrors nd4j SameDiffTests / SameDiffTrainingTest / FlatBufferSerdeTest; the .fb fixtures are the reference's own
libnd4j/tests_cpu/resources files, parsed with the in-tree FlatBuffers reader — data only, nothing executed from them)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from deeplearning4j_amd.autodiff.samediff import SameDiff, TrainingConfig, VariableType
from deeplearning4j_amd.autodiff import listeners as LS
from deeplearning4j_amd.datasets.dataset import DataSet
from deeplearning4j_amd.datasets.iterators import ListDataSetIterator
from deeplearning4j_amd.nd4j.learning import Adam, Sgd

FIX = "/root/reference/libnd4j/tests_cpu/resources"


def test_build_and_execute_graph():
    sd = SameDiff.create("cpu")
    x = sd.placeHolder("x", "FLOAT", -1, 4)
    w = sd.var("w", torch.arange(12.0).reshape(4, 3) / 10)
    b = sd.var("b", torch.tensor([0.1, 0.2, 0.3]))
    z = x.mmul(w).add(b).rename("z")
    out = sd.nn.softmax(z, name="out")
    xin = torch.tensor([[1.0, 2, 3, 4], [0, 1, 0, 1]])
    res = sd.output({"x": xin}, "out", "z")
    ref = xin @ (torch.arange(12.0).reshape(4, 3) / 10) + torch.tensor([0.1, 0.2, 0.3])
    assert torch.allclose(res["z"], ref) and torch.allclose(res["out"], torch.softmax(ref, -1))
    assert sd.inputs() == ["x"] and sd.outputs() == ["out"] and sd.getVariable("w").variableType == VariableType.VARIABLE
    assert out.eval({"x": xin}).shape() == [2, 3]
    with pytest.raises(ValueError):
        sd.output({}, "out")
    # legacy-family nodes + reductions + slicing sugar
    t = sd.math.tanh(z)
    s = (t * t).sum(1).rename("s")
    sl = z[:, 1:3].rename("sl")
    r = sd.output({"x": xin}, ["s", "sl"])
    assert torch.allclose(r["s"], (torch.tanh(ref) ** 2).sum(1)) and torch.equal(r["sl"], ref[:, 1:3])
    assert "Trainable params: 15" in sd.summary()
    # generic registry access
    tk = sd.math.top_k(z, iArgs=[2], nOut=2)
    assert sd.output({"x": xin}, [tk[1].name])[tk[1].name].tolist() == ref.topk(2).indices.tolist()


def test_gradients_match_numeric_and_autograd():
    torch.manual_seed(0)
    sd = SameDiff.create("cpu")
    x = sd.placeHolder("x", "DOUBLE", -1, 3)
    y = sd.placeHolder("y", "DOUBLE", -1, 2)
    w = sd.var("w", torch.randn(3, 2, dtype=torch.float64))
    b = sd.var("b", torch.randn(2, dtype=torch.float64))
    h = sd.nn.tanh(x.mmul(w) + b)
    loss = sd.loss.meanSquaredError(y, h, None, 3, name="loss")
    sd.setLossVariables("loss")
    xv, yv = torch.randn(5, 3, dtype=torch.float64), torch.randn(5, 2, dtype=torch.float64)
    g = sd.calculateGradients({"x": xv, "y": yv}, "w", "b", "x")
    from deeplearning4j_amd.autodiff.validation import GradCheckUtil
    assert GradCheckUtil.checkGradients(sd, {"x": xv, "y": yv}, eps=1e-6, maxRelError=1e-5)
    wt = sd.getArrForVarName("w").tensor().clone().requires_grad_(True)
    ref = ((torch.tanh(xv @ wt + sd.getArrForVarName("b").tensor()) - yv) ** 2).mean()
    ref.backward()
    assert torch.allclose(g["w"].tensor(), wt.grad, atol=1e-10) and g["x"].shape() == [5, 3]
    assert sd.grad("w").getArr().equalsWithEps(g["w"], 1e-12)


def test_training_linear_regression_and_mlp():
    torch.manual_seed(1)
    sd = SameDiff.create("cpu")
    x = sd.placeHolder("in", "FLOAT", -1, 4)
    y = sd.placeHolder("label", "FLOAT", -1, 1)
    w = sd.var("w", torch.zeros(4, 1))
    b = sd.zero("b", 1)
    pred = (x.mmul(w) + b).rename("pred")
    sd.loss.meanSquaredError(y, pred, None, 3, name="loss").markAsLoss()
    sd.setTrainingConfig(TrainingConfig.builder().updater(Sgd(0.1)).dataSetFeatureMapping("in").dataSetLabelMapping("label").build())
    true_w = torch.tensor([[1.0], [-2.0], [0.5], [3.0]])
    X = torch.randn(256, 4)
    Y = X @ true_w + 0.7
    it = ListDataSetIterator(DataSet(X, Y).batchBy(32), 32)
    hl = LS.HistoryListener()
    hist = sd.fit(it, 30, listeners=[hl])
    assert hist.lossCurve.lastMeanLoss()[0] < 1e-3 < hist.lossCurve.meanLoss(0)[0]
    assert torch.allclose(sd.getArrForVarName("w").tensor(), true_w, atol=2e-2) and abs(float(sd.getArrForVarName("b").tensor()) - 0.7) < 2e-2
    assert len(hl.losses) == 30 * 8 and sd.trainingConfig.iterationCount == 240 and sd.trainingConfig.epochCount == 30
    # MLP classifier with Adam + l2, evaluated through sd.evaluate
    sd2 = SameDiff.create("cpu")
    xi = sd2.placeHolder("in", "FLOAT", -1, 2)
    lab = sd2.placeHolder("label", "FLOAT", -1, 2)
    w0, b0 = sd2.var("w0", 2, 16), sd2.zero("b0", 16)
    w1, b1 = sd2.var("w1", 16, 2), sd2.zero("b1", 2)
    logits = sd2.nn.linear(sd2.nn.relu(sd2.nn.linear(xi, w0, b0)), w1, b1)
    sd2.nn.softmax(logits, name="probs")
    sd2.loss.softmaxCrossEntropy(lab, logits, None, 3, name="loss").markAsLoss()
    sd2.setTrainingConfig(TrainingConfig.builder().updater(Adam(0.02)).l2(1e-4).dataSetFeatureMapping("in").dataSetLabelMapping("label").build())
    P = torch.randn(400, 2)
    cls = ((P[:, 0] * P[:, 1]) > 0).long()          # XOR-like quadrants
    it2 = ListDataSetIterator(DataSet(P, torch.eye(2)[cls]).batchBy(50), 50)
    h2 = sd2.fit(it2, 60)
    assert h2.lossCurve.lastMeanLoss()[0] < 0.25 < h2.lossCurve.meanLoss(0)[0]
    from deeplearning4j_amd.eval.evaluation import Evaluation
    ev = sd2.evaluate(it2, "probs", Evaluation())
    assert ev.accuracy() > 0.9 and sd2.calcRegularizationScore() > 0


def test_control_flow_if_and_while():
    sd = SameDiff.create("cpu")
    a = sd.placeHolder("a", "FLOAT")
    out = sd.ifCond(lambda s: a.gt(0.0), lambda s, t: (a * 2.0), lambda s, f: (a - 10.0), name="c")
    # bodies must depend on the switch outputs to be gated: route `a` through them
    sd2 = SameDiff.create("cpu")
    a2 = sd2.placeHolder("a", "FLOAT")
    o2 = sd2.ifCond(lambda s: a2.gt(0.0), lambda s, t: s.base.castTo(t, "FLOAT") * 0.0 + a2 * 2.0,
                    lambda s, f: s.base.castTo(f, "FLOAT") * 0.0 + a2 - 10.0).rename("o")
    assert float(sd2.output({"a": torch.tensor(3.0)}, "o")["o"]) == 6.0
    assert float(sd2.output({"a": torch.tensor(-3.0)}, "o")["o"]) == -13.0
    # while: (i, acc) -> i < n : (i+1, acc + i*i)
    sw = SameDiff.create("cpu")
    n = sw.placeHolder("n", "FLOAT")
    i0, acc0 = sw.constant("i0", torch.tensor(0.0)), sw.constant("acc0", torch.tensor(0.0))
    cf = LS.ControlflowListener()
    sw.setListeners(cf)
    res = sw.whileLoop([i0, acc0], lambda s, v: v[0].lt(n), lambda s, v: [v[0] + 1.0, v[1] + v[0] * v[0]], name="loop")
    r = sw.output({"n": torch.tensor(5.0)}, [res[0].name, res[1].name])
    assert float(r[res[0].name]) == 5.0 and float(r[res[1].name]) == 0 + 1 + 4 + 9 + 16
    r = sw.output({"n": torch.tensor(0.0)}, [res[1].name])
    assert float(r[res[1].name]) == 0.0


def test_flatbuffers_roundtrip_and_listeners(tmp_path):
    torch.manual_seed(2)
    sd = SameDiff.create("cpu")
    x = sd.placeHolder("in", "FLOAT", -1, 3)
    y = sd.placeHolder("label", "FLOAT", -1, 2)
    w = sd.var("w", torch.randn(3, 2))
    pred = sd.nn.sigmoid(x.mmul(w), name="pred")
    sd.loss.logLoss(y, pred, None, 3, 1e-7, name="loss").markAsLoss()
    sd.setTrainingConfig(TrainingConfig.builder().updater(Adam(0.05)).dataSetFeatureMapping("in").dataSetLabelMapping("label").build())
    X, Y = torch.randn(64, 3), (torch.rand(64, 2) > 0.5).float()
    ck = LS.CheckpointListener(tmp_path / "ck", saveEveryNEpochs=1, keepLast=2)
    prof = LS.ProfilingListener(tmp_path / "trace.json")
    bench = LS.OpBenchmarkListener()
    sd.fit(DataSet(X, Y), 4, listeners=[ck, prof, bench, LS.UIListener(tmp_path / "ui.jsonl"), LS.ScoreListener(2)])
    assert len(ck.availableCheckpoints()) == 2 and ck.lastCheckpoint()["epoch"] == 3
    trace = json.load(open(tmp_path / "trace.json"))
    assert any(e["args"]["opName"] == "matmul" for e in trace["traceEvents"])
    assert LS.ProfileAnalyzer.summarize(tmp_path / "trace.json")[0]["count"] >= 4 and bench.agg
    assert len(open(tmp_path / "ui.jsonl").read().splitlines()) == 4
    p = tmp_path / "m.fb"
    sd.save(p, True)
    sd2 = SameDiff.load(p, device="cpu")
    assert list(sd2.variables) == list(sd.variables) and list(sd2.ops) == list(sd.ops)
    assert sd2.lossVariables == ["loss"] and sd2.trainingConfig.iterationCount == 4 and sd2.inputs() == ["in", "label"]
    assert torch.equal(sd2.getArrForVarName("w").tensor(), sd.getArrForVarName("w").tensor())
    assert torch.equal(sd2.updaterState["w"], sd.updaterState["w"])
    a, b = sd.output({"in": X}, "pred")["pred"], sd2.output({"in": X}, "pred")["pred"]
    assert torch.equal(a, b)
    # training resumes identically from the checkpoint (updater state + iteration counter restored)
    sd.fit(DataSet(X, Y), 1)
    sd2.fit(DataSet(X, Y), 1)
    assert torch.allclose(sd.getArrForVarName("w").tensor(), sd2.getArrForVarName("w").tensor(), atol=1e-7)
    sd3 = LS.CheckpointListener.loadLastCheckpoint(tmp_path / "ck")
    assert sd3.trainingConfig.epochCount == 4 or sd3.trainingConfig.epochCount == 3


@pytest.mark.skipif(not os.path.isdir(FIX), reason="reference fixtures not present")
def test_reference_fb_fixtures_parse_and_execute():
    from deeplearning4j_amd.nd4j.ops.custom import op_hash
    names = sorted(glob.glob(FIX + "/*.fb"))
    assert len(names) >= 15
    loaded = {os.path.basename(f): SameDiff.load(f, device="cpu") for f in names}
    # custom-op opNum stored in the files is the 64-bit name hash: the native registry reproduces it bit for bit
    for sd in loaded.values():
        for o in sd.ops.values():
            if o.opType == 21 and not o.opName.startswith("unknown"):
                from deeplearning4j_amd.nd4j.ops.custom import name_by_hash
                canon = name_by_hash(op_hash(o.opName))          # aliases (mmul) store the canonical op's hash
                assert canon is not None and o.opNum == op_hash(canon), o.opName
    sd = loaded["identity_n_2.fb"]
    r = sd.output({}, ["IdentityN", "IdentityN:1"])
    assert torch.equal(r["IdentityN"], sd.getArrForVarName("in_0").tensor()) and torch.equal(r["IdentityN:1"], sd.getArrForVarName("in_1").tensor())
    sd = loaded["channels_last_b1_k2_s1_d1_SAME_crelu.fb"]
    out = sd.output({}, ["conv2d/CRelu/Relu"])["conv2d/CRelu/Relu"]
    x, k = sd.getArrForVarName("in_0").tensor(), sd.getArrForVarName("conv2d/kernel").tensor()
    bias = [sd._arrays[n] for n in sd._arrays if "bias" in n][0]
    conv = torch.nn.functional.conv2d(torch.nn.functional.pad(x.permute(0, 3, 1, 2), [0, 1, 0, 1]), k.permute(3, 2, 0, 1), bias).permute(0, 2, 3, 1)
    assert torch.allclose(out, torch.relu(torch.cat([conv, -conv], -1)), atol=1e-5)
    sd = loaded["cond_true.fb"]
    res = sd.output({}, ["cond/Merge"])["cond/Merge"]
    assert res.numel() >= 1
    sd = loaded["reduce_all_rank2_d0_keep.fb"]
    outs = sd.outputs()
    assert sd.output({}, outs)
    sd = loaded["pad_1D.fb"]
    assert sd.output({}, sd.outputs())
    sd = loaded["scatter_nd_update.fb"]
    assert sd.output({}, sd.outputs())
    sd = loaded["avg_pooling3d.fb"]
    o = sd.output({}, sd.outputs())
    assert list(o.values())[0].dim() == 5
    sd = loaded["simplewhile_1.fb"]
    # while sum(x) < y: x += 2; y -= 1   -> x = 3s, y = 9, output = x + y
    o = sd.output({"input_0": torch.ones(2, 2), "input_1": torch.tensor(10.0)}, ["output"])["output"]
    assert torch.equal(o, torch.full((2, 2), 12.0))
    sd = loaded["while_iter3.fb"]          # i = 0; while i < 3: i += 1
    o = sd.output({}, ["while/Exit", "while/Exit_1"])
    assert float(o["while/Exit"]) == 3.0 and float(o["while/Exit_1"]) == 3.0
    sd = loaded["simplewhile_nested.fb"]
    ph = {n: torch.ones([int(s) if s and s > 0 else 2 for s in (sd.variables[n].shape or [])]) * 3 for n in sd.inputs()}
    res = sd.output(ph, sd.outputs())
    assert all(torch.isfinite(v.float()).all() for v in res.values())


# ------------------------------------------------------------------------------------------------ gradient graph (round 2)
def _mlp_graph():
    from deeplearning4j_amd.autodiff.samediff import SameDiff
    sd = SameDiff.create(device="cpu")
    x = sd.placeHolder("x", "FLOAT", -1, 6)
    y = sd.placeHolder("y", "FLOAT", -1, 3)
    w0 = sd.var("w0", torch.randn(6, 8, generator=torch.Generator().manual_seed(1)) * 0.4)
    b0 = sd.var("b0", torch.zeros(8))
    w1 = sd.var("w1", torch.randn(8, 3, generator=torch.Generator().manual_seed(2)) * 0.4)
    b1 = sd.var("b1", torch.zeros(3))
    g = sd.var("g", torch.ones(8))
    h = sd._op("tanh", [sd._op("biasadd", [x.mmul(w0), b0])[0]])[0]
    h = sd._op("layer_norm", [h, g], iArgs=[1])[0]
    h2 = h.mul(h).add(h)                                # a variable with several consumers: gradient accumulation
    logits = sd._op("xw_plus_b", [h2, w1, b1])[0]
    p = sd._op("softmax", [logits], iArgs=[1])[0]
    diff = p.sub(y)
    loss = diff.mul(diff).sum(1).mean(0)
    loss.rename("loss")
    sd.setLossVariables("loss")
    return sd


def test_grad_function_is_a_graph_of_bp_ops_and_matches_autograd():
    sd = _mlp_graph()
    g = torch.Generator().manual_seed(5)
    ph = {"x": torch.randn(7, 6, generator=g), "y": torch.eye(3)[torch.randint(0, 3, (7,), generator=g)]}
    names = ["w0", "b0", "w1", "b1", "g"]
    ref = sd.calculateGradients(ph, *names)
    gf = sd.createGradFunction(names)
    assert gf is sd.createGradFunction(names)                      # cached
    assert set(gf.grads) == set(names) and all(gf.grads[n].endswith("-grad") for n in names)
    opnames = {o.opName for o in gf.sd.ops.values()}
    assert {"matmul_bp", "tanh_bp", "layer_norm_bp", "xw_plus_b_bp", "softmax_bp", "multiply_bp", "biasadd_bp"} <= opnames
    assert "mergeadd" in opnames                                   # h feeds mul (twice) and add
    losses, got = gf.execute(ph, names)
    for n in names:
        assert torch.allclose(got[n], ref[n].tensor() if hasattr(ref[n], "tensor") else ref[n]._t, atol=1e-5), n
    # the same through the public switch, and training with it
    sd.setGradientMode("graph")
    got2 = sd.calculateGradients(ph, *names)
    for n in names:
        a, b = got2[n], ref[n]
        a = a.tensor() if hasattr(a, "tensor") else a._t
        b = b.tensor() if hasattr(b, "tensor") else b._t
        assert torch.allclose(a, b, atol=1e-5)


def test_grad_function_vjp_fallback_for_ops_without_rule():
    from deeplearning4j_amd.autodiff.samediff import SameDiff
    sd = SameDiff.create(device="cpu")
    x = sd.placeHolder("x", "FLOAT", -1, 4)
    w = sd.var("w", torch.randn(4, 4, generator=torch.Generator().manual_seed(3)))
    h = sd.math.exp(x.mmul(w).mul(0.1))                            # legacy transform + scalar op: no doDiff rule -> __vjp__ nodes
    loss = sd._op("reduce_logsumexp", [h])[0]
    loss.rename("loss")
    sd.setLossVariables("loss")
    ph = {"x": torch.randn(5, 4, generator=torch.Generator().manual_seed(4))}
    ref = sd.calculateGradients(ph, "w")["w"]
    gf = sd.createGradFunction(["w"])
    assert gf.vjp_ops and "matmul_bp" in {o.opName for o in gf.sd.ops.values()}
    _, got = gf.execute(ph, ["w"])
    ref_t = ref.tensor() if hasattr(ref, "tensor") else ref._t
    assert torch.allclose(got["w"], ref_t, atol=1e-5)


def test_inference_session_releases_dead_arrays_and_recycles_buffers():
    """DependencyTracker + ArrayCacheMemoryMgr are live parts of InferenceSession: intermediates are released when their last reader
    has run, and transform outputs are written into recycled buffers (reference: internal/InferenceSession.java arrayUseTracker)."""
    from deeplearning4j_amd.autodiff.samediff import SameDiff
    from deeplearning4j_amd.autodiff.sessions import InferenceSession, ArrayCacheMemoryMgr
    sd = SameDiff.create(device="cpu")
    x = sd.placeHolder("x", "FLOAT", -1, 32)
    w = sd.var("w", torch.randn(32, 32, generator=torch.Generator().manual_seed(0)) * 0.2)
    h = x
    for i in range(6):
        h = sd.math.tanh(h.mmul(w))
        h = sd.math.abs(h) if i % 2 else sd.math.neg(h)
    r = sd._op("reshape", [w], iArgs=[-99, 1024])[0]           # a view of a PARAMETER: must never reach the cache
    out = h.add(r.sum())
    out.rename("out")
    mm = ArrayCacheMemoryMgr(maxCacheBytes=1 << 20)
    sess = InferenceSession(sd, mm)
    xin = torch.randn(16, 32, generator=torch.Generator().manual_seed(1))
    ref = sd.output({"x": xin}, ["out"])["out"].clone()
    w0 = sd._arrays["w"].clone()
    o1 = sess.output(["out"], {"x": xin})["out"]
    assert torch.allclose(o1, ref)
    lt = sess.lifetime
    assert lt.released >= 10 and lt.peakLive <= 4, (lt.released, lt.peakLive)     # 19 intermediates, at most a few alive at once
    assert mm.hits > 0                                                              # later transforms wrote into recycled buffers
    o1 = o1.clone()
    o2 = sess.output(["out"], {"x": xin})["out"]
    assert torch.allclose(o2, o1) and torch.equal(sd._arrays["w"], w0)
