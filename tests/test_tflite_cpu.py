"""TFLite runner: models are assembled with the in-tree FlatBuffers writer (no .tflite fixture ships with the reference and
the TFLite runtime is not installable here — parity with the real interpreter is unpinned) and checked against PyTorch."""
import pytest
import torch
import torch.nn.functional as F

from deeplearning4j_amd.autodiff.flatbuf import S
from deeplearning4j_amd.tflite import TFLiteBuilder, TFLiteModel, TensorFlowLiteRunner


def _cnn(tmp_path=None):
    g = torch.Generator().manual_seed(0)
    w1 = torch.randn(8, 3, 3, 3, generator=g) * 0.2            # OHWI
    b1 = torch.randn(8, generator=g) * 0.1
    wd = torch.randn(1, 3, 3, 8, generator=g) * 0.2            # depthwise [1,kh,kw,C]
    w2 = torch.randn(10, 8 * 4 * 4, generator=g) * 0.1
    b2 = torch.randn(10, generator=g) * 0.1
    b = TFLiteBuilder()
    x = b.tensor("input", [1, 16, 16, 3])
    tw1, tb1 = b.tensor("w1", w1.shape, w1), b.tensor("b1", b1.shape, b1)
    c1 = b.tensor("conv1", [1, 8, 8, 8])
    b.op("CONV_2D", [x, tw1, tb1], [c1], {0: S.Scalar("byte", 0), 1: S.Scalar("int", 2), 2: S.Scalar("int", 2), 3: S.Scalar("byte", 1)})
    twd = b.tensor("wd", wd.shape, wd)
    d1 = b.tensor("dw1", [1, 8, 8, 8])
    b.op("DEPTHWISE_CONV_2D", [c1, twd, -1], [d1], {0: S.Scalar("byte", 0), 1: S.Scalar("int", 1), 2: S.Scalar("int", 1),
                                                    3: S.Scalar("int", 1), 4: S.Scalar("byte", 3)})
    s1 = b.tensor("sum", [1, 8, 8, 8])
    b.op("ADD", [c1, d1], [s1], {0: S.Scalar("byte", 0)})
    p1 = b.tensor("pool", [1, 4, 4, 8])
    b.op("MAX_POOL_2D", [s1], [p1], {0: S.Scalar("byte", 1), 1: S.Scalar("int", 2), 2: S.Scalar("int", 2), 3: S.Scalar("int", 2),
                                     4: S.Scalar("int", 2)})
    shp = b.tensor("shape", [2], torch.tensor([-1, 128]), ttype=2)
    r1 = b.tensor("flat", [1, 128])
    b.op("RESHAPE", [p1, shp], [r1])
    tw2, tb2 = b.tensor("w2", w2.shape, w2), b.tensor("b2", b2.shape, b2)
    f1 = b.tensor("logits", [1, 10])
    b.op("FULLY_CONNECTED", [r1, tw2, tb2], [f1], {0: S.Scalar("byte", 0)})
    o = b.tensor("probs", [1, 10])
    b.op("SOFTMAX", [f1], [o], {0: S.Scalar("float", 1.0)})
    buf = b.finish([x], [o])

    def ref(xin):                                                # NHWC in, plain PyTorch
        xi = xin.permute(0, 3, 1, 2)
        # SAME, stride 2, k 3 on 16 -> out 8, total pad 1: 0 before, 1 after
        y = F.relu(F.conv2d(F.pad(xi, (0, 1, 0, 1)), w1.permute(0, 3, 1, 2), b1, stride=2))
        d = F.conv2d(F.pad(y, (1, 1, 1, 1)), wd.permute(3, 0, 1, 2), None, groups=8).clamp(0, 6)
        p = F.max_pool2d(y + d, 2, 2)
        flat = p.permute(0, 2, 3, 1).reshape(-1, 128)
        return torch.softmax(F.linear(flat, w2, b2), -1)
    return buf, ref


def test_tflite_cnn_matches_pytorch(tmp_path):
    buf, ref = _cnn()
    p = tmp_path / "m.tflite"
    p.write_bytes(buf)
    m = TFLiteModel(str(p))
    assert m.version == 3 and m.opNames() == ["CONV_2D", "DEPTHWISE_CONV_2D", "ADD", "MAX_POOL_2D", "RESHAPE", "FULLY_CONNECTED", "SOFTMAX"]
    r = TensorFlowLiteRunner.builder().modelUri(str(p)).device("cpu").build()
    assert r.getInputNames() == ["input"] and r.getOutputNames() == ["probs"]
    x = torch.randn(4, 16, 16, 3, generator=torch.Generator().manual_seed(1))        # batch dimension is free
    out = r.exec({"input": x})["probs"]
    assert out.shape == (4, 10)
    assert torch.allclose(out, ref(x), atol=1e-5)
    with pytest.raises(ValueError, match="Missing input"):
        r.exec({})
    with pytest.raises(ValueError, match="expects"):
        r.exec({"input": torch.zeros(1, 8, 8, 3)})


def test_tflite_elementwise_reduce_and_shape_ops():
    b = TFLiteBuilder()
    x = b.tensor("x", [2, 4, 4, 2])
    ax = b.tensor("axes", [2], torch.tensor([1, 2]), ttype=2)
    m = b.tensor("mean", [2, 2])
    b.op("MEAN", [x, ax], [m], {0: S.Scalar("bool", False)})
    t = b.tensor("tanh", [2, 2])
    b.op("TANH", [m], [t])
    perm = b.tensor("perm", [2], torch.tensor([1, 0]), ttype=2)
    tr = b.tensor("tr", [2, 2])
    b.op("TRANSPOSE", [t, perm], [tr])
    pads = b.tensor("pads", [2, 2], torch.tensor([[1, 0], [0, 2]]), ttype=2)
    pd = b.tensor("pad", [3, 4])
    b.op("PAD", [tr, pads], [pd])
    cat = b.tensor("cat", [6, 4])
    b.op("CONCATENATION", [pd, pd], [cat], {0: S.Scalar("int", 0), 1: S.Scalar("byte", 1)})
    ap = b.tensor("avg", [2, 2, 2, 2])
    b.op("AVERAGE_POOL_2D", [x], [ap], {0: S.Scalar("byte", 0), 1: S.Scalar("int", 3), 2: S.Scalar("int", 3), 3: S.Scalar("int", 3),
                                        4: S.Scalar("int", 3)})
    r = TensorFlowLiteRunner(b.finish([x], [cat, ap]), device="cpu")
    xs = torch.randn(2, 4, 4, 2, generator=torch.Generator().manual_seed(2))
    out = r.exec({"x": xs})
    ref = F.relu(torch.cat([F.pad(torch.tanh(xs.mean((1, 2))).t(), (0, 2, 1, 0))] * 2, 0))
    assert torch.allclose(out["cat"], ref, atol=1e-6)
    # SAME average pooling averages over valid elements only: k3 s3 on 4 -> out 2, pad (1,1): windows rows [0,1] and [2,3] (+pad)
    xi = xs.permute(0, 3, 1, 2)
    num = F.avg_pool2d(F.pad(xi, (1, 1, 1, 1)), 3, 3)
    den = F.avg_pool2d(F.pad(torch.ones(1, 1, 4, 4), (1, 1, 1, 1)), 3, 3)
    assert torch.allclose(out["avg"], (num / den).permute(0, 2, 3, 1), atol=1e-6)


def test_tflite_rejects_unsupported_and_malformed():
    b = TFLiteBuilder()
    x = b.tensor("x", [1, 4])
    y = b.tensor("y", [1, 4])
    b._codes.append("SIN")
    b.op("SIN", [x], [y])
    buf = b.finish([x], [y])
    assert torch.allclose(TensorFlowLiteRunner(buf, device="cpu").exec({"x": torch.ones(1, 4)})["y"], torch.sin(torch.ones(1, 4)))
    with pytest.raises(ValueError):
        TFLiteModel(b"\x00\x01")
    b2 = TFLiteBuilder()
    x2 = b2.tensor("x", [1, 4])
    b2.op("RELU", [x2], [7])                                    # output index beyond the tensor table
    with pytest.raises(ValueError, match="refers to tensor"):
        TFLiteModel(b2.finish([x2], [x2]))
