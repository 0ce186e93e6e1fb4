"""Vendor-GEMM rule of ops/streams.py on the device: the route mode really lands on the in-tree GEMM and matches the library result,
side-stream regions refuse vendor GEMMs, the device token orders two streams, and ParallelInference workers sharing one device run
their fp32 GEMMs through it (reference: MmulHelper.cu:261, one mutex per device around cuBLAS)."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def test_route_mode_uses_the_in_tree_gemm_and_matches_the_library():
    from deeplearning4j_amd.ops import streams as S
    torch.manual_seed(0)
    for dt, tol in ((torch.float32, 2e-3), (torch.bfloat16, 3e-2)):
        a = torch.randn(96, 64, device=dev(), dtype=dt)
        b = torch.randn(64, 48, device=dev(), dtype=dt)
        w = torch.randn(48, 64, device=dev(), dtype=dt)
        bias = torch.randn(48, device=dev(), dtype=dt)
        want = (torch.mm(a, b), torch.nn.functional.linear(a, w, bias), torch.addmm(bias, a, b), a @ w.t(),
                torch.nn.functional.linear(a.view(4, 24, 64), w))
        g = S.library_gemm_guard("route")
        with g:
            got = (torch.mm(a, b), torch.nn.functional.linear(a, w, bias), torch.addmm(bias, a, b), a @ w.t(),
                   torch.nn.functional.linear(a.view(4, 24, 64), w))
            batched = torch.bmm(a[None], b[None])
        assert g.counts == {"routed": 5, "serialized": 1, "captured": 0}, g.counts
        for x, y in zip(got, want):
            assert x.shape == y.shape and x.dtype == y.dtype
            assert torch.allclose(x.float(), y.float(), atol=tol * 8, rtol=tol), float((x.float() - y.float()).abs().max())
        assert torch.allclose(batched[0].float(), want[0].float(), atol=tol * 8, rtol=tol)
    # operands that need autograd stay with torch (under the token)
    a = torch.randn(8, 16, device=dev(), requires_grad=True)
    g = S.library_gemm_guard("route")
    with g:
        y = torch.mm(a, torch.randn(16, 8, device=dev()))
    y.sum().backward()
    assert g.counts["serialized"] == 1 and a.grad is not None


def test_side_regions_refuse_vendor_gemms():
    from deeplearning4j_amd.ops import streams as S
    a = torch.randn(32, 32, device=dev())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with S.side_region(side, "unit-test"):
        b = a * 2                                   # elementwise work on the side stream is fine
        with pytest.raises(S.LibraryGemmOffPrimaryStream, match="unit-test"):
            torch.mm(a, a)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(b, a * 2)
    # the bidirectional wrapper's fork hands out such a region
    from deeplearning4j_amd.nn.layers.recurrent import _side_stream
    ss = _side_stream(a)
    assert ss.active
    with ss.fork(a):
        with pytest.raises(S.LibraryGemmOffPrimaryStream):
            torch.nn.functional.linear(a, a)
    ss.join()
    torch.cuda.synchronize()


def test_device_token_orders_two_streams():
    from deeplearning4j_amd.ops import streams as S
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.zeros(1 << 22, device=dev())
    torch.cuda.synchronize()
    # stream 1 holds the token for a long chain of updates; stream 2's holder must see all of them although nothing else links the streams
    with torch.cuda.stream(s1), S.library_gemm_token(dev()):
        for _ in range(50):
            x.add_(1.0)
    with torch.cuda.stream(s2), S.library_gemm_token(dev()):
        y = x.clone()
    torch.cuda.synchronize()
    assert float(y.min()) == 50.0 and float(y.max()) == 50.0


def test_parallel_inference_workers_on_one_device_follow_the_rule():
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import DenseLayer, OutputLayer
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    from deeplearning4j_amd.parallel.parallel_inference import ParallelInference, InferenceMode
    conf = (NeuralNetConfiguration.Builder().seed(3).weightInit("XAVIER").list()
            .layer(DenseLayer(nIn=66, nOut=128, activation="RELU")).layer(DenseLayer(nIn=128, nOut=128, activation="TANH"))
            .layer(OutputLayer(nIn=128, nOut=10, activation="SOFTMAX", lossFunction="MCXENT")).build())
    net = MultiLayerNetwork(conf, device=dev()).init()                     # fp32 network: its dense layers would call the vendor GEMM
    xs = [torch.randn(n, 66) for n in (8, 16, 8, 16, 8, 16, 8, 16)]
    want = [net.output(x.to(dev())).float().cpu() for x in xs]
    for capture in (False, True):
        pi = ParallelInference(net, workers=3, inferenceMode=InferenceMode.SEQUENTIAL, graphCapture=capture)
        assert pi._shares_device
        got = [None] * len(xs)

        def call(i):
            o = pi.output(xs[i])
            got[i] = (o[0] if isinstance(o, (list, tuple)) else o).float().cpu()

        try:
            for _ in range(3):
                ts = [threading.Thread(target=call, args=(i,)) for i in range(len(xs))]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join(timeout=60)
                    assert not t.is_alive()
                for g, w in zip(got, want):
                    assert torch.allclose(g, w, atol=2e-3, rtol=2e-2), float((g - w).abs().max())
            st = pi.gemmStatistics()
            assert st["serialized"] + st["captured"] > 0, st     # nIn = 66 is outside the in-tree GEMM's operand rules: that vendor GEMM went through the guard
        finally:
            pi.shutdown()
