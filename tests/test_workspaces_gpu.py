"""Workspaces are live in the networks: an eager GPU iteration runs inside LayerWorkspaceMgr scopes whose workspaces are private
allocator pools, reused cycle after cycle (MultiLayerNetwork.java:1688-1705, ADR 0024)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(ws_mode="ENABLED"):
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import DenseLayer, OutputLayer
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    from deeplearning4j_amd.nd4j.learning import Adam
    conf = (NeuralNetConfiguration.Builder().seed(3).updater(Adam(1e-3)).trainingWorkspaceMode(ws_mode).inferenceWorkspaceMode(ws_mode)
            .list().layer(DenseLayer.Builder().nIn(64).nOut(128).activation("relu").build())
            .layer(DenseLayer.Builder().nIn(128).nOut(128).activation("tanh").build())
            .layer(OutputLayer.Builder().nIn(128).nOut(8).activation("softmax").lossFunction("MCXENT").build()).build())
    net = MultiLayerNetwork(conf, device="cuda")
    net.init()
    return net


def test_training_iterations_allocate_from_the_workspace_pool_and_reuse_it():
    from deeplearning4j_amd.nd4j.workspace import ArrayType, DevicePoolWorkspace, LayerWorkspaceMgr, getWorkspaceManager
    net = _net()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(256, 64, generator=g).cuda()
    y = torch.eye(8)[torch.randint(0, 8, (256,), generator=g)].cuda()
    net.fit(x, y)
    mgr = net.getLayerWorkspaceMgr(True)
    assert mgr is not None and mgr.getWorkspaceName(ArrayType.ACTIVATIONS) == LayerWorkspaceMgr.WS_ALL_LAYERS_ACT
    ws = getWorkspaceManager().getWorkspaceForCurrentThread(wsId=LayerWorkspaceMgr.WS_ALL_LAYERS_ACT)
    assert isinstance(ws, DevicePoolWorkspace) and not ws.isScopeActive()
    c0 = ws.getCyclesCount()              # the named workspace is per thread: other networks of this process cycled it too
    for _ in range(2):
        net.fit(x, y)
    size2, cycles2 = ws.getCurrentSize(), ws.getCyclesCount()
    assert size2 > 0 and cycles2 == c0 + 2
    before = torch.cuda.memory_reserved()
    for _ in range(5):
        net.fit(x, y)
    assert ws.getCurrentSize() == size2 and ws.getCyclesCount() == c0 + 7          # cyclic reuse: the pool does not grow
    assert torch.cuda.memory_reserved() == before
    # an allocation made while the scope is open lands in the pool, one made outside does not
    with mgr.notifyScopeEntered(ArrayType.ACTIVATIONS):
        assert mgr.isWorkspaceOpen(ArrayType.ACTIVATIONS)
        # larger than the whole pool (other networks of this process grew the per-thread workspace too), so no cached block can serve it
        t = torch.empty(size2 + (64 << 20), dtype=torch.uint8, device="cuda")
        assert ws.getCurrentSize() >= size2 + (64 << 20)
    del t
    out = net.output(x)                                                         # inference scope: result outlives the scope
    assert out.shape == (256, 8) and torch.isfinite(out).all()
    assert net.getLayerWorkspaceMgr(False) is not None


def test_workspace_mode_none_uses_no_manager_and_matches_numerically():
    a, b = _net("ENABLED"), _net("NONE")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 64, generator=g).cuda()
    y = torch.eye(8)[torch.randint(0, 8, (64,), generator=g)].cuda()
    for _ in range(3):
        a.fit(x, y)
        b.fit(x, y)
    assert b.getLayerWorkspaceMgr(True) is None
    assert torch.equal(a.params(), b.params())
