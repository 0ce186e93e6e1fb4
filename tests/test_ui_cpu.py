"""StatsListener -> StatsStorage -> UIServer (reference tests: TestStatsListener, TestStatsStorage, TestVertxUI, TestRemoteReceiver)."""
import json
import urllib.request

import torch

from deeplearning4j_amd.datasets.dataset import DataSet
from deeplearning4j_amd.nd4j.learning import Adam
from deeplearning4j_amd.nn.conf.base import InputType
from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
from deeplearning4j_amd.nn.conf.layers import DenseLayer, OutputLayer
from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
from deeplearning4j_amd.ui import (FileStatsStorage, InMemoryStatsStorage, RemoteUIStatsStorageRouter, StatsListener, StatsStorageEvent,
                                   StatsUpdateConfiguration, UIServer)
from deeplearning4j_amd.ui.stats import TYPE_ID


def _net():
    conf = (NeuralNetConfiguration.Builder().seed(1).updater(Adam(0.05)).activation("TANH").list()
            .layer(0, DenseLayer(nIn=4, nOut=6)).layer(1, OutputLayer(nOut=3, activation="SOFTMAX", lossFunction="MCXENT"))
            .setInputType(InputType.feedForward(4)).build())
    net = MultiLayerNetwork(conf)
    net.init()
    return net


def _data():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(24, 4, generator=g)
    y = torch.zeros(24, 3)
    y[torch.arange(24), x[:, :3].argmax(1)] = 1
    return DataSet(x, y)


def _get(url):
    return json.loads(urllib.request.urlopen(url, timeout=5).read().decode())


def test_stats_listener_storage_and_events(tmp_path):
    st = InMemoryStatsStorage()
    events = []
    st.registerStatsStorageListener(lambda e: events.append(e.eventType))
    net = _net()
    lst = StatsListener(st, updateConfig=StatsUpdateConfiguration(numHistogramBins=8))
    net.setListeners(lst)
    ds = _data()
    for _ in range(5):
        net.fit(ds)
    sid = lst.getSessionID()
    assert st.listSessionIDs() == [sid] and st.listTypeIDsForSession(sid) == [TYPE_ID]
    w = st.listWorkerIDsForSession(sid)[0]
    init = st.getStaticInfo(sid, TYPE_ID, w)
    assert init["model"]["numParams"] == net.numParams() and init["model"]["modelClass"] == "MultiLayerNetwork"
    assert init["model"]["paramNames"] == ["0_W", "0_b", "1_W", "1_b"]
    assert st.getNumUpdateRecordsFor(sid) == 5
    ups = st.getAllUpdatesAfter(sid, TYPE_ID, w, -1)
    assert [u.getIterationCount() for u in ups] == sorted(u.getIterationCount() for u in ups)
    last = st.getLatestUpdate(sid, TYPE_ID, w)
    assert abs(last.getScore() - float(net.score())) < 1e-6
    p = last["parameters"]["0_W"]
    W = net.paramTable()["0_W"]
    assert abs(p["meanMagnitude"] - float(W.abs().mean())) < 1e-6 and abs(p["stdev"] - float(W.std(unbiased=False))) < 1e-6
    assert len(p["hist"]) == 8 and sum(p["hist"]) == W.numel()
    assert "0_W" in last["gradients"] and "0_W" in last["updates"] and last["updates"]["0_W"]["meanMagnitude"] > 0
    assert "learningRates" in last.content and "performance" in last.content
    assert events[:3] == [StatsStorageEvent.NewSessionID, StatsStorageEvent.NewWorkerID, StatsStorageEvent.PostMetaData]
    assert events.count(StatsStorageEvent.PostUpdate) == 5 and events.count(StatsStorageEvent.PostStaticInfo) == 1
    # file storage persists and reloads
    f = tmp_path / "stats.db"
    fs = FileStatsStorage(f)
    net2 = _net()
    l2 = StatsListener(fs, listenerFrequency=2)
    net2.setListeners(l2)
    for _ in range(4):
        net2.fit(ds)
    n_rec = fs.getNumUpdateRecordsFor(l2.getSessionID())
    assert n_rec == 2
    fs.close()
    again = FileStatsStorage(f)
    assert again.listSessionIDs() == [l2.getSessionID()] and again.getNumUpdateRecordsFor(l2.getSessionID()) == n_rec
    assert again.getAllStaticInfos(l2.getSessionID(), TYPE_ID)[0]["model"]["numParams"] == net2.numParams()
    again.close()


def test_ui_server_routes_and_remote_receiver():
    ui = UIServer(port=0).start()
    try:
        st = InMemoryStatsStorage()
        ui.attach(st)
        assert ui.isAttached(st)
        net = _net()
        lst = StatsListener(st)
        net.setListeners(lst)
        ds = _data()
        for _ in range(6):
            net.fit(ds)
        base = ui.getAddress()
        assert _get(base + "/train/sessions") == [lst.getSessionID()]
        ov = _get(base + "/train/overview/data?sessionId=" + lst.getSessionID())
        assert len(ov["scores"]) == 6 and ov["model"]["class"] == "MultiLayerNetwork" and "0_W" in ov["updateRatios"]
        assert ov["scores"][0][1] > ov["scores"][-1][1]              # training reduces the score
        md = _get(base + "/train/model/data?sessionId=%s&param=1_W" % lst.getSessionID())
        assert md["paramNames"] == ["0_W", "0_b", "1_W", "1_b"] and "hist" in md["latest"]["1_W"]["parameters"]
        assert "hardware" in _get(base + "/train/system/data")
        html = urllib.request.urlopen(base + "/train/overview", timeout=5).read().decode()
        assert "<svg" in html and "Training overview" in html
        # remote: a second process would post to /remoteReceive
        ui.enableRemoteListener()
        router = RemoteUIStatsStorageRouter(base)
        net2 = _net()
        l2 = StatsListener(router)
        net2.setListeners(l2)
        for _ in range(3):
            net2.fit(ds)
        router.flush()
        router.shutdown()
        assert l2.getSessionID() in _get(base + "/train/sessions")
        assert len(_get(base + "/train/overview/data?sessionId=" + l2.getSessionID())["scores"]) == 3
        ui.detach(st)
        assert lst.getSessionID() not in _get(base + "/train/sessions")
    finally:
        ui.stop()
