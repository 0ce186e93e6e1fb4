"""StatsListener -> StatsStorage -> UIServer (reference tests: TestStatsListener, TestStatsStorage, TestVertxUI, TestRemoteReceiver)."""
import json
import urllib.error
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
        assert "<svg" in html and "DL4J-AMD Training UI" in html and "Model score vs. iteration" in html
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


def test_train_module_model_and_system_pages_sessions_and_language():
    """TrainModule routes beyond the overview (TrainModule.java getModelGraph / getModelData / getSystemData / sessionInfo / setlang)."""
    from deeplearning4j_amd.ui import DefaultI18N
    ui = UIServer(port=0).start()
    try:
        st = InMemoryStatsStorage()
        ui.attach(st)
        net = _net()
        lst = StatsListener(st, updateConfig=StatsUpdateConfiguration(numHistogramBins=6))
        net.setListeners(lst)
        ds = _data()
        for _ in range(4):
            net.fit(ds)
        base, sid = ui.getAddress(), lst.getSessionID()
        g = _get(base + "/train/model/graph?sessionId=" + sid)
        assert g["vertexNames"][0] == "input" and g["vertexTypes"][1:] == ["DenseLayer", "OutputLayer"] and g["vertexInputs"] == [[], [0], [1]]
        assert g["vertexInfo"][1]["nOut"] == "6"
        d = _get(base + "/train/model/data/1?sessionId=" + sid)
        assert d["layerInfo"]["type"] == "DenseLayer" and sorted(d["paramNames"]) == ["W", "b"]
        assert len(d["meanMag"]["param W"]) == 4 and len(d["meanMag"]["update W"]) == 3
        assert len(d["paramHist"]["W"]["counts"]) == 6 and d["paramHist"]["W"]["max"] > d["paramHist"]["W"]["min"]
        assert len(d["learningRates"]) == 4 and abs(d["learningRates"][0][1] - 0.05) < 1e-9
        assert _get(base + "/train/model/data/0?sessionId=" + sid)["meanMag"] == {}                # the input vertex has no parameters
        info = _get(base + "/train/sessions/info")
        assert info[sid]["numParams"] == net.numParams() and info[sid]["numWorkers"] == 1 and info[sid]["lastUpdate"] > 0
        assert _get(base + "/train/sessions/current")["sessionId"] == sid
        assert _get(base + "/train/workers/" + sid) == st.listWorkerIDsForSession(sid)
        sysd = _get(base + "/train/system/data?sessionId=" + sid)
        assert len(sysd["memory"]) == 4 and "hostMaxRssKb" in sysd["memory"][0][1]
        for page in ("/train/model", "/train/system", "/tsne", "/activations", "/samediff"):
            html = urllib.request.urlopen(base + page, timeout=5).read().decode()
            assert "<nav>" in html and "</script>" in html and "%%" not in html and "%(" not in html
        # "/" and "/train" redirect to the overview
        assert urllib.request.urlopen(base + "/", timeout=5).geturl().endswith("/train/overview")
        # language switch changes the rendered navigation, unknown keys fall back to English, unknown languages are refused
        assert _get(base + "/setlang/de") == {"language": "de"}
        html = urllib.request.urlopen(base + "/train/overview", timeout=5).read().decode()
        assert "Übersicht" in html and "Modell-Score pro Iteration" in html and "Standard deviations (log10)" in html
        i18n = DefaultI18N.getInstance()
        assert i18n.getMessage("train.nav.model") == "Modell" and i18n.getMessage("ja", "train.nav.model") == "モデル" and i18n.getMessage("no.such.key") == "no.such.key"
        try:
            urllib.request.urlopen(base + "/setlang/xx", timeout=5)
            raise AssertionError("unknown language accepted")
        except urllib.error.HTTPError as e:
            assert e.code == 500
        _get(base + "/setlang/en")
        assert _get(base + "/nope/route") is None
    except urllib.error.HTTPError as e:
        assert e.code == 404 and "/nope/route" in e.url
    finally:
        ui.stop()


def test_tsne_module_upload_and_coords():
    ui = UIServer(port=0).start()
    try:
        base = ui.getAddress()
        assert _get(base + "/tsne/sessions") == []
        req = urllib.request.Request(base + "/tsne/post/mySession", data=b"0.5,1.5,cat\n-2,3,dog\n\n", method="POST")
        assert json.loads(urllib.request.urlopen(req, timeout=5).read())["rows"] == 2
        req = urllib.request.Request(base + "/tsne/upload", data=b"1,2\n", method="POST")
        urllib.request.urlopen(req, timeout=5)
        assert _get(base + "/tsne/sessions") == ["UploadedFile", "mySession"]
        assert _get(base + "/tsne/coords/mySession") == ["0.5,1.5,cat", "-2,3,dog"]
        ui.tsne.setCoordinates("prog", [[1.0, 2.0], [3.0, 4.5]], labels=["a", "b"])
        assert _get(base + "/tsne/coords/prog") == ["1,2,a", "3,4.5,b"]
    finally:
        ui.stop()


def test_convolutional_listener_posts_activation_grid_png():
    from deeplearning4j_amd.nn.conf.layers import ConvolutionLayer, SubsamplingLayer
    from deeplearning4j_amd.ui import ConvolutionalIterationListener
    from deeplearning4j_amd.ui.weights import TYPE_ID as CONV_TYPE, decode_png_gray, encode_png_gray, tile_activations
    img = (torch.arange(6 * 9).reshape(6, 9) * 3 % 256).to(torch.uint8)
    assert torch.equal(decode_png_gray(encode_png_gray(img)), img)
    t = tile_activations([torch.rand(3, 4, 5), torch.rand(2, 2, 2)])
    assert t.shape == (6 + 4, 3 * 7) and t.dtype == torch.uint8 and int(t.max()) == 255
    conf = (NeuralNetConfiguration.Builder().seed(3).updater(Adam(0.01)).list()
            .layer(0, ConvolutionLayer(nOut=4, kernelSize=(3, 3), activation="RELU"))
            .layer(1, SubsamplingLayer(kernelSize=(2, 2), stride=(2, 2)))
            .layer(2, OutputLayer(nOut=2, activation="SOFTMAX", lossFunction="MCXENT"))
            .setInputType(InputType.convolutional(8, 8, 1)).build())
    net = MultiLayerNetwork(conf)
    net.init()
    g = torch.Generator().manual_seed(0)
    x = torch.rand(6, 1, 8, 8, generator=g)
    y = torch.eye(2)[torch.randint(0, 2, (6,), generator=g)]
    ui = UIServer(port=0).start()
    try:
        st = InMemoryStatsStorage()
        ui.attach(st)
        lst = ConvolutionalIterationListener(st, frequency=2, probeInput=x)
        net.setListeners(lst)
        for _ in range(4):
            net.fit(DataSet(x, y))
        base = ui.getAddress()
        info = _get(base + "/activations/info")
        assert info["iteration"] is not None and len(info["layers"]) >= 2
        assert st.getNumUpdateRecordsFor(lst.getSessionID(), CONV_TYPE) == 2
        png = urllib.request.urlopen(base + "/activations/data", timeout=5).read()
        im = decode_png_gray(png)
        assert im.shape == (info["height"], info["width"]) and int(im.max()) == 255
        assert _get(base + "/train/sessions") == []                      # activation records are not training sessions
    finally:
        ui.stop()


def test_samediff_module_and_components_round_trip(tmp_path):
    from deeplearning4j_amd.autodiff.samediff import SameDiff
    from deeplearning4j_amd.ui import components as C
    sd = SameDiff.create()
    a = sd.placeHolder("in", "FLOAT", -1, 3)
    w = sd.var("w", torch.ones(3, 2))
    sd.nn.softmax(a.mmul(w)).rename("out") if hasattr(sd.nn.softmax(a.mmul(w)), "rename") else None
    path = str(tmp_path / "g.fb")
    sd.save(path)
    ui = UIServer(port=0).start()
    try:
        g = _get(ui.getAddress() + "/samediff/graph?file=" + urllib.request.quote(path))
        assert "error" not in g and {"in", "w"} <= {v["name"] for v in g["variables"]} and any(o["op"] in ("matmul", "mmul") for o in g["ops"])
        assert "error" in _get(ui.getAddress() + "/samediff/graph?file=/no/such/file.fb")
    finally:
        ui.stop()
    line = C.ChartLine.Builder("score", C.StyleChart(width=640, height=300, seriesColors=["#f00"])).addSeries("s", [0, 1, 2], [3.0, 2.0, 1.5]).showLegend(True).build()
    hist = C.ChartHistogram.Builder("h").addBin(0, 1, 5).addBin(1, 2, 7).build()
    table = C.ComponentTable.Builder(C.StyleTable(borderWidthPx=1)).header("k", "v").content([["a", 1], ["b", 2]]).build()
    acc = C.DecoratorAccordion.Builder().title("more").addComponents(C.ComponentText.Builder("hello", C.StyleText(fontSize=12)).build(),
                                                                      C.ChartHorizontalBar.Builder("bars").addValue("x", 1).addValue("y", -2).build()).build()
    area = C.ChartStackedArea.Builder("mem").setXValues([0, 1, 2]).addSeries("a", [1, 2, 3]).addSeries("b", [2, 2, 2]).build()
    tl = C.ChartTimeline.Builder("phases").addLane("worker0", [("fit", 0, 120, "#0a0"), ("eval", 120, 150)]).build()
    sc = C.ChartScatter.Builder("pts").addSeries("p", [0, 1], [1, 0]).build()
    div = C.ComponentDiv(C.StyleDiv(width=100, widthUnit=C.LengthUnit.Percent), [line, hist])
    for c in (line, hist, table, acc, area, tl, sc, div):
        back = C.Component.fromJson(c.toJson())
        assert back == c and type(back) is type(c)
    assert list(json.loads(line.toJson())) == ["ChartLine"] and json.loads(line.toJson())["ChartLine"]["style"]["StyleChart"]["width"] == 640
    try:
        C.ChartLine.Builder("bad").addSeries("s", [0, 1], [1])
        raise AssertionError("length mismatch accepted")
    except ValueError:
        pass
    page = C.StaticPageUtil.renderHTML(line, [table, acc], area, tl, sc, div, title="report <1>")
    assert page.startswith("<!doctype html>") and "report &lt;1&gt;" in page and "\"ChartTimeline\"" in page and "function render" in page
    C.StaticPageUtil.saveHTMLFile(tmp_path / "r.html", line)
    assert (tmp_path / "r.html").read_text().count("ChartLine") >= 2
