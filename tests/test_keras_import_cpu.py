"""Keras import: JSON config -> DL4J configuration, weight conversion (HWIO->OIHW, Flatten row permutation, LSTM gate order),
and the pure-python HDF5 archive.  Keras / h5py are not installed, so the Keras side is a plain PyTorch re-statement of the Keras
layer equations on NHWC data (reference tests: KerasModelEndToEndTest, Keras2ModelConfigurationTest; their .h5 fixtures are
downloaded resources that do not ship with the reference -> parity with real Keras files unpinned)."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from deeplearning4j_amd.modelimport import (Hdf5Archive, Hdf5Writer, InvalidKerasConfigurationException, KerasModelImport,
                                            UnsupportedKerasConfigurationException, write_keras_h5)
from deeplearning4j_amd.nn.conf.layers import BatchNormalization, ConvolutionLayer, DenseLayer, OutputLayer, SubsamplingLayer


def _t(v):
    v = v[0] if isinstance(v, (list, tuple)) else v
    return v if isinstance(v, torch.Tensor) else (v._t if hasattr(v, "_t") else torch.as_tensor(v))


def _seq_cnn_config():
    return {"class_name": "Sequential", "keras_version": "2.4.0", "backend": "tensorflow", "config": {"name": "seq", "layers": [
        {"class_name": "Conv2D", "config": {"name": "conv", "batch_input_shape": [None, 8, 8, 3], "filters": 4, "kernel_size": [3, 3], "strides": [1, 1],
                                            "padding": "same", "data_format": "channels_last", "activation": "relu", "use_bias": True,
                                            "kernel_initializer": {"class_name": "GlorotUniform", "config": {}},
                                            "kernel_regularizer": {"class_name": "L1L2", "config": {"l1": 0.0, "l2": 0.01}}}},
        {"class_name": "BatchNormalization", "config": {"name": "bn", "axis": [3], "momentum": 0.9, "epsilon": 0.001, "center": True, "scale": True}},
        {"class_name": "MaxPooling2D", "config": {"name": "pool", "pool_size": [2, 2], "strides": [2, 2], "padding": "valid"}},
        {"class_name": "Dropout", "config": {"name": "drop", "rate": 0.25}},
        {"class_name": "Flatten", "config": {"name": "flat"}},
        {"class_name": "Dense", "config": {"name": "fc", "units": 5, "activation": "softmax", "use_bias": True}}]}}


def _seq_cnn_weights(rng):
    return {"conv": {"kernel": rng.normal(0, 0.3, (3, 3, 3, 4)).astype(np.float32), "bias": rng.normal(0, 0.1, 4).astype(np.float32)},
            "bn": {"gamma": rng.uniform(0.5, 1.5, 4).astype(np.float32), "beta": rng.normal(0, 0.1, 4).astype(np.float32),
                   "moving_mean": rng.normal(0, 0.2, 4).astype(np.float32), "moving_variance": rng.uniform(0.5, 1.5, 4).astype(np.float32)},
            "fc": {"kernel": rng.normal(0, 0.3, (4 * 4 * 4, 5)).astype(np.float32), "bias": rng.normal(0, 0.1, 5).astype(np.float32)}}


def _keras_cnn_forward(x_nhwc, w):
    x = torch.from_numpy(x_nhwc).permute(0, 3, 1, 2)
    k = torch.from_numpy(w["conv"]["kernel"]).permute(3, 2, 0, 1)
    y = torch.relu(F.conv2d(x, k, torch.from_numpy(w["conv"]["bias"]), padding=1))
    b = w["bn"]
    y = F.batch_norm(y, torch.from_numpy(b["moving_mean"]), torch.from_numpy(b["moving_variance"]), torch.from_numpy(b["gamma"]),
                     torch.from_numpy(b["beta"]), False, 0.0, 1e-3)
    y = F.max_pool2d(y, 2)
    flat = y.permute(0, 2, 3, 1).reshape(y.shape[0], -1)                      # Keras flattens NHWC
    return torch.softmax(flat @ torch.from_numpy(w["fc"]["kernel"]) + torch.from_numpy(w["fc"]["bias"]), 1)


def test_sequential_cnn_config_and_weights_match_keras_semantics():
    rng = np.random.default_rng(0)
    cfg, w = _seq_cnn_config(), _seq_cnn_weights(rng)
    conf = KerasModelImport.importKerasSequentialConfiguration(json.dumps(cfg))
    kinds = [type(c.layer) for c in conf.confs]
    assert kinds[:3] == [ConvolutionLayer, BatchNormalization, SubsamplingLayer] and kinds[-1] is OutputLayer
    assert conf.confs[0].layer.l2 == 0.01 and conf.confs[0].layer.nIn == 3 and conf.confs[-1].layer.nIn == 64
    flat = {"%s/%s" % (l, n): a for l, ws in w.items() for n, a in ws.items()}
    net = KerasModelImport.importKerasSequentialModelAndWeights(cfg, flat)
    x = rng.normal(0, 1, (3, 8, 8, 3)).astype(np.float32)
    got = _t(net.output(torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()))
    assert torch.allclose(got, _keras_cnn_forward(x, w), atol=1e-5)


def test_h5_archive_roundtrip_and_import(tmp_path):
    rng = np.random.default_rng(1)
    cfg, w = _seq_cnn_config(), _seq_cnn_weights(rng)
    p = tmp_path / "model.h5"
    write_keras_h5(p, cfg, w)
    ar = Hdf5Archive(p)
    assert json.loads(ar.read_attribute_as_string("model_config"))["class_name"] == "Sequential"
    assert ar.list("/") == ["model_weights"] and ar.list("/model_weights") == ["bn", "conv", "fc"]
    assert [str(s) for s in ar.attrs("/model_weights")["layer_names"]] == ["conv", "bn", "fc"]
    assert [str(s) for s in ar.attrs("/model_weights/conv")["weight_names"]] == ["conv/kernel:0", "conv/bias:0"]
    k = ar.read_dataset("/model_weights/conv/conv/kernel:0")
    assert k.dtype == np.float32 and np.array_equal(k, w["conv"]["kernel"])
    kw = ar.keras_weights()
    assert set(kw) == {"conv", "bn", "fc"} and np.array_equal(kw["fc"]["fc/bias:0"], w["fc"]["bias"])
    net = KerasModelImport.importKerasSequentialModelAndWeights(str(p))
    x = rng.normal(0, 1, (2, 8, 8, 3)).astype(np.float32)
    assert torch.allclose(_t(net.output(torch.from_numpy(x).permute(0, 3, 1, 2).contiguous())), _keras_cnn_forward(x, w), atol=1e-5)
    # many entries in one group (several SNODs), int + double datasets, numeric attributes
    wr = Hdf5Writer()
    for i in range(300):
        wr.create_dataset("/g/d%03d" % i, np.full((2,), i, dtype=np.int32))
    wr.create_dataset("/g/dbl", np.arange(6, dtype=np.float64).reshape(2, 3))
    wr.set_attr("/g", "version", np.array([3, 1, 4], dtype=np.int64))
    wr.set_attr("/g/dbl", "note", "hello")
    wr.write(tmp_path / "many.h5")
    a2 = Hdf5Archive(tmp_path / "many.h5")
    assert len(a2.list("/g")) == 301 and a2.read_dataset("/g/d257").tolist() == [257, 257]
    assert a2.read_dataset("/g/dbl").tolist() == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]
    assert a2.attrs("/g")["version"].tolist() == [3, 1, 4] and a2.attrs("/g/dbl")["note"] == "hello"
    with pytest.raises(KeyError):
        a2.read_dataset("/g/nope")


def test_lstm_gate_order_and_bidirectional():
    rng = np.random.default_rng(2)
    T, Fin, U = 5, 3, 4
    cfg = {"class_name": "Sequential", "config": {"layers": [
        {"class_name": "LSTM", "config": {"name": "lstm", "batch_input_shape": [None, T, Fin], "units": U, "activation": "tanh",
                                          "recurrent_activation": "sigmoid", "return_sequences": True, "use_bias": True}},
        {"class_name": "Dense", "config": {"name": "out", "units": 2, "activation": "linear"}}]}}
    w = {"lstm/kernel": rng.normal(0, 0.5, (Fin, 4 * U)).astype(np.float32), "lstm/recurrent_kernel": rng.normal(0, 0.5, (U, 4 * U)).astype(np.float32),
         "lstm/bias": rng.normal(0, 0.2, 4 * U).astype(np.float32), "out/kernel": rng.normal(0, 0.5, (U, 2)).astype(np.float32),
         "out/bias": rng.normal(0, 0.1, 2).astype(np.float32)}
    net = KerasModelImport.importKerasSequentialModelAndWeights(cfg, w, loss="MSE")
    x = rng.normal(0, 1, (2, T, Fin)).astype(np.float32)
    # Keras LSTM: gates [i, f, c~, o]
    K, R, b = (torch.from_numpy(w[k]) for k in ("lstm/kernel", "lstm/recurrent_kernel", "lstm/bias"))
    h, c = torch.zeros(2, U), torch.zeros(2, U)
    outs = []
    for t in range(T):
        z = torch.from_numpy(x[:, t]) @ K + h @ R + b
        i, f, g, o = z.split(U, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h @ torch.from_numpy(w["out/kernel"]) + torch.from_numpy(w["out/bias"]))
    ref = torch.stack(outs, 2)                                                   # DL4J time series are [mb, size, T]
    got = _t(net.output(torch.from_numpy(x).permute(0, 2, 1).contiguous()))
    assert got.shape == (2, 2, T) and torch.allclose(got, ref, atol=1e-5)


def test_functional_model_with_merges():
    rng = np.random.default_rng(3)
    cfg = {"class_name": "Functional", "config": {"name": "m", "input_layers": [["in", 0, 0]], "output_layers": [["out", 0, 0]], "layers": [
        {"class_name": "InputLayer", "name": "in", "config": {"name": "in", "batch_input_shape": [None, 6]}, "inbound_nodes": []},
        {"class_name": "Dense", "name": "a", "config": {"name": "a", "units": 4, "activation": "relu"}, "inbound_nodes": [[["in", 0, 0, {}]]]},
        {"class_name": "Dense", "name": "b", "config": {"name": "b", "units": 4, "activation": "tanh"}, "inbound_nodes": [[["in", 0, 0, {}]]]},
        {"class_name": "Add", "name": "add", "config": {"name": "add"}, "inbound_nodes": [[["a", 0, 0, {}], ["b", 0, 0, {}]]]},
        {"class_name": "Concatenate", "name": "cat", "config": {"name": "cat", "axis": -1}, "inbound_nodes": [[["add", 0, 0, {}], ["a", 0, 0, {}]]]},
        {"class_name": "Dense", "name": "out", "config": {"name": "out", "units": 3, "activation": "softmax"}, "inbound_nodes": [[["cat", 0, 0, {}]]]}]}}
    w = {l: {"kernel": rng.normal(0, 0.4, s).astype(np.float32), "bias": rng.normal(0, 0.1, s[1]).astype(np.float32)}
         for l, s in (("a", (6, 4)), ("b", (6, 4)), ("out", (8, 3)))}
    net = KerasModelImport.importKerasModelAndWeights(cfg, w)
    assert net.conf.getNetworkOutputs() == ["out"] and isinstance(net.conf.vertices["out"].layer, OutputLayer)
    x = torch.from_numpy(rng.normal(0, 1, (5, 6)).astype(np.float32))
    lin = lambda n, v: v @ torch.from_numpy(w[n]["kernel"]) + torch.from_numpy(w[n]["bias"])
    a, b = torch.relu(lin("a", x)), torch.tanh(lin("b", x))
    ref = torch.softmax(lin("out", torch.cat([a + b, a], 1)), 1)
    assert torch.allclose(_t(net.output(x)), ref, atol=1e-6)


def test_unsupported_and_invalid_configs():
    with pytest.raises(InvalidKerasConfigurationException):
        KerasModelImport.importKerasSequentialConfiguration({"class_name": "Functional", "config": {}})
    bad = {"class_name": "Sequential", "config": {"layers": [{"class_name": "Lambda", "config": {"name": "l", "batch_input_shape": [None, 4]}}]}}
    with pytest.raises(UnsupportedKerasConfigurationException):
        KerasModelImport.importKerasSequentialConfiguration(bad)
    cfg = _seq_cnn_config()
    with pytest.raises(InvalidKerasConfigurationException):
        KerasModelImport.importKerasSequentialModelAndWeights(cfg, {"conv/kernel": np.zeros((3, 3, 3, 4), np.float32)})   # other layers missing


def test_added_layer_mappers_follow_keras_semantics():
    """Mappers added for parity with the reference importer's layer set (layers/convolutional, local, custom, advanced/activations, core):
    2-D models with ZeroPadding / LocallyConnected2D / LRN / ThresholdedReLU / SpaceToDepth / PoolHelper checked against a PyTorch
    restatement of the Keras equations; 3-D family, Keras-1 aliases, Permute, Lambda and Merge checked at the configuration level."""
    from deeplearning4j_amd.modelimport.keras import KerasLayerMapper
    rng = np.random.default_rng(3)
    cfg = {"class_name": "Sequential", "keras_version": "2.4.0", "backend": "tensorflow", "config": {"name": "seq", "layers": [
        {"class_name": "ZeroPadding2D", "config": {"name": "zp", "batch_input_shape": [None, 6, 6, 2], "padding": [[1, 0], [0, 1]]}},
        {"class_name": "LocallyConnected2D", "config": {"name": "lc", "filters": 3, "kernel_size": [2, 2], "strides": [1, 1], "padding": "valid",
                                                        "activation": "linear", "use_bias": False}},
        {"class_name": "ThresholdedReLU", "config": {"name": "tr", "theta": 0.2}},
        {"class_name": "SpaceToDepth", "config": {"name": "s2d", "block_size": 2}},
        {"class_name": "PoolHelper", "config": {"name": "ph"}},
        {"class_name": "Flatten", "config": {"name": "flat"}},
        {"class_name": "Dense", "config": {"name": "fc", "units": 4, "activation": "linear", "use_bias": True}}]}}
    # zp: 7x7x2 -> lc: 6x6x3 -> s2d: 3x3x12 -> ph: 2x2x12 -> 48
    w = {"lc": {"kernel": rng.normal(0, 0.5, (36, 2 * 2 * 2, 3)).astype(np.float32)},
         "fc": {"kernel": rng.normal(0, 0.3, (48, 4)).astype(np.float32), "bias": rng.normal(0, 0.1, 4).astype(np.float32)}}
    net = KerasModelImport.importKerasSequentialModelAndWeights(json.dumps(cfg), w)
    x = rng.normal(0, 1, (3, 6, 6, 2)).astype(np.float32)
    got = _t(net.output(torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()))
    # Keras semantics on NHWC
    xp = F.pad(torch.from_numpy(x), (0, 0, 0, 1, 1, 0))                                           # pad W right by 1, H top by 1
    patches = xp.unfold(1, 2, 1).unfold(2, 2, 1)                                                    # [n, 6, 6, c, kh, kw]
    patches = patches.permute(0, 1, 2, 4, 5, 3).reshape(3, 36, 8)                                   # feature order (kh, kw, c)
    y = torch.einsum("npf,pfo->npo", patches, torch.from_numpy(w["lc"]["kernel"])).reshape(3, 6, 6, 3)
    y = torch.where(y > 0.2, y, torch.zeros_like(y))
    y = y.reshape(3, 3, 2, 3, 2, 3).permute(0, 1, 3, 2, 4, 5).reshape(3, 3, 3, 12)                  # tf.nn.space_to_depth, NHWC
    y = y[:, 1:, 1:, :]
    ref = y.reshape(3, -1) @ torch.from_numpy(w["fc"]["kernel"]) + torch.from_numpy(w["fc"]["bias"])
    assert torch.allclose(got, ref, atol=1e-4), (got - ref).abs().max()

    m = KerasLayerMapper()
    from deeplearning4j_amd.nn.conf import layers as L
    mp, shp = m.map("Conv3D", {"filters": 4, "kernel_size": 3, "padding": "same", "name": "c3"}, (8, 8, 8, 2))
    assert isinstance(mp.layer, L.Convolution3D) and shp == (8, 8, 8, 4)
    k3 = rng.normal(0, 1, (3, 3, 3, 2, 4)).astype(np.float32)
    assert mp.weights({"kernel": k3}, {})["W"].shape == (4, 2, 3, 3, 3)
    assert m.map("Conv3DTranspose", {"filters": 4, "kernel_size": 2, "strides": 2, "padding": "valid"}, (4, 4, 4, 2))[1] == (8, 8, 8, 4)
    assert m.map("ZeroPadding3D", {"padding": [[1, 1], [2, 2], [0, 1]]}, (8, 8, 8, 2))[1] == (10, 12, 9, 2)
    assert m.map("Cropping3D", {"cropping": 1}, (8, 8, 8, 2))[1] == (6, 6, 6, 2) and m.map("UpSampling3D", {"size": 2}, (4, 4, 4, 2))[1] == (8, 8, 8, 2)
    a2, s2 = m.map("AtrousConvolution2D", {"nb_filter": 5, "nb_row": 3, "nb_col": 3, "atrous_rate": [2, 2], "border_mode": "same"}, (9, 9, 2))
    assert tuple(a2.layer.dilation) == (2, 2) and s2 == (9, 9, 5)
    a1, s1 = m.map("AtrousConvolution1D", {"nb_filter": 5, "filter_length": 3, "atrous_rate": 2, "border_mode": "valid"}, (12, 2))
    assert s1 == (8, 5)
    assert isinstance(m.map("LRN", {"alpha": 1e-3, "beta": 0.5, "k": 1, "n": 3}, (4, 4, 8))[0].layer, L.LocalResponseNormalization)
    assert m.map("LocallyConnected1D", {"filters": 3, "kernel_size": 2}, (7, 2))[1] == (6, 3)
    pm, ps = m.map("Permute", {"dims": [2, 1], "name": "p"}, (7, 3))
    assert ps == (3, 7) and pm.layer.fn(torch.zeros(2, 3, 7)).shape == (2, 7, 3)
    with pytest.raises(UnsupportedKerasConfigurationException):
        m.map("Lambda", {"name": "lam"}, (4,))
    KerasLayerMapper.registerLambdaLayer("lam", lambda x: x * 2, lambda s: s)
    try:
        lm, ls = m.map("Lambda", {"name": "lam"}, (4,))
        assert ls == (4,) and float(lm.layer.fn(torch.ones(1))) == 2.0
    finally:
        KerasLayerMapper.clearLambdaLayers()
    assert m.map("Merge", {"mode": "sum", "name": "mg"}, (4,))[0].vertex is not None
    with pytest.raises(UnsupportedKerasConfigurationException):
        m.map("Merge", {"mode": "cos"}, (4,))
