"""ModelSerializer: DL4J model zip (configuration.json + coefficients.bin + updaterState.bin +
normalizer.bin [+ noParams.marker, objects/<key>]).

Reference: deeplearning4j-nn/.../util/ModelSerializer.java:61-66,140-191 (write), :251-330,476-560
(restore; tries the other network type on mismatch). Binary entries use the Nd4j.write encoding
(nd4j/serde.py). ModelGuesser: util/ModelGuesser.java."""
from __future__ import annotations

import io
import json
import zipfile

import torch

from ..nd4j import serde

CONFIGURATION_JSON = "configuration.json"
COEFFICIENTS_BIN = "coefficients.bin"
UPDATER_BIN = "updaterState.bin"
NORMALIZER_BIN = "normalizer.bin"
NO_PARAMS_MARKER = "noParams.marker"
PREPROCESSOR_BIN = "preprocessor.bin"


class ModelSerializer:
    @staticmethod
    def writeModel(model, path, saveUpdater=True, dataNormalization=None):
        if not model._initialized:
            model.init()
        model.conf.iterationCount, model.conf.epochCount = model.iterationCount, model.epochCount
        target = path if hasattr(path, "write") else str(path)
        with zipfile.ZipFile(target, "w", zipfile.ZIP_DEFLATED) as z:
            z.writestr(CONFIGURATION_JSON, model.conf.toJson())
            if model.numParams() > 0:
                z.writestr(COEFFICIENTS_BIN, serde.to_bytes(model.params()))
            else:
                z.writestr(NO_PARAMS_MARKER, b"")
            st = model.getUpdaterState() if saveUpdater else None
            if st is not None and st.numel() > 0:
                z.writestr(UPDATER_BIN, serde.to_bytes(st))
            if dataNormalization is not None:
                from ..datasets.normalizers import NormalizerSerializer
                b = io.BytesIO()
                NormalizerSerializer.getDefault().write(dataNormalization, b)
                z.writestr(NORMALIZER_BIN, b.getvalue())

    @staticmethod
    def _read(path):
        with zipfile.ZipFile(path if hasattr(path, "read") else str(path), "r") as z:
            names = set(z.namelist())
            conf = z.read(CONFIGURATION_JSON).decode("utf-8")
            params = serde.from_bytes(z.read(COEFFICIENTS_BIN)) if COEFFICIENTS_BIN in names else None
            upd = serde.from_bytes(z.read(UPDATER_BIN)) if UPDATER_BIN in names else None
            norm = z.read(NORMALIZER_BIN) if NORMALIZER_BIN in names else None
        return conf, params, upd, norm

    @staticmethod
    def restoreMultiLayerNetwork(path, loadUpdater=True, device=None):
        from ..nn.conf.configuration import MultiLayerConfiguration
        from ..nn.multilayer import MultiLayerNetwork
        conf_json, params, upd, _ = ModelSerializer._read(path)
        d = json.loads(conf_json)
        if "confs" not in d:
            # ModelSerializer.java:251-330 tries the other configuration type before giving up
            if "vertices" in d:
                return ModelSerializer.restoreComputationGraph(path, loadUpdater, device=device)
            raise ValueError("configuration.json holds neither a MultiLayerConfiguration nor a ComputationGraphConfiguration")
        conf = MultiLayerConfiguration.fromJson(d)
        net = MultiLayerNetwork(conf, device=device)
        net.init(params, True)
        if loadUpdater and upd is not None:
            net.setUpdaterState(upd)
        return net

    @staticmethod
    def restoreComputationGraph(path, loadUpdater=True, device=None):
        from ..nn.conf.configuration import ComputationGraphConfiguration
        from ..nn.graph.computation_graph import ComputationGraph
        conf_json, params, upd, _ = ModelSerializer._read(path)
        d = json.loads(conf_json)
        if "vertices" not in d:
            if "confs" in d:
                return ModelSerializer.restoreMultiLayerNetwork(path, loadUpdater, device=device)
            raise ValueError("configuration.json holds neither a MultiLayerConfiguration nor a ComputationGraphConfiguration")
        conf = ComputationGraphConfiguration.fromJson(conf_json)
        net = ComputationGraph(conf, device=device)
        net.init(params, True)
        if loadUpdater and upd is not None:
            net.setUpdaterState(upd)
        return net

    @staticmethod
    def restoreNormalizerFromFile(path):
        from ..datasets.normalizers import NormalizerSerializer
        _, _, _, norm = ModelSerializer._read(path)
        if norm is None:
            return None
        return NormalizerSerializer.getDefault().restore(io.BytesIO(norm))

    @staticmethod
    def addNormalizerToModel(path, normalizer):
        from ..datasets.normalizers import NormalizerSerializer
        b = io.BytesIO()
        NormalizerSerializer.getDefault().write(normalizer, b)
        with zipfile.ZipFile(str(path), "a") as z:
            z.writestr(NORMALIZER_BIN, b.getvalue())

    @staticmethod
    def addObjectToFile(path, key, obj):
        with zipfile.ZipFile(str(path), "a") as z:
            z.writestr("objects/" + key, json.dumps(obj))

    @staticmethod
    def getObjectFromFile(path, key):
        with zipfile.ZipFile(str(path), "r") as z:
            return json.loads(z.read("objects/" + key).decode("utf-8"))

    @staticmethod
    def listObjectsInFile(path):
        with zipfile.ZipFile(str(path), "r") as z:
            return [n[len("objects/"):] for n in z.namelist() if n.startswith("objects/")]


class ModelGuesser:
    @staticmethod
    def loadModelGuess(path, device=None):
        with zipfile.ZipFile(str(path), "r") as z:
            d = json.loads(z.read(CONFIGURATION_JSON).decode("utf-8"))
        if "confs" in d:
            return ModelSerializer.restoreMultiLayerNetwork(path, True, device)
        return ModelSerializer.restoreComputationGraph(path, True, device)

    @staticmethod
    def loadConfigGuess(path):
        from ..nn.conf.configuration import MultiLayerConfiguration, ComputationGraphConfiguration
        with open(str(path)) as f:
            s = f.read()
        d = json.loads(s)
        return MultiLayerConfiguration.fromJson(d) if "confs" in d else ComputationGraphConfiguration.fromJson(s)

    @staticmethod
    def loadNormalizer(path):
        return ModelSerializer.restoreNormalizerFromFile(path)
