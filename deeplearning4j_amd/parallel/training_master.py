"""TrainingMaster façades: the upstream Spark entry points mapped onto the single-node RCCL engine.

The reference snapshot carries no Spark / parameter-server module (SURVEY.md Appendix D records the upstream API:
``SharedTrainingMaster``, ``ParameterAveragingTrainingMaster``, ``SparkDl4jMultiLayer``, ``SparkComputationGraph``).
Upstream a TrainingMaster decides how an RDD of DataSets is split over executors and how replicas exchange updates
(Aeron UDP mesh with threshold-encoded gradients, or periodic parameter averaging through the Spark driver). On an
MI355X node "the cluster" is the 8 GPUs of the node, one process per GPU, and the exchange is a dense bucketed RCCL
all-reduce over xGMI (parallel/comm.py) — threshold encoding only pays on slow links and is accepted but not applied
unless ``encodingDebugMode`` asks for it. Both masters configure a :class:`ParallelWrapper`:

* SharedTrainingMaster          -> TrainingMode.SHARED_GRADIENTS (all-reduce of the flat gradient every iteration)
* ParameterAveragingTrainingMaster -> TrainingMode.AVERAGING (all-reduce of parameters every ``averagingFrequency`` iterations)

``SparkDl4jMultiLayer(sc, network_or_conf, trainingMaster).fit(data)`` takes a DataSetIterator or a list of DataSets
(the "RDD"); ``sc`` is ignored. Launch with ``python -m torch.distributed.run --nproc-per-node N`` for N GPUs.
"""
from __future__ import annotations

from typing import Optional

from .parallel_wrapper import ParallelWrapper, TrainingMode


class TrainingMaster:
    trainingMode = TrainingMode.SHARED_GRADIENTS

    def __init__(self, **kw):
        self.batchSizePerWorker = kw.get("batchSizePerWorker", 16)
        self.workersPerNode = kw.get("workersPerNode", -1)
        self.averagingFrequency = kw.get("averagingFrequency", 1)
        self.thresholdAlgorithm = kw.get("thresholdAlgorithm")
        self.residualPostProcessor = kw.get("residualPostProcessor")
        self.collectTrainingStats = kw.get("collectTrainingStats", False)
        self.saveUpdater = kw.get("saveUpdater", True)
        self.workerPrefetchNumBatches = kw.get("workerPrefetchNumBatches", 2)
        self.encodingDebugMode = kw.get("encodingDebugMode", False)
        self.rddDataSetNumExamples = kw.get("rddDataSetNumExamples", 1)
        self.ignoredOptions = dict(kw.get("ignoredOptions", {}))
        self._stats = []

    def wrapper(self, model) -> ParallelWrapper:
        b = (ParallelWrapper.Builder(model).trainingMode(self.trainingMode).averagingFrequency(self.averagingFrequency)
             .averageUpdaters(self.saveUpdater).prefetchBuffer(self.workerPrefetchNumBatches))
        if self.thresholdAlgorithm is not None:
            b.thresholdAlgorithm(self.thresholdAlgorithm)
        if self.residualPostProcessor is not None:
            b.residualPostProcessor(self.residualPostProcessor)
        if self.workersPerNode and self.workersPerNode > 0:
            b.workers(self.workersPerNode)
        return b.build()

    def setCollectTrainingStats(self, b):
        self.collectTrainingStats = bool(b)

    def getTrainingStats(self):
        return list(self._stats)


class _MasterBuilder:
    _cls = TrainingMaster

    def __init__(self, *args, **kw):
        # upstream: Builder(rddDataSetNumExamples) | Builder(thresholdAlgorithm, rddDataSetNumExamples) | Builder(voidConfiguration, n)
        self._kw = dict(kw)
        for a in args:
            if isinstance(a, int):
                self._kw["rddDataSetNumExamples"] = a
            elif hasattr(a, "calculateThreshold"):
                self._kw["thresholdAlgorithm"] = a

    # options that change behaviour here
    _APPLIED = ("batchSizePerWorker", "workersPerNode", "averagingFrequency", "thresholdAlgorithm", "residualPostProcessor",
                "collectTrainingStats", "saveUpdater", "workerPrefetchNumBatches", "encodingDebugMode",
                "rddDataSetNumExamples")
    # upstream options that are meaningless without Spark / Aeron (storage levels, repartitioning, mesh transport):
    # accepted so upstream code builds unchanged, recorded in ``ignoredOptions`` and otherwise without effect
    _ACCEPTED = ("rddTrainingApproach", "exportDirectory", "storageLevel", "storageLevelStreams", "repartionData",
                 "repartitionStrategy", "repartitioner", "aggregationDepth", "rngSeed", "unicastPort", "networkMask",
                 "controllerAddress", "transport", "meshBuildMode", "workerTogglePeriodicGC", "workerPeriodicGCFrequency",
                 "maxChunkSize", "debugLongerIterations", "shakeFrequency", "voidConfiguration", "trainingHooks",
                 "updatesThreshold", "thresholdStep", "minThreshold", "stepTrigger", "stepDelay", "numWorkers")

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        if name not in self._APPLIED and name not in self._ACCEPTED:
            raise AttributeError("%s.Builder has no option '%s' (applied: %s; accepted without effect: %s)"
                                 % (self._cls.__name__, name, ", ".join(self._APPLIED), ", ".join(self._ACCEPTED)))

        def setter(*v):
            val = v[0] if len(v) == 1 else (v or True)
            if name in self._APPLIED:
                self._kw[name] = val
            else:
                self._kw.setdefault("ignoredOptions", {})[name] = val
            return self
        return setter

    def build(self):
        return self._cls(**self._kw)


class SharedTrainingMaster(TrainingMaster):
    """Gradient sharing: upstream's threshold-encoded Aeron mesh, here a dense RCCL all-reduce each iteration."""
    trainingMode = TrainingMode.SHARED_GRADIENTS

    class Builder(_MasterBuilder):
        pass


class ParameterAveragingTrainingMaster(TrainingMaster):
    """Periodic parameter (and updater-state) averaging every ``averagingFrequency`` minibatches."""
    trainingMode = TrainingMode.AVERAGING

    class Builder(_MasterBuilder):
        pass


SharedTrainingMaster.Builder._cls = SharedTrainingMaster
ParameterAveragingTrainingMaster.Builder._cls = ParameterAveragingTrainingMaster


class _ListIterator:
    def __init__(self, items):
        self._items, self._k = list(items), 0

    def hasNext(self):
        return self._k < len(self._items)

    def next(self):
        self._k += 1
        return self._items[self._k - 1]

    def reset(self):
        self._k = 0

    def resetSupported(self):
        return True


class _SparkFacade:
    _net_cls = None

    def __init__(self, sparkContext, network, trainingMaster: Optional[TrainingMaster] = None):
        from ..nn.graph.computation_graph import ComputationGraph
        from ..nn.multilayer import MultiLayerNetwork
        if not isinstance(network, (MultiLayerNetwork, ComputationGraph)):          # a configuration: build the network
            network = self._make(network)
        self.network = network
        self.trainingMaster = trainingMaster or SharedTrainingMaster()
        self._wrapper = None

    def _make(self, conf):
        raise NotImplementedError

    def fit(self, data, numEpochs: int = 1):
        if self._wrapper is None:
            self._wrapper = self.trainingMaster.wrapper(self.network)
        it = data if hasattr(data, "hasNext") else _ListIterator(data)
        import time
        t0 = time.time()
        self._wrapper.fit(it, numEpochs)
        if self.trainingMaster.collectTrainingStats:
            self.trainingMaster._stats.append({"fitTimeMs": int((time.time() - t0) * 1000), "epochs": numEpochs,
                                               "score": float(self.network.score())})
        return self.network

    def getNetwork(self):
        return self.network

    def getTrainingMaster(self):
        return self.trainingMaster

    def getScore(self):
        return float(self.network.score())

    def close(self):
        if self._wrapper is not None:
            self._wrapper.shutdown()
            self._wrapper = None


class SparkDl4jMultiLayer(_SparkFacade):
    def _make(self, conf):
        from ..nn.multilayer import MultiLayerNetwork
        net = MultiLayerNetwork(conf)
        net.init()
        return net


class SparkComputationGraph(_SparkFacade):
    def _make(self, conf):
        from ..nn.graph.computation_graph import ComputationGraph
        net = ComputationGraph(conf)
        net.init()
        return net
