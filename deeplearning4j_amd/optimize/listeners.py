"""TrainingListeners (reference: deeplearning4j-nn/.../optimize/listeners/*.java, optimize/api/TrainingListener.java).

The iteration is stream-asynchronous, so listeners that need the score force the one device->host
read themselves; PerformanceListener measures over its reporting window with a device sync only at
report time (the reference times a single iteration with 1 ms resolution — PerformanceListener.java:94-118)."""
from __future__ import annotations

import logging
import os
import time
from typing import List

import torch

log = logging.getLogger("deeplearning4j_amd")


class TrainingListener:
    def iterationDone(self, model, iteration, epoch):
        pass

    def onEpochStart(self, model):
        pass

    def onEpochEnd(self, model):
        pass

    def onForwardPass(self, model, activations=None):
        pass

    def onGradientCalculation(self, model):
        pass

    def onBackwardPass(self, model):
        pass


BaseTrainingListener = TrainingListener


class ScoreIterationListener(TrainingListener):
    def __init__(self, printIterations=10):
        self.printIterations = max(1, int(printIterations))

    def iterationDone(self, model, iteration, epoch):
        if iteration % self.printIterations == 0:
            log.info("Score at iteration %d is %s", iteration, model.score())
            print("Score at iteration %d is %s" % (iteration, model.score()))


class CollectScoresListener(TrainingListener):
    def __init__(self, frequency=1):
        self.frequency = max(1, frequency)
        self.listIteration: List[int] = []
        self.listScore: List[float] = []

    def iterationDone(self, model, iteration, epoch):
        if iteration % self.frequency == 0:
            self.listIteration.append(iteration)
            self.listScore.append(model.score())


class CollectScoresIterationListener(CollectScoresListener):
    def getScoreVsIter(self):
        return list(zip(self.listIteration, self.listScore))

    def exportScores(self, path, delimiter=","):
        with open(path, "w") as f:
            f.write("Iteration%sScore\n" % delimiter)
            for i, s in zip(self.listIteration, self.listScore):
                f.write("%d%s%r\n" % (i, delimiter, s))


class PerformanceListener(TrainingListener):
    """samples/sec, batches/sec, iteration time, ETL time over a reporting window."""

    def __init__(self, frequency=1, reportScore=False, reportGC=False, printer=print):
        self.frequency, self.reportScore = max(1, int(frequency)), reportScore
        self._t0 = None
        self._samples = 0
        self._iters = 0
        self.printer = printer
        self.lastSamplesPerSec = float("nan")
        self.lastBatchesPerSec = float("nan")
        self.lastIterationMs = float("nan")

    def iterationDone(self, model, iteration, epoch):
        if self._t0 is None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self._t0 = time.perf_counter()
            return
        self._samples += model.batchSize()
        self._iters += 1
        if self._iters % self.frequency == 0:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dt = time.perf_counter() - self._t0
            self.lastSamplesPerSec = self._samples / dt
            self.lastBatchesPerSec = self._iters / dt
            self.lastIterationMs = 1000.0 * dt / self._iters
            msg = "ETL: %d ms; iteration %d; iteration time: %.3f ms; samples/sec: %.3f; batches/sec: %.3f;" % (
                getattr(model, "lastEtlTime", 0), iteration, self.lastIterationMs, self.lastSamplesPerSec, self.lastBatchesPerSec)
            if self.reportScore:
                msg += " score: %s;" % model.score()
            if self.printer:
                self.printer(msg)
            self._samples = self._iters = 0
            self._t0 = time.perf_counter()


class TimeIterationListener(TrainingListener):
    def __init__(self, iterationCount, frequency=1):
        self.total, self.frequency, self.start = iterationCount, frequency, time.time()

    def iterationDone(self, model, iteration, epoch):
        if iteration % self.frequency == 0 and iteration > 0:
            el = time.time() - self.start
            remaining = el / iteration * (self.total - iteration)
            print("Remaining time : %d mn - End expected : %s" % (remaining / 60, time.ctime(time.time() + remaining)))


class EvaluativeListener(TrainingListener):
    def __init__(self, iterator, frequency=1, invocationType="EPOCH_END", evaluations=None):
        self.iterator, self.frequency, self.invocationType = iterator, frequency, invocationType
        self.evaluations = evaluations
        self.results = []
        self._count = 0

    def _run(self, model):
        from ..eval.evaluation import Evaluation
        evs = self.evaluations or [Evaluation()]
        for e in evs:
            e.reset()
        model.doEvaluation(self.iterator, *evs)
        self.results.append(evs[0])
        print(evs[0].stats())

    def iterationDone(self, model, iteration, epoch):
        if self.invocationType == "ITERATION_END" and iteration % self.frequency == 0:
            self._run(model)

    def onEpochEnd(self, model):
        self._count += 1
        if self.invocationType == "EPOCH_END" and self._count % self.frequency == 0:
            self._run(model)


class SleepyTrainingListener(TrainingListener):
    def __init__(self, timerIteration=0, timerEE=0, timerES=0):
        self.timerIteration, self.timerEE, self.timerES = timerIteration, timerEE, timerES

    def iterationDone(self, model, iteration, epoch):
        if self.timerIteration:
            time.sleep(self.timerIteration / 1000.0)

    def onEpochEnd(self, model):
        if self.timerEE:
            time.sleep(self.timerEE / 1000.0)

    def onEpochStart(self, model):
        if self.timerES:
            time.sleep(self.timerES / 1000.0)


class CheckpointListener(TrainingListener):
    """Periodic checkpoints with retention policy and checkpointInfo.txt index
    (optimize/listeners/CheckpointListener.java)."""

    def __init__(self, directory, saveEveryNEpochs=None, saveEveryNIterations=None, saveEveryMs=None,
                 keepAll=True, keepLast=None, keepEvery=None, saveUpdater=True, logSaving=True, deleteExisting=False):
        self.dir = str(directory)
        os.makedirs(self.dir, exist_ok=True)
        self.nEpochs, self.nIter, self.ms = saveEveryNEpochs, saveEveryNIterations, saveEveryMs
        self.keepAll, self.keepLast, self.keepEvery = keepAll and keepLast is None, keepLast, keepEvery
        self.saveUpdater, self.logSaving = saveUpdater, logSaving
        self._last_save = time.time()
        self._num = 0
        self._checkpoints = []
        info = os.path.join(self.dir, "checkpointInfo.txt")
        if deleteExisting and os.path.exists(info):
            for cp in CheckpointListener.availableCheckpoints(self.dir):
                try:
                    os.remove(os.path.join(self.dir, cp["filename"]))
                except OSError:
                    pass
            os.remove(info)
        if os.path.exists(info):
            self._checkpoints = CheckpointListener.availableCheckpoints(self.dir)
            self._num = max([c["checkpointNum"] for c in self._checkpoints], default=-1) + 1

    class Builder:
        def __init__(self, directory):
            self.kw = {"directory": directory}

        def saveEveryEpoch(self):
            self.kw["saveEveryNEpochs"] = 1; return self

        def saveEveryNEpochs(self, n):
            self.kw["saveEveryNEpochs"] = n; return self

        def saveEveryNIterations(self, n):
            self.kw["saveEveryNIterations"] = n; return self

        def saveEvery(self, amount, unit="SECONDS"):
            mult = {"MILLISECONDS": 1, "SECONDS": 1000, "MINUTES": 60000, "HOURS": 3600000}[unit]
            self.kw["saveEveryMs"] = amount * mult; return self

        def keepAll(self):
            self.kw["keepAll"] = True; return self

        def keepLast(self, n):
            self.kw["keepLast"] = n; return self

        def keepLastAndEvery(self, n, every):
            self.kw["keepLast"], self.kw["keepEvery"] = n, every; return self

        def saveUpdater(self, b):
            self.kw["saveUpdater"] = b; return self

        def deleteExisting(self, b):
            self.kw["deleteExisting"] = b; return self

        def logSaving(self, b):
            self.kw["logSaving"] = b; return self

        def build(self):
            return CheckpointListener(**self.kw)

    def iterationDone(self, model, iteration, epoch):
        if self.nIter and iteration > 0 and iteration % self.nIter == 0:
            self._save(model, iteration, epoch)
        elif self.ms and (time.time() - self._last_save) * 1000 >= self.ms:
            self._save(model, iteration, epoch)

    def onEpochEnd(self, model):
        ep = model.getEpochCount() + 1
        if self.nEpochs and ep % self.nEpochs == 0:
            self._save(model, model.getIterationCount(), ep)

    def _save(self, model, iteration, epoch):
        kind = "ComputationGraph" if hasattr(model, "topologicalOrder") else "MultiLayerNetwork"
        fn = "checkpoint_%d_%s.zip" % (self._num, kind)
        model.save(os.path.join(self.dir, fn), self.saveUpdater)
        cp = {"checkpointNum": self._num, "timestamp": int(time.time() * 1000), "iteration": iteration, "epoch": epoch,
              "modelType": kind, "filename": fn}
        self._checkpoints.append(cp)
        info = os.path.join(self.dir, "checkpointInfo.txt")
        new = not os.path.exists(info)
        with open(info, "a") as f:
            if new:
                f.write("checkpointNum,timestamp,iteration,epoch,modelType,filename\n")
            f.write("%(checkpointNum)d,%(timestamp)d,%(iteration)d,%(epoch)d,%(modelType)s,%(filename)s\n" % cp)
        self._num += 1
        self._last_save = time.time()
        if self.logSaving:
            log.info("Model checkpoint saved: epoch %d, iteration %d, path: %s", epoch, iteration, fn)
        if not self.keepAll and self.keepLast is not None:
            keep = set(c["checkpointNum"] for c in self._checkpoints[-self.keepLast:])
            for c in list(self._checkpoints):
                n = c["checkpointNum"]
                if n in keep or (self.keepEvery and n % self.keepEvery == 0):
                    continue
                p = os.path.join(self.dir, c["filename"])
                if os.path.exists(p):
                    os.remove(p)
                self._checkpoints.remove(c)

    @staticmethod
    def availableCheckpoints(directory):
        info = os.path.join(str(directory), "checkpointInfo.txt")
        out = []
        if not os.path.exists(info):
            return out
        with open(info) as f:
            lines = f.read().strip().split("\n")[1:]
        for ln in lines:
            p = ln.split(",")
            if len(p) < 6:
                continue
            cp = {"checkpointNum": int(p[0]), "timestamp": int(p[1]), "iteration": int(p[2]), "epoch": int(p[3]),
                  "modelType": p[4], "filename": p[5]}
            if os.path.exists(os.path.join(str(directory), cp["filename"])):
                out.append(cp)
        return out

    @staticmethod
    def lastCheckpoint(directory):
        cps = CheckpointListener.availableCheckpoints(directory)
        return cps[-1] if cps else None

    @staticmethod
    def loadCheckpointMLN(directory, checkpointNum):
        from ..nn.multilayer import MultiLayerNetwork
        for c in CheckpointListener.availableCheckpoints(directory):
            if c["checkpointNum"] == checkpointNum:
                return MultiLayerNetwork.load(os.path.join(str(directory), c["filename"]), True)
        raise FileNotFoundError("checkpoint %d" % checkpointNum)

    @staticmethod
    def loadCheckpointCG(directory, checkpointNum):
        from ..nn.graph.computation_graph import ComputationGraph
        for c in CheckpointListener.availableCheckpoints(directory):
            if c["checkpointNum"] == checkpointNum:
                return ComputationGraph.load(os.path.join(str(directory), c["filename"]), True)
        raise FileNotFoundError("checkpoint %d" % checkpointNum)

    @staticmethod
    def loadLastCheckpointMLN(directory):
        return CheckpointListener.loadCheckpointMLN(directory, CheckpointListener.lastCheckpoint(directory)["checkpointNum"])

    @staticmethod
    def loadLastCheckpointCG(directory):
        return CheckpointListener.loadCheckpointCG(directory, CheckpointListener.lastCheckpoint(directory)["checkpointNum"])


class FailureTestingListener(TrainingListener):
    """Fault injection (optimize/listeners/FailureTestingListener.java:41-316)."""

    class FailureMode:
        OOM, SYSTEM_EXIT_1, ILLEGAL_STATE, INFINITE_SLEEP = "OOM", "SYSTEM_EXIT_1", "ILLEGAL_STATE", "INFINITE_SLEEP"

    class CallType:
        ANY, EPOCH_START, EPOCH_END, FORWARD_PASS, GRADIENT_CALC, BACKWARD_PASS, ITER_DONE = (
            "ANY", "EPOCH_START", "EPOCH_END", "FORWARD_PASS", "GRADIENT_CALC", "BACKWARD_PASS", "ITER_DONE")

    class FailureTrigger:
        def initialize(self):
            pass

        def triggerFailure(self, callType, iteration, epoch, model):
            raise NotImplementedError

    class IterationEpochTrigger(FailureTrigger):
        def __init__(self, isEpoch, count):
            self.isEpoch, self.count = isEpoch, count

        def triggerFailure(self, callType, iteration, epoch, model):
            return (epoch if self.isEpoch else iteration) == self.count

    class RandomProb(FailureTrigger):
        def __init__(self, callType, prob, seed=0):
            import random
            self.callType, self.prob, self.rng = callType, prob, random.Random(seed)

        def triggerFailure(self, callType, iteration, epoch, model):
            return (self.callType in ("ANY", callType)) and self.rng.random() < self.prob

    class TimeSinceInitializedTrigger(FailureTrigger):
        def __init__(self, msSinceInit):
            self.ms = msSinceInit
            self.t0 = None

        def initialize(self):
            self.t0 = time.time()

        def triggerFailure(self, callType, iteration, epoch, model):
            return (time.time() - self.t0) * 1000 >= self.ms

    class UserNameTrigger(FailureTrigger):
        def __init__(self, userName):
            self.userName = userName

        def triggerFailure(self, callType, iteration, epoch, model):
            import getpass
            return getpass.getuser() == self.userName

    class HostNameTrigger(FailureTrigger):
        def __init__(self, hostName):
            self.hostName = hostName

        def triggerFailure(self, callType, iteration, epoch, model):
            import socket
            return socket.gethostname() == self.hostName

    class And(FailureTrigger):
        def __init__(self, *triggers):
            self.triggers = triggers

        def initialize(self):
            for t in self.triggers:
                t.initialize()

        def triggerFailure(self, *a):
            return all(t.triggerFailure(*a) for t in self.triggers)

    class Or(And):
        def triggerFailure(self, *a):
            return any(t.triggerFailure(*a) for t in self.triggers)

    def __init__(self, failureMode, trigger):
        self.failureMode, self.trigger = failureMode, trigger
        trigger.initialize()

    def _call(self, ct, model):
        if self.trigger.triggerFailure(ct, model.getIterationCount(), model.getEpochCount(), model):
            m = self.failureMode
            if m == "OOM":
                raise MemoryError("FailureTestingListener: simulated OOM")
            if m == "SYSTEM_EXIT_1":
                raise SystemExit(1)
            if m == "ILLEGAL_STATE":
                raise RuntimeError("FailureTestingListener - IllegalStateException")
            if m == "INFINITE_SLEEP":
                while True:
                    time.sleep(3600)

    def iterationDone(self, model, iteration, epoch):
        self._call("ITER_DONE", model)

    def onEpochStart(self, model):
        self._call("EPOCH_START", model)

    def onEpochEnd(self, model):
        self._call("EPOCH_END", model)

    def onGradientCalculation(self, model):
        self._call("GRADIENT_CALC", model)
