"""SameDiff listeners (reference: nd4j-api autodiff/listeners/** — Listener/BaseListener, At, Operation,
impl/{ScoreListener,HistoryListener,UIListener}, checkpoint/CheckpointListener, profiler/ProfilingListener
(chrome-trace JSON) + comparison/ProfileAnalyzer, debugging/{ExecDebuggingListener,ArraySavingListener,
OpBenchmarkListener,ControlflowListener})."""
from __future__ import annotations

import json
import os
import time
from collections import defaultdict
from typing import Dict, List, Optional

import torch


class Operation:
    TRAINING, TRAINING_VALIDATION, INFERENCE, EVALUATION = "TRAINING", "TRAINING_VALIDATION", "INFERENCE", "EVALUATION"


class At:
    def __init__(self, epoch=0, iteration=0, operation=Operation.INFERENCE, trainingThreadNum=0, javaThreadNum=0):
        self.epoch_, self.iteration_, self.operation_ = epoch, iteration, operation
        self.trainingThreadNum, self.javaThreadNum = trainingThreadNum, javaThreadNum

    def epoch(self): return self.epoch_
    def iteration(self): return self.iteration_
    def operation(self): return self.operation_

    def __repr__(self):
        return "At(epoch=%d, iter=%d, op=%s)" % (self.epoch_, self.iteration_, self.operation_)


class Listener:
    def isActive(self, operation): return True
    def requiredVariables(self, sd): return []
    def epochStart(self, sd, at): pass
    def epochEnd(self, sd, at, lossCurve, epochTimeMillis): return "CONTINUE"
    def validationDone(self, sd, at, validationData): return "CONTINUE"
    def iterationStart(self, sd, at, data, etlTimeMs): pass
    def iterationDone(self, sd, at, dataSet, loss): pass
    def operationStart(self, sd, op): pass
    def operationEnd(self, sd, op): pass
    def preOpExecution(self, sd, at, op, inputs): pass
    def opExecution(self, sd, at, op, inputs, outputs, nanos=0): pass
    def activationAvailable(self, sd, at, batch, op, varName, activation): pass
    def preUpdate(self, sd, at, variable, update): pass


BaseListener = Listener


class ScoreListener(Listener):
    def __init__(self, frequency=10, reportEpochs=True, reportIterPerformance=True):
        self.frequency, self.reportEpochs, self.reportIterPerformance = frequency, reportEpochs, reportIterPerformance
        self.log: List[str] = []
        self._t = None
        self._ex = 0

    def iterationStart(self, sd, at, data, etl):
        if self._t is None:
            self._t = time.time()

    def iterationDone(self, sd, at, ds, loss):
        try:
            self._ex += ds.numExamples()
        except Exception:
            pass
        if self.frequency > 0 and at.iteration() % self.frequency == 0:
            dt = max(time.time() - (self._t or time.time()), 1e-9)
            msg = "Loss at epoch %d, iteration %d: %s" % (at.epoch(), at.iteration(), ", ".join("%.6f" % l for l in loss))
            if self.reportIterPerformance and self._ex:
                msg += " (%.1f examples/sec)" % (self._ex / dt)
            self.log.append(msg)
            print(msg)
            self._t, self._ex = time.time(), 0

    def epochEnd(self, sd, at, lossCurve, ms):
        if self.reportEpochs:
            msg = "Epoch %d complete in %d ms; mean loss %s" % (at.epoch(), ms, lossCurve.lastMeanLoss())
            self.log.append(msg)
            print(msg)
        return "CONTINUE"


class HistoryListener(Listener):
    def __init__(self, trainEvaluations=None):
        self.losses: List[List[float]] = []
        self.epochLosses = []

    def iterationDone(self, sd, at, ds, loss):
        self.losses.append(list(loss))

    def epochEnd(self, sd, at, lossCurve, ms):
        self.epochLosses.append(lossCurve.lastMeanLoss())
        return "CONTINUE"


class CheckpointListener(Listener):
    """Periodic ``.fb`` checkpoints with keep-last / keep-every policies (checkpoint/CheckpointListener.java)."""

    def __init__(self, rootDir, saveEveryNEpochs=None, saveEveryNIterations=None, keepLast=None, keepLastAndEvery=None, saveUpdaterState=True,
                 deleteExisting=False):
        self.dir = str(rootDir)
        os.makedirs(self.dir, exist_ok=True)
        self.everyEp, self.everyIt, self.keepLast, self.keepEvery = saveEveryNEpochs, saveEveryNIterations, keepLast, keepLastAndEvery
        self.saveUpdater = saveUpdaterState
        self.saved: List[dict] = []
        if deleteExisting:
            for f in os.listdir(self.dir):
                if f.startswith("checkpoint-") and f.endswith(".fb"):
                    os.remove(os.path.join(self.dir, f))

    def _save(self, sd, at):
        n = len(self.saved)
        path = os.path.join(self.dir, "checkpoint-%d_epoch-%d_iter-%d.fb" % (n, at.epoch(), at.iteration()))
        sd.save(path, self.saveUpdater)
        self.saved.append({"num": n, "epoch": at.epoch(), "iteration": at.iteration(), "path": path, "time": time.time()})
        with open(os.path.join(self.dir, "checkpointInfo.txt"), "w") as f:
            for c in self.saved:
                f.write("%d,%d,%d,%s\n" % (c["num"], c["epoch"], c["iteration"], os.path.basename(c["path"])))
        if self.keepLast:
            for c in self.saved[:-self.keepLast]:
                keep = self.keepEvery and c["num"] % self.keepEvery == 0
                if not keep and os.path.exists(c["path"]):
                    os.remove(c["path"])

    def iterationDone(self, sd, at, ds, loss):
        if self.everyIt and (at.iteration() + 1) % self.everyIt == 0:
            self._save(sd, at)

    def epochEnd(self, sd, at, lossCurve, ms):
        if self.everyEp and (at.epoch() + 1) % self.everyEp == 0:
            self._save(sd, at)
        return "CONTINUE"

    def availableCheckpoints(self):
        return [c for c in self.saved if os.path.exists(c["path"])]

    def lastCheckpoint(self):
        a = self.availableCheckpoints()
        return a[-1] if a else None

    @staticmethod
    def loadLastCheckpoint(rootDir):
        from .samediff import SameDiff
        info = os.path.join(str(rootDir), "checkpointInfo.txt")
        last = None
        for line in open(info):
            name = line.strip().split(",")[-1]
            if os.path.exists(os.path.join(str(rootDir), name)):
                last = name
        return SameDiff.load(os.path.join(str(rootDir), last))


class ProfilingListener(Listener):
    """Chrome-trace (``chrome://tracing`` / Perfetto) JSON of op execution (profiler/ProfilingListener.java).
    Device time is measured with stream events when ``deviceTiming`` is on, host wall time otherwise."""

    def __init__(self, outputFile, warmup=0, maxIterations=None, operations=None, deviceTiming=False):
        self.path, self.warmup, self.maxIter, self.ops = str(outputFile), warmup, maxIterations, operations
        self.deviceTiming = deviceTiming and torch.cuda.is_available()
        self.events: List[dict] = []
        self._start = {}
        self._n = 0
        self._t0 = time.perf_counter_ns()

    def isActive(self, operation):
        return self.ops is None or operation in self.ops

    def iterationDone(self, sd, at, ds, loss):
        self._n += 1

    def _on(self):
        return self._n >= self.warmup and (self.maxIter is None or self._n < self.warmup + self.maxIter)

    def preOpExecution(self, sd, at, op, inputs):
        if self._on():
            if self.deviceTiming:
                torch.cuda.synchronize()
            self._start[op.name] = time.perf_counter_ns()

    def opExecution(self, sd, at, op, inputs, outputs, nanos=0):
        if not self._on() or op.name not in self._start:
            return
        if self.deviceTiming:
            torch.cuda.synchronize()
        t0 = self._start.pop(op.name)
        t1 = time.perf_counter_ns()
        self.events.append({"name": op.name, "cat": "op", "ph": "X", "ts": (t0 - self._t0) / 1000.0, "dur": (t1 - t0) / 1000.0,
                            "pid": os.getpid(), "tid": 0,
                            "args": {"opName": op.opName, "iteration": at.iteration() if at else 0,
                                     "inputShapes": [list(x.shape) for x in inputs if isinstance(x, torch.Tensor)],
                                     "outputShapes": [list(x.shape) for x in outputs if isinstance(x, torch.Tensor)]}})

    def operationEnd(self, sd, op):
        self.flush()

    def flush(self):
        with open(self.path, "w") as f:
            json.dump({"traceEvents": self.events, "displayTimeUnit": "ms"}, f)


class ProfileAnalyzer:
    """Aggregate / compare chrome traces (profiler/comparison/ProfileAnalyzer.java)."""

    @staticmethod
    def summarize(path, by="opName"):
        ev = json.load(open(str(path)))["traceEvents"]
        agg = defaultdict(lambda: [0, 0.0])
        for e in ev:
            k = e["args"].get("opName") if by == "opName" else e["name"]
            agg[k][0] += 1
            agg[k][1] += e["dur"]
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        return [{"key": k, "count": c, "totalUs": d, "meanUs": d / c} for k, (c, d) in rows]

    @staticmethod
    def compare(pathA, pathB, by="opName"):
        a = {r["key"]: r for r in ProfileAnalyzer.summarize(pathA, by)}
        b = {r["key"]: r for r in ProfileAnalyzer.summarize(pathB, by)}
        out = []
        for k in sorted(set(a) | set(b)):
            ta, tb = a.get(k, {}).get("totalUs", 0.0), b.get(k, {}).get("totalUs", 0.0)
            out.append({"key": k, "aUs": ta, "bUs": tb, "ratio": (tb / ta) if ta else float("inf")})
        return sorted(out, key=lambda r: -max(r["aUs"], r["bUs"]))


class ExecDebuggingListener(Listener):
    class PrintMode:
        OPS_ONLY, SHAPES_ONLY, REPRODUCE = "OPS_ONLY", "SHAPES_ONLY", "REPRODUCE"

    def __init__(self, printMode="OPS_ONLY", maxIterations=-1, logIter=False):
        self.mode, self.maxIter, self.logIter = printMode, maxIterations, logIter
        self.lines: List[str] = []

    def opExecution(self, sd, at, op, inputs, outputs, nanos=0):
        if self.maxIter > 0 and at and at.iteration() >= self.maxIter:
            return
        s = "%s(%s)" % (op.opName, op.name)
        if self.mode != "OPS_ONLY":
            s += " in=%s out=%s" % ([list(x.shape) for x in inputs if isinstance(x, torch.Tensor)],
                                    [list(x.shape) for x in outputs if isinstance(x, torch.Tensor)])
        if self.mode == "REPRODUCE":
            s += " iArgs=%s tArgs=%s bArgs=%s" % (op.iArgs, op.tArgs, op.bArgs)
        self.lines.append(s)
        print(s)


class ArraySavingListener(Listener):
    def __init__(self, directory):
        self.dir = str(directory)
        os.makedirs(self.dir, exist_ok=True)
        self.count = 0

    def opExecution(self, sd, at, op, inputs, outputs, nanos=0):
        from ..nd4j.factory import Nd4j
        from ..nd4j.ndarray import INDArray
        for n, o in zip(op.outputs, outputs):
            if isinstance(o, torch.Tensor):
                Nd4j.saveBinary(INDArray(o.detach()), os.path.join(self.dir, "%d_%s.bin" % (self.count, n.replace("/", "__").replace(":", "_"))))
                self.count += 1

    @staticmethod
    def compare(dirA, dirB, eps=1e-5):
        from ..nd4j.factory import Nd4j
        bad = []
        for f in sorted(os.listdir(str(dirA))):
            pb = os.path.join(str(dirB), f)
            if not os.path.exists(pb):
                bad.append((f, "missing"))
                continue
            if not Nd4j.readBinary(os.path.join(str(dirA), f)).equalsWithEps(Nd4j.readBinary(pb), eps):
                bad.append((f, "differs"))
        return bad


class OpBenchmarkListener(Listener):
    def __init__(self, operation=Operation.INFERENCE, printActivations=False):
        self.operation = operation
        self.agg: Dict[str, dict] = {}

    def opExecution(self, sd, at, op, inputs, outputs, nanos=0):
        a = self.agg.setdefault(op.name, {"opName": op.opName, "count": 0, "nanos": 0})
        a["count"] += 1
        a["nanos"] += nanos

    def getAggregateModeMap(self):
        return self.agg

    def printStats(self):
        rows = sorted(self.agg.items(), key=lambda kv: -kv[1]["nanos"])
        s = "\n".join("%-40s %-24s %6d calls %10.3f ms" % (k, v["opName"], v["count"], v["nanos"] / 1e6) for k, v in rows)
        print(s)
        return s


class ControlflowListener(Listener):
    """Records control-flow op firings (debugging/ControlflowListener.java)."""

    def __init__(self):
        self.events = []

    def opExecution(self, sd, at, op, inputs, outputs, nanos=0):
        if op.opName.lower() in ("switch", "merge", "enter", "exit", "next_iteration", "loop_cond"):
            self.events.append((op.opName, op.name))


class UIListener(Listener):
    """Writes loss / parameter-norm time series to a JSONL log consumable by the stats storage (ui/stats.py).
    (reference: impl/UIListener.java writing LogFileWriter protobuf-ish event files)."""

    def __init__(self, logFile, lossFrequency=1, paramFrequency=10):
        self.path, self.lossFreq, self.paramFreq = str(logFile), lossFrequency, paramFrequency

    def iterationDone(self, sd, at, ds, loss):
        rec = {"iter": at.iteration(), "epoch": at.epoch(), "time": time.time()}
        if self.lossFreq and at.iteration() % self.lossFreq == 0:
            rec["loss"] = list(loss)
        if self.paramFreq and at.iteration() % self.paramFreq == 0:
            rec["paramNorms"] = {n: float(sd._arrays[n].float().norm()) for n in sd.trainableVariables()}
        with open(self.path, "a") as f:
            f.write(json.dumps(rec) + "\n")
