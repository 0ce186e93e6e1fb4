"""StatsListener + StatsStorage (reference: deeplearning4j-ui-model/.../ui/model/stats/{BaseStatsListener,StatsListener}.java,
api/{StatsInitializationReport,StatsReport,StatsUpdateConfiguration}.java, storage/{InMemoryStatsStorage,FileStatsStorage,
mapdb/MapDBStatsStorage}.java, core api/storage/{StatsStorage,StatsStorageRouter,StatsStorageEvent,Persistable}.java).

The reference encodes reports with SBE into MapDB; here a report is a JSON document and FileStatsStorage is a single sqlite file
(stdlib), which keeps the same (sessionID, typeID, workerID, timestamp) addressing and listener notifications.

All per-parameter statistics (mean magnitudes, mean/stdev, histograms) are computed on device in one pass per tensor group and
brought to the host with a single transfer per report, so attaching the listener does not add a sync per parameter."""
from __future__ import annotations

import json
import os
import platform
import sqlite3
import threading
import time
import uuid
from typing import Callable, Dict, List, Optional

import torch

from ..optimize.listeners import TrainingListener

TYPE_ID = "StatsListener"


class StatsUpdateConfiguration:
    def __init__(self, reportingFrequency=1, collectPerformanceStats=True, collectMemoryStats=True, collectLearningRates=True,
                 collectHistogramsParameters=True, collectHistogramsGradients=True, collectHistogramsUpdates=True, collectHistogramsActivations=False,
                 numHistogramBins=20, collectMeanParameters=True, collectMeanGradients=True, collectMeanUpdates=True, collectStdevParameters=True,
                 collectStdevGradients=True, collectStdevUpdates=True, collectMeanMagnitudesParameters=True, collectMeanMagnitudesGradients=True,
                 collectMeanMagnitudesUpdates=True):
        self.__dict__.update({k: v for k, v in locals().items() if k != "self"})


class Persistable:
    def __init__(self, sessionID, typeID, workerID, timestamp, content: dict):
        self.sessionID, self.typeID, self.workerID, self.timestamp, self.content = sessionID, typeID, workerID, int(timestamp), content

    def getSessionID(self):
        return self.sessionID

    def getTypeID(self):
        return self.typeID

    def getWorkerID(self):
        return self.workerID

    def getTimeStamp(self):
        return self.timestamp

    def encode(self) -> bytes:
        return json.dumps(self.content).encode("utf-8")

    def __getitem__(self, k):
        return self.content[k]

    def get(self, k, d=None):
        return self.content.get(k, d)


class StatsInitializationReport(Persistable):
    """Static info: software (versions, backend), hardware (devices, memory), model (class, #params, #layers, param names)."""


class StatsReport(Persistable):
    """Per-iteration info: score, lr, performance, memory, per-parameter summaries and histograms."""

    def getScore(self):
        return self.content.get("score")

    def getIterationCount(self):
        return self.content.get("iteration")


class StatsStorageEvent:
    NewSessionID, NewTypeID, NewWorkerID, PostMetaData, PostStaticInfo, PostUpdate = \
        "NewSessionID", "NewTypeID", "NewWorkerID", "PostMetaData", "PostStaticInfo", "PostUpdate"

    def __init__(self, storage, eventType, sessionID, typeID, workerID, timestamp):
        self.storage, self.eventType, self.sessionID, self.typeID, self.workerID, self.timestamp = storage, eventType, sessionID, typeID, workerID, timestamp


class StatsStorage:
    """Router + queryable storage."""

    def __init__(self):
        self._listeners: List[Callable[[StatsStorageEvent], None]] = []
        self._lock = threading.RLock()

    # -- router API
    def putStorageMetaData(self, meta: Persistable):
        self._put("meta", meta, StatsStorageEvent.PostMetaData)

    def putStaticInfo(self, info: Persistable):
        self._put("static", info, StatsStorageEvent.PostStaticInfo)

    def putUpdate(self, update: Persistable):
        self._put("update", update, StatsStorageEvent.PostUpdate)

    # -- listeners
    def registerStatsStorageListener(self, l):
        self._listeners.append(l)

    def deregisterStatsStorageListener(self, l):
        if l in self._listeners:
            self._listeners.remove(l)

    def _notify(self, kind, p: Persistable, new_session, new_worker):
        evs = []
        if new_session:
            evs.append(StatsStorageEvent.NewSessionID)
        if new_worker:
            evs.append(StatsStorageEvent.NewWorkerID)
        evs.append(kind)
        for e in evs:
            ev = StatsStorageEvent(self, e, p.sessionID, p.typeID, p.workerID, p.timestamp)
            for l in list(self._listeners):
                (l.notify if hasattr(l, "notify") else l)(ev)

    # -- to implement
    def _put(self, table, p, event):
        raise NotImplementedError

    def close(self):
        pass

    def isClosed(self):
        return False


class InMemoryStatsStorage(StatsStorage):
    def __init__(self):
        super().__init__()
        self._data: Dict[str, List[Persistable]] = {"meta": [], "static": [], "update": []}

    def _put(self, table, p, event):
        with self._lock:
            new_s = p.sessionID not in self.listSessionIDs()
            new_w = p.workerID not in self.listWorkerIDsForSession(p.sessionID)
            if table == "static":
                self._data[table] = [q for q in self._data[table] if (q.sessionID, q.typeID, q.workerID) != (p.sessionID, p.typeID, p.workerID)]
            self._data[table].append(p)
        self._notify(event, p, new_s, new_w)

    def _all(self):
        return self._data["meta"] + self._data["static"] + self._data["update"]

    def listSessionIDs(self):
        return sorted({p.sessionID for p in self._all()})

    def sessionExists(self, s):
        return s in self.listSessionIDs()

    def listTypeIDsForSession(self, s):
        return sorted({p.typeID for p in self._all() if p.sessionID == s})

    def listWorkerIDsForSession(self, s):
        return sorted({p.workerID for p in self._all() if p.sessionID == s})

    def listWorkerIDsForSessionAndType(self, s, t):
        return sorted({p.workerID for p in self._all() if p.sessionID == s and p.typeID == t})

    def getStaticInfo(self, s, t, w):
        for p in self._data["static"]:
            if (p.sessionID, p.typeID, p.workerID) == (s, t, w):
                return p
        return None

    def getAllStaticInfos(self, s, t):
        return [p for p in self._data["static"] if p.sessionID == s and p.typeID == t]

    def getStorageMetaData(self, s, t):
        for p in self._data["meta"]:
            if p.sessionID == s and p.typeID == t:
                return p
        return None

    def _updates(self, s, t, w=None):
        return sorted((p for p in self._data["update"] if p.sessionID == s and p.typeID == t and (w is None or p.workerID == w)),
                      key=lambda p: p.timestamp)

    def getNumUpdateRecordsFor(self, s, t=None, w=None):
        return len([p for p in self._data["update"] if p.sessionID == s and (t is None or p.typeID == t) and (w is None or p.workerID == w)])

    def getAllUpdatesAfter(self, s, t, w_or_ts, ts=None):
        w, ts = (None, w_or_ts) if ts is None else (w_or_ts, ts)
        return [p for p in self._updates(s, t, w) if p.timestamp > ts]

    def getLatestUpdate(self, s, t, w):
        u = self._updates(s, t, w)
        return u[-1] if u else None

    def getUpdate(self, s, t, w, ts):
        for p in self._updates(s, t, w):
            if p.timestamp == ts:
                return p
        return None

    def getLatestUpdateAllWorkers(self, s, t):
        return [u for u in (self.getLatestUpdate(s, t, w) for w in self.listWorkerIDsForSessionAndType(s, t)) if u is not None]

    def getAllUpdateTimes(self, s, t, w):
        return [p.timestamp for p in self._updates(s, t, w)]


class FileStatsStorage(InMemoryStatsStorage):
    """Persistent storage in one sqlite file; loads existing content on open (FileStatsStorage.java semantics)."""

    def __init__(self, path):
        super().__init__()
        self.path = os.fspath(path)
        self._db = sqlite3.connect(self.path, check_same_thread=False)
        self._db.execute("CREATE TABLE IF NOT EXISTS records (tbl TEXT, session TEXT, type TEXT, worker TEXT, ts INTEGER, cls TEXT, body BLOB)")
        self._db.commit()
        self._closed = False
        cls_map = {"StatsInitializationReport": StatsInitializationReport, "StatsReport": StatsReport}
        for tbl, s, t, w, ts, cls, body in self._db.execute("SELECT tbl, session, type, worker, ts, cls, body FROM records ORDER BY rowid"):
            self._data[tbl].append(cls_map.get(cls, Persistable)(s, t, w, ts, json.loads(body)))

    def _put(self, table, p, event):
        with self._lock:
            if table == "static":
                self._db.execute("DELETE FROM records WHERE tbl='static' AND session=? AND type=? AND worker=?", (p.sessionID, p.typeID, p.workerID))
            self._db.execute("INSERT INTO records VALUES (?,?,?,?,?,?,?)",
                             (table, p.sessionID, p.typeID, p.workerID, p.timestamp, type(p).__name__, p.encode()))
            self._db.commit()
        super()._put(table, p, event)

    def close(self):
        self._db.close()
        self._closed = True

    def isClosed(self):
        return self._closed


class RemoteUIStatsStorageRouter:
    """POSTs reports to a remote UIServer (``/remoteReceive``) from a background thread with retry
    (RemoteUIStatsStorageRouter.java)."""

    def __init__(self, address, maxRetryCount=10, retryDelayMS=1000, retryBackoffFactor=2.0):
        self.url = address.rstrip("/") + "/remoteReceive"
        self.maxRetry, self.delay, self.backoff = maxRetryCount, retryDelayMS / 1000.0, retryBackoffFactor
        import queue
        self._q: "queue.Queue" = queue.Queue()
        self._shutdown = False
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _send(self, kind, p: Persistable):
        self._q.put({"type": kind, "class": type(p).__name__, "sessionID": p.sessionID, "typeID": p.typeID, "workerID": p.workerID,
                     "timestamp": p.timestamp, "content": p.content})

    def putStorageMetaData(self, m): self._send("meta", m)
    def putStaticInfo(self, s): self._send("static", s)
    def putUpdate(self, u): self._send("update", u)

    def _run(self):
        import urllib.request
        while not self._shutdown:
            item = self._q.get()
            if item is None:
                return
            delay = self.delay
            for _ in range(self.maxRetry + 1):
                try:
                    req = urllib.request.Request(self.url, data=json.dumps(item).encode(), headers={"Content-Type": "application/json"})
                    urllib.request.urlopen(req, timeout=5).read()
                    break
                except Exception:
                    time.sleep(delay)
                    delay *= self.backoff

    def flush(self, timeout=10.0):
        t0 = time.time()
        while not self._q.empty() and time.time() - t0 < timeout:
            time.sleep(0.02)
        time.sleep(0.05)

    def shutdown(self):
        self._shutdown = True
        self._q.put(None)


# ----------------------------------------------------------------------------- listener
def _summ(t: torch.Tensor, bins: int, hist: bool):
    f = t.detach().float().reshape(-1)
    if f.numel() == 0:
        return None
    out = {"meanMagnitude": f.abs().mean(), "mean": f.mean(), "stdev": f.std(unbiased=False) if f.numel() > 1 else f.new_zeros(())}
    if hist:
        lo, hi = f.min(), f.max()
        out["min"], out["max"] = lo, hi
        out["hist"] = torch.histc(f, bins=bins, min=float(lo), max=float(hi)) if float(hi) > float(lo) else torch.zeros(bins, device=f.device)
    return out


class StatsListener(TrainingListener):
    def __init__(self, router, listenerFrequency=1, updateConfig: Optional[StatsUpdateConfiguration] = None, sessionID=None, workerID=None):
        self.router = router
        self.cfg = updateConfig or StatsUpdateConfiguration(reportingFrequency=listenerFrequency)
        self.sessionID = sessionID or str(uuid.uuid4())
        self.workerID = workerID or "%s_%d" % (platform.node() or "worker", os.getpid())
        self._init_done = False
        self._last_time = None
        self._last_params = None
        self._examples = 0
        self._minibatches = 0
        self._t0 = None

    def getSessionID(self):
        return self.sessionID

    def getWorkerID(self):
        return self.workerID

    def _static(self, model):
        dev = model.ctx.device
        hw = {"device": str(dev), "deviceName": torch.cuda.get_device_name(dev) if dev.type == "cuda" else platform.processor() or "cpu",
              "deviceTotalMemory": torch.cuda.get_device_properties(dev).total_memory if dev.type == "cuda" else None,
              "nDevices": torch.cuda.device_count() if dev.type == "cuda" else 0, "cpuCores": os.cpu_count()}
        sw = {"os": platform.platform(), "python": platform.python_version(), "torch": torch.__version__,
              "hip": getattr(torch.version, "hip", None), "backend": "MI355X/HIP" if dev.type == "cuda" else "CPU", "hostname": platform.node()}
        names = list(model.paramTable().keys())
        layers = model._ordered_layers()
        mi = {"modelClass": type(model).__name__, "numParams": model.numParams(), "numLayers": len(layers), "paramNames": names,
              "layerNames": [getattr(l.conf, "layerName", None) or str(i) for i, l in enumerate(layers)],
              "configJson": model.conf.toJson() if hasattr(model.conf, "toJson") else None}
        try:
            from .modules import buildGraphInfo
            mi["graph"] = buildGraphInfo(model).toDict()
        except Exception:          # the model page then falls back to the plain layer list
            mi["graph"] = None
        now = int(time.time() * 1000)
        self.router.putStorageMetaData(Persistable(self.sessionID, TYPE_ID, self.workerID, now,
                                                   {"initTypeClass": "StatsInitializationReport", "updateTypeClass": "StatsReport"}))
        self.router.putStaticInfo(StatsInitializationReport(self.sessionID, TYPE_ID, self.workerID, now, {"software": sw, "hardware": hw, "model": mi}))
        self._init_done = True
        self._t0 = time.time()

    def onEpochStart(self, model):
        if not self._init_done:
            self._static(model)

    def iterationDone(self, model, iteration, epoch):
        if not self._init_done:
            self._static(model)
        c = self.cfg
        mb = model.batchSize() if hasattr(model, "batchSize") else 0
        self._examples += int(mb or 0)
        self._minibatches += 1
        if c.reportingFrequency > 1 and iteration % c.reportingFrequency != 0:
            return
        now = time.time()
        rep = {"iteration": int(iteration), "epoch": int(epoch), "score": float(model.score()), "timestamp": int(now * 1000)}
        if c.collectPerformanceStats:
            dt = (now - self._last_time) if self._last_time else None
            rep["performance"] = {"totalRuntimeMs": int((now - self._t0) * 1000), "totalExamples": self._examples,
                                  "totalMinibatches": self._minibatches,
                                  "examplesPerSecond": (mb * c.reportingFrequency / dt) if dt else None,
                                  "minibatchesPerSecond": (c.reportingFrequency / dt) if dt else None,
                                  "etlTimeMs": getattr(model, "lastEtlTime", 0)}
        self._last_time = now
        if c.collectMemoryStats:
            dev = model.ctx.device
            mem = {}
            if dev.type == "cuda":
                free, total = torch.cuda.mem_get_info(dev)
                mem = {"deviceCurrentBytes": torch.cuda.memory_allocated(dev), "deviceReservedBytes": torch.cuda.memory_reserved(dev),
                       "deviceMaxBytes": total, "deviceFreeBytes": free}
            try:
                import resource
                mem["hostMaxRssKb"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
            except Exception:
                pass
            rep["memory"] = mem
        if c.collectLearningRates:
            try:
                rep["learningRates"] = {str(i): model.getLearningRate(i) for i in range(len(model._ordered_layers()))
                                        if model._ordered_layers()[i].param_specs()}
            except Exception:
                rep["learningRates"] = {}
        table = model.paramTable()
        grads = None
        try:
            grads = model.gradient().gradientForVariable()
        except Exception:
            grads = None
        pending: Dict[str, Dict[str, dict]] = {"parameters": {}, "gradients": {}, "updates": {}}
        for k, p in table.items():
            if c.collectMeanMagnitudesParameters or c.collectHistogramsParameters:
                pending["parameters"][k] = _summ(p, c.numHistogramBins, c.collectHistogramsParameters)
            if grads is not None and k in grads and (c.collectMeanMagnitudesGradients or c.collectHistogramsGradients):
                pending["gradients"][k] = _summ(grads[k], c.numHistogramBins, c.collectHistogramsGradients)
            if self._last_params is not None and k in self._last_params and (c.collectMeanMagnitudesUpdates or c.collectHistogramsUpdates):
                pending["updates"][k] = _summ(self._last_params[k] - p, c.numHistogramBins, c.collectHistogramsUpdates)
        if c.collectMeanMagnitudesUpdates or c.collectHistogramsUpdates:
            self._last_params = {k: p.detach().clone() for k, p in table.items()}
        # one host transfer for everything
        flat, index = [], []
        for grp, d in pending.items():
            for k, s in d.items():
                if s is None:
                    continue
                for stat, v in s.items():
                    index.append((grp, k, stat, v.numel()))
                    flat.append(v.reshape(-1))
        if flat:
            host = torch.cat(flat).cpu().tolist()
            off = 0
            for grp, k, stat, n in index:
                val = host[off:off + n]
                off += n
                rep.setdefault(grp, {}).setdefault(k, {})[stat] = val if stat == "hist" else val[0]
        self.router.putUpdate(StatsReport(self.sessionID, TYPE_ID, self.workerID, rep["timestamp"], rep))
