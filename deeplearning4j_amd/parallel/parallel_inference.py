"""ParallelInference / InplaceParallelInference (reference: deeplearning4j-parallelwrapper/.../parallelism/
{ParallelInference,InplaceParallelInference}.java, inference/{InferenceMode,LoadBalanceMode}.java,
inference/observers/{BasicInferenceObservable,BatchedInferenceObservable}.java).

MI355X design: one worker thread per replica, each pinned to a HIP device with its *own stream*, so replicas on the same GPU
overlap and replicas on different GPUs run fully independently; the model replica is made by a device-to-device copy of the
flat parameter vector (one xGMI transfer per GPU, not per layer).  BATCHED mode coalesces concurrent requests into one
forward up to ``batchLimit`` examples (dynamic batching) and scatters the rows back to the callers.
"""
from __future__ import annotations

import queue
import threading
import time
from typing import List, Optional, Sequence

import torch


class InferenceMode:
    SEQUENTIAL, BATCHED, INPLACE = "SEQUENTIAL", "BATCHED", "INPLACE"


class LoadBalanceMode:
    ROUND_ROBIN, FIFO = "ROUND_ROBIN", "FIFO"


class _Observable:
    """One caller's request (BasicInferenceObservable): inputs in, outputs or exception out."""
    __slots__ = ("inputs", "masks", "n", "_event", "_out", "_err")

    def __init__(self, inputs, masks=None):
        self.inputs, self.masks = inputs, masks
        self.n = int(inputs[0].shape[0])
        self._event = threading.Event()
        self._out, self._err = None, None

    def set(self, out):
        self._out = out
        self._event.set()

    def fail(self, err):
        self._err = err
        self._event.set()

    def get(self, timeout=None):
        if not self._event.wait(timeout):
            raise TimeoutError("inference request timed out")
        if self._err is not None:
            raise self._err
        return self._out


def _replicate(model, device):
    net = type(model)(model.conf.clone(), device=device)
    net.init(model.params().reshape(-1).to(device), True)
    return net


def _forward(net, inputs, masks=None):
    from ..nn.multilayer import MultiLayerNetwork
    with torch.no_grad():
        if isinstance(net, MultiLayerNetwork):
            y = net.output(inputs[0], train=False, featuresMask=masks[0] if masks else None)
            return [y]
        y = net.output(*inputs, train=False, featureMasks=masks) if masks else net.output(*inputs, train=False)
        return list(y) if isinstance(y, (list, tuple)) else [y]


def _raw(t):
    return t.tensor() if hasattr(t, "tensor") and callable(t.tensor) else t


class _Worker(threading.Thread):
    def __init__(self, pi: "ParallelInference", idx: int, device: torch.device, q: "queue.Queue"):
        super().__init__(name="InferenceWorker-%d" % idx, daemon=True)
        self.pi, self.idx, self.device, self.q = pi, idx, device, q
        self.replica = None
        self.stream = None
        self.counter = 0
        self._model_version = -1
        self.ready = threading.Event()
        self.init_error = None
        self._lib_gemm = {}            # input signature -> vendor GEMMs inside the captured graph (None = not seen yet)
        self.gemm_counts = {}

    def _refresh_model(self):
        v = self.pi._model_version
        if v != self._model_version:
            self.replica = _replicate(self.pi.model, self.device)
            if self.pi.graphCapture and self.device.type == "cuda" and hasattr(self.replica, "enableInferenceCapture"):
                self.replica.enableInferenceCapture(True)      # launch-bound small batches: replay the forward as a HIP graph
            self._model_version = v

    def run(self):
        try:
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
                self.stream = torch.cuda.Stream(self.device)
            self._refresh_model()
        except Exception as e:                     # surface replica-creation failures to the constructor
            self.init_error = e
            self.ready.set()
            return
        self.ready.set()
        while True:
            item = self.q.get()
            if item is None:
                return
            batch: List[_Observable] = [item]
            if self.pi.inferenceMode == InferenceMode.BATCHED:
                total = item.n
                deadline = time.time() + self.pi.maxWaitMs / 1000.0 if self.pi.maxWaitMs > 0 else None
                while total < self.pi.batchLimit:
                    try:
                        nxt = self.q.get_nowait() if deadline is None else self.q.get(timeout=max(0.0, deadline - time.time()))
                    except queue.Empty:
                        break
                    if nxt is None:
                        self.q.put(None)
                        break
                    if not self._compatible(item, nxt):
                        self.q.put(nxt)
                        break
                    batch.append(nxt)
                    total += nxt.n
            try:
                self._refresh_model()
                self._process(batch)
            except Exception as e:
                for o in batch:
                    o.fail(e)

    def _guarded_forward(self, ins, masks):
        """The replica's forward under the vendor-GEMM rule of ops/streams.py: workers of one device each own a stream, and two vendor
        GEMMs running at the same time on two streams can dead-lock the device (the reference holds a per-device mutex around cuBLAS for
        the same reason, MmulHelper.cu:261). 2-D GEMMs are routed to the in-tree kernel; any other vendor GEMM runs under the device
        token; a forward whose HIP graph recorded vendor GEMMs is replayed under the token as a whole. Input signatures not seen yet
        run under the token too (their first call captures, and replays, a graph whose contents are not known before)."""
        if self.stream is None or not self.pi._shares_device:
            return [_raw(y) for y in _forward(self.replica, ins, masks)]
        from ..ops import streams as _st
        sig = tuple((tuple(x.shape), x.dtype) for x in ins)
        state = self._lib_gemm.get(sig)
        guard = _st.library_gemm_guard("route", "ParallelInference worker")
        if state is False:
            with guard:
                outs = [_raw(y) for y in _forward(self.replica, ins, masks)]
        else:
            with _st.library_gemm_token(self.device), guard:
                outs = [_raw(y) for y in _forward(self.replica, ins, masks)]
        for k, v in guard.counts.items():
            self.gemm_counts[k] = self.gemm_counts.get(k, 0) + v
        if guard.counts["captured"]:
            self._lib_gemm[sig] = True
        elif state is None:
            # decided once the signature has been captured (the eager warm-up calls before it say nothing about the graph), or at once
            # when this replica does not capture at all
            cap = getattr(self.replica, "_infer_cap", None)
            e = cap.entries.get(sig) if cap is not None and not cap.disabled else None
            if cap is None or cap.disabled or (e is not None and e.get("graph") is not None):
                self._lib_gemm[sig] = False
        return outs

    @staticmethod
    def _compatible(a: _Observable, b: _Observable):
        if len(a.inputs) != len(b.inputs) or (a.masks is None) != (b.masks is None):
            return False
        return all(x.shape[1:] == y.shape[1:] and x.dtype == y.dtype for x, y in zip(a.inputs, b.inputs))

    def _process(self, batch: List[_Observable]):
        dev = self.device
        ins = [torch.cat([_raw(o.inputs[k]) for o in batch], 0) if len(batch) > 1 else _raw(batch[0].inputs[k]) for k in range(len(batch[0].inputs))]
        masks = None
        if batch[0].masks is not None:
            masks = [None if batch[0].masks[k] is None else torch.cat([_raw(o.masks[k]) for o in batch], 0) for k in range(len(batch[0].masks))]
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _nullctx()
        with ctx:
            ins = [x.to(dev, non_blocking=True) for x in ins]
            if masks:
                masks = [None if m is None else m.to(dev, non_blocking=True) for m in masks]
            n = int(ins[0].shape[0])
            if self.pi.graphCapture and dev.type == "cuda" and masks is None and n > 0:
                # pad the dynamic batch to the next power of two so that a handful of captured graphs serve every batch size
                # (inference is per-example independent; the padded rows are dropped when the outputs are sliced below)
                tgt = 1 << (n - 1).bit_length()
                if tgt != n:
                    ins = [torch.cat([x, x[-1:].expand(tgt - n, *x.shape[1:])], 0) for x in ins]
            outs = self._guarded_forward(ins, masks)
            if self.stream is not None:
                self.stream.synchronize()
        self.counter += len(batch)
        off = 0
        for o in batch:
            o.set([y[off:off + o.n] for y in outs])
            off += o.n


class _nullctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class ParallelInference:
    DEFAULT_NUM_WORKERS, DEFAULT_BATCH_LIMIT, DEFAULT_QUEUE_LIMIT = None, 32, 64

    def __init__(self, model, workers=None, inferenceMode=InferenceMode.BATCHED, batchLimit=32, queueLimit=64,
                 loadBalanceMode=LoadBalanceMode.FIFO, devices: Optional[Sequence] = None, maxWaitMs=0, graphCapture=True):
        self.graphCapture = bool(graphCapture)
        self.model = model
        self.inferenceMode, self.loadBalanceMode = inferenceMode, loadBalanceMode
        self.batchLimit, self.queueLimit, self.maxWaitMs = int(batchLimit), int(queueLimit), maxWaitMs
        if devices is None:
            n = torch.cuda.device_count() if model.ctx.device.type == "cuda" else 0
            devices = [torch.device("cuda", i) for i in range(n)] if n else [model.ctx.device]
        devices = [torch.device(d) for d in devices]
        self.workers_n = int(workers) if workers else len(devices)
        self._shares_device = self.workers_n > len(set(devices))      # several workers (= streams) on one device: ops/streams.py rule applies
        self._model_version = 0
        self._rr = 0
        self._lock = threading.Lock()
        self._shutdown = False
        self._workers: List[_Worker] = []
        if inferenceMode == InferenceMode.INPLACE:
            # caller's thread runs the forward on a replica it borrows (InplaceParallelInference semantics)
            self._replicas = [_replicate(model, devices[i % len(devices)]) for i in range(self.workers_n)]
            self._free: "queue.Queue" = queue.Queue()
            for r in self._replicas:
                self._free.put(r)
            return
        fifo = loadBalanceMode == LoadBalanceMode.FIFO
        shared = queue.Queue(maxsize=self.queueLimit) if fifo else None
        for i in range(self.workers_n):
            q = shared if fifo else queue.Queue(maxsize=self.queueLimit)
            w = _Worker(self, i, devices[i % len(devices)], q)
            self._workers.append(w)
            w.start()
        for w in self._workers:
            w.ready.wait()
            if w.init_error is not None:
                self.shutdown()
                raise w.init_error

    # -- Builder
    class Builder:
        def __init__(self, model):
            self._model, self._kw = model, {}

        def inferenceMode(self, m):
            self._kw["inferenceMode"] = m
            return self

        def loadBalanceMode(self, m):
            self._kw["loadBalanceMode"] = m
            return self

        def workers(self, n):
            self._kw["workers"] = n
            return self

        def batchLimit(self, n):
            self._kw["batchLimit"] = n
            return self

        def queueLimit(self, n):
            self._kw["queueLimit"] = n
            return self

        def devices(self, d):
            self._kw["devices"] = d
            return self

        def maxWaitMs(self, ms):
            self._kw["maxWaitMs"] = ms
            return self

        def build(self):
            return ParallelInference(self._model, **self._kw)

    # -- API
    def output(self, *inputs, masks=None, timeout=None):
        """Blocking inference; thread-safe.  Returns a tensor for one-output models, else a list."""
        if self._shutdown:
            raise RuntimeError("ParallelInference has been shut down")
        ins = [torch.as_tensor(_raw(x)) for x in (inputs[0] if len(inputs) == 1 and isinstance(inputs[0], (list, tuple)) else inputs)]
        if self.inferenceMode == InferenceMode.INPLACE:
            r = self._free.get()
            try:
                dev = r.ctx.device
                outs = [_raw(y) for y in _forward(r, [x.to(dev) for x in ins], None if masks is None else [None if m is None else m.to(dev) for m in masks])]
            finally:
                self._free.put(r)
            return outs[0] if len(outs) == 1 else outs
        obs = _Observable(ins, masks)
        self._enqueue(obs)
        outs = obs.get(timeout)
        return outs[0] if len(outs) == 1 else outs

    def outputAsync(self, *inputs, masks=None) -> _Observable:
        ins = [torch.as_tensor(_raw(x)) for x in inputs]
        obs = _Observable(ins, masks)
        self._enqueue(obs)
        return obs

    def _enqueue(self, obs):
        if self.loadBalanceMode == LoadBalanceMode.FIFO:
            self._workers[0].q.put(obs)
        else:
            with self._lock:
                w = self._workers[self._rr % len(self._workers)]
                self._rr += 1
            w.q.put(obs)

    def updateModel(self, model):
        """Swap in new parameters; workers re-replicate before their next request (ParallelInference.updateModel)."""
        self.model = model
        with self._lock:
            self._model_version += 1
        if self.inferenceMode == InferenceMode.INPLACE:
            for r in self._replicas:
                r.setParams(model.params().reshape(-1).to(r.ctx.device))

    def getCurrentModelsFromWorkers(self):
        if self.inferenceMode == InferenceMode.INPLACE:
            return list(self._replicas)
        return [w.replica for w in self._workers]

    def getWorkerCounter(self, i):
        return self._workers[i].counter

    def gemmStatistics(self):
        """Vendor-GEMM accounting of the workers (ops/streams.py): routed to the in-tree kernel / serialised by the device token /
        recorded into a HIP graph (such graphs replay under the token)."""
        tot = {"routed": 0, "serialized": 0, "captured": 0}
        for w in self._workers:
            for k, v in w.gemm_counts.items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def shutdown(self):
        if self._shutdown:
            return
        self._shutdown = True
        seen = set()
        for w in self._workers:
            if id(w.q) in seen and self.loadBalanceMode == LoadBalanceMode.FIFO:
                w.q.put(None)
                continue
            seen.add(id(w.q))
            w.q.put(None)
        for w in self._workers:
            w.join(timeout=5)

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass


class InplaceParallelInference(ParallelInference):
    """Caller-thread inference over a pool of replicas (InplaceParallelInference.java)."""

    def __init__(self, model, workers=None, devices=None, **kw):
        super().__init__(model, workers=workers, inferenceMode=InferenceMode.INPLACE, devices=devices, **kw)


# reference class names (parallelism/inference/observers/{BasicInferenceObservable,BatchedInferenceObservable}.java, InferenceWorker)
InferenceWorker = _Worker
BasicInferenceObservable = _Observable
BatchedInferenceObservable = _Observable
