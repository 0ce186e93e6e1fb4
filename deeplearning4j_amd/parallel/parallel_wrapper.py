"""ParallelWrapper — data-parallel training, re-created on RCCL (the class is absent from the
reference snapshot; API from upstream DL4J).

One process per GPU (launched by torchrun / torch.distributed.run). ``trainingMode``:
* SHARED_GRADIENTS (default): every iteration the flat gradient vector is all-reduced (averaged)
  in buckets — overlapped with backward in eager mode — then each rank applies the identical fused
  update, so parameters stay bitwise equal across ranks. BatchNorm running statistics ride in the
  gradient vector (BatchNormalization.java:170-186,321-353) and are averaged for free.
* AVERAGING: each rank trains ``averagingFrequency`` iterations locally, then parameters (and
  updater state if ``averageUpdaters``) are all-reduced — upstream's periodic parameter averaging.
Threshold-encoding knobs are accepted for source compatibility; on xGMI the dense all-reduce wins, so they only take
effect with ``compatEncoding(True)`` (sparse ±threshold messages with residuals, all-gathered and decoded by every rank).
"""
from __future__ import annotations

import torch

from . import comm


class TrainingMode:
    AVERAGING, SHARED_GRADIENTS, CUSTOM = "AVERAGING", "SHARED_GRADIENTS", "CUSTOM"


class ParallelWrapper:
    class Builder:
        def __init__(self, model):
            self._kw = {"model": model}

        def workers(self, n):
            self._kw["workers"] = n; return self

        def prefetchBuffer(self, n):
            self._kw["prefetchBuffer"] = n; return self

        def averagingFrequency(self, n):
            self._kw["averagingFrequency"] = n; return self

        def reportScoreAfterAveraging(self, b):
            self._kw["reportScoreAfterAveraging"] = b; return self

        def averageUpdaters(self, b):
            self._kw["averageUpdaters"] = b; return self

        def trainingMode(self, m):
            self._kw["trainingMode"] = m; return self

        def workspaceMode(self, m):
            return self

        def thresholdAlgorithm(self, a):
            self._kw["thresholdAlgorithm"] = a; return self

        def residualPostProcessor(self, r):
            self._kw["residualPostProcessor"] = r; return self

        def gradientsAccumulator(self, a):
            self._kw["gradientsAccumulator"] = a; return self

        def modelParamsSupplier(self, supplier):
            """Callable returning the flat parameter vector every replica starts from (upstream ModelParamsSupplier)."""
            self._kw["modelParamsSupplier"] = supplier; return self

        def updaterParamsSupplier(self, supplier):
            self._kw["updaterParamsSupplier"] = supplier; return self

        def compatEncoding(self, b=True):
            """Exchange threshold-encoded sparse updates with residuals (upstream SHARED_GRADIENTS wire behaviour) instead
            of the dense all-reduce. Uses ``gradientsAccumulator`` or builds an EncodedGradientsAccumulator from
            ``thresholdAlgorithm`` / ``residualPostProcessor``."""
            self._kw["compatEncoding"] = bool(b); return self

        def bucketBytes(self, n):
            self._kw["bucketBytes"] = n; return self

        def overlap(self, b):
            self._kw["overlap"] = b; return self

        def wireDataType(self, dt):
            """Gradient all-reduce wire format: None (the gradient vector's own type) or "BFLOAT16" (half the xGMI bytes)."""
            self._kw["wireDataType"] = dt; return self

        def watchdogTimeout(self, seconds):
            """Seconds without progress (while collectives are outstanding) after which this rank exits with
            comm.CommWatchdog.EXIT_CODE; 0 disables the watchdog."""
            self._kw["watchdogTimeout"] = seconds; return self

        def build(self):
            return ParallelWrapper(**self._kw)

    def __init__(self, model, workers=None, prefetchBuffer=2, averagingFrequency=1, reportScoreAfterAveraging=True,
                 averageUpdaters=True, trainingMode=TrainingMode.SHARED_GRADIENTS, thresholdAlgorithm=None,
                 residualPostProcessor=None, gradientsAccumulator=None, bucketBytes=None, overlap=True,
                 modelParamsSupplier=None, updaterParamsSupplier=None, compatEncoding=False, wireDataType=None, watchdogTimeout=120.0):
        self.model = model
        self.averagingFrequency = max(1, int(averagingFrequency))
        self.averageUpdaters = averageUpdaters
        self.trainingMode = trainingMode
        self.prefetchBuffer = prefetchBuffer
        self.reportScore = reportScoreAfterAveraging
        self.thresholdAlgorithm, self.residualPostProcessor = thresholdAlgorithm, residualPostProcessor
        self.overlap = overlap
        self._stop = False
        comm.init_process_group()
        self.world = comm.world_size()
        if workers is not None and workers != self.world and comm.rank() == 0 and self.world > 1:
            print("ParallelWrapper: workers(%d) ignored, world size is %d (one process per GPU)" % (workers, self.world))
        if not model._initialized:
            model.init()
        for supplier, target in ((modelParamsSupplier, "_params"), (updaterParamsSupplier, "_updater_state")):
            if supplier is not None:
                v = supplier() if callable(supplier) else supplier
                if v is not None:
                    v = getattr(v, "_t", v)
                    dst = getattr(model, target)
                    if v.numel() != dst.numel():
                        raise ValueError("%s supplier returned %d values, the model has %d" % (target, v.numel(), dst.numel()))
                    dst.copy_(torch.as_tensor(v).reshape(dst.shape).to(dst.device, dst.dtype))
        # replicas start identical: broadcast params + updater state from rank 0
        comm.broadcast_(model._params, 0)
        comm.broadcast_(model._updater_state, 0)
        model._refresh_replicas()
        wire = None
        if wireDataType is not None and str(wireDataType).upper() in ("BFLOAT16", "BF16", "TORCH.BFLOAT16"):
            wire = torch.bfloat16
        self.watchdog = comm.CommWatchdog(watchdogTimeout) if (watchdogTimeout and (self.world > 1 or comm._forced())) else None
        self.buckets = comm.GradientBuckets(model._grads, bucketBytes, wire_dtype=wire, watchdog=self.watchdog)
        self._local_iters = 0
        self.accumulator = None
        if compatEncoding or gradientsAccumulator is not None:
            from ..nd4j.compression import EncodedGradientsAccumulator
            self.accumulator = gradientsAccumulator or EncodedGradientsAccumulator(thresholdAlgorithm, residualPostProcessor)
        if trainingMode == TrainingMode.SHARED_GRADIENTS:
            model._grad_sync = self._sync_gradients
            # encoded exchange: variable-length messages whose sizes are read on the host, so no bucket overlap; with HIP-graph
            # training it runs eagerly between the captured forward/backward graph and the captured update graph
            if self.accumulator is None and overlap and (self.world > 1 or comm._forced()):
                model._on_backward_progress = self.buckets.ready
                model._on_backward_begin = self.buckets.begin
                model._dp_buckets = self.buckets      # lets the HIP-graph step cut its capture at bucket boundaries

    # gradient sharing
    def _sync_gradients(self):
        if self.accumulator is not None:
            g = self.model._grads
            self.accumulator.applyUpdate(self.accumulator.storeUpdate(g), g)
            return
        if self.world == 1 and not comm._forced():
            return
        if (self.overlap and getattr(self.model, "_on_backward_progress", None) is not None
                and self.model._graph_step is None):
            self.buckets.finish()
        else:
            self.buckets.reduce_all()

    def _maybe_average_params(self):
        self._local_iters += 1
        if self.trainingMode != TrainingMode.AVERAGING or self._local_iters % self.averagingFrequency:
            return
        m = self.model
        comm.all_reduce_avg_(m._params)
        if self.averageUpdaters and m._state_size:
            comm.all_reduce_avg_(m._updater_state)
        m._refresh_replicas()

    def fit(self, iterator, numEpochs=1):
        """Each rank consumes its own shard: batch k of every complete round of ``world`` batches goes to rank k % world.
        A ragged tail is dropped on every rank (ShardedDataSetIterator), so the iteration count is uniform and no
        all-reduce is ever left without its peers."""
        from ..datasets.iterators import ShardedDataSetIterator
        m = self.model
        src = iterator if self.world == 1 else ShardedDataSetIterator(iterator, comm.rank(), self.world)
        for _ in range(numEpochs):
            if src.resetSupported():
                src.reset()
            while src.hasNext() and not self._stop:
                if self.watchdog is not None:
                    self.watchdog.busy(True)
                m.fit(src.next())
                self._maybe_average_params()
                if self.watchdog is not None:
                    self.watchdog.busy(False)
            m.incrementEpochCount()
        return self

    def setListeners(self, *ls):
        self.model.setListeners(*ls)

    def stopFit(self):
        self._stop = True

    def shutdown(self):
        if self.watchdog is not None:
            self.watchdog.stop()
        self.model._grad_sync = None
        self.model._on_backward_progress = None
        self.model._dp_buckets = None

    close = shutdown

    def params_checksum(self):
        """Bit-exact digest of the flat parameter vector: int64 wrap-around sum of the raw 32-bit words plus a position-weighted
        sum (catches permuted values). One [2] int64 tensor per rank instead of the whole vector."""
        p = self.model._params.detach().contiguous()
        w = p.view(torch.int32).to(torch.int64) if p.element_size() == 4 else p.view(torch.int16).to(torch.int64)
        idx = torch.arange(1, w.numel() + 1, device=w.device, dtype=torch.int64) & 0xFFFF
        return torch.stack([w.sum(), (w * idx).sum()])

    def checksums_in_sync(self):
        """All-gather of :meth:`params_checksum`; True when every rank holds bit-identical parameters."""
        d = self.params_checksum()
        if self.world == 1:
            return True
        ds = comm.all_gather(d)
        return all(bool(torch.equal(x, ds[0])) for x in ds[1:])

    def comm_report(self, steps=None):
        """What bench.py prints so an N-GPU run can be verified from its output line alone: backend, the world size an all-reduce
        of ones actually saw, bucket layout, wire bytes per step and the parameter digest agreement."""
        import torch.distributed as dist
        seen = 1
        backend = "none"
        if dist.is_initialized():
            backend = dist.get_backend()
            one = torch.ones(1, device=self.model._params.device, dtype=torch.float32)
            dist.all_reduce(one, op=dist.ReduceOp.SUM)
            seen = int(round(float(one.item())))
        b = self.buckets
        rep = {"backend": "rccl" if backend == "nccl" and torch.version.hip else backend, "world_size_seen": seen,
               "buckets": len(b.ranges), "bucket_mb": round(b.bucket_bytes / float(1 << 20), 2),
               "head_mb": round(b.head_bytes / float(1 << 20), 2),
               "wire_dtype": str(b.wire_dtype or self.model._grads.dtype).replace("torch.", ""),
               "bytes_per_step": int(b.bytes_launched // steps) if steps else int(b.bytes_launched),
               "collectives_per_step": (b.collectives // steps) if steps else b.collectives,
               "params_in_sync": bool(self.checksums_in_sync())}
        return rep

    def params_in_sync(self, atol=0.0):
        """Debug check: parameters bitwise (or atol) equal across ranks."""
        if self.world == 1:
            return True
        ps = comm.all_gather(self.model._params)
        return all(bool((p - ps[0]).abs().max() <= atol) for p in ps[1:])
