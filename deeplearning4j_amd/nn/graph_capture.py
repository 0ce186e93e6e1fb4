"""HIP-graph capture of the training iteration (forward + loss + backward [+ fused updater]).

The reference drains both CUDA streams before every custom op (CudaExecutioner.java:1621-1625,1803),
so an iteration is thousands of synchronous launches. Here, for static shapes, the iteration is
captured once into hipGraphs and replayed: launch overhead disappears and the GPU runs
back-to-back kernels. Everything the iteration reads that changes between steps lives in device
memory (input/label staging buffers, updater iteration counter, learning-rate segment table, learning-rate SCHEDULE table).

Data-parallel mode splits the capture: A = forward+backward, B = fused update, with the RCCL
gradient all-reduce issued eagerly between the replays. When the wrapper provides gradient buckets
(``net._dp_buckets``) A itself is cut into segments at the points of backward where a bucket of the
flat gradient vector becomes final: after replaying segment k the bucket's all-reduce is launched on
the communication stream and runs over xGMI while segment k+1 computes — the same overlap the eager
path gets from the backward-progress hook, without giving up graph replay.
"""
from __future__ import annotations

import torch


class CutPlanner:
    """Where the captured backward is cut so gradient buckets can be handed to RCCL while the rest of backward replays.

    Every cut costs a graph-launch boundary (the 1-rank RCCL rehearsal measured +5 % with one cut per bucket), so at most
    ``DL4J_AMD_DP_MAX_CUTS`` cuts are made (default 2), each releasing a GROUP of finished buckets; the head bucket (offset 0) is
    never worth a cut — nothing is left to overlap it with, ``finish()`` launches it. One rank has nobody to overlap with: no
    cuts at all unless ``DL4J_AMD_DP_CUTS_AT_ONE_RANK=1`` (rehearsal of the cut path itself).

    ``buckets`` needs ``ranges`` and ``count_ready(min_done_offset)`` only (unit-tested on CPU with a stand-in)."""

    def __init__(self, buckets, max_cuts=None, world=None):
        import os
        if world is None:
            from ..parallel import comm
            world = comm.world_size()
        if max_cuts is None:
            max_cuts = int(os.environ.get("DL4J_AMD_DP_MAX_CUTS", "2"))
        if world == 1 and os.environ.get("DL4J_AMD_DP_CUTS_AT_ONE_RANK") != "1":
            max_cuts = 0
        self.buckets = buckets
        self.cuttable = max(0, len(buckets.ranges) - 1)
        self.group = max(1, -(-self.cuttable // max_cuts)) if max_cuts > 0 else self.cuttable + 1
        self.launched = 0
        self.cuts = 0

    def release(self, min_done_offset) -> int:
        """Number of buckets to release with a cut NOW (0: keep capturing)."""
        nb = min(self.buckets.count_ready(min_done_offset), self.cuttable) - self.launched
        if nb >= self.group or (nb > 0 and self.launched + nb == self.cuttable and self.group <= self.cuttable):
            self.launched += nb
            self.cuts += 1
            return nb
        return 0


class CapturedTrainStep:
    def __init__(self, net, warmup=3):
        self.net = net
        self.warmup = warmup
        self.graph_a = self.graph_b = None
        self.segments = None
        self._cap_stream = None
        self._x = self._y = None
        self._seen = 0
        self._sig = None

    def _compute(self, x, y):
        net = self.net
        if hasattr(net, "topologicalOrder"):
            net.computeGradientAndScore([x], [y])
        else:
            net.computeGradientAndScore(x, y)

    def run(self, x, y):
        net = self.net
        staged = False
        if self._x is not None and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 \
                and tuple(x.shape) == tuple(self._x.shape) and net.ctx.channels_last:
            # image batch straight into the static input: NCHW fp32 -> NHWC compute type in one kernel, no staging copy
            from ..ops import spatial as SP
            staged = SP.nchw_f32_to_nhwc(x, self._x.dtype, out=self._x) is not None
        if not staged:
            x = net._prep_features(x)
        y = net._prep_labels(y)
        if staged and (self._sig is None or self._sig[2:] != (tuple(y.shape), y.dtype)):
            x, staged = net._prep_features(x), False          # the labels changed shape: buffers are rebuilt below from prepared inputs
        sig = ((tuple(self._x.shape), self._x.dtype) if staged else (tuple(x.shape), x.dtype)) + (tuple(y.shape), y.dtype)
        if self._sig != sig:
            self.graph_a = self.graph_b = None
            self._seen, self._sig = 0, sig
            self._x = torch.empty_like(x)
            self._y = torch.empty_like(y)
        if not staged:
            self._x.copy_(x, non_blocking=True)
        self._y.copy_(y, non_blocking=True)
        if self.graph_a is None:
            if self._seen < self.warmup:
                self._seen += 1
                hook, net._on_backward_progress = net._on_backward_progress, None
                self._compute(self._x, self._y)
                net._on_backward_progress = hook
                net._apply_update(net._last_batch)
                return
            # thread_local: the RCCL watchdog thread's event queries must not invalidate the capture
            torch.cuda.synchronize()
            buckets = getattr(net, "_dp_buckets", None)
            if buckets is not None and net._grad_sync is not None:
                self.segments = self._capture_segmented(buckets)
                ga = self.segments[0][0]
            else:
                net._on_backward_progress = None   # eager overlap hooks cannot run inside a captured region
                self.segments = None
                ga = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, stream=self._capture_stream(), capture_error_mode="thread_local"):
                    self._compute(self._x, self._y)
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=ga.pool(), stream=self._capture_stream(), capture_error_mode="thread_local"):
                net._apply_update(net._last_batch, sync=False)
            self.graph_a, self.graph_b = ga, gb
        if self.segments is not None:
            buckets = net._dp_buckets
            buckets.begin()
            for g, nb in self.segments:
                g.replay()
                if nb:
                    buckets.launch_next(nb)
            buckets.finish()
        else:
            self.graph_a.replay()
            if net._grad_sync is not None:
                net._grad_sync()
        if net._has_schedule:
            # learning-rate schedules live in a device table indexed by the device iteration counter; the host only tops the
            # table up when its window runs out (ops/updater.SegmentedUpdater.ensure_schedule)
            net._updater.ensure_schedule(net.iterationCount, net.epochCount)
        self.graph_b.replay()

    def _capture_stream(self):
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream()
            # persistent per-stream scratch the kernels need must exist BEFORE the capture starts (a buffer allocated while
            # capturing would live in the graph's private pool): the ticket counters of the in-kernel BatchNorm reductions
            from ..ops import batchnorm as BN
            with torch.cuda.stream(self._cap_stream):
                BN.ticket_ws(self.net.ctx.device)
        return self._cap_stream

    def _capture_segmented(self, buckets):
        """Capture forward+backward as a chain of graphs cut where gradient buckets become final.

        Returns [(graph, n_buckets_ready_after_it), ...]; all graphs share one memory pool and are replayed in capture order."""
        net = self.net
        segs = []
        state = {"g": None, "pool": torch.cuda.graph_pool_handle()}

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=state["pool"], capture_error_mode="thread_local")
            state["g"] = g

        planner = CutPlanner(buckets)

        def progress(min_done_offset):
            nb = planner.release(min_done_offset)
            if nb > 0:
                from ..ops import streams
                streams.join()                  # side-stream wgrads belong to this segment: join before the cut
                state["g"].capture_end()
                segs.append((state["g"], nb))
                begin()

        hook, begin_hook = net._on_backward_progress, net._on_backward_begin
        net._on_backward_progress, net._on_backward_begin = progress, None
        side = self._capture_stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                begin()
                try:
                    self._compute(self._x, self._y)
                finally:
                    state["g"].capture_end()
                segs.append((state["g"], 0))
        finally:
            net._on_backward_progress, net._on_backward_begin = hook, begin_hook
        torch.cuda.current_stream().wait_stream(side)
        return segs


import threading

_CAPTURE_LOCK = threading.Lock()


class CapturedInference:
    """HIP-graph replay of the inference forward pass, one graph per input signature (shapes + dtypes).

    Small-batch inference is launch bound (the zoo ResNet-50 forward is ~220 launches: 2.8 ms at batch 32 and 3.0 ms at
    batch 256 when issued eagerly), which is exactly the serving case of ParallelInference (reference:
    parallelism/ParallelInference.java). The first ``warmup`` calls of a signature run eagerly (lazy allocations, library
    handles); then the forward is captured once and later calls copy the inputs into the static buffers and replay.
    """

    def __init__(self, warmup=2, max_graphs=8):
        self.warmup, self.max_graphs = warmup, max_graphs
        self.entries = {}
        self.disabled = False
        self._stream = None

    def run(self, fwd, xs):
        """fwd(list of tensors) -> list of tensors. Returns the outputs (fresh tensors), or None when the caller should run eagerly."""
        if self.disabled or not xs or not all(isinstance(x, torch.Tensor) and x.is_cuda for x in xs):
            return None
        key = tuple((tuple(x.shape), x.dtype) for x in xs)
        e = self.entries.get(key)
        if e is None:
            if len(self.entries) >= self.max_graphs:
                return None
            e = self.entries[key] = {"seen": 0, "graph": None}
        if e["graph"] is None:
            e["seen"] += 1
            if e["seen"] <= self.warmup:
                return None
            try:
                if self._stream is None:
                    self._stream = torch.cuda.Stream()
                static_in = [x.clone() for x in xs]
                with _CAPTURE_LOCK:           # several host threads (ParallelInference workers) capture one at a time
                    self._stream.wait_stream(torch.cuda.current_stream())
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self._stream, capture_error_mode="thread_local"):
                        static_out = fwd(static_in)
                e.update(graph=g, static_in=static_in, static_out=static_out)
            except Exception as ex:           # an op that synchronises with the host: stay eager for this network
                import warnings
                warnings.warn("inference graph capture disabled: %s" % (ex,))
                self.disabled = True
                torch.cuda.current_stream().synchronize()
                self._retired = list(self.entries.values())     # graphs of other signatures are kept alive, not destroyed mid-flight
                self.entries = {}
                return None
        for s, x in zip(e["static_in"], xs):
            s.copy_(x)
        e["graph"].replay()
        return [o.clone() for o in e["static_out"]]
