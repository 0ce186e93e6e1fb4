"""HIP-graph capture of the training iteration (forward + loss + backward [+ fused updater]).

The reference drains both CUDA streams before every custom op (CudaExecutioner.java:1621-1625,1803),
so an iteration is thousands of synchronous launches. Here, for static shapes, the iteration is
captured once into hipGraphs and replayed: launch overhead disappears and the GPU runs
back-to-back kernels. Everything the iteration reads that changes between steps lives in device
memory (input/label staging buffers, updater iteration counter, learning-rate segment table).

Data-parallel mode splits the capture in two (A = forward+backward, B = fused update) with the
RCCL gradient all-reduce issued eagerly between the replays on the same stream.
"""
from __future__ import annotations

import torch


class CapturedTrainStep:
    def __init__(self, net, warmup=3):
        self.net = net
        self.warmup = warmup
        self.graph_a = self.graph_b = None
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
        x = net._prep_features(x)
        y = net._prep_labels(y)
        sig = (tuple(x.shape), x.dtype, tuple(y.shape), y.dtype)
        if self._sig != sig:
            self.graph_a = self.graph_b = None
            self._seen, self._sig = 0, sig
            self._x = torch.empty_like(x)
            self._y = torch.empty_like(y)
        self._x.copy_(x, non_blocking=True)
        self._y.copy_(y, non_blocking=True)
        if self.graph_a is None:
            if self._seen < self.warmup or net._has_schedule:
                # LR schedules re-upload the segment table from the host every step: stay eager
                self._seen += 1
                hook, net._on_backward_progress = net._on_backward_progress, None
                self._compute(self._x, self._y)
                net._on_backward_progress = hook
                net._apply_update(net._last_batch)
                return
            net._on_backward_progress = None   # overlap hooks cannot run inside a captured region
            # thread_local: the RCCL watchdog thread's event queries must not invalidate the capture
            torch.cuda.synchronize()
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                self._compute(self._x, self._y)
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
                net._apply_update(net._last_batch, sync=False)
            self.graph_a, self.graph_b = ga, gb
        self.graph_a.replay()
        if net._grad_sync is not None:
            net._grad_sync()
        self.graph_b.replay()
