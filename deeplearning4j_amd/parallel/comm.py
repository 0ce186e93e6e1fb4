"""Collective communication layer: RCCL over xGMI through torch.distributed (backend "nccl" IS RCCL
on ROCm); gloo for CPU tests. One process per GPU.

Replaces the reference's single-kernel peer averaging (`average(..., propagate=true)`,
libnd4j/include/loops/cuda/specials/averagingKernel.cu:29-76, NativeOps.cu:1677-1704: ONE device
reads all peers over P2P, no overlap) with bucketed all-reduce of the flat gradient vector. Bucket
sizes are chosen for xGMI's point-to-point links (7 x ~153 GB/s per GPU): few, large messages
(default 32 MiB) so the per-collective launch/sync cost is amortised and RCCL can stripe over links.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend: Optional[str] = None, timeout_s: int = 600):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* env (no-op for world size 1 unless forced)."""
    if dist.is_initialized():
        return
    rank, local_rank, world = env_world()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    import datetime
    kw = {}
    if backend == "nccl" and torch.cuda.is_available():
        kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
    if backend == "nccl":
        # RCCL error channel (ncclCommGetAsyncError): c10d's own watchdog thread polls the communicator and, in
        # this mode, tears the process down on an asynchronous error or a collective that exceeds the timeout instead of hanging
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        os.environ.setdefault("TORCH_NCCL_DUMP_ON_TIMEOUT", "0")
    dist.init_process_group(backend=backend, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=timeout_s), **kw)


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def broadcast_(t: torch.Tensor, src=0):
    if world_size() > 1:
        dist.broadcast(t, src)
    return t


def _forced():
    """DL4J_AMD_FORCE_COLLECTIVES=1: issue the collectives even at world size 1 (single-GPU rehearsal of the RCCL call pattern)."""
    return os.environ.get("DL4J_AMD_FORCE_COLLECTIVES") == "1" and dist.is_initialized()


def all_reduce_avg_(t: torch.Tensor, async_op=False):
    """In-place average across ranks."""
    if world_size() == 1 and not _forced():
        return None
    if dist.get_backend() == "nccl":
        # one rank (the forced rehearsal): the average IS the sum, and SUM spares RCCL's single-rank pre-multiply kernel
        # (oneRankReduce, 31 us per bucket in profiles/r3_final_steps.md's data-parallel twin)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM if world_size() == 1 else dist.ReduceOp.AVG, async_op=async_op)
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)
    if async_op:
        return _ScaledWork(w, t, 1.0 / world_size())
    t.mul_(1.0 / world_size())
    return None


class _ScaledWork:
    def __init__(self, work, t, s):
        self.work, self.t, self.s = work, t, s

    def wait(self):
        self.work.wait()
        self.t.mul_(self.s)


def all_reduce_sum_(t, async_op=False):
    if world_size() == 1:
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


def all_gather(t: torch.Tensor) -> List[torch.Tensor]:
    if world_size() == 1:
        return [t]
    out = [torch.empty_like(t) for _ in range(world_size())]
    dist.all_gather(out, t)
    return out


def reduce_scatter_avg(flat: torch.Tensor) -> torch.Tensor:
    """Reduce-scatter of a flat vector whose length is divisible by world size; returns this rank's shard."""
    w = world_size()
    if w == 1:
        return flat
    shard = torch.empty(flat.numel() // w, dtype=flat.dtype, device=flat.device)
    dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM)
    return shard.mul_(1.0 / w)


class CommWatchdog:
    """Host-side failure detector for collective training (both backends): the training loop ``tick()``s once per iteration and
    registers the async collectives it is waiting on; a daemon thread polls them. A work that reports an exception, or a loop that
    makes no progress for ``timeout_s`` while collectives are outstanding (a peer died, a link is down), ends THIS process with
    exit code ``EXIT_CODE`` after printing what it was waiting on — a clean non-zero exit the launcher can act on, instead of a
    600 s hang per surviving rank. Nothing is re-executed in place (no exec from a process that owns a GPU)."""
    EXIT_CODE = 75

    def __init__(self, timeout_s: float = 120.0, poll_s: float = 0.5, on_failure=None):
        import threading
        import time
        self.timeout_s, self.poll_s = float(timeout_s), float(poll_s)
        self.on_failure = on_failure
        self._last = time.monotonic()
        self._works = []
        self._busy = False
        self._stop = threading.Event()
        self._lock = threading.Lock()
        self.failure = None
        self._thread = threading.Thread(target=self._run, name="dl4j-comm-watchdog", daemon=True)
        self._thread.start()

    def tick(self):
        import time
        self._last = time.monotonic()

    def busy(self, flag: bool):
        """True while the loop is inside a region that depends on its peers (between iteration start and gradient sync end)."""
        self._busy = bool(flag)
        self.tick()

    def watch(self, work):
        if work is not None:
            with self._lock:
                self._works.append(getattr(work, "work", work))

    def clear(self):
        with self._lock:
            self._works = []

    def _run(self):
        import time
        while not self._stop.wait(self.poll_s):
            reason = None
            with self._lock:
                works = list(self._works)
            for w in works:
                exc = self._failed(w)
                if exc is not None:
                    reason = "collective failed: %r" % (exc,)
                    break
            # "no progress" only counts while a watched collective is still incomplete: a long first iteration (native build,
            # graph capture, warm-up), a slow listener or an evaluation pass with nothing outstanding is not a dead peer
            pending = 0
            for w in works:
                try:
                    if not w.is_completed():
                        pending += 1
                except Exception:
                    pending += 1
            if reason is None and self._busy and pending > 0 and time.monotonic() - self._last > self.timeout_s:
                reason = "no progress for %.0f s with %d collective(s) outstanding (a peer rank is gone or a link is down)" % (
                    time.monotonic() - self._last, pending)
            if reason is not None:
                self.failure = reason
                self._fail(reason)
                return

    @staticmethod
    def _failed(w):
        """None while ``w`` is pending or has completed without an error, else what is known about the failure.
        ``Work.exception()`` cannot be trusted on its own: torch 2.10's binding has no converter for its return type, so the call
        raises TypeError for EVERY completed work, failed or not (seen in a 2-rank rehearsal, where the poll caught a finished work
        and the watchdog ended two healthy ranks). A TypeError from it therefore means "ask differently": ``is_success()``."""
        try:
            if not w.is_completed():
                return None
        except Exception as e:                        # querying a broken communicator may itself raise
            return e
        try:
            return w.exception()
        except TypeError:
            pass
        except Exception as e:
            return e
        try:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ok = w.is_success()
        except Exception:
            return None                               # no way to ask: c10d's own error handling (init_process_group) stays in charge
        return None if ok else RuntimeError("the collective completed without success")

    def _fail(self, reason):
        import sys
        msg = "[dl4j-comm-watchdog] rank %d/%d: %s -> exiting with code %d" % (rank(), world_size(), reason, self.EXIT_CODE)
        print(msg, file=sys.stderr, flush=True)
        if self.on_failure is not None:
            self.on_failure(reason)
            return
        os._exit(self.EXIT_CODE)

    def stop(self):
        self._stop.set()


class GradientBuckets:
    """Contiguous ranges of the flat gradient vector reduced as separate collectives.

    Backward fills the flat vector from its tail (last layer) to its head, so bucket ``k`` can be
    launched as soon as every parameter at offset >= bucket_start has its gradient — the overlap
    hook calls :meth:`ready` with the lowest finished offset."""

    def __init__(self, flat_grads: torch.Tensor, bucket_bytes: Optional[int] = None, head_bytes: Optional[int] = None, wire_dtype=None,
                 watchdog: Optional["CommWatchdog"] = None):
        # sizes are tunable without code changes (the 8-GPU node is only seen by the round-end driver): DL4J_AMD_DP_BUCKET_MB /
        # DL4J_AMD_DP_HEAD_MB in MiB; an explicit argument wins
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get("DL4J_AMD_DP_BUCKET_MB", "32")) * (1 << 20))
        if head_bytes is None:
            head_bytes = int(float(os.environ.get("DL4J_AMD_DP_HEAD_MB", "4")) * (1 << 20))
        self.bucket_bytes, self.head_bytes = int(bucket_bytes), int(head_bytes)
        self.bytes_launched = 0          # wire bytes handed to collectives since construction (bench.py reports per step)
        self.collectives = 0
        # wire_dtype=torch.bfloat16: the buckets travel as bf16 (half the xGMI bytes; every rank rounds its own gradient once,
        # RCCL sums in bf16, the average lands back in the fp32 vector) — an option, the default wire format is the vector's own
        self.wire_dtype = wire_dtype if (wire_dtype is not None and wire_dtype != flat_grads.dtype) else None
        self._stage = torch.empty(flat_grads.numel(), dtype=self.wire_dtype, device=flat_grads.device) if self.wire_dtype else None
        self.watchdog = watchdog
        n = flat_grads.numel()
        per = max(1, bucket_bytes // flat_grads.element_size())
        self.flat = flat_grads
        bounds = list(range(0, n, per)) + [n]
        # The HEAD bucket (offset 0: the first layers) completes last — its collective starts when backward has finished
        # and is fully exposed — so it is kept small: over xGMI a 4 MiB ring all-reduce costs ~50 us where 32 MiB costs ~0.3 ms.
        head = max(1, head_bytes // flat_grads.element_size())
        if len(bounds) >= 2 and bounds[1] > 2 * head:
            bounds.insert(1, head)
        self.ranges = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
        self._next = len(self.ranges) - 1
        self._works = []
        self.comm_stream = torch.cuda.Stream() if flat_grads.is_cuda else None

    def begin(self):
        self._next = len(self.ranges) - 1
        self._works = []

    def ready(self, min_done_offset: int):
        """Launch every not-yet-launched bucket whose range is complete."""
        while self._next >= 0 and self.ranges[self._next][0] >= min_done_offset:
            if self.flat.is_cuda:
                from ..ops import streams
                streams.join()                  # wgrads issued on the side stream are part of this bucket
            self._launch(self._next)
            self._next -= 1

    def count_ready(self, min_done_offset: int) -> int:
        """Number of buckets (counted from the tail) whose range lies entirely at or above ``min_done_offset``."""
        n = 0
        for s, _ in reversed(self.ranges):
            if s < min_done_offset:
                break
            n += 1
        return n

    def launch_next(self, n: int = 1):
        """Launch the next ``n`` buckets in tail-to-head order (captured-graph path: readiness was decided at capture time)."""
        while n > 0 and self._next >= 0:
            self._launch(self._next)
            self._next -= 1
            n -= 1

    def _launch(self, k):
        s, e = self.ranges[k]
        seg = self.flat[s:e]
        st = self._stage[s:e] if self._stage is not None else None
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                if st is not None:
                    st.copy_(seg)
                w = all_reduce_avg_(st if st is not None else seg, async_op=True)
                if st is not None:
                    # An async collective runs on the process group's own stream, not on comm_stream: only Work.wait() orders
                    # the CURRENT stream (comm_stream here) after it. On the stream-ordered backend wait() does not block the
                    # host, so the copy-back is queued behind the reduce and the next bucket can still be launched at once.
                    if w is not None:
                        w.wait()
                    seg.copy_(st)
        else:
            if st is not None:
                st.copy_(seg)
            w = all_reduce_avg_(st if st is not None else seg, async_op=True)
            if st is not None:                 # host backend (gloo): completion is not stream-ordered
                if w is not None:
                    w.wait()
                    w = None
                seg.copy_(st)
        self.bytes_launched += (e - s) * (st if st is not None else seg).element_size()
        self.collectives += 1
        if w is not None:
            self._works.append(w)
            if self.watchdog is not None:
                self.watchdog.watch(w)
        if self.watchdog is not None:
            self.watchdog.tick()

    def finish(self):
        self.ready(0)
        for w in self._works:
            w.wait()
        if self.watchdog is not None:
            self.watchdog.clear()
            self.watchdog.tick()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.begin()          # self-resetting: the next iteration starts from the tail bucket again

    def reduce_all(self):
        """Non-overlapped path: reduce every bucket on the current stream."""
        for s, e in self.ranges:
            self.bytes_launched += (e - s) * (self._stage if self._stage is not None else self.flat).element_size()
            self.collectives += 1
            if self._stage is not None:
                st = self._stage[s:e]
                st.copy_(self.flat[s:e])
                all_reduce_avg_(st)
                self.flat[s:e].copy_(st)
            else:
                all_reduce_avg_(self.flat[s:e])
