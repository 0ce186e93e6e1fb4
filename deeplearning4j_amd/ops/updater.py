"""Segmented fused updater over flat params / grads / state (see native/csrc/updaters.hip).

``SegmentedUpdater`` is built once from the network's UpdaterBlocks (contiguous parameter ranges
sharing one IUpdater config — reference: deeplearning4j-nn/.../nn/updater/BaseMultiLayerUpdater.java:83-141)
and applies {÷minibatch, L1/L2, updater, weight decay, params -= update} in ONE launch on GPU.
The CPU path applies the identical formulas with PyTorch ops per segment (oracle for tests).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from ..native import lib as L

_UPD_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
KIND = {"NoOp": 0, "Sgd": 1, "Nesterovs": 2, "Adam": 3, "RmsProp": 4, "AdaGrad": 5, "AdaDelta": 6, "AdaMax": 7,
        "Nadam": 8, "AMSGrad": 9, "AdaBelief": 10}
STATE_MULT = {0: 0, 1: 0, 2: 1, 3: 2, 4: 1, 5: 1, 6: 2, 7: 2, 8: 2, 9: 3, 10: 2}
CHUNK = 2048


@dataclass
class Segment:
    off: int
    length: int
    state_off: int
    kind: int
    lr: float = 0.0
    a: float = 0.0
    b: float = 0.0
    eps: float = 0.0
    l1: float = 0.0
    l2: float = 0.0
    wd: float = 0.0
    divide_by_batch: bool = True
    wd_apply_lr: bool = True
    update_only: bool = False

    def pack(self) -> bytes:
        flags = (1 if self.divide_by_batch else 0) | (2 if self.wd_apply_lr else 0) | (4 if self.update_only else 0)
        return struct.pack("<qqqiifffffffxxxx", self.off, self.length, self.state_off, self.kind, flags, self.lr,
                           self.a, self.b, self.eps, self.l1, self.l2, self.wd)


def _block_tables(lengths):
    seg_ids, chunks = [], []
    for si, n in enumerate(lengths):
        nb = (n + CHUNK - 1) // CHUNK
        seg_ids.append(np.full(nb, si, dtype=np.int32))
        chunks.append(np.arange(nb, dtype=np.int32))
    if not seg_ids:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    return np.concatenate(seg_ids), np.concatenate(chunks)


class SegmentedUpdater:
    def __init__(self, segments: List[Segment], device):
        self.segments = segments
        self.device = torch.device(device)
        self._tables_dirty = True
        self._seg_dev = None
        bs, bc = _block_tables([s.length for s in segments])
        self.num_blocks = int(bs.shape[0])
        self._blk_seg = torch.from_numpy(bs).to(self.device)
        self._blk_chunk = torch.from_numpy(bc).to(self.device)
        self.iter_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        # device-resident learning-rate schedule (GPU): [window, nseg] table + first iteration it covers
        self._sched_fn = None
        self._sched_table = None
        self._sched_base = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._sched_host = None          # (base, epoch) the table was filled for
        self._upload()

    def _upload(self):
        raw = b"".join(s.pack() for s in self.segments)
        arr = torch.frombuffer(bytearray(raw), dtype=torch.uint8) if raw else torch.zeros(0, dtype=torch.uint8)
        if self._seg_dev is None or self._seg_dev.numel() != arr.numel():
            self._seg_dev = arr.to(self.device)
        else:
            self._seg_dev.copy_(arr, non_blocking=True)
        self._tables_dirty = False

    def set_learning_rates(self, lrs):
        """Update per-segment learning rates (ISchedule support); table re-uploaded lazily."""
        changed = False
        for s, lr in zip(self.segments, lrs):
            if lr is not None and s.lr != lr:
                s.lr = float(lr)
                changed = True
        if changed:
            self._tables_dirty = True

    def set_iteration(self, it: int):
        self.iter_dev.fill_(int(it))

    SCHED_WINDOW = 2048

    def set_schedule(self, fn):
        """``fn(iteration, epoch) -> [lr or None per segment]``. On the GPU the next SCHED_WINDOW iterations are evaluated once and
        kept in device memory (the updater kernel indexes the table with its device iteration counter), so a captured training
        step stays valid while the learning rate changes; the CPU path sets the rates per step."""
        self._sched_fn = fn
        self._sched_host = None

    def ensure_schedule(self, iteration: int, epoch: int):
        """Host side of the device schedule: (re)fill the table when ``iteration`` left the window or the epoch changed. Call it
        before every step / graph replay; it is a no-op almost always. Also mirrors the current rates into ``segments``."""
        if self._sched_fn is None:
            return
        cur = self._sched_fn(iteration, epoch)
        for s, lr in zip(self.segments, cur):
            if lr is not None:
                s.lr = float(lr)
        if self.device.type != "cuda":
            self._tables_dirty = True
            return
        h = self._sched_host
        if h is not None and h[1] == epoch and h[0] <= iteration < h[0] + self.SCHED_WINDOW:
            return
        nseg = len(self.segments)
        rows = np.full((self.SCHED_WINDOW, nseg), np.nan, dtype=np.float32)
        for k in range(self.SCHED_WINDOW):
            for j, lr in enumerate(self._sched_fn(iteration + k, epoch)):
                if lr is not None:
                    rows[k, j] = lr
        t = torch.from_numpy(rows)
        if self._sched_table is None:
            self._sched_table = t.to(self.device)
        else:
            self._sched_table.copy_(t, non_blocking=False)
        self._sched_base.fill_(int(iteration))
        self._sched_host = (iteration, epoch)

    def step(self, params, grads, state, batch_size: int, write_update: bool = False, shadow=None):
        if self._tables_dirty:
            self._upload()
        inv_b = 1.0 / float(max(batch_size, 1))
        if L.use_hip(params) and params.dtype in _UPD_DT and grads.dtype == params.dtype and state.dtype == params.dtype:
            sched = self._sched_table if self._sched_fn is not None else None
            L.check(L.hip().dl4j_segmented_update_ex(
                L.ptr(params), L.ptr(grads), L.ptr(state), L.ptr(shadow if params.dtype == torch.float32 else None),
                L.ptr(self._seg_dev), L.ptr(self._blk_seg), L.ptr(self._blk_chunk), L.c_int(self.num_blocks), L.ptr(self.iter_dev),
                L.c_float(inv_b), L.c_int(1 if write_update else 0), L.c_int(1), L.c_int(_UPD_DT[params.dtype]), L.ptr(sched),
                L.ptr(self._sched_base), L.c_int(self.SCHED_WINDOW if sched is not None else 0), L.c_int(len(self.segments)),
                L.stream()), "segmented_update")
            return
        t = float(self.iter_dev.item() + 1)
        for s in self.segments:
            apply_segment_reference(s, params, grads, state, inv_b, t, write_update)
        self.iter_dev += 1


def apply_segment_reference(s: Segment, params, grads, state, inv_b, t, write_update=False):
    """PyTorch implementation of one segment (CPU path, fp64 gradient checks and GPU test oracle)."""
    n = s.length
    g = grads[s.off:s.off + n].clone()
    w = params[s.off:s.off + n] if not s.update_only else torch.zeros_like(g)
    if s.divide_by_batch:
        g = g * inv_b
    if s.l1:
        g = g + s.l1 * torch.sign(w)
    if s.l2:
        g = g + s.l2 * w
    st = state[s.state_off:s.state_off + STATE_MULT[s.kind] * n] if STATE_MULT[s.kind] else None
    k = s.kind
    eps = 1e-7 if (s.eps == 0.0 and k >= 3) else s.eps

    def fix(alpha):
        if alpha != alpha or alpha == 0.0 or alpha in (float("inf"), float("-inf")):
            return eps
        return alpha
    if k == 0:
        u = g
    elif k == 1:
        u = s.lr * g
    elif k == 2:
        vprev = s.a * st[:n]
        v = vprev - s.lr * g
        st[:n] = v
        u = vprev - (s.a + 1.0) * v
    elif k in (3, 9, 10):
        alpha = fix(s.lr * (1.0 - s.b ** t) ** 0.5 / (1.0 - s.a ** t)) if (1.0 - s.a ** t) != 0 else eps
        m = s.a * st[:n] + (1.0 - s.a) * g
        st[:n] = m
        if k == 3:
            v = s.b * st[n:2 * n] + (1.0 - s.b) * g * g
            st[n:2 * n] = v
            u = m * alpha / (torch.sqrt(v) + eps)
        elif k == 9:
            v = s.b * st[n:2 * n] + (1.0 - s.b) * g * g
            hh = torch.maximum(st[2 * n:3 * n], v)
            st[n:2 * n] = v
            st[2 * n:3 * n] = hh
            u = alpha * m / (torch.sqrt(hh) + eps)
        else:
            d = g - m
            sv = s.b * st[n:2 * n] + (1.0 - s.b) * d * d + eps
            st[n:2 * n] = sv
            u = alpha * m / (torch.sqrt(sv) + eps)
    elif k == 4:
        r = s.a * st[:n] + (1.0 - s.a) * g * g
        st[:n] = r
        u = s.lr * g / (torch.sqrt(r) + eps)
    elif k == 5:
        hh = st[:n] + g * g
        st[:n] = hh
        u = s.lr * g / (torch.sqrt(hh) + eps)
    elif k == 6:
        msg = s.a * st[:n] + (1.0 - s.a) * g * g
        msdx0 = st[n:2 * n].clone()
        u = g * torch.sqrt(msdx0 + eps) / torch.sqrt(msg + eps)
        st[:n] = msg
        st[n:2 * n] = s.a * msdx0 + (1.0 - s.a) * u * u
    elif k == 7:
        alpha = fix(s.lr / (1.0 - s.a ** t)) if (1.0 - s.a ** t) != 0 else eps
        m = s.a * st[:n] + (1.0 - s.a) * g
        ui = torch.maximum(s.b * st[n:2 * n], g.abs()) + 1e-32
        st[:n] = m
        st[n:2 * n] = ui
        u = m * alpha / ui
    elif k == 8:
        m = s.a * st[:n] + (1.0 - s.a) * g
        v = s.b * st[n:2 * n] + (1.0 - s.b) * g * g
        st[:n] = m
        st[n:2 * n] = v
        u = s.lr * ((s.a * m + (1.0 - s.a) * g) / (1.0 - s.a ** t)) / (torch.sqrt(v) + eps)
    else:
        raise ValueError(k)
    if s.wd:
        u = u + s.wd * (s.lr if s.wd_apply_lr else 1.0) * w
    if s.update_only:
        grads[s.off:s.off + n] = u
    else:
        params[s.off:s.off + n] = w - u
        if write_update:
            grads[s.off:s.off + n] = u


# ---------------------------------------------------------------------------------------------
class CastTable:
    """One-launch refresh of compute-dtype (bf16) weight replicas from the fp32 master vector,
    with optional OIHW -> OHWI (channels-last filter) permutation per entry."""

    def __init__(self, entries, device):
        # entries: list of (src_off, dst_off, len, I, K[, O, mode]); mode 1 = [O][I][K] -> [I][K flipped][O]
        self.entries = [tuple(e) + (0, 0) if len(e) == 5 else tuple(e) for e in entries]
        assert all(e[2] < (1 << 31) for e in self.entries), "cast table entries are single weight tensors (< 2^31 elements)"
        self.device = torch.device(device)
        raw = b"".join(struct.pack("<qqqiiii", *e) for e in self.entries)
        self._seg = (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device) if raw
                     else torch.zeros(0, dtype=torch.uint8, device=self.device))
        # mode-1 entries are processed as 64 x 64 transpose tiles: one block per tile (the chunk index of the table is the tile index)
        bs, bc = _block_tables([e[2] if e[6] != 1 else self._tiles(e) * CHUNK for e in self.entries])
        self.num_blocks = int(bs.shape[0])
        self._blk_seg = torch.from_numpy(bs).to(self.device)
        self._blk_chunk = torch.from_numpy(bc).to(self.device)

    @staticmethod
    def _tiles(e):
        _, _, n, I, K, O, _ = e
        assert O > 0 and n == O * I * K, "mode-1 cast entries describe a whole [O, I, K] filter"
        return ((O + 63) // 64) * ((I * K + 63) // 64)

    def run(self, src_f32, dst_bf16):
        if L.use_hip(src_f32) and src_f32.dtype == torch.float32 and dst_bf16.dtype in (torch.float32, torch.bfloat16, torch.float16):
            code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dst_bf16.dtype]
            L.check(L.hip().dl4j_cast_table_dt(L.ptr(src_f32), L.ptr(dst_bf16), L.ptr(self._seg), L.ptr(self._blk_seg),
                                               L.ptr(self._blk_chunk), L.c_int(self.num_blocks), L.c_int(code), L.stream()),
                    "cast_table")
            return
        for (so, do, n, I, K, O, mode) in self.entries:
            s = src_f32[so:so + n]
            if mode == 1:
                s = torch.flip(s.view(O, I, K), dims=[2]).permute(1, 2, 0).reshape(-1)
            elif K > 1:
                s = s.view(-1, I, K).permute(0, 2, 1).reshape(-1)
            dst_bf16[do:do + n] = s.to(dst_bf16.dtype)


class GradNormTable:
    """Gradient normalization of the whole network in two launches (native/csrc/updaters.hip gn_*): ``ranges`` =
    [(offset, length, mode, threshold)] over the flat gradient vector, mode 1 renormalize-L2, 2 clip element-wise, 3 clip-L2; one
    range per layer or per parameter tensor (GradientNormalization.java, BaseMultiLayerUpdater.preApply:405-453)."""
    MODES = {"RenormalizeL2PerLayer": (1, "layer"), "RenormalizeL2PerParamType": (1, "param"),
             "ClipElementWiseAbsoluteValue": (2, "param"), "ClipL2PerLayer": (3, "layer"), "ClipL2PerParamType": (3, "param")}

    def __init__(self, ranges, device):
        self.ranges = [tuple(r) for r in ranges]
        self.device = torch.device(device)
        raw = b"".join(struct.pack("<qqif", int(o), int(n), int(m), float(t)) for (o, n, m, t) in self.ranges)
        self._seg = (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device) if raw
                     else torch.zeros(0, dtype=torch.uint8, device=self.device))
        bs, bc = _block_tables([r[1] for r in self.ranges])
        self.num_blocks = int(bs.shape[0])
        self._blk_seg = torch.from_numpy(bs).to(self.device)
        self._blk_chunk = torch.from_numpy(bc).to(self.device)
        self._acc = torch.zeros(max(len(self.ranges), 1), dtype=torch.float32, device=self.device)

    def run(self, grads):
        if not self.ranges:
            return
        if L.use_hip(grads) and grads.dtype in _UPD_DT:
            L.check(L.hip().dl4j_grad_normalize(L.ptr(grads), L.ptr(self._seg), L.c_int(len(self.ranges)), L.ptr(self._blk_seg),
                                                L.ptr(self._blk_chunk), L.c_int(self.num_blocks), L.ptr(self._acc),
                                                L.c_int(_UPD_DT[grads.dtype]), L.stream()), "grad_normalize")
            return
        for (o, n, m, thr) in self.ranges:
            g = grads[o:o + n]
            if m == 2:
                g.clamp_(-thr, thr)
                continue
            nrm = torch.sqrt((g.double() ** 2).sum())
            if m == 1:
                g.div_(nrm.clamp_min(1e-12).to(g.dtype))
            else:
                g.mul_(torch.where(nrm > thr, thr / nrm.clamp_min(1e-12), torch.ones_like(nrm)).to(g.dtype))


def global_sumsq(x: torch.Tensor) -> torch.Tensor:
    """sum(x^2) over a flat fp32 vector as a 1-element device tensor (no host sync)."""
    if L.use_hip(x) and x.dtype == torch.float32:
        partial = torch.empty(1024, device=x.device, dtype=torch.float32)
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        L.check(L.hip().dl4j_sumsq(L.ptr(x), L.c_int64(x.numel()), L.ptr(partial), L.ptr(out), L.stream()), "sumsq")
        return out
    return (x.double() * x.double()).sum().to(x.dtype).reshape(1)
