"""Stacked LSTM layers as one pipeline over several HIP streams.

Reference behaviour: MultiLayerNetwork runs a stack of LSTM / GravesLSTM layers one after the other, each over the whole sequence
(nn/multilayer/MultiLayerNetwork.java ffToLayerActivationsInWs -> nn/layers/recurrent/LSTMHelpers.java:211-369 activateHelper, :531-714
backpropGradientHelper). The numbers computed here are the same; the schedule is MI355X-first:

* One fused recurrent launch per time step fills n / 8 workgroups (64 of 256 CUs at the 512-unit char-RNN shape) and a step cannot start
  before the previous one has finished, so a single layer can never use more than a quarter of the chip.
* Layer l + 1 at step t only needs layer l at step t. The sequence is cut into chunks of ``CHUNK`` steps; chunk c of layer l + 1 is issued
  on its own stream as soon as chunk c of layer l is done, so up to ``len(layers)`` recurrences run side by side (a wavefront over
  layer x chunk). The backward pass is the mirror image: top layer first, last chunk first, the input gradient of a chunk (dG . W^T)
  handed down as soon as the chunk's recursion is done.
* Library GEMMs (first layer's input projection, weight gradients) are only ever issued on the current stream: two vendor fp32 GEMMs
  running at the same time on two streams dead-locked the device (profiles/r2_lstm_persistent.md). The side streams carry nothing but
  kernels of native/csrc, which never wait on another workgroup: the step kernels of lstm_persist.hip and, for the per-chunk input
  projection (H_chunk . W + b) and input gradient (dG_chunk . W^T) between two layers, the in-tree MFMA GEMM (conv_mfma.hip dl4j_gemm_nt_ex via
  ops/gemm.py). Issuing those on the consumer's own stream keeps the hand-off between two layers at one event per chunk (a detour
  over the current stream for a library GEMM costs two cross-stream hops per stage: measured 12.8 ms instead of the figure in
  profiles/r2_lstm_persistent.md).
* Every buffer is allocated on the current stream before the first fork and kept in the layers' saved state past the join, so the caching
  allocator never recycles a block a side stream still uses. The fork / join pattern is legal inside a hipGraph capture.

The per-layer saved state is what ``LSTMImpl.forward`` leaves (plus a ``stack`` mark), so a stack forwarded here can also be
back-propagated layer by layer (``LSTMImpl.backward``); ``backward`` here only takes stacks that ``forward`` here has run.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

from ...native import lib as L
from ...ops import gemm as G
from ...nd4j.activations import ActivationSigmoid, ActivationTanH
from ...ops import lstm as LSTMOP

ENABLED = os.environ.get("DL4J_AMD_LSTM_STACK", "1") != "0"
DIAG = os.environ.get("DL4J_AMD_LSTM_DIAG", "1") != "0"          # bf16 stacks of equal width: one launch per anti-diagonal (see _diag_ok)
CHUNK = int(os.environ.get("DL4J_AMD_LSTM_STACK_CHUNK", "0"))     # 0: about a sixth of the sequence, at least 4 steps
_TLS = threading.local()      # side streams and the event pool are per host thread (ParallelWrapper / ParallelInference workers share a device)


def _tls(name):
    d = getattr(_TLS, name, None)
    if d is None:
        d = {}
        setattr(_TLS, name, d)
    return d


def _streams(dev, k):
    s = _tls("streams").setdefault(dev, [])
    while len(s) < k:
        s.append(torch.cuda.Stream(device=dev))
    return s[:k]


class _Events:
    """A per-device pool of events, handed out in order and reused from the start by the next pass."""

    def __init__(self, dev):
        self.pool = _tls("events").setdefault(dev, [])
        self.i = 0

    def record(self, stream):
        if self.i == len(self.pool):
            self.pool.append(torch.cuda.Event())
        e = self.pool[self.i]
        self.i += 1
        e.record(stream)
        return e


def _chunks(T, own_gemm=True):
    # measured on the 3 x 512 char-RNN (T = 50): bf16 best at 6 chunks, fp32 (library GEMM hand-off, longer steps) at 10
    tc = CHUNK if CHUNK > 0 else max(4, -(-T // (6 if own_gemm else 10)))
    return [(s, min(T, s + tc)) for s in range(0, T, tc)]


def find_runs(layers, plan, pre):
    """{first index: [indices]} of maximal runs of >= 2 consecutive plain LSTM / GravesLSTM layers with nothing between them."""
    from .recurrent import LSTMImpl
    kinds = {i: k for k, i in plan}
    runs, cur = {}, []

    def close():
        if len(cur) >= 2:
            runs[cur[0]] = list(cur)
        del cur[:]
    for i, l in enumerate(layers):
        if not (type(l) is LSTMImpl and not l.reverse and kinds.get(i) == "layer"):
            close()
            continue
        if cur and i in pre:            # a preprocessor sits in front of this layer: it can only start a run
            close()
        cur.append(i)
    close()
    return runs


def ok(layers, x, train):
    """Every layer takes the fused per-step route at this shape and nothing but the recurrence sits between the layers."""
    if not (ENABLED and x.is_cuda and x.dim() == 3 and LSTMOP._PERSIST and LSTMOP._PERSIST_MODE == "steps"):
        return False
    mb, _, T = x.shape
    if T < 8:
        return False
    cd = layers[0].ctx.cdtype
    if cd not in (torch.float32, torch.bfloat16):
        return False
    sup = L.hip().dl4j_lstm_persist_supported if cd == torch.bfloat16 else L.hip().dl4j_lstm_steps_f32_supported
    for l in layers:
        c = l.conf
        if l.ctx.cdtype != cd or (train and getattr(c, "idropout", None) is not None):
            return False
        if not (isinstance(c.activationFn, ActivationTanH) and isinstance(c.gateActivationFn, ActivationSigmoid)):
            return False
        RW = l._p("RW")
        RW4 = RW[:, :4 * c.nOut]
        if RW4.dtype != cd or not RW4.t().is_contiguous() or RW4.data_ptr() % 16 != 0 or l._p("W").dtype != cd:
            return False
        if not sup(L.c_int(mb), L.c_int(c.nOut)):
            return False
        W = l._p("W")
        if l is not layers[0] and not (W.t().is_contiguous() and W.data_ptr() % 16 == 0 and W.shape[0] % 8 == 0):
            return False          # 'f'-order W is the [4n, nIn] NT operand of the in-tree GEMM
    return True


def _sp(stream):
    return L.c_void_p(stream.cuda_stream)


def _parr(ts):
    """C array of device pointers (null for None)."""
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def _diag_ok(lay, mb, bf16):
    """One launch per anti-diagonal of the layer x time wavefront (native/csrc/lstm_persist.hip lstm_diag_*_kernel): bf16 / fp32 stacks
    of 2-4 layers of the same width (a multiple of 128) whose upper layers read an input that is a multiple of 64 wide."""
    if not (DIAG and 2 <= len(lay) <= 4):
        return False
    n = lay[0]["n"]
    if any(d["n"] != n for d in lay) or any(d["nIn"] % 64 != 0 for d in lay[1:]):
        return False
    return bool(L.hip().dl4j_lstm_diag_supported(L.c_int(mb), L.c_int(n), L.c_int(len(lay))))


def _gemm_nt_on(stream, x, w, y, bias):
    """y [M, N] = x [M, K] . w [N, K]^T (+ fp32 bias) with the in-tree MFMA GEMM on an explicit stream; nothing is allocated."""
    M, K = x.shape
    N = w.shape[0]
    t = G.pick_tile(M, N, K, x.dtype)
    if t == 4:
        t = 0
    d = (ctypes.c_int64 * 14)(M, N, K, x.stride(0), w.stride(0), y.stride(0), 0, 0, 0, 1, t, G._DT[x.dtype], 0, 0)
    L.check(L.hip().dl4j_gemm_nt_ex(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(bias), L.ptr(None), ctypes.c_float(1.0), L.ptr(None), L.ptr(None),
                                    L.c_int(0), d, _sp(stream)), "lstm_stack_gemm")


def forward(layers, x, train, mode=None):
    """x [mb, nIn, T] -> activations of the last layer [mb, n, T]; leaves every layer's saved state / tBPTT state as LSTMImpl.forward does."""
    cd = layers[0].ctx.cdtype
    if x.dtype != cd:
        x = x.to(cd)
    mb, nIn, T = x.shape
    dev = x.device
    bf16 = cd == torch.bfloat16
    cur = torch.cuda.current_stream(dev)
    S = _streams(dev.index if dev.index is not None else torch.cuda.current_device(), len(layers))
    ev = _Events(dev.index if dev.index is not None else torch.cuda.current_device())
    own = bf16        # hand-off GEMM: in-tree kernel on the consumer's stream (bf16) / library GEMM on the current stream (fp32: faster there)
    chunks = _chunks(T, own)
    nc, nl = len(chunks), len(layers)
    lay = []
    xt = x.permute(2, 0, 1).reshape(T * mb, nIn)
    for li, l in enumerate(layers):
        n = l.conf.nOut
        W, RW, b = l._p("W"), l._p("RW"), l._p("b").reshape(-1)
        RW4 = RW[:, :4 * n]
        peep = pf = po = pg = None
        if RW.shape[1] > 4 * n:
            peep = (RW[:, 4 * n], RW[:, 4 * n + 1], RW[:, 4 * n + 2])
            pf, po, pg = (p.contiguous().float() for p in peep)
        init = l.tbpttState.get("s") if mode == "tbptt" else None
        if init is not None:
            h0, c0 = init[0].to(cd).contiguous(), init[1].to(cd).contiguous()
        else:
            h0 = torch.zeros(mb, n, dtype=cd, device=dev)
            c0 = torch.zeros(mb, n, dtype=cd, device=dev)
        d = dict(l=l, n=n, W=W, Wnt=W.t(), bf=b.float().contiguous(), RW4=RW4, b=b.to(cd), peep=peep, pf=pf, po=po, pg=pg, h0=h0, c0=c0,
                 xw=torch.empty(T, mb, 4 * n, device=dev, dtype=cd), H=torch.empty(T, mb, n, device=dev, dtype=cd),
                 Cs=torch.empty(T, mb, n, device=dev, dtype=cd), SV=torch.empty(T, mb, 6 * n, device=dev, dtype=cd) if train else None,
                 bar=torch.empty(2, device=dev, dtype=torch.int32), xt=xt if li == 0 else lay[li - 1]["H"].view(T * mb, -1),
                 nIn=nIn if li == 0 else lay[li - 1]["n"])
        lay.append(d)
    torch.addmm(lay[0]["b"], xt, lay[0]["W"], out=lay[0]["xw"].view(T * mb, -1))
    diag = _diag_ok(lay, mb, bf16)
    if diag:
        has_peep = lay[0]["pf"] is not None
        if any((d["pf"] is not None) != has_peep for d in lay):
            diag = False
    if diag:
        for d in lay[1:]:
            d["Wnt_c"] = d["Wnt"] if d["Wnt"].is_contiguous() else d["Wnt"].contiguous()
        nins = (ctypes.c_int * nl)(*[d["nIn"] for d in lay])
        L.check(L.hip().dl4j_lstm_diag_fwd(
            L.ptr(lay[0]["xw"]), _parr([d["RW4"] for d in lay]), _parr([None] + [d["Wnt_c"] for d in lay[1:]]),
            _parr([None] + [lay[i - 1]["H"] for i in range(1, nl)]), _parr([None] + [d["bf"] for d in lay[1:]]),
            _parr([d["h0"] for d in lay]), _parr([d["c0"] for d in lay]),
            _parr([d["pf"] for d in lay]) if has_peep else None, _parr([d["po"] for d in lay]) if has_peep else None,
            _parr([d["pg"] for d in lay]) if has_peep else None, _parr([d["H"] for d in lay]), _parr([d["Cs"] for d in lay]),
            _parr([d["SV"] for d in lay]) if train else None, nins, L.c_int(nl), L.c_int(T), L.c_int(mb), L.c_int(lay[0]["n"]),
            L.c_int(0 if bf16 else 1), L.ptr(None), L.c_int(0), _sp(cur)),
            "lstm_diag_fwd")
        return _finish_forward(lay, train, mode, mb, T)
    rec = [[None] * nc for _ in range(nl)]
    proj = [[None] * nc for _ in range(nl)]
    e0 = ev.record(cur)
    hip = L.hip()
    try:
        for dgn in range(nc + nl - 1):
            for li in range(nl):
                ci = dgn - li
                if ci < 0 or ci >= nc:
                    continue
                d = lay[li]
                s0, s1 = chunks[ci]
                st = S[li]
                if li == 0:
                    if ci == 0:
                        st.wait_event(e0)
                elif own:
                    st.wait_event(rec[li - 1][ci])
                    if ci == 0:
                        st.wait_event(e0)
                    _gemm_nt_on(st, lay[li - 1]["H"][s0:s1].view((s1 - s0) * mb, -1), d["Wnt"], d["xw"][s0:s1].view((s1 - s0) * mb, -1), d["bf"])
                else:
                    st.wait_event(proj[li][ci])
                if bf16:
                    rc = hip.dl4j_lstm_steps_fwd_range(L.ptr(d["xw"]), L.ptr(d["RW4"]), L.c_int64(d["n"]), L.ptr(d["h0"]), L.ptr(d["c0"]),
                                                       L.ptr(d["pf"]), L.ptr(d["po"]), L.ptr(d["pg"]), L.ptr(None), L.ptr(d["H"]), L.ptr(d["Cs"]),
                                                       L.ptr(d["SV"]), L.ptr(d["bar"]), L.c_int(T), L.c_int(mb), L.c_int(d["n"]), L.c_int(0),
                                                       L.c_int(s0), L.c_int(s1), _sp(st))
                else:
                    rc = hip.dl4j_lstm_steps_fwd_f32_range(L.ptr(d["xw"]), L.ptr(d["RW4"]), L.c_int64(d["n"]), L.ptr(d["h0"]), L.ptr(d["c0"]),
                                                           L.ptr(d["pf"]), L.ptr(d["po"]), L.ptr(d["pg"]), L.ptr(None), L.ptr(d["H"]),
                                                           L.ptr(d["Cs"]), L.ptr(d["SV"]), L.c_int(T), L.c_int(mb), L.c_int(d["n"]), L.c_int(0),
                                                           L.c_int(s0), L.c_int(s1), _sp(st))
                L.check(rc, "lstm_stack_fwd")
                rec[li][ci] = ev.record(st)
                if not own and li + 1 < nl:
                    # library GEMM hand-off: the next layer's projection of this chunk, issued on the current stream right behind the chunk
                    nx = lay[li + 1]
                    cur.wait_event(rec[li][ci])
                    torch.addmm(nx["b"], d["H"][s0:s1].view((s1 - s0) * mb, -1), nx["W"], out=nx["xw"][s0:s1].view((s1 - s0) * mb, -1))
                    proj[li + 1][ci] = ev.record(cur)
    finally:
        for st in S:
            cur.wait_stream(st)
    return _finish_forward(lay, train, mode, mb, T)


def _finish_forward(lay, train, mode, mb, T):
    for d in lay:
        l = d["l"]
        st = {"H": d["H"], "Cs": d["Cs"], "SV": d["SV"], "h0": d["h0"], "c0": d["c0"], "peep": (d["pf"], d["po"], d["pg"]), "maskT": None,
              "stack": True}
        if train:
            l._saved, l._xt, l._dims, l._peep = ("native", st), d["xt"], (mb, d["nIn"], T, d["n"]), d["peep"]
        l._drop_mask = None
        l.maskArray = None
        l._last_state = (d["H"][T - 1], d["Cs"][T - 1])
        if mode == "tbptt":
            l.tbpttState["s"] = tuple(t.detach() for t in l._last_state)
    return lay[-1]["H"].permute(1, 2, 0)


def can_backward(layers):
    """The forward pass of every layer went through ``forward`` above (so shapes and data types are the supported ones)."""
    return ENABLED and all(isinstance(l._saved, tuple) and l._saved[0] == "native" and l._saved[1].get("stack") and l._saved[1].get("SV") is not None
                           for l in layers)


def backward(layers, eps):
    """eps [mb, n_top, T] -> dL/dx of the first layer [mb, nIn, T] (or None when it is not needed); writes every layer's gradients."""
    cd = layers[0].ctx.cdtype
    bf16 = cd == torch.bfloat16
    dev = eps.device
    di = dev.index if dev.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(dev)
    nl = len(layers)
    S = _streams(di, nl)
    ev = _Events(di)
    mb, _, T, _ = layers[-1]._dims
    own = bf16
    chunks = _chunks(T, own)
    nc = len(chunks)
    lay = []
    for li, l in enumerate(layers):
        _, nIn, _, n = l._dims
        st = l._saved[1]
        RW4 = l._p("RW")[:, :4 * n]
        pf, po, pg = st["peep"]
        lay.append(dict(l=l, n=n, nIn=nIn, st=st, W=l._p("W"), Wc=l._p("W").contiguous() if (li > 0 and own) else None, rw4t=RW4.contiguous(), pf=pf, po=po, pg=pg,
                        dG=torch.empty(T, mb, 4 * n, device=dev, dtype=cd), dc=torch.empty(2, mb, n, device=dev, dtype=torch.float32),
                        dpeep=torch.zeros(3, n, device=dev, dtype=torch.float32) if pf is not None else None,
                        bar=torch.empty(2, device=dev, dtype=torch.int32),
                        E=eps.permute(2, 0, 1).to(cd).contiguous() if li == nl - 1 else torch.empty(T, mb, n, device=dev, dtype=cd)))
    if _diag_ok(lay, mb, bf16) and len({d["pf"] is not None for d in lay}) == 1:
        has_peep = lay[0]["pf"] is not None
        for li in range(nl - 1):
            up = lay[li + 1]
            lay[li]["wct"] = up["Wc"] if up["Wc"] is not None else up["W"].contiguous()
        L.check(L.hip().dl4j_lstm_diag_bwd(
            L.ptr(lay[-1]["E"]), _parr([d["rw4t"] for d in lay]), _parr([lay[li].get("wct") for li in range(nl)]),
            _parr([d["st"]["SV"] for d in lay]), _parr([d["st"]["Cs"] for d in lay]), _parr([d["st"]["c0"] for d in lay]),
            _parr([d["pf"] for d in lay]) if has_peep else None, _parr([d["po"] for d in lay]) if has_peep else None,
            _parr([d["pg"] for d in lay]) if has_peep else None, _parr([d["dG"] for d in lay]),
            _parr([d["dpeep"] for d in lay]) if has_peep else None, _parr([d["dc"] for d in lay]), L.c_int(nl), L.c_int(T), L.c_int(mb),
            L.c_int(lay[0]["n"]), L.c_int(0 if bf16 else 1), L.ptr(None), L.c_int(0), _sp(cur)), "lstm_diag_bwd")
        return _finish_backward(lay, nl)
    e0 = ev.record(cur)
    done = [[None] * nc for _ in range(nl)]        # dG of (layer, chunk) is complete
    ready = [[None] * nc for _ in range(nl)]       # E of (layer, chunk) is complete (library hand-off)
    hip = L.hip()
    try:
        for dgn in range(nc + nl - 1):
            for k in range(nl):                    # k = 0: top layer
                cj = dgn - k
                if cj < 0 or cj >= nc:
                    continue
                li, ci = nl - 1 - k, nc - 1 - cj
                d = lay[li]
                s0, s1 = chunks[ci]
                st = S[li]
                if cj == 0:
                    st.wait_event(e0)
                if k > 0:
                    # dL/dH of this chunk = input gradient of the layer above (dG . W^T)
                    up = lay[li + 1]
                    if own:
                        st.wait_event(done[li + 1][ci])
                        _gemm_nt_on(st, up["dG"][s0:s1].view((s1 - s0) * mb, -1), up["Wc"], d["E"][s0:s1].view((s1 - s0) * mb, -1), None)
                    else:
                        st.wait_event(ready[li][ci])
                sv = d["st"]
                if bf16:
                    rc = hip.dl4j_lstm_steps_bwd_range(L.ptr(d["E"]), L.ptr(d["rw4t"]), L.c_int64(4 * d["n"]), L.ptr(sv["SV"]), L.ptr(sv["Cs"]),
                                                       L.ptr(sv["c0"]), L.ptr(d["pf"]), L.ptr(d["po"]), L.ptr(d["pg"]), L.ptr(None), L.ptr(d["dG"]),
                                                       L.ptr(d["dpeep"]), L.ptr(d["bar"]), L.ptr(d["dc"]), L.ptr(None), L.c_int(T), L.c_int(mb),
                                                       L.c_int(d["n"]), L.c_int(0), L.c_int(s0), L.c_int(s1), _sp(st))
                else:
                    rc = hip.dl4j_lstm_steps_bwd_f32_range(L.ptr(d["E"]), L.ptr(d["rw4t"]), L.c_int64(4 * d["n"]), L.ptr(sv["SV"]),
                                                           L.ptr(sv["Cs"]), L.ptr(sv["c0"]), L.ptr(d["pf"]), L.ptr(d["po"]), L.ptr(d["pg"]),
                                                           L.ptr(None), L.ptr(d["dG"]), L.ptr(d["dpeep"]), L.ptr(d["dc"]), L.ptr(None), L.c_int(T),
                                                           L.c_int(mb), L.c_int(d["n"]), L.c_int(0), L.c_int(s0), L.c_int(s1), _sp(st))
                L.check(rc, "lstm_stack_bwd")
                done[li][ci] = ev.record(st)
                if not own and li > 0:
                    cur.wait_event(done[li][ci])
                    torch.mm(d["dG"][s0:s1].view((s1 - s0) * mb, -1), d["W"].t(), out=lay[li - 1]["E"][s0:s1].view((s1 - s0) * mb, -1))
                    ready[li - 1][ci] = ev.record(cur)
    finally:
        for st in S:
            cur.wait_stream(st)
    return _finish_backward(lay, nl)


def _finish_backward(lay, nl):
    dx = None
    for li in range(nl - 1, -1, -1):
        d = lay[li]
        l = d["l"]
        dp = None if d["dpeep"] is None else [d["dpeep"][0], d["dpeep"][1], d["dpeep"][2]]
        l._saved = []
        out = l.backward_finish((d["dG"], LSTMOP._hprev(d["st"], False), dp), want_dx=(li == 0))
        if li == 0:
            dx = out
    return dx
