"""Recurrent layers: LSTM / GravesLSTM (peepholes) / GravesBidirectionalLSTM / SimpleRnn + wrappers.

Reference: nn/layers/recurrent/LSTMHelpers.java:211-369 (forward, Java time loop with 2 GEMMs per
step) and :531-714 (backward, ~6 GEMMs per step). MI355X-first restructuring:

* the input projection for ALL timesteps is one GEMM ([T*mb, nIn] x [nIn, 4n]);
* per step only the recurrent GEMM (h_{t-1} x RW) + one fused pointwise cell kernel remain;
* backward keeps per step one GEMM (dGates x RW^T) + one fused pointwise kernel, while
  dW, dRW and dX are three big GEMMs over all timesteps after the loop.

Parameter layout is DL4J's (LSTMParamInitializer.java:122-145): W [nIn,4n], RW [n,4n(+3)], b [4n],
'f' order, gate column blocks [a(block input) | f | o | g(input gate)], Graves peepholes
[wFF, wOO, wGG] in the last 3 RW columns.
"""
from __future__ import annotations

import contextlib
import threading

import torch

from ...nd4j.activations import ActivationSigmoid, ActivationTanH
from ...ops import lstm as LSTMOP
from .base import LayerImpl, AutogradLayerImpl


class LSTMImpl(LayerImpl):
    reverse = False
    prefix = ""

    def __init__(self, conf, ctx):
        super().__init__(conf, ctx)
        self.stateMap = {}     # rnnTimeStep state
        self.tbpttState = {}

    def compute_replicas(self):
        k = self.prefix
        return [("W" + k, "plain"), ("RW" + k, "plain"), ("b" + k, "plain")]

    def type(self):
        return "RECURRENT"

    def _p(self, key):
        return self.w(key + self.prefix)

    # state management (MultiLayerNetwork.rnnTimeStep / tBPTT)
    def rnnClearPreviousState(self):
        self.stateMap, self.tbpttState = {}, {}

    def rnnGetPreviousState(self):
        return dict(self.stateMap)

    def rnnSetPreviousState(self, s):
        self.stateMap = dict(s)

    def forward(self, x, train=False, initial_state=None, store_state=False, fork=None):
        """x: [mb, nIn, T] -> [mb, n, T]. ``fork`` (a _side_stream): the native recurrence is issued on the side stream, the input
        projection (a library GEMM) stays on the current one."""
        x = self._dropout_in(x, train)
        if x.dtype != self.ctx.cdtype:
            x = x.to(self.ctx.cdtype)
        c = self.conf
        n = c.nOut
        mb, nIn, T = x.shape
        W, RW, b = self._p("W"), self._p("RW"), self._p("b").reshape(-1)
        RW4 = RW[:, :4 * n]
        peep = None
        if RW.shape[1] > 4 * n:
            peep = (RW[:, 4 * n], RW[:, 4 * n + 1], RW[:, 4 * n + 2])  # wFF, wOO, wGG
        xt = x.permute(2, 0, 1).reshape(T * mb, nIn)                 # [T*mb, nIn]
        xw = torch.addmm(b.to(xt.dtype), xt, W).view(T, mb, 4 * n)  # one GEMM for all steps
        if initial_state is not None:
            h, cs = initial_state
        else:
            h = torch.zeros(mb, n, dtype=x.dtype, device=x.device)
            cs = torch.zeros(mb, n, dtype=x.dtype, device=x.device)
        mask = self.maskArray
        if LSTMOP.native_seq_ok(xw, RW4, n, c.activationFn, c.gateActivationFn):
            # whole sequence issued from C++ (native/csrc/rnn_seq.hip): recurrent GEMM on the in-tree MFMA kernel + fused cell
            with (fork.fork(x, xw, h, cs) if fork is not None else contextlib.nullcontext()):
                H, st = LSTMOP.seq_forward(xw, RW4, h, cs, peep, mask, train, self.reverse)
            if train:
                self._saved, self._xt, self._dims, self._peep = ("native", st), xt, (mb, nIn, T, n), peep
            t_last = 0 if self.reverse else T - 1
            h, cs = H[t_last], st["Cs"][t_last]
            if store_state:
                self.stateMap = {"prevActivation": h.detach(), "prevMemCell": cs.detach()}
            self._last_state = (h, cs)
            return H.permute(1, 2, 0)
        H = torch.empty(T, mb, n, dtype=x.dtype, device=x.device)
        saved = [] if train else None
        order = range(T - 1, -1, -1) if self.reverse else range(T)
        for t in order:
            gates = torch.addmm(xw[t], h, RW4)
            m = mask[:, t].reshape(mb, 1).to(x.dtype) if mask is not None else None
            h_new, c_new, sv = LSTMOP.cell_forward(gates, cs, peep, c.activationFn, c.gateActivationFn, m, train)
            if train:
                saved.append((t, h, cs, sv, m))
            h, cs = h_new, c_new
            H[t] = h
        if train:
            self._saved, self._xt, self._dims, self._peep = saved, xt, (mb, nIn, T, n), peep
        if store_state:
            self.stateMap = {"prevActivation": h.detach(), "prevMemCell": cs.detach()}
        self._last_state = (h, cs)
        return H.permute(1, 2, 0)

    def backward(self, eps):
        """eps: [mb, n, T] -> dx [mb, nIn, T]; writes dW, dRW, db."""
        return self.backward_finish(self.backward_recur(eps))

    def backward_recur(self, eps, fork=None):
        """The time recursion of the backward pass: (dG [T, mb, 4n], Hprev [T, mb, n], dpeep). With ``fork`` the native loop is issued on the
        side stream."""
        c = self.conf
        mb, nIn, T, n = self._dims
        W, RW = self._p("W"), self._p("RW")
        RW4 = RW[:, :4 * n]
        E = eps.permute(2, 0, 1).to(W.dtype)                           # [T, mb, n]
        if isinstance(self._saved, tuple) and self._saved[0] == "native":
            with (fork.fork(eps, E, self._saved) if fork is not None else contextlib.nullcontext()):
                dG, Hprev, dpeep = LSTMOP.seq_backward(E, RW4, self._saved[1], self.reverse)
            self._saved = []
        else:
            dG = torch.empty(T, mb, 4 * n, dtype=W.dtype, device=W.device)
            Hprev = torch.empty(T, mb, n, dtype=W.dtype, device=W.device)
            dpeep = [torch.zeros(n, dtype=torch.float32, device=W.device) for _ in range(3)] if self._peep is not None else None
        dh_next = torch.zeros(mb, n, dtype=W.dtype, device=W.device)
        dc_next = torch.zeros(mb, n, dtype=W.dtype, device=W.device)
        for (t, h_prev, c_prev, sv, m) in reversed(self._saved):
            dh = E[t] + dh_next
            dgates, dc_prev, dp = LSTMOP.cell_backward(dh, dc_next, c_prev, sv, self._peep, c.activationFn,
                                                      c.gateActivationFn, m)
            dG[t] = dgates
            Hprev[t] = h_prev
            dh_next = torch.matmul(dgates, RW4.t())
            dc_next = dc_prev
            if dp is not None:
                for i in range(3):
                    dpeep[i] += dp[i]
        return dG, Hprev, dpeep

    def backward_finish(self, rec, want_dx=True):
        """Weight / bias / input gradients from the recursion's outputs (library GEMMs, current stream)."""
        dG, Hprev, dpeep = rec
        mb, nIn, T, n = self._dims
        k = self.prefix
        W = self._p("W")
        dG2 = dG.view(T * mb, 4 * n)
        gW, gRW, gb = self.grads["W" + k], self.grads["RW" + k], self.grads["b" + k]
        # 16-bit GPU layers: the in-tree split-K weight-gradient kernel writes fp32 straight into the gradient views (no library GEMM, no
        # conversion pass) and the first launch also accumulates the bias gradient (column sums of dG) from the tiles it loads anyway
        from ...ops import dense as DENSE
        db_done = False

        def own(x2, g):
            nonlocal db_done
            gt = g.t()
            if not (dG2.is_cuda and dG2.dtype in (torch.bfloat16, torch.float16) and x2.dtype == dG2.dtype and DENSE.dw_ok(dG2, x2, gt)):
                return False
            take_db = (not db_done and gb.dtype == torch.float32 and gb.is_contiguous() and DENSE.dw_takes_db(dG2, x2))
            if take_db:
                gb.zero_()
            DENSE.dw(dG2, x2, gt, False, db_out=gb.view(-1) if take_db else None)
            db_done = db_done or take_db
            return True
        xt = self._xt if self._xt.is_contiguous() else None
        if xt is None or not own(xt, gW):
            gW.copy_(torch.matmul(self._xt.t(), dG2))
        hp2 = Hprev.view(T * mb, n)
        if dpeep is not None or not own(hp2, gRW):
            gRW[:, :4 * n].copy_(torch.matmul(hp2.t(), dG2))
        if dpeep is not None:
            gRW[:, 4 * n].copy_(dpeep[0]); gRW[:, 4 * n + 1].copy_(dpeep[1]); gRW[:, 4 * n + 2].copy_(dpeep[2])
        if not db_done:
            gb.copy_(dG2.sum(0, dtype=torch.float32).reshape(1, -1))
        self._saved = None
        if not want_dx or not getattr(self, "needs_input_grad", True):
            return None
        dx = torch.matmul(dG2, W.t()).view(T, mb, nIn).permute(1, 2, 0)
        return self._dropout_back(dx)

    def clear(self):
        super().clear()
        self._saved = None
        self._xt = None


_SIDE_TLS = threading.local()     # one side stream per (host thread, device)


class _side_stream:
    """Fork / join of a second HIP stream so that the two directions of a bidirectional layer overlap on the chip (each per-step launch
    of the recurrence fills 64 of 256 CUs at the char-RNN shape). ``fork()`` makes the side stream wait for everything queued on the
    current one and returns the context that issues on it; ``join()`` makes the current stream wait for the side stream. The pattern is
    legal inside a hipGraph capture (the side stream joins the capture through the first wait and is merged back before the capture
    ends). Tensors allocated on the side stream are released only after the join and every later use of the side stream starts with a
    wait on the current one, so the caching allocator's per-stream reuse stays ordered.

    Only this framework's own recurrence kernels are issued on the side stream. Library GEMMs stay on the current stream: the fp32
    input-projection GEMM of the vendor library is a persistent stream-K style kernel whose workgroups wait for each other, and two of
    them launched from two streams at the same time were measured to dead-lock the device (each holds CUs the other needs;
    profiles/r2_lstm_persistent.md). The kernels of native/csrc never wait on another workgroup in their per-step form, so they can share
    the chip with anything. On the CPU, or with ``DL4J_AMD_BIDIR_STREAMS=0``, ``active`` is False and callers run in line."""

    def __init__(self, x):
        import os
        self.cur = self.side = None
        if x.is_cuda and os.environ.get("DL4J_AMD_BIDIR_STREAMS", "1") != "0":
            dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
            pool = getattr(_SIDE_TLS, "streams", None)
            if pool is None:
                pool = _SIDE_TLS.streams = {}
            if dev not in pool:
                pool[dev] = torch.cuda.Stream(device=dev)
            self.side = pool[dev]
            self.cur = torch.cuda.current_stream(dev)
        self.forked = False

    @property
    def active(self):
        return self.side is not None

    def fork(self, *inputs):
        """``inputs``: tensors (or containers of them, e.g. the forward state with its current-stream h0 / c0) allocated on the current
        stream that the side stream reads. They are held until the join, so that the caching
        allocator cannot hand their blocks to later current-stream work while the side stream still reads them."""
        self.side.wait_stream(self.cur)
        self.forked = True
        self.held = inputs
        from ...ops.streams import side_region
        return side_region(self.side, 'bidirectional side-stream')

    def join(self):
        if self.forked:
            self.cur.wait_stream(self.side)
            self.forked = False
        self.held = ()


def _overlap_forward(f, b, xf, xb, train):
    """(yf, yb) of two independent LSTMImpl directions; b's recurrence on the side stream when both run the native loop."""
    ss = _side_stream(xf)
    if not ss.active or not (isinstance(f, LSTMImpl) and isinstance(b, LSTMImpl)):
        return f.forward(xf, train), b.forward(xb, train)
    try:
        yb = b.forward(xb, train, fork=ss)
        yf = f.forward(xf, train)
    finally:
        ss.join()
    return yf, yb


def _overlap_backward(f, b, ef, eb):
    ss = _side_stream(ef)
    if not ss.active or not (isinstance(f, LSTMImpl) and isinstance(b, LSTMImpl)):
        return f.backward(ef), b.backward(eb)
    try:
        rec = b.backward_recur(eb, fork=ss)
        dxf = f.backward(ef)
    finally:
        ss.join()
    return dxf, b.backward_finish(rec)


class GravesBidirectionalLSTMImpl(LayerImpl):
    """Two Graves LSTMs (forward / backward in time), outputs ADDED (GravesBidirectionalLSTM.java)."""

    def __init__(self, conf, ctx):
        super().__init__(conf, ctx)
        self.f = LSTMImpl(conf, ctx); self.f.prefix = "F"
        self.b = LSTMImpl(conf, ctx); self.b.prefix = "B"; self.b.reverse = True

    def type(self):
        return "RECURRENT"

    def compute_replicas(self):
        return self.f.compute_replicas() + self.b.compute_replicas()

    def set_views(self, params, grads):
        super().set_views(params, grads)
        for l in (self.f, self.b):
            l.params, l.grads, l.cparams = params, grads, self.cparams

    def forward(self, x, train=False):
        self.f.cparams = self.b.cparams = self.cparams
        self.f.maskArray = self.b.maskArray = self.maskArray
        x = self._dropout_in(x, train)
        yf, yb = _overlap_forward(self.f, self.b, x, x, train)   # the two recurrences overlap on two HIP streams
        return yf + yb

    def backward(self, eps):
        dx, dxb = _overlap_backward(self.f, self.b, eps, eps)
        return self._dropout_back(dx + dxb) if dx is not None else None

    def rnnClearPreviousState(self):
        pass


class SimpleRnnImpl(AutogradLayerImpl):
    """out_t = act(x_t W + out_{t-1} RW + b) (recurrent/SimpleRnn.java)."""

    def type(self):
        return "RECURRENT"

    def __init__(self, conf, ctx):
        super().__init__(conf, ctx)
        self.stateMap = {}

    def rnnClearPreviousState(self):
        self.stateMap = {}

    def rnnGetPreviousState(self):
        return dict(self.stateMap)

    def rnnSetPreviousState(self, s):
        self.stateMap = dict(s)

    def _forward(self, x, P, train, initial_state=None):
        mb, nIn, T = x.shape
        n = self.conf.nOut
        W, RW, b = P["W"], P["RW"], P["b"].reshape(-1)
        h = initial_state if initial_state is not None else torch.zeros(mb, n, dtype=x.dtype, device=x.device)
        outs = []
        ln = self.conf.hasLayerNorm
        for t in range(T):
            a = x[:, :, t] @ W
            r = h @ RW
            if ln:
                g = P["g"].reshape(-1)
                a = torch.nn.functional.layer_norm(a, (n,), g[:n], None, 1e-5)
                r = torch.nn.functional.layer_norm(r, (n,), g[n:], None, 1e-5)
            h = self.conf.activationFn.forward(a + r + b, train)
            if self.maskArray is not None:
                h = h * self.maskArray[:, t].reshape(mb, 1).to(h.dtype)
            outs.append(h)
        self._last_state = (h,)
        return torch.stack(outs, dim=2)

    def forward(self, x, train=False, initial_state=None, store_state=False):
        if initial_state is not None or store_state:
            with torch.no_grad():
                st = initial_state[0] if initial_state is not None else None
                y = self._forward(x, {k: self.w(k) for k in self.params}, False, st)
            if store_state:
                self.stateMap = {"prevActivation": self._last_state[0].detach()}
            return y
        return super().forward(x, train)


class BidirectionalImpl(LayerImpl):
    """Wrapper running a recurrent layer forwards and backwards in time (recurrent/Bidirectional.java)."""

    def __init__(self, conf, ctx):
        super().__init__(conf, ctx)
        from . import registry
        self.fwd = registry.create(conf.fwd, ctx)
        self.bwd = registry.create(conf.bwd, ctx)

    def type(self):
        return "RECURRENT"

    def compute_replicas(self):
        return [("f" + k, l) for k, l in self.fwd.compute_replicas()] + [("b" + k, l) for k, l in self.bwd.compute_replicas()]

    def set_views(self, params, grads):
        super().set_views(params, grads)
        self.fwd.set_views({k[1:]: v for k, v in params.items() if k.startswith("f")},
                           {k[1:]: v for k, v in grads.items() if k.startswith("f")})
        self.bwd.set_views({k[1:]: v for k, v in params.items() if k.startswith("b")},
                           {k[1:]: v for k, v in grads.items() if k.startswith("b")})

    def _sync_replicas(self):
        self.fwd.cparams = {k[1:]: v for k, v in self.cparams.items() if k.startswith("f")}
        self.bwd.cparams = {k[1:]: v for k, v in self.cparams.items() if k.startswith("b")}

    def forward(self, x, train=False):
        self._sync_replicas()
        self.fwd.maskArray = self.maskArray
        self.bwd.maskArray = None if self.maskArray is None else torch.flip(self.maskArray, dims=[1])
        # the two directions do not depend on each other until the merge: their recurrences overlap on two HIP streams
        yf, yb = _overlap_forward(self.fwd, self.bwd, x, torch.flip(x, dims=[2]), train)
        yb = torch.flip(yb, dims=[2])
        m = self.conf.mode
        self._yf, self._yb = (yf, yb) if m == "MUL" else (None, None)
        if m == "CONCAT":
            return torch.cat([yf, yb], dim=1)
        if m == "ADD":
            return yf + yb
        if m == "MUL":
            return yf * yb
        return 0.5 * (yf + yb)

    def backward(self, eps):
        m = self.conf.mode
        if m == "CONCAT":
            n = eps.shape[1] // 2
            ef, eb = eps[:, :n], eps[:, n:]
        elif m == "ADD":
            ef = eb = eps
        elif m == "MUL":
            ef, eb = eps * self._yb, eps * self._yf
        else:
            ef = eb = 0.5 * eps
        dxf, dxb = _overlap_backward(self.fwd, self.bwd, ef.contiguous(), torch.flip(eb, dims=[2]).contiguous())
        if dxf is None:
            return None
        return dxf + torch.flip(dxb, dims=[2])

    def rnnClearPreviousState(self):
        for l in (self.fwd, self.bwd):
            if hasattr(l, "rnnClearPreviousState"):
                l.rnnClearPreviousState()


class _WrapperImpl(LayerImpl):
    def __init__(self, conf, ctx):
        super().__init__(conf, ctx)
        from . import registry
        self.underlying = registry.create(conf.underlying, ctx)

    def type(self):
        return self.underlying.type()

    def compute_replicas(self):
        return self.underlying.compute_replicas()

    def set_views(self, params, grads):
        super().set_views(params, grads)
        self.underlying.set_views(params, grads)

    def _sync(self):
        self.underlying.cparams = self.cparams
        self.underlying.maskArray = self.maskArray
        self.underlying.needs_input_grad = getattr(self, "needs_input_grad", True)

    def forward(self, x, train=False):
        self._sync()
        return self.underlying.forward(x, train)

    def backward(self, eps):
        return self.underlying.backward(eps)

    def clear(self):
        super().clear()
        self.underlying.clear()

    def __getattr__(self, name):
        if name.startswith("rnn"):
            return getattr(self.underlying, name)
        raise AttributeError(name)


class LastTimeStepImpl(_WrapperImpl):
    def type(self):
        return "FEED_FORWARD"

    def forward(self, x, train=False):
        self._sync()
        y = self.underlying.forward(x, train)   # [mb, n, T]
        self._T = y.shape[2]
        m = self.maskArray
        if m is None:
            self._idx = None
            return y[:, :, -1]
        idx = (m.shape[1] - 1 - torch.flip(m > 0, dims=[1]).int().argmax(dim=1)).long()
        self._idx = idx
        return y[torch.arange(y.shape[0], device=y.device), :, idx]

    def backward(self, eps):
        mb, n = eps.shape
        e = torch.zeros(mb, n, self._T, dtype=eps.dtype, device=eps.device)
        if self._idx is None:
            e[:, :, -1] = eps
        else:
            e[torch.arange(mb, device=eps.device), :, self._idx] = eps
        return self.underlying.backward(e)

    def feedForwardMaskArray(self, mask, minibatch=None):
        self.maskArray = mask
        return None


class TimeDistributedImpl(_WrapperImpl):
    def type(self):
        return "RECURRENT"

    def forward(self, x, train=False):
        self._sync()
        mb, n, T = x.shape
        self._mbT = (mb, T)
        y2 = self.underlying.forward(x.permute(0, 2, 1).reshape(mb * T, n), train)
        return y2.reshape(mb, T, -1).permute(0, 2, 1)

    def backward(self, eps):
        mb, T = self._mbT
        d2 = self.underlying.backward(eps.permute(0, 2, 1).reshape(mb * T, -1))
        if d2 is None:
            return None
        return d2.reshape(mb, T, -1).permute(0, 2, 1)


class MaskZeroLayerImpl(_WrapperImpl):
    def forward(self, x, train=False):
        # mask timesteps whose input is entirely == maskingValue
        m = (x != self.conf.maskingValue).any(dim=1).to(x.dtype)   # [mb, T]
        self.maskArray = m
        self._sync()
        return self.underlying.forward(x * m.unsqueeze(1), train)


class FrozenLayerImpl(_WrapperImpl):
    """Parameters are never updated; no gradients computed for them (misc/FrozenLayer.java)."""

    def forward(self, x, train=False):
        self._sync()
        return self.underlying.forward(x, False)

    def backward(self, eps):
        for g in self.grads.values():
            g.zero_()
        return None


class FrozenLayerWithBackpropImpl(_WrapperImpl):
    def forward(self, x, train=False):
        self._sync()
        return self.underlying.forward(x, train)

    def backward(self, eps):
        dx = self.underlying.backward(eps)
        for g in self.grads.values():
            g.zero_()
        return dx


class MaskLayerImpl(LayerImpl):
    def forward(self, x, train=False):
        m = self.maskArray
        if m is None:
            return x
        if x.dim() == 3:
            return x * m.reshape(x.shape[0], 1, x.shape[2]).to(x.dtype)
        return x * m.reshape(x.shape[0], *([1] * (x.dim() - 1))).to(x.dtype)

    def backward(self, eps):
        return self.forward(eps)


class RepeatVectorImpl(LayerImpl):
    def forward(self, x, train=False):
        return x.unsqueeze(2).expand(x.shape[0], x.shape[1], self.conf.n)

    def backward(self, eps):
        return eps.sum(2)
