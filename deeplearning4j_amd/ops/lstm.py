"""LSTM cell pointwise math (DL4J gate order [a | f | o | g], Graves peepholes optional).

Equations: SURVEY Appendix B.7 / nn/layers/recurrent/LSTMHelpers.java:227-369.
GPU fast path (tanh/sigmoid, fp32/bf16): fused HIP kernels in native/csrc/lstm.hip — one launch per
timestep for all gate math instead of ~20 elementwise ops. Generic path: PyTorch with arbitrary
IActivation (used on CPU, fp64 gradient checks and exotic activations).
"""
from __future__ import annotations

import torch

from ..nd4j.activations import ActivationSigmoid, ActivationTanH
from ..native import lib as L


def _fast(gates, act, gate_act):
    return (L.use_hip(gates) and gates.dtype in (torch.float32, torch.bfloat16) and isinstance(act, ActivationTanH)
            and isinstance(gate_act, ActivationSigmoid) and gates.is_contiguous() and hasattr(L.hip(), "dl4j_lstm_cell_fwd"))


def cell_forward(gates, c_prev, peep, act, gate_act, mask, train):
    mb, n4 = gates.shape
    n = n4 // 4
    if _fast(gates, act, gate_act):
        h = torch.empty(mb, n, dtype=gates.dtype, device=gates.device)
        c = torch.empty_like(h)
        sv = torch.empty(mb, 6 * n, dtype=gates.dtype, device=gates.device) if train else None  # a f o g c tanh(c)
        pf, po, pg = peep if peep is not None else (None, None, None)
        if peep is not None:
            pf, po, pg = pf.contiguous().float(), po.contiguous().float(), pg.contiguous().float()
        cp = c_prev.contiguous()
        mk = mask.reshape(-1).float().contiguous() if mask is not None else None
        L.check(L.hip().dl4j_lstm_cell_fwd(L.ptr(gates), L.ptr(cp), L.ptr(pf), L.ptr(po), L.ptr(pg), L.ptr(mk),
                                           L.ptr(h), L.ptr(c), L.ptr(sv), L.c_int(mb), L.c_int(n),
                                           L.c_int(L.dtype_code(gates)), L.stream()), "lstm_cell_fwd")
        return h, c, ("fast", sv, (pf, po, pg))
    za, zf, zo, zg = gates[:, :n], gates[:, n:2 * n], gates[:, 2 * n:3 * n], gates[:, 3 * n:]
    if peep is not None:
        wFF, wOO, wGG = (p.to(gates.dtype) for p in peep)
        zf = zf + c_prev * wFF
        zg = zg + c_prev * wGG
    a = act.forward(za, train)
    f = gate_act.forward(zf, train)
    g = gate_act.forward(zg, train)
    c = f * c_prev + g * a
    if peep is not None:
        zo = zo + c * wOO
    o = gate_act.forward(zo, train)
    ac = act.forward(c, train)
    h = o * ac
    sv = ("ref", (za, zf, zo, zg, a, f, o, g, c, ac)) if train else None
    if mask is not None:
        h = h * mask
        c = c * mask
    return h, c, sv


def cell_backward(dh, dc_next, c_prev, sv, peep, act, gate_act, mask):
    if sv[0] == "fast":
        _, s, (pf, po, pg) = sv
        mb, n6 = s.shape
        n = n6 // 6
        dgates = torch.empty(mb, 4 * n, dtype=s.dtype, device=s.device)
        dc_prev = torch.empty(mb, n, dtype=s.dtype, device=s.device)
        dpeep = torch.zeros(3, n, dtype=torch.float32, device=s.device) if pf is not None else None
        mk = mask.reshape(-1).float().contiguous() if mask is not None else None
        L.check(L.hip().dl4j_lstm_cell_bwd(L.ptr(dh.contiguous()), L.ptr(dc_next.contiguous()), L.ptr(c_prev.contiguous()),
                                           L.ptr(s), L.ptr(pf), L.ptr(po), L.ptr(pg), L.ptr(mk), L.ptr(dgates),
                                           L.ptr(dc_prev), L.ptr(dpeep), L.c_int(mb), L.c_int(n),
                                           L.c_int(L.dtype_code(s)), L.stream()), "lstm_cell_bwd")
        return dgates, dc_prev, (None if dpeep is None else (dpeep[0], dpeep[1], dpeep[2]))
    za, zf, zo, zg, a, f, o, g, c, ac = sv[1]
    if mask is not None:
        dh = dh * mask
        dc_next = dc_next * mask
    dzo = gate_act.backprop(zo, dh * ac)
    dc = dc_next + act.backprop(c, dh * o)
    if peep is not None:
        wFF, wOO, wGG = (p.to(dh.dtype) for p in peep)
        dc = dc + dzo * wOO
    dza = act.backprop(za, dc * g)
    dzf = gate_act.backprop(zf, dc * c_prev)
    dzg = gate_act.backprop(zg, dc * a)
    dc_prev = dc * f
    dp = None
    if peep is not None:
        dc_prev = dc_prev + dzf * wFF + dzg * wGG
        dp = ((dzf * c_prev).sum(0).float(), (dzo * c).sum(0).float(), (dzg * c_prev).sum(0).float())
    return torch.cat([dza, dzf, dzo, dzg], dim=1), dc_prev, dp
