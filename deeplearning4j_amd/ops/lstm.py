"""LSTM cell pointwise math (DL4J gate order [a | f | o | g], Graves peepholes optional).

Equations: SURVEY Appendix B.7 / nn/layers/recurrent/LSTMHelpers.java:227-369.
GPU fast path (tanh/sigmoid, fp32/bf16): fused HIP kernels in native/csrc/lstm.hip — one launch per
timestep for all gate math instead of ~20 elementwise ops. Generic path: PyTorch with arbitrary
IActivation (used on CPU, fp64 gradient checks and exotic activations).
"""
from __future__ import annotations

import ctypes
import os

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


# ------------------------------------------------------------------------------------------------------ native sequence loops
def native_seq_ok(xw, RW4, n, act, gate_act):
    """Whole-sequence C++ loop (native/csrc/rnn_seq.hip): GPU, fp32 / bf16, tanh + sigmoid, n a multiple of the 16-byte chunk, and
    the DL4J 'f'-order recurrent weights whose memory image is the [4n, n] NT operand."""
    if not (L.use_hip(xw) and xw.dtype in (torch.float32, torch.bfloat16) and isinstance(act, ActivationTanH)
            and isinstance(gate_act, ActivationSigmoid) and xw.is_contiguous()):
        return False
    ch = 4 if xw.dtype == torch.float32 else 8
    return n % ch == 0 and RW4.dtype == xw.dtype and RW4.t().is_contiguous() and RW4.data_ptr() % 16 == 0


_PERSIST = os.environ.get("DL4J_AMD_LSTM_PERSIST", "1") == "1"
_PERSIST_CHECK = os.environ.get("DL4J_AMD_LSTM_PERSIST_CHECK", "0") == "1"
# "steps": the fused kernels launched once per time step (default: a kernel boundary is the cheapest all-to-all hand-off of h on
# MI355X, profiles/r2_lstm_persistent.md); "grid": one launch per sequence with in-kernel hand-offs between the workgroups
_PERSIST_MODE = os.environ.get("DL4J_AMD_LSTM_PERSIST_MODE", "steps")


def persist_ok(xw, RW4, mb, n):
    """bf16 sequences whose shape the persistent kernel supports (n % 8 == 0, n <= 2048, mb <= 128, weights + hidden state fit LDS)."""
    return _PERSIST and xw.dtype == torch.bfloat16 and RW4.dtype == torch.bfloat16 and bool(L.hip().dl4j_lstm_persist_supported(L.c_int(mb), L.c_int(n)))


def steps_f32_ok(xw, RW4, mb, n):
    """fp32 sequences the fused fp32 step kernels support (n % 64 == 0, n <= 2048, mb <= 128)."""
    return _PERSIST and xw.dtype == torch.float32 and RW4.dtype == torch.float32 and bool(L.hip().dl4j_lstm_steps_f32_supported(L.c_int(mb), L.c_int(n)))


_DIAG1 = os.environ.get("DL4J_AMD_LSTM_DIAG1", "1") != "0"


def diag1_ok(xw, mb, n):
    """Sequences on the one-tile-per-workgroup step kernels (lstm_diag_*_kernel with a single layer; masks and the reverse direction
    included): bound by operand bytes per CU, they move a third of what the 8-unit x all-rows step kernels move."""
    return (_DIAG1 and _PERSIST and _PERSIST_MODE == "steps" and xw.dtype in (torch.bfloat16, torch.float32)
            and bool(L.hip().dl4j_lstm_diag_supported(L.c_int(mb), L.c_int(n), L.c_int(1))))


def _p1(t):
    return (ctypes.c_void_p * 1)(None if t is None else t.data_ptr())


def seq_forward(xw, RW4, h0, c0, peep, mask, train, reverse):
    """xw [T, mb, 4n]. Returns (H [T, mb, n], state dict for seq_backward)."""
    T, mb, n4 = xw.shape
    n = n4 // 4
    dev, dt = xw.device, xw.dtype
    H = torch.empty(T, mb, n, device=dev, dtype=dt)
    Cs = torch.empty(T, mb, n, device=dev, dtype=dt)
    SV = torch.empty(T, mb, 6 * n, device=dev, dtype=dt) if train else None
    ws = torch.empty(mb, n4, device=dev, dtype=dt)
    pf = po = pg = None
    if peep is not None:
        pf, po, pg = (p.contiguous().float() for p in peep)
    maskT = mask.to(dev).t().contiguous().float() if mask is not None else None
    h0c, c0c = h0.to(dt).contiguous(), c0.to(dt).contiguous()
    if diag1_ok(xw, mb, n) and RW4.dtype == dt:
        nins = (ctypes.c_int * 1)(0)
        L.check(L.hip().dl4j_lstm_diag_fwd(L.ptr(xw), _p1(RW4), _p1(None), _p1(None), _p1(None), _p1(h0c), _p1(c0c),
                                           _p1(pf) if pf is not None else None, _p1(po) if pf is not None else None,
                                           _p1(pg) if pf is not None else None, _p1(H), _p1(Cs), _p1(SV) if SV is not None else None, nins,
                                           L.c_int(1), L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(0 if dt == torch.bfloat16 else 1),
                                           L.ptr(maskT), L.c_int(1 if reverse else 0), L.stream()), "lstm_diag_fwd")
        return H, {"H": H, "Cs": Cs, "SV": SV, "h0": h0c, "c0": c0c, "peep": (pf, po, pg), "maskT": maskT}
    if persist_ok(xw, RW4, mb, n):
        # one launch for the whole sequence (native/csrc/lstm_persist.hip): weights resident in LDS, cell state in registers
        bar = torch.empty(2, device=dev, dtype=torch.int32)
        fn = L.hip().dl4j_lstm_persist_fwd if _PERSIST_MODE == "grid" else L.hip().dl4j_lstm_steps_fwd
        L.check(fn(L.ptr(xw), L.ptr(RW4), L.c_int64(n), L.ptr(h0c), L.ptr(c0c), L.ptr(pf), L.ptr(po), L.ptr(pg), L.ptr(maskT), L.ptr(H), L.ptr(Cs),
                   L.ptr(SV), L.ptr(bar), L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(1 if reverse else 0), L.stream()), "lstm_fused_fwd")
        if _PERSIST_MODE == "grid" and _PERSIST_CHECK and int(bar[1]) != 0:
            raise RuntimeError("lstm_persist_fwd: inter-workgroup hand-off timed out (workgroups were not co-resident)")
        return H, {"H": H, "Cs": Cs, "SV": SV, "h0": h0c, "c0": c0c, "peep": (pf, po, pg), "maskT": maskT}
    if steps_f32_ok(xw, RW4, mb, n):
        L.check(L.hip().dl4j_lstm_steps_fwd_f32(L.ptr(xw), L.ptr(RW4), L.c_int64(n), L.ptr(h0c), L.ptr(c0c), L.ptr(pf), L.ptr(po), L.ptr(pg),
                                                L.ptr(maskT), L.ptr(H), L.ptr(Cs), L.ptr(SV), L.c_int(T), L.c_int(mb), L.c_int(n),
                                                L.c_int(1 if reverse else 0), L.stream()), "lstm_steps_fwd_f32")
        return H, {"H": H, "Cs": Cs, "SV": SV, "h0": h0c, "c0": c0c, "peep": (pf, po, pg), "maskT": maskT}
    L.check(L.hip().dl4j_lstm_seq_fwd(L.ptr(xw), L.ptr(RW4), L.c_int64(n), L.ptr(h0c), L.ptr(c0c), L.ptr(pf), L.ptr(po), L.ptr(pg),
                                      L.ptr(maskT), L.ptr(H), L.ptr(Cs), L.ptr(SV), L.ptr(ws), L.c_int(T), L.c_int(mb), L.c_int(n),
                                      L.c_int(L.dtype_code(xw)), L.c_int(1 if reverse else 0), L.stream()), "lstm_seq_fwd")
    return H, {"H": H, "Cs": Cs, "SV": SV, "h0": h0c, "c0": c0c, "peep": (pf, po, pg), "maskT": maskT}


def seq_backward(E, RW4, st, reverse, dc_last=None):
    """E [T, mb, n] = dL/dH. Returns (dG [T, mb, 4n], Hprev [T, mb, n], [dwFF, dwOO, dwGG] or None)."""
    T, mb, n = E.shape
    dev, dt = E.device, E.dtype
    E = E.contiguous()
    dcl = None if dc_last is None else dc_last.float().contiguous()      # gradient into the last cell state (fused step kernels only)
    st.pop("dc0_grad", None)
    rw4t = RW4.contiguous()                       # [n, 4n]: the NT operand of dh_prev = dgates . RW4^T
    dG = torch.empty(T, mb, 4 * n, device=dev, dtype=dt)
    pf, po, pg = st["peep"]
    dpeep = torch.zeros(3, n, device=dev, dtype=torch.float32) if pf is not None else None
    if dcl is None and st["SV"] is not None and rw4t.dtype == dt and diag1_ok(E, mb, n):
        dc_ws = torch.empty(2, mb, n, device=dev, dtype=torch.float32)
        L.check(L.hip().dl4j_lstm_diag_bwd(L.ptr(E), _p1(rw4t), _p1(None), _p1(st["SV"]), _p1(st["Cs"]), _p1(st["c0"]),
                                           _p1(pf) if pf is not None else None, _p1(po) if pf is not None else None,
                                           _p1(pg) if pf is not None else None, _p1(dG), _p1(dpeep) if pf is not None else None, _p1(dc_ws),
                                           L.c_int(1), L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(0 if dt == torch.bfloat16 else 1),
                                           L.ptr(st["maskT"]), L.c_int(1 if reverse else 0), L.stream()), "lstm_diag_bwd")
        st["dc0_grad"] = dc_ws[(T - 1) % 2]
        return dG, _hprev(st, reverse), (None if dpeep is None else [dpeep[0], dpeep[1], dpeep[2]])
    if persist_ok(E, rw4t, mb, n) and st["SV"] is not None:
        bar = torch.empty(2, device=dev, dtype=torch.int32)
        if _PERSIST_MODE == "grid":
            L.check(L.hip().dl4j_lstm_persist_bwd(L.ptr(E), L.ptr(rw4t), L.c_int64(4 * n), L.ptr(st["SV"]), L.ptr(st["Cs"]), L.ptr(st["c0"]),
                                                  L.ptr(pf), L.ptr(po), L.ptr(pg), L.ptr(st["maskT"]), L.ptr(dG), L.ptr(dpeep), L.ptr(bar),
                                                  L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(1 if reverse else 0), L.stream()), "lstm_persist_bwd")
            if _PERSIST_CHECK and int(bar[1]) != 0:
                raise RuntimeError("lstm_persist_bwd: grid barrier timed out (workgroups were not co-resident)")
        else:
            dc_ws = torch.empty(2, mb, n, device=dev, dtype=torch.float32)
            L.check(L.hip().dl4j_lstm_steps_bwd(L.ptr(E), L.ptr(rw4t), L.c_int64(4 * n), L.ptr(st["SV"]), L.ptr(st["Cs"]), L.ptr(st["c0"]),
                                                L.ptr(pf), L.ptr(po), L.ptr(pg), L.ptr(st["maskT"]), L.ptr(dG), L.ptr(dpeep), L.ptr(bar),
                                                L.ptr(dc_ws), L.ptr(dcl), L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(1 if reverse else 0),
                                                L.stream()), "lstm_steps_bwd")
            st["dc0_grad"] = dc_ws[(T - 1) % 2]
        return dG, _hprev(st, reverse), (None if dpeep is None else [dpeep[0], dpeep[1], dpeep[2]])
    if steps_f32_ok(E, rw4t, mb, n) and st["SV"] is not None:
        dc_ws = torch.empty(2, mb, n, device=dev, dtype=torch.float32)
        L.check(L.hip().dl4j_lstm_steps_bwd_f32(L.ptr(E), L.ptr(rw4t), L.c_int64(4 * n), L.ptr(st["SV"]), L.ptr(st["Cs"]), L.ptr(st["c0"]),
                                                L.ptr(pf), L.ptr(po), L.ptr(pg), L.ptr(st["maskT"]), L.ptr(dG), L.ptr(dpeep), L.ptr(dc_ws),
                                                L.ptr(dcl), L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(1 if reverse else 0), L.stream()),
                "lstm_steps_bwd_f32")
        st["dc0_grad"] = dc_ws[(T - 1) % 2]
        return dG, _hprev(st, reverse), (None if dpeep is None else [dpeep[0], dpeep[1], dpeep[2]])
    dh_ws = torch.empty(mb, n, device=dev, dtype=dt)
    dc_ws = torch.empty(2, mb, n, device=dev, dtype=dt)
    L.check(L.hip().dl4j_lstm_seq_bwd(L.ptr(E), L.ptr(rw4t), L.c_int64(4 * n), L.ptr(st["SV"]), L.ptr(st["Cs"]), L.ptr(st["c0"]),
                                      L.ptr(pf), L.ptr(po), L.ptr(pg), L.ptr(st["maskT"]), L.ptr(dG), L.ptr(dpeep), L.ptr(dh_ws),
                                      L.ptr(dc_ws), L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(L.dtype_code(E)),
                                      L.c_int(1 if reverse else 0), L.stream()), "lstm_seq_bwd")
    return dG, _hprev(st, reverse), (None if dpeep is None else [dpeep[0], dpeep[1], dpeep[2]])


def _hprev(st, reverse):
    """Hidden state that fed every step: H shifted by one step with h0 at the sequence start."""
    H, h0 = st["H"], st["h0"]
    if reverse:
        return torch.cat([H[1:], h0.unsqueeze(0)], 0)
    return torch.cat([h0.unsqueeze(0), H[:-1]], 0)


# ------------------------------------------------------------------------------------------------------ GRU (libnd4j gru / gruCell)
def gru_native_ok(x, n):
    ch = 4 if x.dtype == torch.float32 else 8
    return L.use_hip(x) and x.dtype in (torch.float32, torch.bfloat16) and n % ch == 0


def gru_cell(xz, hz, hprev, train=True):
    """Fused pointwise GRU cell: returns (h, sv = [r | u | c] or None)."""
    mb, n = hprev.shape
    h = torch.empty_like(hprev)
    sv = torch.empty(mb, 3 * n, device=hprev.device, dtype=hprev.dtype) if train else None
    L.check(L.hip().dl4j_gru_cell_fwd(L.ptr(xz.contiguous()), L.ptr(hz.contiguous()), L.ptr(hprev.contiguous()), L.ptr(h), L.ptr(sv),
                                      L.c_int(mb), L.c_int(n), L.c_int(L.dtype_code(hprev)), L.stream()), "gru_cell_fwd")
    return h, sv


def gru_cell_backward(dh, sv, hz, hprev):
    """Returns (dxz, dhz, dhp_direct): gradients of the two pre-activation operands and the direct part of dL/dh_prev."""
    mb, n = hprev.shape
    dxz = torch.empty(mb, 3 * n, device=dh.device, dtype=dh.dtype)
    dhz = torch.empty_like(dxz)
    dhp = torch.empty_like(hprev)
    L.check(L.hip().dl4j_gru_cell_bwd(L.ptr(dh.contiguous()), L.ptr(sv), L.ptr(hz.contiguous()), L.ptr(hprev.contiguous()), L.ptr(dxz),
                                      L.ptr(dhz), L.ptr(dhp), L.c_int(mb), L.c_int(n), L.c_int(L.dtype_code(dh)), L.stream()),
            "gru_cell_bwd")
    return dxz, dhz, dhp


def gru_fused_ok(xz, n):
    """bf16 / fp32 sequences whose shape the fused step kernels (native/csrc/gru_seq.hip) take: one launch per step in each direction."""
    return (os.environ.get("DL4J_AMD_GRU_FUSED", "1") != "0" and L.use_hip(xz) and xz.dtype in (torch.bfloat16, torch.float32) and xz.dim() == 3
            and bool(L.hip().dl4j_gru_steps_supported(L.c_int(xz.shape[1]), L.c_int(n))))


class _GruSeqFused(torch.autograd.Function):
    """gru layer on the fused step kernels: per time step ONE launch does the three-gate recurrent GEMM (MFMA), the gate math and the
    blend; backwards ONE launch does dHZ_{t+1} . Wh^T and the cell backward. Weight gradients are big GEMMs over all steps."""

    @staticmethod
    def forward(ctx, xz, h0, Wh):
        T, mb, n3 = xz.shape
        n = n3 // 3
        train = any(ctx.needs_input_grad)          # (grad mode is off inside forward: is_grad_enabled() would always say no)
        wh_t = Wh.t().contiguous()                               # [3n, n]: row = gate column, K contiguous
        H = torch.empty(T, mb, n, device=xz.device, dtype=xz.dtype)
        SV = torch.empty(T, mb, 4 * n, device=xz.device, dtype=xz.dtype) if train else None
        h0c, xzc = h0.contiguous(), xz.contiguous()
        L.check(L.hip().dl4j_gru_steps_fwd(L.ptr(xzc), L.ptr(wh_t), L.c_int64(n), L.ptr(h0c), L.ptr(H), L.ptr(SV), L.c_int(T), L.c_int(mb),
                                           L.c_int(n), L.c_int(0 if xz.dtype == torch.bfloat16 else 1), L.stream()), "gru_steps_fwd")
        ctx.save_for_backward(H, SV, h0c, Wh)
        return H

    @staticmethod
    def backward(ctx, dH):
        H, SV, h0, Wh = ctx.saved_tensors
        if SV is None:
            raise RuntimeError("gru: backward called on a forward pass that saved no activations")
        T, mb, n = H.shape
        dXZ = torch.empty(T, mb, 3 * n, device=H.device, dtype=H.dtype)
        dHZ = torch.empty_like(dXZ)
        carry = torch.empty(2, mb, n, device=H.device, dtype=torch.float32)
        whc = Wh.contiguous()                                    # [n, 3n]: row j = gate weights leaving unit j
        rc = L.hip().dl4j_gru_steps_bwd(L.ptr(dH.contiguous()), L.ptr(whc), L.c_int64(3 * n), L.ptr(SV), L.ptr(H), L.ptr(h0), L.ptr(dXZ),
                                        L.ptr(dHZ), L.ptr(carry), L.c_int(T), L.c_int(mb), L.c_int(n),
                                        L.c_int(0 if H.dtype == torch.bfloat16 else 1), L.stream())
        if rc < 0:
            raise RuntimeError("dl4j_gru_steps_bwd failed with status %d" % rc)
        dh0 = (carry[rc] + (dHZ[0] @ whc.t()).float()).to(H.dtype)
        Hprev = torch.cat([h0.unsqueeze(0), H[:-1]], 0)
        dWh = Hprev.reshape(T * mb, n).t() @ dHZ.reshape(T * mb, 3 * n)
        return dXZ, dh0, dWh


class _GruSeq(torch.autograd.Function):
    """gru layer on the native loop: forward = C++ time loop (recurrent GEMM on the MFMA kernel + fused cell), backward = fused cell
    backward per step + one GEMM per step for dh, the weight gradients as big GEMMs over all steps."""

    @staticmethod
    def forward(ctx, xz, h0, Wh):
        T, mb, n3 = xz.shape
        n = n3 // 3
        dev, dt = xz.device, xz.dtype
        wh_t = Wh.t().contiguous()                               # [3n, n] NT operand
        H = torch.empty(T, mb, n, device=dev, dtype=dt)
        HZ = torch.empty(T, mb, n3, device=dev, dtype=dt)
        SV = torch.empty(T, mb, n3, device=dev, dtype=dt)
        h0c = h0.contiguous()
        L.check(L.hip().dl4j_gru_seq_fwd(L.ptr(xz.contiguous()), L.ptr(wh_t), L.c_int64(n), L.ptr(h0c), L.ptr(H), L.ptr(HZ), L.ptr(SV),
                                         None, L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(L.dtype_code(xz)), L.stream()),
                "gru_seq_fwd")
        ctx.save_for_backward(H, HZ, SV, h0c, Wh)
        return H

    @staticmethod
    def backward(ctx, dH):
        H, HZ, SV, h0, Wh = ctx.saved_tensors
        T, mb, n = H.shape
        dXZ = torch.empty_like(HZ)
        dHZ = torch.empty_like(HZ)
        dh = torch.zeros(mb, n, device=H.device, dtype=H.dtype)
        whc = Wh.contiguous()                                    # [n, 3n]: dh_prev += dhz . Wh^T  ==  gemm_nt(dhz, Wh)
        from . import gemm as G
        for t in range(T - 1, -1, -1):
            hprev = H[t - 1] if t > 0 else h0
            dxz, dhz, dhp = gru_cell_backward(dH[t] + dh, SV[t], HZ[t], hprev)
            dXZ[t], dHZ[t] = dxz, dhz
            dh = G.gemm_nt(dhz, whc, residual=dhp) if G.supported(dhz, whc) else dhp + dhz @ whc.t()
        Hprev = torch.cat([h0.unsqueeze(0), H[:-1]], 0)
        dWh = Hprev.reshape(T * mb, n).t() @ dHZ.reshape(T * mb, 3 * n)
        return dXZ, dh, dWh


def gru_sequence(x, h0, Wx, Wh, b):
    """x [T, mb, nIn] -> H [T, mb, n] (libnd4j `gru`)."""
    xz = x @ Wx
    if b is not None:
        xz = xz + b
    if gru_fused_ok(xz, Wh.shape[0]):
        return _GruSeqFused.apply(xz, h0, Wh)
    return _GruSeq.apply(xz, h0, Wh)


# ------------------------------------------------------------------------------------------------------ lstmLayer op on the fused kernels
def _lib_to_kernel_perm(n, device):
    """Column permutation from libnd4j's gate order [i | f | c' | o] (lstmLayer.cpp:200-249) to the kernels' DL4J order
    [candidate | forget | output | input]."""
    a = torch.arange(n, device=device)
    return torch.cat([a + 2 * n, a + n, a + 3 * n, a])


class _LstmLayerFn(torch.autograd.Function):
    """One direction of the libnd4j ``lstmLayer`` op (sigmoid gates, tanh cell / output activation, no clipping, full-length
    sequences) on the fused sequence kernels — the role cuDNN plays for the reference (platform/cudnn/lstmLayer.cu). Inputs in libnd4j
    layout: xs [T, B, nIn], Wx [nIn, 4n], Wr [n, 4n], bias [4n] or None, h0 / c0 [B, n], Wp [3n] or None. Returns (H [T, B, n], hL, cL)."""

    @staticmethod
    def forward(ctx, xs, Wx, Wr, bias, h0, c0, Wp, reverse):
        T, B, nIn = xs.shape
        n = Wr.shape[0]
        perm = _lib_to_kernel_perm(n, xs.device)
        Wxp, Wrp = Wx.index_select(1, perm), Wr.index_select(1, perm)
        xw = xs.reshape(T * B, nIn) @ Wxp
        if bias is not None:
            xw = xw + bias.index_select(0, perm)
        xw = xw.reshape(T, B, 4 * n).contiguous()
        RW4 = Wrp.t().contiguous().t()                   # [n, 4n] view of a [4n, n] K-contiguous image
        peep = None if Wp is None else (Wp[n:2 * n], Wp[2 * n:], Wp[:n])          # (forget, output, input) peepholes
        H, st = seq_forward(xw, RW4, h0, c0, peep, None, True, bool(reverse))
        last = 0 if reverse else T - 1
        ctx.save_for_backward(xs, Wxp, perm)
        ctx.st, ctx.RW4, ctx.reverse, ctx.has_bias, ctx.has_peep, ctx.last = st, RW4, bool(reverse), bias is not None, Wp is not None, last
        return H, H[last].clone(), st["Cs"][last].clone()

    @staticmethod
    def backward(ctx, gH, ghL, gcL):
        xs, Wxp, perm = ctx.saved_tensors
        T, B, nIn = xs.shape
        n = ctx.RW4.shape[0]
        E = torch.zeros(T, B, n, device=xs.device, dtype=xs.dtype) if gH is None else gH.to(xs.dtype).contiguous().clone()
        if ghL is not None:
            E[ctx.last] += ghL.to(xs.dtype)
        dG, Hprev, dpeep = seq_backward(E, ctx.RW4, ctx.st, ctx.reverse, dc_last=gcL)
        dc0 = ctx.st.get("dc0_grad")
        if dc0 is None and (gcL is not None or ctx.needs_input_grad[5]):
            raise NotImplementedError("lstmLayer: cell-state gradients need the fused step kernels (DL4J_AMD_LSTM_PERSIST_MODE=steps)")
        dG2 = dG.reshape(T * B, 4 * n)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(4 * n, device=perm.device)
        dWx = (xs.reshape(T * B, nIn).t() @ dG2).index_select(1, inv)
        dWr = (Hprev.reshape(T * B, n).t() @ dG2).index_select(1, inv)
        db = dG2.float().sum(0).to(xs.dtype).index_select(0, inv) if ctx.has_bias else None
        dx = (dG2 @ Wxp.t()).reshape(T, B, nIn)
        dWp = torch.cat([dpeep[2], dpeep[0], dpeep[1]]).to(xs.dtype) if ctx.has_peep else None
        first = T - 1 if ctx.reverse else 0                       # the step that consumed h0 / c0
        dh0 = (dG[first] @ ctx.RW4.t()) if ctx.needs_input_grad[4] else None      # RW4 [n, 4n] in kernel gate order, like dG
        dc0g = dc0.to(xs.dtype) if (ctx.needs_input_grad[5] and dc0 is not None) else None
        return dx, dWx, dWr, db, dh0, dc0g, dWp, None


def lstm_layer_fused_ok(x, n, mb):
    """Shapes / dtypes the fused lstmLayer route supports."""
    if not (L.use_hip(x) and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    fn = L.hip().dl4j_lstm_steps_f32_supported if x.dtype == torch.float32 else L.hip().dl4j_lstm_persist_supported
    return _PERSIST and bool(fn(L.c_int(mb), L.c_int(n)))


def lstm_layer_direction(xs, Wx, Wr, bias, h0, c0, Wp, reverse):
    T, B, _ = xs.shape
    n = Wr.shape[0]
    h0 = xs.new_zeros(B, n) if h0 is None else h0
    c0 = xs.new_zeros(B, n) if c0 is None else c0
    return _LstmLayerFn.apply(xs.contiguous(), Wx, Wr, bias, h0, c0, Wp, reverse)
