"""Second half of the op-validation case table: independent references for the ops that round 2 only *executed* (updaters beyond
Adam, every ``*_loss_grad``, recurrent cells and sequences, image / colour-space ops, space<->batch/depth permutes, im2col family,
string ops, test ops ...). Same contract as op_cases.py: a reference is NumPy / a closed form / explicit Python loops / an ATen
module with copied weights (torch.nn.LSTM, GRU, RNN) - never the op's own implementation module.

Reference: platform-tests/.../opvalidation/{LayerOpValidation,LossOpValidation,RnnOpValidation,MiscOpValidation,ShapeOpValidation,
TransformOpValidation}.java; libnd4j helpers cited next to the cases whose conventions are not obvious.
"""
from __future__ import annotations

import colorsys
import itertools
import math

import numpy as np
import torch
import torch.nn.functional as F

from .op_cases import Case, I, R, d64


def _np(x):
    return x.detach().double().cpu().numpy()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ------------------------------------------------------------------------------------------------ losses
def _loss_grad_ref(per):
    """d/d(pred, weights, labels) of sum(per(pred, labels) * w) by fp64 autograd of the closed form written here."""
    def ref(p, w, l):
        pp, ww, ll = (v.double().clone().requires_grad_(True) for v in (p, w, l))
        y = (per(pp, ll) * ww).sum()
        gs = torch.autograd.grad(y, [pp, ww, ll], allow_unused=True)
        return [g if g is not None else torch.zeros_like(v) for g, v in zip(gs, (pp, ww, ll))]
    return ref


def _pair_mse(p, l):
    d = p - l
    n = d.shape[1]
    diff = d.unsqueeze(2) - d.unsqueeze(1)               # every ordered pair (i, j) of one row
    return (diff * diff).sum((1, 2)) / (n * (n - 1))


_PER = {
    "absolute_difference_loss": lambda p, l: (p - l).abs(),
    "mean_sqerr_loss": lambda p, l: (p - l) ** 2,
    "huber_loss": lambda p, l: torch.where((p - l).abs() <= 0.3, 0.5 * (p - l) ** 2, 0.3 * (p - l).abs() - 0.045),
    "log_loss": lambda p, l: -(l * torch.log(p + 1e-7) + (1 - l) * torch.log(1 - p + 1e-7)),
    "hinge_loss": lambda p, l: torch.clamp_min(1 - (2 * l - 1) * p, 0),
    "log_poisson_loss": lambda p, l: torch.exp(p) - l * p,
    "cosine_distance_loss": lambda p, l: 1 - (p * l).sum(1, keepdim=True),
    "mean_pairwssqerr_loss": _pair_mse,
    "sigm_cross_entropy_loss": lambda z, l: torch.clamp_min(z, 0) - z * l + torch.log1p(torch.exp(-z.abs())),
    "softmax_cross_entropy_loss": lambda z, l: -(l * (z - torch.logsumexp(z, 1, keepdim=True))).sum(1, keepdim=True),
}


def _ctc_bruteforce(logits, labels, blank=0):
    """-log sum over every alignment that collapses to `labels` of prod_t softmax(logits)[t, a_t] (T and C tiny)."""
    lp = logits - torch.logsumexp(logits, -1, keepdim=True)
    T, Cn = lp.shape
    total = None
    for path in itertools.product(range(Cn), repeat=T):
        col, prev = [], None
        for a in path:
            if a != prev and a != blank:
                col.append(a)
            prev = a
        if col == labels:
            s = sum(lp[t, a] for t, a in enumerate(path))
            total = s if total is None else torch.logaddexp(total, s)
    return -total


def _ctc_exact_nbest(logits, blank, nbest):
    lp = torch.log_softmax(logits.double(), -1)
    T, Cn = lp.shape
    acc = {}
    for path in itertools.product(range(Cn), repeat=T):
        col, prev = [], None
        for a in path:
            if a != prev and a != blank:
                col.append(a)
            prev = a
        s = float(sum(lp[t, a] for t, a in enumerate(path)))
        k = tuple(col)
        acc[k] = s if k not in acc else float(np.logaddexp(acc[k], s))
    return sorted(acc.items(), key=lambda kv: -kv[1])[:nbest]


# ------------------------------------------------------------------------------------------------ image helpers
def _per_pixel(x, fn):
    a = _np(x)
    out = np.zeros_like(a)
    for idx in np.ndindex(*a.shape[:-1]):
        out[idx] = fn(*a[idx])
    return _t(out)


_YIQ = np.array([[0.299, 0.587, 0.114], [0.59590059, -0.27455667, -0.32134392], [0.21153661, -0.52273617, 0.31119955]])
_YUV = np.array([[0.299, 0.587, 0.114], [-0.14714119, -0.28886916, 0.43601035], [0.61497538, -0.51496512, -0.10001026]])


def _hue(delta):
    def f(r, g, b):
        h, s, v = colorsys.rgb_to_hsv(r, g, b)
        return colorsys.hsv_to_rgb((h + delta) % 1.0, s, v)
    return f


def _sat(factor):
    def f(r, g, b):
        h, s, v = colorsys.rgb_to_hsv(r, g, b)
        return colorsys.hsv_to_rgb(h, min(max(s * factor, 0.0), 1.0), v)
    return f


def _bilinear_tf1(x, oh, ow):
    """TF1 resize_bilinear without align_corners / half_pixel_centers: src = dst * (in / out)."""
    a = _np(x)
    N, H, W, Cn = a.shape
    out = np.zeros((N, oh, ow, Cn))
    for y in range(oh):
        sy = y * H / oh
        y0 = int(math.floor(sy)); y1 = min(y0 + 1, H - 1); wy = sy - y0
        for xx in range(ow):
            sx = xx * W / ow
            x0 = int(math.floor(sx)); x1 = min(x0 + 1, W - 1); wx = sx - x0
            out[:, y, xx] = (a[:, y0, x0] * (1 - wy) * (1 - wx) + a[:, y0, x1] * (1 - wy) * wx + a[:, y1, x0] * wy * (1 - wx) + a[:, y1, x1] * wy * wx)
    return _t(out)


def _fake_quant_ref(x, mn, mx, bits=8, narrow=False):
    a, lo, hi = _np(x), np.broadcast_to(_np(mn), x.shape[-1:] if mn.dim() else ()), np.broadcast_to(_np(mx), x.shape[-1:] if mx.dim() else ())
    out = np.zeros_like(a)
    qmin, qmax = (1.0 if narrow else 0.0), float(2 ** bits - 1)
    for idx in np.ndindex(*a.shape):
        l, h = (float(lo[idx[-1]]), float(hi[idx[-1]])) if np.ndim(lo) else (float(lo), float(hi))
        scale = (h - l) / (qmax - qmin)
        zp = qmin - l / scale
        nz = qmin if zp < qmin else (qmax if zp > qmax else float(np.round(zp)))
        nmin, nmax = (qmin - nz) * scale, (qmax - nz) * scale
        c = min(max(a[idx], nmin), nmax)
        out[idx] = math.floor((c - nmin) / scale + 0.5) * scale + nmin
    return _t(out)


# ------------------------------------------------------------------------------------------------ layout permutes (NumPy)
def _s2d(x, b):
    a = _np(x)
    N, H, W, Cn = a.shape
    return _t(a.reshape(N, H // b, b, W // b, b, Cn).transpose(0, 1, 3, 2, 4, 5).reshape(N, H // b, W // b, b * b * Cn))


def _d2s(x, b):
    a = _np(x)
    N, H, W, Cn = a.shape
    return _t(a.reshape(N, H, W, b, b, Cn // (b * b)).transpose(0, 1, 3, 2, 4, 5).reshape(N, H * b, W * b, Cn // (b * b)))


def _s2b(x, b):
    a = _np(x)
    N, H, W, Cn = a.shape
    return _t(a.reshape(N, H // b, b, W // b, b, Cn).transpose(2, 4, 0, 1, 3, 5).reshape(b * b * N, H // b, W // b, Cn))


def _b2s(x, b):
    a = _np(x)
    NB, H, W, Cn = a.shape
    N = NB // (b * b)
    return _t(a.reshape(b, b, N, H, W, Cn).transpose(2, 3, 0, 4, 1, 5).reshape(N, H * b, W * b, Cn))


def _im2col_ref(x, k, s):
    a = _np(x)
    N, Cn, H, W = a.shape
    oh, ow = (H - k) // s + 1, (W - k) // s + 1
    out = np.zeros((N, Cn, k, k, oh, ow))
    for i in range(k):
        for j in range(k):
            for y in range(oh):
                for xx in range(ow):
                    out[:, :, i, j, y, xx] = a[:, :, y * s + i, xx * s + j]
    return _t(out)


def _col2im_ref(c, s, H, W):
    a = _np(c)
    N, Cn, k, _, oh, ow = a.shape
    out = np.zeros((N, Cn, H, W))
    for i in range(k):
        for j in range(k):
            for y in range(oh):
                for xx in range(ow):
                    out[:, :, y * s + i, xx * s + j] += a[:, :, i, j, y, xx]
    return _t(out)


def _patches_ref(x, k, s):
    a = _np(x)                                   # NHWC, VALID
    N, H, W, Cn = a.shape
    oh, ow = (H - k) // s + 1, (W - k) // s + 1
    out = np.zeros((N, oh, ow, k * k * Cn))
    for y in range(oh):
        for xx in range(ow):
            out[:, y, xx] = a[:, y * s:y * s + k, xx * s:xx * s + k, :].reshape(N, -1)
    return _t(out)


def _dilation_same_ref(x, w):
    a, f = _np(x), _np(w)                        # stride 1, rate 1, SAME: total pad k-1, the smaller half in front
    N, H, W, Cn = a.shape
    kH, kW, _ = f.shape
    pt, pl = (kH - 1) // 2, (kW - 1) // 2
    out = np.full((N, H, W, Cn), -np.inf)
    for y in range(H):
        for xx in range(W):
            for i in range(kH):
                for j in range(kW):
                    yy, xj = y - pt + i, xx - pl + j
                    if 0 <= yy < H and 0 <= xj < W:
                        out[:, y, xx] = np.maximum(out[:, y, xx], a[:, yy, xj] + f[i, j])
    return _t(out)


def _lrn_ref(x, depth, bias, alpha, beta):
    a = _np(x)
    Cn = a.shape[1]
    out = np.zeros_like(a)
    for c in range(Cn):
        lo, hi = max(0, c - depth), min(Cn - 1, c + depth)
        out[:, c] = a[:, c] / (bias + alpha * (a[:, lo:hi + 1] ** 2).sum(1)) ** beta
    return _t(out)


# ------------------------------------------------------------------------------------------------ recurrent references
def _torch_lstm(x, Wx, Wr, b, h0=None, c0=None, reverse=False):
    """torch.nn.LSTM with the op's weights (gate order i, f, g, o on both sides; libnd4j helpers/impl/lstmLayer.cpp:200-249)."""
    nIn, n = Wx.shape[0], Wr.shape[0]
    m = torch.nn.LSTM(nIn, n).double()
    with torch.no_grad():
        m.weight_ih_l0.copy_(d64(Wx).t()); m.weight_hh_l0.copy_(d64(Wr).t())
        m.bias_ih_l0.copy_(d64(b)); m.bias_hh_l0.zero_()
    xs = d64(x).flip(0) if reverse else d64(x)
    B = x.shape[1]
    st = (d64(h0).unsqueeze(0) if h0 is not None else torch.zeros(1, B, n, dtype=torch.float64),
          d64(c0).unsqueeze(0) if c0 is not None else torch.zeros(1, B, n, dtype=torch.float64))
    y, (h, c) = m(xs, st)
    return (y.flip(0) if reverse else y).detach(), h[0].detach(), c[0].detach()


def _torch_gru(x, h0, Wx, Wh, b):
    nIn, n = Wx.shape[0], Wh.shape[0]
    m = torch.nn.GRU(nIn, n).double()
    with torch.no_grad():
        m.weight_ih_l0.copy_(d64(Wx).t()); m.weight_hh_l0.copy_(d64(Wh).t())
        m.bias_ih_l0.copy_(d64(b)); m.bias_hh_l0.zero_()
    return m(d64(x), d64(h0).unsqueeze(0))[0].detach()


def _torch_rnn(x, Wx, Wh, b, reverse=False):
    nIn, n = Wx.shape[0], Wh.shape[0]
    m = torch.nn.RNN(nIn, n, nonlinearity="tanh").double()
    with torch.no_grad():
        m.weight_ih_l0.copy_(d64(Wx).t()); m.weight_hh_l0.copy_(d64(Wh).t())
        m.bias_ih_l0.copy_(d64(b)[:n]); m.bias_hh_l0.copy_(d64(b)[n:])
    xs = d64(x).flip(0) if reverse else d64(x)
    y, h = m(xs)
    return (y.flip(0) if reverse else y).detach(), h[0].detach()


def _sig(v):
    return 1.0 / (1.0 + np.exp(-v))


def _gru_cell_ref(x, h, Wru, Wc, bru, bc, lib=np):
    """gruCell.cpp: r, u from [x, h] Wru; c from [x, r*h] Wc; h' = u*h + (1-u)*c. `lib` = np for values, torch for autograd."""
    cat = (lambda a, b: np.concatenate([a, b], -1)) if lib is np else (lambda a, b: torch.cat([a, b], -1))
    sg = _sig if lib is np else torch.sigmoid
    n = h.shape[-1]
    ru = sg(cat(x, h) @ Wru + bru)
    r, u = ru[..., :n], ru[..., n:]
    c = lib.tanh(cat(x, r * h) @ Wc + bc)
    return r, u, c, u * h + (1 - u) * c


def _sru_ref(x, w, b, c0):
    xa, wa, ba, c = _np(x), _np(w), _np(b), _np(c0)      # x [B, K, T], w [3K, K]
    B, K, T = xa.shape
    hs, cs = np.zeros((B, K, T)), np.zeros((B, K, T))
    for t in range(T):
        z = xa[:, :, t] @ wa.T
        f, r = _sig(z[:, K:2 * K] + ba[:K]), _sig(z[:, 2 * K:] + ba[K:])
        c = f * c + (1 - f) * z[:, :K]
        hs[:, :, t] = r * np.tanh(c) + (1 - r) * xa[:, :, t]
        cs[:, :, t] = c
    return [_t(hs), _t(cs)]


def _sru_bi_ref(x, w, b, c0, lib_torch=False):
    """x [T, B, 2K], w [2K, 6K] (per direction: candidate, forget, reset), b [4K], c0 [B, 2K]; direction 1 runs backwards in time."""
    X, W, Bv, C0 = (d64(v) for v in (x, w, b, c0))
    T, B, K2 = X.shape
    K = K2 // 2
    z = (X @ W).reshape(T, B, 2, 3 * K)
    hs = [[None, None] for _ in range(T)]
    cs = [[None, None] for _ in range(T)]
    for dr in range(2):
        c = C0[:, dr * K:(dr + 1) * K]
        bf, br = Bv[dr * 2 * K:dr * 2 * K + K], Bv[dr * 2 * K + K:(dr + 1) * 2 * K]
        for t in (range(T) if dr == 0 else range(T - 1, -1, -1)):
            zz = z[t, :, dr]
            f, r = torch.sigmoid(zz[:, K:2 * K] + bf), torch.sigmoid(zz[:, 2 * K:] + br)
            c = f * c + (1 - f) * zz[:, :K]
            hs[t][dr] = r * torch.tanh(c) + (1 - r) * X[t, :, dr * K:(dr + 1) * K]
            cs[t][dr] = c
    h = torch.stack([torch.cat(p, -1) for p in hs])
    cc = torch.stack([torch.cat(p, -1) for p in cs])
    return h, cc


def _autograd(fn, ins, gy, pick=0):
    leaves = [d64(v).clone().requires_grad_(True) for v in ins]
    y = fn(*leaves)
    y = y[pick] if isinstance(y, (tuple, list)) else y
    gs = torch.autograd.grad(y, leaves, d64(gy), allow_unused=True)
    return [g if g is not None else torch.zeros_like(v) for g, v in zip(gs, leaves)]


# ------------------------------------------------------------------------------------------------ the table
def extend(C, add):
    x34, y34 = R(3, 4, seed=1), R(3, 4, seed=2)
    pos = R(3, 4, seed=3, lo=0.5, hi=2.5)

    # ---- losses: gradients of every weighted-sum loss against autograd of the closed form
    for nm, per in _PER.items():
        c0 = C[nm][0]
        add(nm + "_grad", Case(c0.ins, i=c0.i, t=c0.t, ref=_loss_grad_ref(per), rtol=1e-4, atol=1e-6, tag="autograd-of-closed-form"))
    lab, pred, one = R(3, 4, seed=111, lo=0, hi=1), R(3, 4, seed=112, lo=0.05, hi=0.95), torch.ones(1)
    add("mean_pairwssqerr_loss", Case([pred, one, lab], i=[1], ref=lambda p, w, l: _pair_mse(d64(p), d64(l)).sum(), rtol=1e-5, tag="pair-mean"))
    oh = torch.eye(4)[[0, 2, 3]]
    add("softmax_cross_entropy_loss_with_logits_grad", Case([x34, oh], ref=lambda z, l: [
        torch.softmax(d64(z), 1) * d64(l).sum(1, keepdim=True) - d64(l), -(d64(z) - torch.logsumexp(d64(z), 1, keepdim=True))], rtol=1e-5))
    add("sparse_softmax_cross_entropy_loss_with_logits_grad", Case([I(0, 2, 3), x34], ref=lambda l, z: torch.softmax(d64(z), 1) - torch.eye(4, dtype=torch.float64)[l]))
    add("weighted_cross_entropy_with_logits", Case([lab, x34, torch.ones(4) * 2], ref=lambda t_, z, w: (
        -(d64(w) * d64(t_) * torch.log(torch.sigmoid(d64(z))) + (1 - d64(t_)) * torch.log(1 - torch.sigmoid(d64(z))))), rtol=1e-5, atol=1e-6))
    lg = R(1, 5, 4, seed=113)
    add("ctc_loss_grad", Case([torch.tensor([[1, 2]]), lg, I(2), I(5)], i=[0],
                              ref=lambda t_, z, tl, il: _autograd(lambda zz: _ctc_bruteforce(zz[0], [1, 2]), [z], torch.ones(())), rtol=1e-4, atol=1e-6,
                              tag="brute-force alignments"))
    peaked = torch.tensor([[[0.1, 3.0, 0.2], [2.5, 0.1, 0.3], [0.2, 0.1, 2.8]]])

    def _beam_ref(z, sl):
        best = _ctc_exact_nbest(z[0], 0, 2)
        paths = torch.zeros(1, 2, 3, dtype=torch.int32)
        for k, (seq, _) in enumerate(best):
            paths[0, k, :len(seq)] = torch.tensor(seq, dtype=torch.int32)
        return [paths, torch.tensor([[s for _, s in best]]), torch.tensor([[len(seq) for seq, _ in best]], dtype=torch.int32)]
    add("ctc_beam", Case([peaked, I(3)], i=[0, 27, 2], b=[True], ref=_beam_ref, rtol=1e-4, atol=1e-5, tag="exact enumeration"))

    # ---- updaters with non-trivial state: closed forms of the reference's updater classes (nd4j learning/*Updater.java)
    g_, sa, sb, sc = R(6, seed=201), R(6, seed=202, lo=0.01, hi=0.2), R(6, seed=203, lo=0.01, hi=0.2), R(6, seed=204, lo=0.01, hi=0.2)

    def _adadelta(g, msg, msdx, rho=0.95, eps=1e-6):
        g, msg, msdx = d64(g), d64(msg), d64(msdx)
        m2 = rho * msg + (1 - rho) * g * g
        u = g * torch.sqrt(msdx + eps) / torch.sqrt(m2 + eps)
        return [u, m2, rho * msdx + (1 - rho) * u * u]
    add("ada_delta_updater", Case([g_, sa, sb], t=[0.95, 1e-6], ref=_adadelta, rtol=1e-5))

    def _adamax(g, u, m, lr=0.002, b1=0.9, b2=0.999, it=2):
        g, u, m = d64(g), d64(u), d64(m)
        m2 = b1 * m + (1 - b1) * g
        u2 = torch.maximum(b2 * u, g.abs()) + 1e-32
        return [lr / (1 - b1 ** (it + 1)) * m2 / u2, u2, m2]
    add("ada_max_updater", Case([g_, sa, sb], t=[0.002, 0.9, 0.999, 1e-8], i=[2], ref=_adamax, rtol=1e-5))

    def _nadam(g, v, m, lr=0.002, b1=0.9, b2=0.999, eps=1e-8, it=2):
        g, v, m = d64(g), d64(v), d64(m)
        m2, v2 = b1 * m + (1 - b1) * g, b2 * v + (1 - b2) * g * g
        return [lr * (b1 * m2 + (1 - b1) * g) / (1 - b1 ** (it + 1)) / (torch.sqrt(v2) + eps), v2, m2]
    add("nadam_updater", Case([g_, sa, sb], t=[0.002, 0.9, 0.999, 1e-8], i=[2], ref=_nadam, rtol=1e-5))

    def _amsgrad(g, v, m, h, lr=0.001, b1=0.9, b2=0.999, eps=1e-8, it=2):
        g, v, m, h = d64(g), d64(v), d64(m), d64(h)
        m2, v2 = b1 * m + (1 - b1) * g, b2 * v + (1 - b2) * g * g
        h2 = torch.maximum(h, v2)
        a = lr * math.sqrt(1 - b2 ** (it + 1)) / (1 - b1 ** (it + 1))
        return [a * m2 / (torch.sqrt(h2) + eps), v2, m2, h2]
    add("ams_grad_updater", Case([g_, sa, sb, sc], t=[0.001, 0.9, 0.999, 1e-8], i=[2], ref=_amsgrad, rtol=1e-5))

    def _adabelief(g, s, m, lr=0.001, b1=0.9, b2=0.999, eps=1e-8, it=2):
        g, s, m = d64(g), d64(s), d64(m)
        m2 = b1 * m + (1 - b1) * g
        s2 = b2 * s + (1 - b2) * (g - m2) ** 2 + eps
        a = lr * math.sqrt(1 - b2 ** (it + 1)) / (1 - b1 ** (it + 1))
        return [a * m2 / (torch.sqrt(s2) + eps), s2, m2]
    add("adabelief_updater", Case([g_, sa, sb], t=[0.001, 0.9, 0.999, 1e-8], i=[2], ref=_adabelief, rtol=1e-5))

    # ---- colour spaces (colorsys per pixel; the TF matrices for YIQ / YUV)
    rgb = R(2, 4, 4, 3, seed=115, lo=0, hi=1)
    add("rgb_to_hsv", Case([rgb], ref=lambda x: _per_pixel(x, colorsys.rgb_to_hsv), rtol=1e-4, atol=1e-5))
    hsv = R(2, 4, 4, 3, seed=116, lo=0.05, hi=0.95)
    add("hsv_to_rgb", Case([hsv], ref=lambda x: _per_pixel(x, colorsys.hsv_to_rgb), rtol=1e-4, atol=1e-5))
    add("adjust_hue", Case([rgb], t=[0.1], ref=lambda x: _per_pixel(x, _hue(0.1)), rtol=1e-4, atol=1e-5))
    add("adjust_saturation", Case([rgb], t=[0.5], ref=lambda x: _per_pixel(x, _sat(0.5)), rtol=1e-4, atol=1e-5))
    add("rgb_to_yiq", Case([rgb], ref=lambda x: _t(_np(x) @ _YIQ.T), rtol=1e-4, atol=1e-5))
    add("rgb_to_yuv", Case([rgb], ref=lambda x: _t(_np(x) @ _YUV.T), rtol=1e-4, atol=1e-5))
    add("yiq_to_rgb", Case([rgb], ref=lambda x: _t(_np(x) @ np.linalg.inv(_YIQ).T), rtol=1e-3, atol=1e-4))
    add("yuv_to_rgb", Case([rgb], ref=lambda x: _t(_np(x) @ np.linalg.inv(_YUV).T), rtol=1e-3, atol=1e-4))
    add("image_resize", Case([rgb, I(8, 8, dtype=torch.int32)], i=[1], b=[False, False],
                             ref=lambda x, s: _t(_np(x).repeat(2, 1).repeat(2, 2)), tag="nearest x2"))
    add("resize_images", Case([rgb, I(8, 8, dtype=torch.int32)], i=[0], b=[False, False], ref=lambda x, s: _bilinear_tf1(x, 8, 8), rtol=1e-4, atol=1e-5))
    box_img = torch.zeros(1, 4, 4, 3)
    want = torch.zeros(1, 4, 4, 3, dtype=torch.float64)
    for rr in range(3):                       # box (0.1, 0.1, 0.8, 0.8) on a 4x4 grid: rows / columns 0..2, edges only
        for cc in range(3):
            if rr in (0, 2) or cc in (0, 2):
                want[0, rr, cc] = torch.tensor([1.0, 0.0, 0.0], dtype=torch.float64)
    add("draw_bounding_boxes", Case([box_img, torch.tensor([[[0.1, 0.1, 0.8, 0.8]]]), torch.tensor([[1.0, 0.0, 0.0]])], ref=lambda im, bx, col: want, tag="one box"))
    # ranges whose zero point is not a rounding tie (with [-1, 1] it is 127.5 exactly and fp32 / fp64 arithmetic nudge differently)
    add("fake_quant_with_min_max_vars", Case([x34, torch.tensor(-1.0), torch.tensor(1.5)], i=[8], b=[False], ref=lambda x, a, b: _fake_quant_ref(x, a, b), atol=1e-6))
    add("fake_quant_with_min_max_vars_per_channel", Case([x34, -torch.ones(4) * torch.tensor([1.0, 0.5, 2.0, 1.5]), torch.tensor([1.5, 1.5, 0.5, 2.0])], i=[8], b=[False],
                                                         ref=lambda x, a, b: _fake_quant_ref(x, a, b), atol=1e-6))

    # ---- layout permutes
    add("space_to_depth", Case([R(1, 4, 4, 2, seed=21)], i=[2, 1], ref=lambda x: _s2d(x, 2)))
    add("depth_to_space", Case([R(1, 2, 2, 8, seed=22)], i=[2, 1], ref=lambda x: _d2s(x, 2)))
    s2b_in, z22 = torch.arange(32.0).reshape(1, 4, 4, 2), torch.zeros(2, 2, dtype=torch.int64)
    add("space_to_batch", Case([s2b_in, z22], i=[2], ref=lambda x, p: _s2b(x, 2)))
    add("space_to_batch_nd", Case([s2b_in, I(2, 2), z22], ref=lambda x, b, p: _s2b(x, 2)))
    b2s_in = torch.arange(32.0).reshape(4, 2, 2, 2)
    add("batch_to_space", Case([b2s_in, z22], i=[2], ref=lambda x, c: _b2s(x, 2)))
    add("batch_to_space_nd", Case([b2s_in, I(2, 2), z22], ref=lambda x, b, c: _b2s(x, 2)))
    img = R(2, 3, 8, 8, seed=43)
    add("im2col", Case([img], i=[2, 2, 2, 2, 0, 0, 1, 1, 0], t=[0.0], ref=lambda x: _im2col_ref(x, 2, 2)))
    add("col2im", Case([R(2, 3, 2, 2, 4, 4, seed=58)], i=[2, 2, 0, 0, 8, 8, 1, 1, 0], ref=lambda c: _col2im_ref(c, 2, 8, 8)))
    add("col2im", Case([R(1, 2, 3, 3, 3, 3, seed=59)], i=[1, 1, 0, 0, 5, 5, 1, 1, 0], ref=lambda c: _col2im_ref(c, 1, 5, 5), tag="overlapping"))
    add("extract_image_patches", Case([R(1, 6, 6, 2, seed=59)], i=[2, 2, 2, 2, 1, 1, 0], ref=lambda x: _patches_ref(x, 2, 2)))
    add("dilation2d", Case([R(1, 6, 6, 2, seed=60), R(2, 2, 2, seed=61)], i=[1, 1, 1, 1, 1, 1, 1, 1, 1], ref=_dilation_same_ref))
    add("lrn", Case([img], t=[1.0, 1e-4, 0.75], i=[2], ref=lambda x: _lrn_ref(x, 2, 1.0, 1e-4, 0.75), rtol=1e-5))
    add("max_pool_with_argmax", Case([img], i=[2, 2, 2, 2, 0, 0, 1, 1, 0, 0, 0], ref=lambda x: [
        d64(x).reshape(2, 3, 4, 2, 4, 2).amax((3, 5)),
        # flat index inside each [H, W] plane of the maximum of every 2x2 window
        torch.stack([torch.stack([torch.stack([torch.stack([
            torch.tensor(max(((int(y * 2 + a_) * 8 + int(xx * 2 + b_)) for a_ in range(2) for b_ in range(2)),
                             key=lambda f: float(x[n, c].reshape(-1)[f]))) for xx in range(4)]) for y in range(4)]) for c in range(3)]) for n in range(2)])]))
    add("fused_batch_norm", Case([R(2, 4, 4, 3, seed=66), R(3, seed=64), R(3, seed=65), torch.zeros(3), torch.ones(3)], i=[0, 1], t=[1e-3], ref=lambda x, g, b, m, v: [
        (d64(x) - d64(x).mean((0, 1, 2))) / torch.sqrt(d64(x).var((0, 1, 2), unbiased=False) + 1e-3) * d64(g) + d64(b), d64(x).mean((0, 1, 2)),
        d64(x).var((0, 1, 2), unbiased=False)], rtol=1e-4, atol=1e-5))

    # ---- attention
    q, v, k = R(2, 3, 4, seed=72), R(2, 5, 4, seed=73), R(2, 5, 4, seed=74)

    def _att(qq, vv, kk):
        logits = qq @ kk.transpose(1, 2)
        w = torch.exp(logits - logits.amax(-1, keepdim=True))
        w = w / w.sum(-1, keepdim=True)
        return w @ vv, w, logits
    add("dot_product_attention_v2", Case([q, v, k], ref=lambda a, b, c: list(_att(d64(a), d64(b), d64(c))), rtol=1e-5))
    gy = R(2, 3, 4, seed=301)
    add("dot_product_attention_v2_bp", Case([q, v, k, gy], ref=lambda a, b, c, g: _autograd(lambda *z: _att(*z)[0], [a, b, c], g), rtol=1e-4, atol=1e-6))

    def _mh(qq, kk, vv, Wq, Wk, Wv, Wo):
        outs = []
        for bi in range(qq.shape[0]):
            heads = []
            for h in range(Wq.shape[0]):
                pq, pk, pv = Wq[h] @ qq[bi], Wk[h] @ kk[bi], Wv[h] @ vv[bi]          # [p, T]
                w = torch.softmax(pk.t() @ pq / math.sqrt(pk.shape[0]), 0)          # [Tk, Tq], normalised over keys
                heads.append(pv @ w)
            outs.append(Wo.t() @ torch.cat(heads, 0))
        return torch.stack(outs)
    mh_ins = [R(2, 4, 3, seed=75), R(2, 4, 5, seed=76), R(2, 4, 5, seed=77), R(2, 3, 4, seed=78), R(2, 3, 4, seed=79), R(2, 3, 4, seed=80), R(6, 4, seed=81)]
    add("multi_head_dot_product_attention", Case(mh_ins, i=[1, 0], ref=lambda *z: _mh(*[d64(t_) for t_ in z]), rtol=1e-4, atol=1e-6))

    # ---- recurrent
    n, nIn, B, T = 5, 4, 2, 3
    Wx, Wr, bb = R(nIn, 4 * n, seed=85) * 0.3, R(n, 4 * n, seed=86) * 0.3, R(4 * n, seed=87) * 0.1
    xs, h0, c0 = R(T, B, nIn, seed=88), R(B, n, seed=83), R(B, n, seed=84)
    add("lstmLayer", Case([xs, Wx, Wr, bb], i=[0, 0, 2, 0, 0], t=[0.0], b=[True, False, False, False, False, True, True, True],
                          ref=lambda x, a, r_, b_: list(_torch_lstm(x, a, r_, b_)), rtol=1e-5, tag="torch.nn.LSTM"))
    add("lstmLayer", Case([xs, Wx, Wr, bb, h0, c0], i=[0, 1, 2, 0, 0], t=[0.0], b=[True, False, True, True, False, True, True, True],
                          ref=lambda x, a, r_, b_, h, c: list(_torch_lstm(x, a, r_, b_, h, c, reverse=True)), rtol=1e-5, tag="backward direction, initial state"))

    def _lstm_bp_ref(x, a, r_, b_, gy_):
        nIn_, n_ = a.shape[0], r_.shape[0]
        m = torch.nn.LSTM(nIn_, n_).double()
        with torch.no_grad():
            m.weight_ih_l0.copy_(d64(a).t()); m.weight_hh_l0.copy_(d64(r_).t()); m.bias_ih_l0.copy_(d64(b_)); m.bias_hh_l0.zero_()
        xx = d64(x).clone().requires_grad_(True)
        y = m(xx)[0]
        gx, gwi, gwh, gbi = torch.autograd.grad(y, [xx, m.weight_ih_l0, m.weight_hh_l0, m.bias_ih_l0], d64(gy_))
        return [gx, gwi.t(), gwh.t(), gbi]
    add("lstmLayer_bp", Case([xs, Wx, Wr, bb, R(T, B, n, seed=302)], i=[0, 0, 2, 0, 0], t=[0.0], b=[True, False, False, False, False, True, False, False],
                             ref=_lstm_bp_ref, rtol=1e-4, atol=1e-6, tag="torch.nn.LSTM autograd"))

    def _cell_ref(x, a, r_, b_, h, c):
        m = torch.nn.LSTMCell(a.shape[0], r_.shape[0]).double()
        with torch.no_grad():
            m.weight_ih.copy_(d64(a).t()); m.weight_hh.copy_(d64(r_).t()); m.bias_ih.copy_(d64(b_)); m.bias_hh.zero_()
        return [t_.detach() for t_ in m(d64(x), (d64(h), d64(c)))]
    xc = R(B, nIn, seed=82)
    add("lstmLayerCell", Case([xc, Wx, Wr, bb, h0, c0], i=[2, 0, 0], t=[0.0], b=[True, False], ref=_cell_ref, rtol=1e-5, tag="torch.nn.LSTMCell"))

    def _cell_bp_ref(x, a, r_, b_, h, c, g):
        def f(x_, a_, r2, b2, h_, c_):
            z = x_ @ a_ + h_ @ r2 + b2
            i_, f_, g_2, o_ = torch.sigmoid(z[:, :n]), torch.sigmoid(z[:, n:2 * n]), torch.tanh(z[:, 2 * n:3 * n]), torch.sigmoid(z[:, 3 * n:])
            return o_ * torch.tanh(f_ * c_ + i_ * g_2)
        return _autograd(f, [x, a, r_, b_, h, c], g)
    add("lstmLayerCellBp", Case([xc, Wx, Wr, bb, h0, c0, R(B, n, seed=303)], i=[2, 0, 0], t=[0.0], b=[True, False], ref=_cell_bp_ref, rtol=1e-4, atol=1e-6))

    def _classic_cell(x, h, c, a, r_, wc, wp, b_):
        # helpers/cpu/lstm.cpp:81-100: z split as (input, forget, cell, output); no peepholes / projection in this case
        z = _np(x) @ _np(a) + _np(h) @ _np(r_) + _np(b_)
        i_, f_, g_2, o_ = _sig(z[:, :n]), _sig(z[:, n:2 * n]), np.tanh(z[:, 2 * n:3 * n]), _sig(z[:, 3 * n:])
        cn = f_ * _np(c) + i_ * g_2
        return [_t(o_ * np.tanh(cn)), _t(cn)]
    add("lstmCell", Case([xc, h0, c0, Wx, Wr, torch.zeros(3 * n), torch.eye(n), bb], i=[0, 0], t=[0.0, 0.0, 0.0], ref=_classic_cell, rtol=1e-5, tag="gate order i,f,c,o"))

    Wblk, bblk = R(nIn + n, 4 * n, seed=304) * 0.3, R(4 * n, seed=305) * 0.1
    wci, wcf, wco = R(n, seed=306) * 0.2, R(n, seed=307) * 0.2, R(n, seed=308) * 0.2

    def _block_cell(x, cl, yl, W, a1, a2, a3, b_, peep=True, fb=0.5):
        # helpers/cpu/lstm.cpp:185-235: weights ordered [input gate, block input, forget gate, output gate]
        m = np.concatenate([_np(x), _np(yl)], 1) @ _np(W) + _np(b_)
        zi, zz, zf, zo = m[:, :n], m[:, n:2 * n], m[:, 2 * n:3 * n], m[:, 3 * n:]
        if peep:
            zi, zf = zi + _np(cl) * _np(a1), zf + _np(cl) * _np(a2)
        i_, z_, f_ = _sig(zi), np.tanh(zz), _sig(zf + fb)
        c = z_ * i_ + f_ * _np(cl)
        o_ = _sig(zo + (c * _np(a3) if peep else 0.0))
        return [_t(v_) for v_ in (i_, c, f_, o_, z_, np.tanh(c), o_ * np.tanh(c))]
    add("lstmBlockCell", Case([xc, c0, h0, Wblk, wci, wcf, wco, bblk], i=[1], t=[0.5, 0.0], ref=_block_cell, rtol=1e-5, tag="peepholes + forget bias"))

    def _block_seq(x, cl, yl, W, a1, a2, a3, b_):
        c, y = cl, yl
        steps = []
        for t_ in range(x.shape[0]):
            r_ = _block_cell(x[t_], c, y, W, a1, a2, a3, b_)
            c, y = r_[1], r_[6]
            steps.append(r_)
        return [torch.stack([s_[k_] for s_ in steps]) for k_ in range(7)]
    add("lstmBlock", Case([torch.tensor(T), xs, c0, h0, Wblk, wci, wcf, wco, bblk], i=[1, 0], t=[0.5, 0.0], ref=lambda mt, *z: _block_seq(*z), rtol=1e-4, atol=1e-5,
                          tag="time-major sequence of block cells"))

    def _classic_seq(x, h, c, a, r_, wc, wp, b_):
        hs, cs = [], []
        for t_ in range(x.shape[0]):
            h, c = _classic_cell(x[t_], h, c, a, r_, wc, wp, b_)
            hs.append(h)
            cs.append(c)
        return [torch.stack(hs), torch.stack(cs)]
    add("lstm", Case([xs, h0, c0, Wx, Wr, torch.zeros(3 * n), torch.eye(n), bb], i=[0, 0], t=[0.0, 0.0, 0.0], ref=_classic_seq, rtol=1e-5, tag="sequence of classic cells"))

    Wru, Wc_, bru, bc_ = R(nIn + n, 2 * n, seed=91) * 0.3, R(nIn + n, n, seed=92) * 0.3, R(2 * n, seed=93) * 0.1, R(n, seed=94) * 0.1
    xg, hg = R(B, nIn, seed=89), R(B, n, seed=90)
    add("gruCell", Case([xg, hg, Wru, Wc_, bru, bc_], ref=lambda *z: [_t(v_) for v_ in _gru_cell_ref(*[_np(t_) for t_ in z])], rtol=1e-5))
    zB = torch.zeros(B, n)
    add("gruCell_bp", Case([xg, hg, Wru, Wc_, bru, bc_, zB, zB, zB, R(B, n, seed=309)], ref=lambda *z: _autograd(lambda *w: _gru_cell_ref(*w, lib=torch)[3], list(z[:6]), z[9]),
                           rtol=1e-4, atol=1e-6))
    Wgx, Wgh, bg = R(nIn, 3 * n, seed=96) * 0.3, R(n, 3 * n, seed=97) * 0.3, R(3 * n, seed=98) * 0.1
    add("gru", Case([R(T, B, nIn, seed=95), hg, Wgx, Wgh, bg], ref=_torch_gru, rtol=1e-5, tag="torch.nn.GRU"))
    sx, sw, sb_, sc0 = R(B, n, 3, seed=103), R(3 * n, n, seed=104) * 0.3, R(2 * n, seed=105) * 0.1, R(B, n, seed=106)
    add("sru", Case([sx, sw, sb_, sc0, torch.ones(B, n)], ref=lambda x, w, b_, c, m: _sru_ref(x, w, b_, c), rtol=1e-5))
    add("sruCell", Case([R(B, n, seed=99), R(B, n, seed=100), R(n, 3 * n, seed=101) * 0.3, R(2 * n, seed=102) * 0.1],
                        ref=lambda x, c, w, b_: [t_[:, :, 0] for t_ in _sru_ref(x.unsqueeze(-1), w.t(), b_, c)], rtol=1e-5))

    def _sru_bp_ref(x, w, b_, c, gh):
        def f(x_, w_, b2, c_):
            K = x_.shape[1]
            cc, hs = c_, []
            for t_ in range(x_.shape[2]):
                z = x_[:, :, t_] @ w_.t()
                f_, r_ = torch.sigmoid(z[:, K:2 * K] + b2[:K]), torch.sigmoid(z[:, 2 * K:] + b2[K:])
                cc = f_ * cc + (1 - f_) * z[:, :K]
                hs.append(r_ * torch.tanh(cc) + (1 - r_) * x_[:, :, t_])
            return torch.stack(hs, -1)
        return _autograd(f, [x, w, b_, c], gh)
    add("sru_bp", Case([sx, sw, sb_, sc0, _sru_ref(sx, sw, sb_, sc0)[1].float(), torch.zeros(B, n), R(B, n, 3, seed=310), torch.ones(B, n)],
                       ref=lambda x, w, b_, c, cs_, gc, gh, m: _sru_bp_ref(x, w, b_, c, gh), rtol=1e-4, atol=1e-6))
    K = 3
    bx, bw, bbv, bc0 = R(4, B, 2 * K, seed=311), R(2 * K, 6 * K, seed=312) * 0.3, R(4 * K, seed=313) * 0.1, R(B, 2 * K, seed=314)
    add("sru_bi", Case([bx, bw, bbv, bc0, torch.ones(B, 2 * K)], ref=lambda x, w, b_, c, m: [t_.detach() for t_ in _sru_bi_ref(x, w, b_, c)], rtol=1e-5))
    add("sru_bi_bp", Case([bx, bw, bbv, bc0, _sru_bi_ref(bx, bw, bbv, bc0)[1].float(), torch.zeros(B, 2 * K), R(4, B, 2 * K, seed=315), torch.ones(B, 2 * K)],
                          ref=lambda x, w, b_, c, ct, gc, g, m: _autograd(lambda *z: _sru_bi_ref(*z)[0], [x, w, b_, c], g), rtol=1e-4, atol=1e-6))
    rx, rWx, rWh, rb = R(T, B, nIn, seed=107), R(nIn, n, seed=108) * 0.3, R(n, n, seed=109) * 0.3, R(2 * n, seed=110) * 0.1
    add("static_rnn", Case([rx, rWx, rWh, rb], ref=lambda x, a, h, b_: list(_torch_rnn(x, a, h, b_)), rtol=1e-5, tag="torch.nn.RNN"))
    add("dynamic_rnn", Case([rx, rWx, rWh, rb], i=[1], ref=lambda x, a, h, b_: list(_torch_rnn(x, a, h, b_)), rtol=1e-5, tag="time major"))
    add("dynamic_rnn", Case([rx.permute(1, 0, 2).contiguous(), rWx, rWh, rb], i=[0],
                            ref=lambda x, a, h, b_: (lambda y, hl: [y.permute(1, 0, 2), hl])(*_torch_rnn(x.permute(1, 0, 2), a, h, b_)), rtol=1e-5, tag="batch major"))
    rWx2, rWh2, rb2 = R(nIn, n, seed=316) * 0.3, R(n, n, seed=317) * 0.3, R(2 * n, seed=318) * 0.1

    def _birnn(x, a, h, b_, a2, h2, b2):
        yf, hf = _torch_rnn(x, a, h, b_)
        yb, hb = _torch_rnn(x, a2, h2, b2, reverse=True)
        return [torch.cat([yf, yb], -1), hf, hb]
    add("static_bidirectional_rnn", Case([rx, rWx, rWh, rb, rWx2, rWh2, rb2], ref=_birnn, rtol=1e-5))
    add("dynamic_bidirectional_rnn", Case([rx, rWx, rWh, rb, rWx2, rWh2, rb2], i=[1], ref=_birnn, rtol=1e-5))

    # ---- linear algebra
    m4 = R(4, 4, seed=38)
    spd = m4 @ m4.t() + 4 * torch.eye(4)
    add("log_matrix_determinant", Case([spd], ref=lambda a: [torch.tensor(v_) for v_ in np.linalg.slogdet(_np(a))], rtol=1e-4))

    def _lu_ref(a):
        import scipy.linalg
        lu_, piv = scipy.linalg.lu_factor(_np(a))
        perm = list(range(a.shape[0]))
        for k_, p_ in enumerate(piv):
            perm[k_], perm[p_] = perm[p_], perm[k_]
        return [_t(lu_), torch.tensor(perm, dtype=torch.int32)]
    add("lu", Case([m4], ref=_lu_ref, rtol=1e-4, atol=1e-5, tag="scipy lu_factor"))

    def _eig_canon(outs):
        w, v_ = outs[0].double().cpu(), outs[1].double().cpu()
        order = sorted(range(w.shape[0]), key=lambda k_: (round(float(w[k_, 0]), 6), round(float(w[k_, 1]), 6)))
        w, v_ = w[order], v_[:, order]
        vc = torch.complex(v_[..., 0], v_[..., 1])
        for k_ in range(vc.shape[1]):                 # unit norm, largest component real positive
            col = vc[:, k_]
            piv = col[col.abs().argmax()]
            vc[:, k_] = col / col.norm() * (piv.conj() / piv.abs())
        return [w, torch.stack([vc.real, vc.imag], -1)]

    def _eig_ref(a):
        w, v_ = np.linalg.eig(_np(a))
        return [torch.stack([_t(w.real.copy()), _t(np.imag(w).astype(np.float64))], -1), torch.stack([_t(v_.real.copy()), _t(np.imag(v_).astype(np.float64))], -1)]
    add("eig", Case([R(4, 4, seed=320)], ref=_eig_ref, canon=_eig_canon, rtol=1e-4, atol=1e-5, tag="numpy eig, canonical order / phase"))
    tA, tB = R(2, 3, 4, seed=29), R(4, 3, 5, seed=30)
    add("tensormmul_bp", Case([tA, tB, torch.tensordot(tA, tB, dims=([1, 2], [1, 0])), R(2, 5, seed=321)], i=[2, 1, 2, 2, 1, 0],
                              ref=lambda a, b_, c_, g: _autograd(lambda x_, y_: torch.tensordot(x_, y_, dims=([1, 2], [1, 0])), [a, b_], g), rtol=1e-5))
    A0, A1, B0, B1, G0, G1 = R(3, 4, seed=31), R(3, 4, seed=32), R(4, 2, seed=33), R(4, 2, seed=34), R(3, 2, seed=322), R(3, 2, seed=323)
    add("batched_gemm_bp", Case([torch.tensor(1.5), torch.tensor(0.0), A0, A1, B0, B1, G0, G1], i=[0, 0, 3, 2, 4, 3, 4, 3, 2],
                                ref=lambda al, be, a0, a1, b0, b1, g0, g1: [1.5 * d64(g0) @ d64(b0).t(), 1.5 * d64(g1) @ d64(b1).t(), 1.5 * d64(a0).t() @ d64(g0),
                                                                            1.5 * d64(a1).t() @ d64(g1)], rtol=1e-5))
    add("reduce_dot_bp", Case([x34, y34, R(3, seed=11)], i=[1], ref=lambda a, b_, g: [d64(g).unsqueeze(1) * d64(b_), d64(g).unsqueeze(1) * d64(a)]))
    add("norm", Case([x34], t=[2.0], ref=lambda x: torch.tensor(np.linalg.norm(_np(x)))))
    add("norm", Case([x34], t=[1.0], i=[1], ref=lambda x: _t(np.abs(_np(x)).sum(1)), tag="L1 rows"))
    add("percentile", Case([x34], t=[50.0, 2.0], ref=lambda x: torch.tensor(np.percentile(_np(x), 50.0, method="nearest"))))
    add("percentile", Case([x34], t=[30.0, 0.0], i=[1], ref=lambda x: _t(np.percentile(_np(x), 30.0, axis=1, method="lower")), tag="lower, rows"))
    add("histogram", Case([x34], i=[4], ref=lambda x: _t(np.histogram(_np(x), bins=4, range=(float(x.min()), float(x.max())))[0])))
    add("normalize_moments", Case([torch.tensor(12.0), R(4, seed=12), R(4, seed=13, lo=10, hi=20)], t=[0.0],
                                  ref=lambda c, m, v_: [d64(m) / 12.0, d64(v_) / 12.0 - (d64(m) / 12.0) ** 2]))
    add("sufficient_statistics", Case([x34, I(0)], ref=lambda x, dd: [torch.tensor(3.0, dtype=torch.float64), d64(x).sum(0), (d64(x) ** 2).sum(0)]))
    add("knn_mindistance", Case([R(4, seed=117), R(4, seed=118) - 2, R(4, seed=119) + 2, ], ref=lambda x, lo, hi: torch.tensor(
        math.sqrt(sum(max(float(l_ - p_), float(p_ - h_), 0.0) ** 2 for p_, l_, h_ in zip(x, lo, hi))))))
    add("knn_mindistance", Case([torch.tensor([3.0, -4.0]), torch.tensor([0.0, 0.0]), torch.tensor([1.0, 1.0])], ref=lambda x, lo, hi: torch.tensor(math.sqrt(4.0 + 16.0)), tag="outside the box"))
    add("cell_contains", Case([torch.tensor([0.0, 0.0]), torch.tensor([1.0, 2.0]), torch.tensor([0.5, -1.5])], i=[2], ref=lambda c, w, p: torch.tensor(True)))
    add("cell_contains", Case([torch.tensor([0.0, 0.0]), torch.tensor([1.0, 2.0]), torch.tensor([1.5, 0.0])], i=[2], ref=lambda c, w, p: torch.tensor(False), tag="outside"))
    gains = x34.abs()
    add("barnes_gains", Case([gains, x34, y34], ref=lambda g, gx, e: _t(np.maximum(np.where(np.sign(_np(gx)) != np.sign(_np(e)), _np(g) + 0.2, _np(g) * 0.8), 0.01))))

    # ---- shape / bookkeeping ops with an exact expected value
    add("Assert", Case([torch.ones(3)], ref=lambda x: d64(x)))
    add("check_numerics", Case([x34, torch.zeros(1)], ref=lambda x, m: d64(x)))
    add("expose", Case([x34], ref=lambda x: d64(x)))
    add("print_variable", Case([x34], ref=lambda x: d64(x)))
    add("print_affinity", Case([x34], ref=lambda x: d64(x)))
    add("testcustom", Case([x34], ref=lambda x: d64(x)))
    add("test_scalar", Case([torch.tensor([1.0])], ref=lambda x: torch.tensor([3.0])))
    add("testop2i2o", Case([x34, y34], ref=lambda a, b_: [d64(a) + 1, d64(b_) + 2]))
    add("testreduction", Case([x34], i=[1], ref=lambda x: _t(_np(x).sum(1))))
    add("test_output_reshape", Case([x34], ref=lambda x: d64(x).reshape(-1)))
    add("noop", Case([x34], ref=lambda x: []))
    add("order", Case([x34.t()], i=[ord("c")], ref=lambda x: d64(x), check=None, tag="values kept"))
    add("flatten_2d", Case([R(2, 3, 4, seed=9)], i=[2], ref=lambda x: d64(x).reshape(6, 4)))
    add("set_shape", Case([x34, I(2, 6)], ref=lambda x, s: _t(_np(x).reshape(2, 6))))
    add("tear", Case([x34], i=[1], ref=lambda x: [d64(x)[k_] for k_ in range(3)]))
    add("create", Case([I(2, 3)], i=[ord("c"), 5], ref=lambda s: torch.zeros(2, 3, dtype=torch.float64)))
    add("create_view", Case([x34, I(1, 3, 1), I(2)], ref=lambda x, a, b_: d64(x)[1:3, 2]))
    add("eye", Case([], i=[3], ref=lambda: _t(np.eye(3))))
    add("eye", Case([], i=[99, 2, 3], ref=lambda: _t(np.eye(2, 3)), tag="rectangular"))
    add("meshgrid", Case([I(1.0, 2.0, dtype=torch.float32), I(3.0, 4.0, 5.0, dtype=torch.float32)], i=[0],
                         ref=lambda a, b_: [_t(v_) for v_ in np.meshgrid(_np(a), _np(b_), indexing="ij")]))
    add("meshgrid", Case([I(1.0, 2.0, dtype=torch.float32), I(3.0, 4.0, 5.0, dtype=torch.float32)], i=[1],
                         ref=lambda a, b_: [_t(v_) for v_ in np.meshgrid(_np(a), _np(b_), indexing="xy")], tag="cartesian"))
    add("cast", Case([x34 * 10], i=[9], ref=lambda x: _t(_np(x).astype(np.int32)), tag="to INT32 (truncation)"))
    add("cast", Case([x34], i=[6], ref=lambda x: d64(x), tag="to DOUBLE"))
    add("to_uint32", Case([pos * 100], ref=lambda x: _t(_np(x).astype(np.int64))))
    add("to_uint64", Case([pos * 100], ref=lambda x: _t(_np(x).astype(np.int64))))
    add("bitcast", Case([x34], i=[7], ref=lambda x: _t(x.numpy().view(np.int8).reshape(3, 4, 4))))
    add("bitcast", Case([x34], i=[9], ref=lambda x: _t(x.numpy().view(np.int32)), tag="same width"))
    add("min_max_datatype", Case([], i=[5, 1], ref=lambda: torch.tensor(float(np.finfo(np.float32).max), dtype=torch.float64)))
    add("min_max_datatype", Case([], i=[9, 0], ref=lambda: torch.tensor(int(np.iinfo(np.int32).min)), tag="int32 min"))
    add("choose", Case([x34], i=[3], t=[0.0], ref=lambda x: [d64(x)[x > 0], torch.tensor(int((x > 0).sum()))], tag="greater than"))
    add("firas_sparse", Case([x34], i=[1, 2], ref=lambda x: d64(x)[:, 1:3]))
    add("broadcastgradientargs", Case([I(3, 1), I(1, 4)], ref=lambda a, b_: [torch.tensor([1]), torch.tensor([0])]))
    add("broadcastgradientargs", Case([I(2, 3, 4), I(4)], ref=lambda a, b_: [torch.tensor([], dtype=torch.int64), torch.tensor([0, 1])], tag="rank extension"))
    add("compat_sparse_to_dense", Case([torch.tensor([[0, 1], [2, 3]]), I(3, 4), I(5.0, 6.0, dtype=torch.float32), torch.tensor(1.0)],
                                       ref=lambda ix, s, v_, dflt: torch.tensor([[1, 5.0, 1, 1], [1, 1, 1, 1], [1, 1, 1, 6.0]], dtype=torch.float64)))
    xu = torch.arange(12.0).reshape(3, 4)
    add("scatter_update", Case([xu, torch.tensor([[10.0, 20.0, 30.0, 40.0], [1.0, 2.0, 3.0, 4.0]])], i=[0, 1, 1, 2, 2, 0],
                               ref=lambda x, u: torch.tensor([[1.0, 3, 5, 7], [4, 5, 6, 7], [18.0, 29, 40, 51]], dtype=torch.float64), tag="add to rows 2 and 0"))

    def _hash_ref(x):
        bits = x.numpy().view(np.int32).reshape(-1).tolist()
        nn = len(bits)
        tot = 0
        for k_, v_ in enumerate(bits):
            w_ = pow(31, (nn - 1 - k_) % 62, 1 << 64)
            tot = (tot + v_ * w_) % (1 << 64)
        return torch.tensor(tot - (1 << 64) if tot >= (1 << 63) else tot, dtype=torch.int64)
    add("hashcode", Case([x34], ref=_hash_ref, tag="31-polynomial over the raw bits, 64-bit wrap"))
    xz = x34.clone()
    xz[0, 1] = 0.0
    xz[2, 3] = 0.0
    add("dropout_bp", Case([xz, (xz != 0).float(), y34], t=[0.5], i=[0], ref=lambda x, m, g: torch.where(d64(m) != 0, d64(g) / 0.5, torch.zeros_like(d64(g)))))
    add("alpha_dropout_bp", Case([xz, y34], t=[0.5, 1.7, 1.0, 0.0], i=[0], ref=lambda x, g: torch.where(d64(x) != 0, d64(g) * 1.7, torch.zeros_like(d64(g)))))

    # ---- gradients of the piecewise-linear arithmetic ops: central differences of the NumPy forward (inputs away from the jumps)
    def _num_bp(fwd):
        def ref(a, b_, g):
            A, Bv, G = _np(a), _np(b_), _np(g)
            eps = 1e-6
            outs = []
            for which, X in ((0, A), (1, Bv)):
                gr = np.zeros_like(X)
                for idx in np.ndindex(*X.shape):
                    P, M = X.copy(), X.copy()
                    P[idx] += eps
                    M[idx] -= eps
                    fp, fm = (fwd(P, Bv), fwd(M, Bv)) if which == 0 else (fwd(A, P), fwd(A, M))
                    gr[idx] = ((fp - fm) * G).sum() / (2 * eps)
                outs.append(_t(gr))
            return outs
        return ref
    pa, pb = R(3, 4, seed=3, lo=0.5, hi=2.5), R(3, 4, seed=4, lo=0.5, hi=2.5)
    for nm, fwd in (("floordiv_bp", lambda a, b_: np.floor(a / b_)), ("floormod_bp", lambda a, b_: a - np.floor(a / b_) * b_), ("mod_bp", np.fmod),
                    ("reversemod_bp", lambda a, b_: np.fmod(b_, a)), ("assign_bp", lambda a, b_: b_ + 0 * a)):
        add(nm, Case([pa, pb, y34], ref=_num_bp(fwd), rtol=1e-4, atol=1e-5, tag="central differences"))
    add("resize_bilinear", Case([rgb, I(8, 8, dtype=torch.int32)], b=[False, False], ref=lambda x, s_: _bilinear_tf1(x, 8, 8), rtol=1e-4, atol=1e-5, tag="TF1 sampling"))
    add("resize_nearest_neighbor", Case([rgb, I(8, 8, dtype=torch.int32)], b=[False, False], ref=lambda x, s_: _t(_np(x).repeat(2, 1).repeat(2, 2)), tag="x2"))

    # ---- strings
    from ..nd4j.ops.impl_misc import StringArray
    add("split_string", Case([StringArray(["a b", "c d e"]), StringArray([" "])], ref=lambda a, dl: [StringArray(["a", "b", "c", "d", "e"])]))
    add("compat_string_split", Case([StringArray(["a b", "c d e"]), StringArray([" "])],
                                    ref=lambda a, dl: [torch.tensor([[0, 0], [0, 1], [1, 0], [1, 1], [1, 2]]), StringArray(["a", "b", "c", "d", "e"])]))


def _seed_scenario(device):
    from ..nd4j.ops.custom import exec_op
    exec_op("set_seed", [torch.tensor([1234])])
    got = int(exec_op("get_seed", [])[0])
    return None if got == 1234 else "get_seed returned %d after set_seed(1234)" % got


EXTRA_SCENARIOS = [({"set_seed", "get_seed"}, _seed_scenario)]
