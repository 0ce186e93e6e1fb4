"""Platform helpers: custom ops routed to the hand-written gfx950 helper kernels when ``isUsable`` holds.

Reference: PLATFORM_IMPL / PLATFORM_CHECK (ops/declarable/PlatformHelper.h; cuDNN helpers in platform/cudnn/*.cu) and
the dispatch in DeclarableOp.cpp:692-739. The generic implementations (impl_core / impl_nn / impl_misc) stay the
oracle: tests run every op below with helpers allowed and disallowed and compare.
"""
from __future__ import annotations

import torch

from ...native import lib as L
from ...ops import helpers as H
from .custom import platform

_F = (torch.float32, torch.bfloat16)


def _fx(x):
    return isinstance(x, torch.Tensor) and H.ok(x)


def _last_axis(i, x, pos=0):
    return ((int(i[pos]) if len(i) > pos else -1) % max(x.dim(), 1)) == x.dim() - 1


# ----------------------------------------------------------------------------- softmax family
@platform("softmax", lambda ins, i, t, b, d: _fx(ins[0]) and ins[0].dim() > 2 and _last_axis(i, ins[0]))
def _softmax_nd(ins, i, t, b, d):
    return H.softmax_last(ins[0], False)


@platform("log_softmax", lambda ins, i, t, b, d: _fx(ins[0]) and ins[0].dim() >= 2 and _last_axis(i, ins[0]))
def _log_softmax(ins, i, t, b, d):
    return H.softmax_last(ins[0], True)


def _sm_bp_ok(ins, i, t, b, d):
    return (_fx(ins[0]) and _fx(ins[1]) and ins[0].dim() >= 2 and ins[0].shape == ins[1].shape and ins[0].dtype == ins[1].dtype
            and _last_axis(i, ins[0]))


@platform("softmax_bp", _sm_bp_ok)
def _softmax_bp(ins, i, t, b, d):
    return H.softmax_bwd_last(ins[0], ins[1], False)


@platform("log_softmax_bp", _sm_bp_ok)
def _log_softmax_bp(ins, i, t, b, d):
    return H.softmax_bwd_last(ins[0], ins[1], True)


# ----------------------------------------------------------------------------- layer norm / bias / prelu / lrn / clip / dropout
def _ln_ok(ins, i, t, b, d):
    x = ins[0]
    if not _fx(x) or x.dim() < 2 or (b and b[0] and x.dim() > 2):
        return False
    axes = [a % x.dim() for a in i] if i else [x.dim() - 1]
    return axes == [x.dim() - 1] and ins[1].numel() == x.shape[-1] and (len(ins) < 3 or ins[2].numel() == x.shape[-1])


@platform("layer_norm", _ln_ok)
def _layer_norm(ins, i, t, b, d):
    return H.layer_norm(ins[0], ins[1], ins[2] if len(ins) > 2 else None, float(t[0]) if t else 1e-5)


def _ln_bp_ok(ins, i, t, b, d):
    # (x, gain, [bias], eps_grad)
    return _ln_ok(ins[:-1], i, t, b, d) and _fx(ins[-1]) and ins[-1].shape == ins[0].shape and ins[-1].dtype == ins[0].dtype


@platform("layer_norm_bp", _ln_bp_ok)
def _layer_norm_bp(ins, i, t, b, d):
    x, gain, g = ins[0], ins[1], ins[-1]
    _, mean, rstd = H.layer_norm(x, None, None, float(t[0]) if t else 1e-5, want_stats=True)
    dx, dgamma, dbeta = H.layer_norm_bwd(x, g, gain, mean, rstd)
    outs = [dx, dgamma.to(gain.dtype).reshape(gain.shape)]
    if len(ins) > 3:
        outs.append(dbeta.to(ins[2].dtype).reshape(ins[2].shape))
    return tuple(outs)


def _bias_ok(ins, i, t, b, d):
    x, bias = ins[0], ins[1]
    if not _fx(x) or x.dim() < 2:
        return False
    nchw = bool(b[0]) if b else (bool(i[0]) if i else False)
    ax = 1 if (nchw and x.dim() > 2) else x.dim() - 1
    return bias.numel() == x.shape[ax]


@platform("biasadd", _bias_ok)
def _biasadd(ins, i, t, b, d):
    x = ins[0]
    nchw = bool(b[0]) if b else (bool(i[0]) if i else False)
    return H.bias_add(x, ins[1], 1 if (nchw and x.dim() > 2) else -1)


@platform("prelu", lambda ins, i, t, b, d: _fx(ins[0]) and ins[0].dim() >= 2 and ins[1].numel() == ins[0].shape[1]
          and (ins[1].dim() == 1 or all(int(s) == 1 for s in ins[1].shape[1:])))
def _prelu(ins, i, t, b, d):
    return H.prelu(ins[0], ins[1], 1)


@platform("lrn", lambda ins, i, t, b, d: _fx(ins[0]) and ins[0].dim() == 4)
def _lrn(ins, i, t, b, d):
    x = ins[0]                                                  # NCHW logical; LRN runs across C
    bias, alpha, beta = (list(t) + [1.0, 1.0, 0.5])[:3] if t else (1.0, 1.0, 0.5)
    xl = x.permute(0, 2, 3, 1).contiguous()
    return H.lrn_last(xl, int(i[0]) if i else 5, bias, alpha, beta).permute(0, 3, 1, 2).contiguous()


@platform("clipbynorm", lambda ins, i, t, b, d: _fx(ins[0]) and (not i or sorted(k % ins[0].dim() for k in i) == list(range(ins[0].dim())))
          and (bool(t) or len(ins) > 1))
def _clipbynorm(ins, i, t, b, d):
    return H.clip_by_norm(ins[0], float(t[0]) if t else float(ins[1]))


@platform("dropout", lambda ins, i, t, b, d: _fx(ins[0]))
def _dropout(ins, i, t, b, d):
    from ..rng import getRandom
    x = ins[0]
    p = float(t[0]) if t else 0.5
    key, node, off = getRandom().reserve(x.numel())          # advances the generator exactly like a legacy random fill
    y, mask = H.dropout(x, p, key, (node ^ (off * 0x9E3779B97F4A7C15)) & (2 ** 64 - 1))
    return y, mask.to(x.dtype)


# ----------------------------------------------------------------------------- indexed
def _idx(t):
    return isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in (torch.int32, torch.int64) and t.is_contiguous() and t.numel() > 0


@platform("embedding_lookup", lambda ins, i, t, b, d: L.use_hip(ins[0]) and ins[0].dim() == 2 and ins[0].is_contiguous()
          and ins[0].element_size() in (1, 2, 4, 8) and _idx(ins[1]))
def _embedding_lookup(ins, i, t, b, d):
    return H.gather_rows(ins[0], ins[1])


@platform("gather", lambda ins, i, t, b, d: len(ins) > 1 and L.use_hip(ins[0]) and ins[0].dim() >= 1 and ins[0].is_contiguous()
          and (int(i[0]) if i else 0) % ins[0].dim() == 0 and ins[0].element_size() in (1, 2, 4, 8) and _idx(ins[1])
          and ins[0].numel() > 0)
def _gather0(ins, i, t, b, d):
    x, idx = ins[0], ins[1]
    V = x.shape[0]
    out = H.gather_rows(x.reshape(V, -1), idx)
    return out.reshape(tuple(idx.shape) + tuple(x.shape[1:]))


@platform("onehot", lambda ins, i, t, b, d: _idx(ins[0]) and (int(i[0]) if i else -1) in (-1, ins[0].dim()) and (len(i) > 1 or len(ins) > 1)
          and (not d or str(d[0]).upper() in ("FLOAT", "FLOAT32", "BFLOAT16")))
def _onehot(ins, i, t, b, d):
    depth = int(i[1]) if len(i) > 1 else int(ins[1])
    on = t[0] if t else (float(ins[2]) if len(ins) > 2 else 1.0)
    off = t[1] if len(t) > 1 else (float(ins[3]) if len(ins) > 3 else 0.0)
    dt = torch.bfloat16 if (d and str(d[0]).upper() == "BFLOAT16") else torch.float32
    return H.one_hot(ins[0], depth, on, off, dt)


def _seg_ok(sorted_ids):
    def ok(ins, i, t, b, d):
        return _fx(ins[0]) and _idx(ins[1]) and ins[1].dim() == 1 and ins[1].numel() == ins[0].shape[0] and (sorted_ids or i or len(ins) > 2)
    return ok


@platform("unsorted_segment_sum", _seg_ok(False))
def _unsorted_segment_sum(ins, i, t, b, d):
    return H.segment_sum(ins[0], ins[1], int(i[0]) if i else int(ins[2]))


@platform("segment_sum", _seg_ok(True))
def _segment_sum(ins, i, t, b, d):
    n = int(ins[1].max()) + 1          # output shape is data dependent (same host read as the generic path)
    return H.segment_sum(ins[0], ins[1], n)


def _cumsum_ok(ins, i, t, b, d):
    x = ins[0]
    if not _fx(x) or x.dim() < 1:
        return False
    dims = list(i[2:]) if len(i) > 2 else None
    return dims is not None and len(dims) == 1 and int(dims[0]) % x.dim() == x.dim() - 1


@platform("cumsum", _cumsum_ok)
def _cumsum(ins, i, t, b, d):
    return H.cumsum_last(ins[0], bool(i[0]) if i else False, bool(i[1]) if len(i) > 1 else False)


def _merge_ok(ins, i, t, b, d):
    return (len(ins) >= 2 and all(_fx(x) for x in ins) and all(x.shape == ins[0].shape and x.dtype == ins[0].dtype for x in ins))


for _name, _mode in (("mergeadd", "add"), ("mergeavg", "avg"), ("mergemax", "max")):
    platform(_name, _merge_ok)(lambda ins, i, t, b, d, _m=_mode: H.merge(list(ins), _m))


# ----------------------------------------------------------------------------- im2col / col2im / upsampling / resize
@platform("im2col", lambda ins, i, t, b, d: _fx(ins[0]) and ins[0].dim() == 4 and len(i) >= 8 and not (len(i) > 8 and i[8]))
def _im2col(ins, i, t, b, d):
    kH, kW, sH, sW, pH, pW, dH, dW = (int(v) for v in i[:8])
    return H.im2col(ins[0], kH, kW, sH, sW, pH, pW, dH, dW, float(t[0]) if t else 0.0)


@platform("col2im", lambda ins, i, t, b, d: _fx(ins[0]) and ins[0].dim() == 6 and len(i) >= 8)
def _col2im(ins, i, t, b, d):
    sH, sW, pH, pW, iH, iW, dH, dW = (int(v) for v in i[:8])
    return H.col2im(ins[0], iH, iW, sH, sW, pH, pW, dH, dW)


@platform("upsampling2d", lambda ins, i, t, b, d: L.use_hip(ins[0]) and ins[0].dim() == 4 and ins[0].is_contiguous() and len(i) >= 2
          and (len(i) < 3 or i[2]) and ins[0].element_size() in (1, 2, 4, 8) and ins[0].numel() > 0)
def _upsampling2d(ins, i, t, b, d):
    return H.upsample2d(ins[0], int(i[0]), int(i[1]))


@platform("upsampling2d_bp", lambda ins, i, t, b, d: _fx(ins[1]) and ins[1].dim() == 4 and (not i or i[0]) and ins[0].dim() == 4
          and ins[1].shape[2] % ins[0].shape[2] == 0 and ins[1].shape[3] % ins[0].shape[3] == 0)
def _upsampling2d_bp(ins, i, t, b, d):
    x, g = ins[0], ins[1]
    return H.upsample2d_bwd(g, g.shape[2] // x.shape[2], g.shape[3] // x.shape[3])


def _resize_ok(ins, i, t, b, d):
    return _fx(ins[0]) and ins[0].dim() == 4 and (len(ins) > 1 or len(i) >= 2)


def _resize(mode):
    def f(ins, i, t, b, d):
        x = ins[0]
        if len(ins) > 1:
            h, w = [int(v) for v in ins[1].reshape(-1).tolist()]
        else:
            h, w = int(i[0]), int(i[1])
        return H.resize_nhwc(x, h, w, mode, bool(b[0]) if b else False, bool(b[1]) if len(b) > 1 else False)
    return f


platform("resize_bilinear", _resize_ok)(_resize("bilinear"))
platform("resize_nearest_neighbor", _resize_ok)(_resize("nearest"))


# ----------------------------------------------------------------------------- index-map family
def _mv(x):
    return (isinstance(x, torch.Tensor) and L.use_hip(x) and x.element_size() in (1, 2, 4, 8) and 1 <= x.dim() <= 8 and x.numel() > 0
            and x.numel() < 2 ** 31 and not x.is_complex())


def _pads_of(ins):
    return [(int(lo), int(hi)) for lo, hi in ins[1].long().reshape(-1, 2).tolist()]


def _pad_ok(ins, i, t, b, d):
    if not (_mv(ins[0]) and len(ins) > 1 and ins[1].numel() == 2 * ins[0].dim()):
        return False
    mode = int(i[0]) if i else 0
    pads = _pads_of(ins)
    if mode == 1:
        return all(lo < s and hi < s and lo >= 0 and hi >= 0 for s, (lo, hi) in zip(ins[0].shape, pads))
    if mode == 2:
        return all(lo <= s and hi <= s and lo >= 0 and hi >= 0 for s, (lo, hi) in zip(ins[0].shape, pads))
    return mode == 0 and all(s + lo + hi > 0 for s, (lo, hi) in zip(ins[0].shape, pads))


@platform("pad", _pad_ok)
def _pad(ins, i, t, b, d):
    mode = {0: "constant", 1: "reflect", 2: "symmetric"}[int(i[0]) if i else 0]
    val = t[0] if t else (float(ins[2]) if len(ins) > 2 else 0.0)
    return H.pad_nd(ins[0], _pads_of(ins), mode, val)


@platform("mirror_pad", lambda ins, i, t, b, d: _pad_ok(ins, [2 if (i and i[0] == 1) else 1], t, b, d))
def _mirror_pad(ins, i, t, b, d):
    return H.pad_nd(ins[0], _pads_of(ins), "symmetric" if (i and i[0] == 1) else "reflect")


@platform("reverse", lambda ins, i, t, b, d: _mv(ins[0]) and (bool(i) or len(ins) == 1))
def _reverse(ins, i, t, b, d):
    return H.reverse_dims(ins[0], list(i) if i else list(range(ins[0].dim())))


@platform("tile", lambda ins, i, t, b, d: _mv(ins[0]) and bool(i) and len(i) >= ins[0].dim() and len(i) <= 8 and all(int(r) >= 1 for r in i))
def _tile(ins, i, t, b, d):
    return H.tile_nd(ins[0], list(i))


@platform("roll", lambda ins, i, t, b, d: _mv(ins[0]) and len(i) >= 2)
def _roll(ins, i, t, b, d):
    dims = list(i[1:])
    return H.roll_nd(ins[0], [int(i[0])] * len(dims), dims)


@platform("concat", lambda ins, i, t, b, d: not (b and b[0]) and len(ins) >= 2 and all(_mv(x) for x in ins)
          and all(x.dim() == ins[0].dim() and x.dtype == ins[0].dtype for x in ins))
def _concat(ins, i, t, b, d):
    return H.concat_nd(list(ins), int(i[0]) if i else 0)
