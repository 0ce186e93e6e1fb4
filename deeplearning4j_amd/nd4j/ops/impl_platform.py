"""Platform helpers: custom ops routed to the hand-written gfx950 helper kernels when ``isUsable`` holds.

Reference: PLATFORM_IMPL / PLATFORM_CHECK (ops/declarable/PlatformHelper.h; cuDNN helpers in platform/cudnn/*.cu) and
the dispatch in DeclarableOp.cpp:692-739. The generic implementations (impl_core / impl_nn / impl_misc) stay the
oracle: tests run every op below with helpers allowed and disallowed and compare.
"""
from __future__ import annotations

import torch

from ...native import lib as L
from ...ops import helpers as H
from .custom import op, platform

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


@platform("layer_norm", _ln_ok, differentiable=True)
def _layer_norm(ins, i, t, b, d):
    eps = float(t[0]) if t else 1e-5
    if torch.is_grad_enabled() and any(isinstance(v, torch.Tensor) and v.requires_grad for v in ins):
        return H.LayerNormFn.apply(ins[0], ins[1], ins[2] if len(ins) > 2 else None, eps)
    return H.layer_norm(ins[0], ins[1], ins[2] if len(ins) > 2 else None, eps)


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


class _EmbeddingFn(torch.autograd.Function):
    """Row gather forward; the backward scatter-adds the rows of dy into an fp32 table: directly into the variable's view of the flat
    gradient vector when the table is linked (ops/linear.py Links) instead of materialising, sorting and copying a dense [V, D] gradient."""

    @staticmethod
    def forward(ctx, table, idx):
        from ...ops import linear as LIN
        ctx.save_for_backward(idx)
        ctx.meta = (table.shape, table.dtype, LIN.link_of(table))
        return H.gather_rows(table, idx)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        shape, dtype, lk = ctx.meta
        src = dy if (dy.is_contiguous() and dy.dtype in (torch.float32, torch.bfloat16)) else dy.float().contiguous()
        if lk is not None:
            lk.hits += 1
            H.scatter_add_rows_(lk.grad.view(shape), idx, src)
            return None, None
        g = torch.zeros(shape, device=dy.device, dtype=torch.float32)
        H.scatter_add_rows_(g, idx, src)
        return g.to(dtype), None


@platform("embedding_lookup", lambda ins, i, t, b, d: L.use_hip(ins[0]) and ins[0].dim() == 2 and ins[0].is_contiguous()
          and ins[0].element_size() in (1, 2, 4, 8) and _idx(ins[1]) and ins[1].is_contiguous(), differentiable=True)
def _embedding_lookup(ins, i, t, b, d):
    if torch.is_grad_enabled() and ins[0].requires_grad and ins[0].dtype in (torch.float32, torch.bfloat16):
        return _EmbeddingFn.apply(ins[0], ins[1])
    return H.gather_rows(ins[0].detach(), ins[1])


@platform("gather", lambda ins, i, t, b, d: len(ins) > 1 and L.use_hip(ins[0]) and ins[0].dim() >= 1 and ins[0].is_contiguous()
          and (int(i[0]) if i else 0) % ins[0].dim() == 0 and ins[0].element_size() in (1, 2, 4, 8) and _idx(ins[1])
          and ins[0].numel() > 0, differentiable=True)
def _gather0(ins, i, t, b, d):
    x, idx = ins[0], ins[1]
    V = x.shape[0]
    if torch.is_grad_enabled() and x.requires_grad:
        if x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16):
            return _EmbeddingFn.apply(x, idx)               # row gather = embedding lookup (TF-imported GatherV2 on a table)
        return x[idx]
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


# ----------------------------------------------------------------------------- selection / indexed reductions (helpers_select.hip)
def _reg_segment(kind):
    platform("unsorted_segment_" + kind, _seg_ok(False))(
        lambda ins, i, t, b, d: H.segment_reduce(ins[0], ins[1], int(i[0]) if i else int(ins[2]), kind))
    platform("segment_" + kind, _seg_ok(True))(
        lambda ins, i, t, b, d: H.segment_reduce(ins[0], ins[1], int(ins[1].max()) + 1, kind))


for _kind in ("max", "min", "prod", "mean"):
    _reg_segment(_kind)


def _nd_ok(x, idx):
    return (L.use_hip(x) and x.is_contiguous() and x.numel() > 0 and x.element_size() in (1, 2, 4, 8) and _idx(idx) and idx.dim() >= 2
            and 1 <= idx.shape[-1] <= min(8, x.dim()))


@platform("gather_nd", lambda ins, i, t, b, d: _nd_ok(ins[0], ins[1]))
def _gather_nd(ins, i, t, b, d):
    return H.gather_nd(ins[0], ins[1])


def _scatter_nd_ok(ins, i, t, b, d):
    return (len(ins) >= 3 and _fx(ins[0]) and _nd_ok(ins[0], ins[1]) and isinstance(ins[2], torch.Tensor) and ins[2].is_cuda
            and ins[2].is_floating_point())


for _kind in ("add", "sub", "update"):
    platform("scatter_nd_" + _kind, _scatter_nd_ok)(lambda ins, i, t, b, d, _k=_kind: H.scatter_nd(ins[0], ins[1], ins[2], _k))


@platform("scatter_nd", lambda ins, i, t, b, d: len(ins) >= 3 and _idx(ins[0]) and ins[0].dim() >= 2 and _fx(ins[1])
          and 1 <= ins[0].shape[-1] <= min(8, ins[2].numel()))
def _scatter_nd_new(ins, i, t, b, d):
    shape = [int(v) for v in ins[2].tolist()]
    return H.scatter_nd(torch.zeros(shape, dtype=ins[1].dtype, device=ins[1].device), ins[0], ins[1], "add")


@platform("top_k", lambda ins, i, t, b, d: _fx(ins[0]) and ins[0].dim() >= 1 and 0 < (int(i[0]) if i else int(ins[1])) <= min(ins[0].shape[-1], 256))
def _top_k(ins, i, t, b, d):
    from .impl_core import dtype_arg
    v, ix = H.top_k(ins[0], int(i[0]) if i else int(ins[1]))
    return v, (ix.to(dtype_arg(d, None, 0, torch.int32)) if d else ix)


@platform("histogram_fixed_width", lambda ins, i, t, b, d: _fx(ins[0]) and (len(ins) > 2 or bool(i)) and float(ins[1].reshape(-1)[1]) > float(ins[1].reshape(-1)[0]))
def _hist_fw(ins, i, t, b, d):
    rng = ins[1].reshape(-1).float()
    return H.histogram_fixed_width(ins[0], float(rng[0]), float(rng[1]), int(ins[2]) if len(ins) > 2 else int(i[0]))


@platform("lrn_bp", lambda ins, i, t, b, d: _fx(ins[0]) and ins[0].dim() == 4 and ins[0].shape[1] <= 4096 and _fx(ins[1])
          and ins[1].shape == ins[0].shape and ins[1].dtype == ins[0].dtype)
def _lrn_bp(ins, i, t, b, d):
    bias, alpha, beta = (list(t) + [1.0, 1.0, 0.5])[:3] if t else (1.0, 1.0, 0.5)
    xl = ins[0].permute(0, 2, 3, 1).contiguous()
    gl = ins[1].permute(0, 2, 3, 1).contiguous()
    return H.lrn_bwd_last(xl, gl, int(i[0]) if i else 5, bias, alpha, beta).permute(0, 3, 1, 2).contiguous()


def _sru_ok(ins, i, t, b, d):
    if len(ins) < 4 or not all(isinstance(v, torch.Tensor) and L.use_hip(v) and v.dtype == torch.float32 for v in ins[:4]):
        return False
    x, w, bias, c0 = ins[:4]
    return x.dim() == 3 and w.shape == (3 * x.shape[1], x.shape[1]) and bias.numel() == 2 * x.shape[1] and c0.shape == x.shape[:2]


@platform("sru", _sru_ok)
def _sru(ins, i, t, b, d):
    h, c, _, _ = H.sru_fwd(ins[0], ins[1], ins[2], ins[3], ins[4] if len(ins) > 4 else None)
    return h, c


@platform("sru_bp", lambda ins, i, t, b, d: _sru_ok(ins, i, t, b, d) and len(ins) == 5 and ins[4].shape == ins[0].shape)
def _sru_bp(ins, i, t, b, d):
    return H.sru_bwd(ins[0], ins[1], ins[2], ins[3], ins[4])


def _color_ok(ins, i, t, b, d):
    x = ins[0]
    return _fx(x) and x.dim() >= 1 and x.shape[-1] == 3 and ((int(i[0]) if i else -1) % x.dim()) == x.dim() - 1


platform("rgb_to_hsv", _color_ok)(lambda ins, i, t, b, d: H.color_convert(ins[0], "rgb_to_hsv"))
platform("hsv_to_rgb", _color_ok)(lambda ins, i, t, b, d: H.color_convert(ins[0], "hsv_to_rgb"))


def _reg_color_matrix():
    from . import impl_misc as M
    inv = lambda m: torch.linalg.inv(torch.tensor(m, dtype=torch.float64)).tolist()
    for name, mat in (("rgb_to_yuv", M._YUV), ("yuv_to_rgb", inv(M._YUV)), ("rgb_to_yiq", M._YIQ), ("yiq_to_rgb", inv(M._YIQ))):
        platform(name, _color_ok)(lambda ins, i, t, b, d, _m=mat: H.color_convert(ins[0], "matrix", _m))


_reg_color_matrix()


def _nms_ok(ins, i, t, b, d):
    return (len(ins) >= 2 and L.use_hip(ins[0]) and ins[0].dim() == 2 and ins[0].shape[1] == 4 and ins[0].is_floating_point()
            and ins[1].is_cuda and ins[1].numel() == ins[0].shape[0] and (bool(i) or len(ins) > 2))


@platform("non_max_suppression", _nms_ok)
def _nms(ins, i, t, b, d):
    max_out = int(i[0]) if i else int(ins[2])
    thr = t[0] if t else (float(ins[3]) if len(ins) > 3 else 0.5)
    sthr = t[1] if len(t) > 1 else (float(ins[4]) if len(ins) > 4 else float("-inf"))
    return H.nms(ins[0], ins[1].reshape(-1), max_out, thr, sthr)


platform("non_max_suppression_v3", _nms_ok)(_nms)


@platform("non_max_suppression_overlaps", lambda ins, i, t, b, d: len(ins) >= 2 and L.use_hip(ins[0]) and ins[0].dim() == 2
          and ins[0].shape[0] == ins[0].shape[1] == ins[1].numel() and ins[0].is_floating_point() and (bool(i) or len(ins) > 2))
def _nms_ov(ins, i, t, b, d):
    return H.nms(None, ins[1].reshape(-1), int(i[0]) if i else int(ins[2]), t[0] if t else 0.5, t[1] if len(t) > 1 else float("-inf"),
                 overlaps=ins[0])


# ----------------------------------------------------------------------------- spatial (helpers_spatial.hip)
def _dw_parse(ins, i):
    from .impl_nn import _conv2d_args, _same_pad
    c = _conv2d_args(i)
    x, w = ins[0], ins[1]
    kH, kW, iC, mC = w.shape
    if c["same"]:
        ph, pw = _same_pad(x.shape[2], kH, c["s"][0], c["dl"][0]), _same_pad(x.shape[3], kW, c["s"][1], c["dl"][1])
        p_lo, out = [ph[0], pw[0]], (-(-x.shape[2] // c["s"][0]), -(-x.shape[3] // c["s"][1]))
    else:
        p_lo, out = list(c["p"]), None
    return c, p_lo, out


def _dw_ok(ins, i, t, b, d):
    from .impl_nn import _conv2d_args
    c = _conv2d_args(i)
    x, w = ins[0], ins[1]
    return (_fx(x) and x.dim() == 4 and isinstance(w, torch.Tensor) and w.is_cuda and w.dim() == 4 and w.dtype == x.dtype and not c["nhwc"]
            and c["wfmt"] == 0 and w.shape[2] == x.shape[1] and min(c["s"]) > 0)


@platform("depthwise_conv2d", _dw_ok)
def _depthwise(ins, i, t, b, d):
    c, p_lo, out = _dw_parse(ins, i)
    w = ins[1]
    kH, kW, iC, mC = w.shape
    wk = w.permute(2, 3, 0, 1).reshape(iC * mC, kH, kW).contiguous()
    return H.depthwise_fwd(ins[0], wk, ins[2] if len(ins) > 2 else None, c["s"], p_lo, c["dl"], out)


@platform("depthwise_conv2d_bp", lambda ins, i, t, b, d: len(ins) >= 3 and _dw_ok(ins[:-1], i, t, b, d) and _fx(ins[-1]) and ins[-1].dtype == ins[0].dtype)
def _depthwise_bp(ins, i, t, b, d):
    *fw, g = ins
    c, p_lo, out = _dw_parse(fw, i)
    x, w = fw[0], fw[1]
    kH, kW, iC, mC = w.shape
    wk = w.permute(2, 3, 0, 1).reshape(iC * mC, kH, kW).contiguous()
    dx, dw, db = H.depthwise_bwd(x, wk, g, c["s"], p_lo, c["dl"], want_bias=len(fw) > 2)
    dwk = dw.reshape(iC, mC, kH, kW).permute(2, 3, 0, 1).contiguous().to(w.dtype)
    return (dx, dwk) + ((db.reshape(fw[2].shape).to(fw[2].dtype),) if len(fw) > 2 else ())


def _p3_parse(i):
    a = list(i) + [0] * (15 - len(i))
    return dict(k=a[0:3], s=a[3:6], p=a[6:9], dl=[max(v, 1) for v in a[9:12]], same=bool(a[12]), extra=a[13], ndhwc=a[14] == 1)


def _p3_ok(kind):
    def ok(ins, i, t, b, d):
        c = _p3_parse(i)
        return (_fx(ins[0]) and ins[0].dim() == 5 and not c["ndhwc"] and not c["same"] and min(c["k"]) > 0 and min(c["s"]) > 0
                and (kind == "max" or c["dl"] == [1, 1, 1]) and all(2 * c["p"][a] <= c["k"][a] for a in range(3)))
    return ok


def _reg_pool3d(kind):
    @platform(kind + "pool3dnew", _p3_ok(kind))
    def fwd(ins, i, t, b, d):
        c = _p3_parse(i)
        return H.pool3d(ins[0], c["k"], c["s"], c["p"], c["dl"], kind, include_pad=(c["extra"] == 0))

    @platform(kind + "pool3dnew_bp", lambda ins, i, t, b, d: len(ins) == 2 and _p3_ok(kind)(ins, i, t, b, d) and _fx(ins[1]) and ins[1].dtype == ins[0].dtype)
    def bwd(ins, i, t, b, d):
        c = _p3_parse(i)
        x, g = ins
        y = H.pool3d(x, c["k"], c["s"], c["p"], c["dl"], kind, include_pad=(c["extra"] == 0))
        return H.pool3d_bwd(x, y, g, c["k"], c["s"], c["p"], c["dl"], kind, include_pad=(c["extra"] == 0))


_reg_pool3d("max")
_reg_pool3d("avg")


@platform("upsampling3d", lambda ins, i, t, b, d: L.use_hip(ins[0]) and ins[0].dim() == 5 and ins[0].is_contiguous() and len(i) >= 3
          and (len(i) < 4 or bool(i[3])) and ins[0].element_size() in (1, 2, 4, 8) and min(i[:3]) > 0)
def _upsampling3d(ins, i, t, b, d):
    return H.upsample3d(ins[0], int(i[0]), int(i[1]), int(i[2]))


@platform("upsampling3d_bp", lambda ins, i, t, b, d: _fx(ins[1]) and ins[1].dim() == 5 and (not i or bool(i[0]))
          and all(ins[1].shape[2 + a] % ins[0].shape[2 + a] == 0 for a in range(3)))
def _upsampling3d_bp(ins, i, t, b, d):
    x, g = ins[0], ins[1]
    return H.upsample3d_bwd(g, *[g.shape[2 + a] // x.shape[2 + a] for a in range(3)])


def _nhwc_window(x, kH, kW, sH, sW, rH, rW, same):
    from .impl_nn import _same_pad
    H_, W_ = x.shape[1], x.shape[2]
    if same:
        ph, pw = _same_pad(H_, kH, sH, rH), _same_pad(W_, kW, sW, rW)
        return [ph[0], pw[0]], (-(-H_ // sH), -(-W_ // sW))
    return [0, 0], ((H_ - (kH - 1) * rH - 1) // sH + 1, (W_ - (kW - 1) * rW - 1) // sW + 1)


def _dil_args(i):
    rates = list(i[1:5]) if len(i) >= 9 else [1, 1, 1, 1]
    strides = list(i[5:9]) if len(i) >= 9 else [1, 1, 1, 1]
    return bool(i[0]) if i else False, rates[1], rates[2], strides[1], strides[2]


def _dil_ok(ins, i, t, b, d):
    x, w = ins[0], ins[1]
    if not (_fx(x) and x.dim() == 4 and _fx(w) and w.dim() == 3 and w.dtype == x.dtype and w.shape[2] == x.shape[3]):
        return False
    same, rH, rW, sH, sW = _dil_args(i)
    _, out = _nhwc_window(x, w.shape[0], w.shape[1], sH, sW, rH, rW, same)
    return min(rH, rW, sH, sW) > 0 and min(out) > 0


@platform("dilation2d", _dil_ok)
def _dilation2d(ins, i, t, b, d):
    x, w = ins[0], ins[1]
    same, rH, rW, sH, sW = _dil_args(i)
    p_lo, out = _nhwc_window(x, w.shape[0], w.shape[1], sH, sW, rH, rW, same)
    return H.dilation2d(x, w, (sH, sW), p_lo, (rH, rW), out)


def _patches_ok(ins, i, t, b, d):
    x = ins[0]
    if not (L.use_hip(x) and x.dim() == 4 and x.is_contiguous() and x.element_size() in (1, 2, 4, 8) and len(i) >= 6 and min(i[:6]) > 0):
        return False
    _, out = _nhwc_window(x, i[0], i[1], i[2], i[3], i[4], i[5], bool(i[6]) if len(i) > 6 else False)
    return min(out) > 0


@platform("extract_image_patches", _patches_ok)
def _extract_patches(ins, i, t, b, d):
    x = ins[0]
    kH, kW, sH, sW, rH, rW = [int(v) for v in i[:6]]
    p_lo, out = _nhwc_window(x, kH, kW, sH, sW, rH, rW, bool(i[6]) if len(i) > 6 else False)
    return H.extract_patches(x, kH, kW, (sH, sW), p_lo, (rH, rW), out)


# ----------------------------------------------------------------------------- more colour / scatter / space-batch helpers
platform("adjust_hue", lambda ins, i, t, b, d: _color_ok(ins, i, t, b, d) and (bool(t) or len(ins) > 1))(
    lambda ins, i, t, b, d: H.color_convert(ins[0], "adjust_hue", param=t[0] if t else float(ins[1])))
platform("adjust_saturation", lambda ins, i, t, b, d: _color_ok(ins, i, t, b, d) and (bool(t) or len(ins) > 1))(
    lambda ins, i, t, b, d: H.color_convert(ins[0], "adjust_saturation", param=t[0] if t else float(ins[1])))


def _scatter_red_ok(ins, i, t, b, d):
    return (len(ins) >= 3 and _fx(ins[0]) and ins[0].dim() >= 1 and _idx(ins[1]) and isinstance(ins[2], torch.Tensor) and ins[2].is_cuda
            and ins[2].is_floating_point() and ins[2].numel() == ins[1].numel() * (ins[0].numel() // ins[0].shape[0]))


for _kind, _name in (("max", "scatter_max"), ("min", "scatter_min"), ("mul", "scatter_mul")):
    platform(_name, _scatter_red_ok)(lambda ins, i, t, b, d, _k=_kind: H.scatter_reduce_rows(ins[0], ins[1].reshape(-1), ins[2], _k))


def _perm_ok(x):
    return L.use_hip(x) and x.is_contiguous() and x.numel() > 0 and x.element_size() in (1, 2, 4, 8)


def _s2d_ok(ins, i, t, b, d):
    x = ins[0]
    if not (_perm_ok(x) and x.dim() == 4 and i and int(i[0]) > 0):
        return False
    bs, nhwc = int(i[0]), (len(i) > 1 and i[1] == 1)
    h, w = (x.shape[1], x.shape[2]) if nhwc else (x.shape[2], x.shape[3])
    return h % bs == 0 and w % bs == 0


@platform("space_to_depth", _s2d_ok)
def _space_to_depth(ins, i, t, b, d):
    x, bs, nhwc = ins[0], int(i[0]), (len(i) > 1 and i[1] == 1)
    if nhwc:
        n, h, w, c = x.shape      # generic: NHWC -> NCHW -> [n, bs, bs, c, h/bs, w/bs] -> NHWC, composed into one permute
        v = x.reshape(n, h // bs, bs, w // bs, bs, c)
        return H.permute_copy(v, [0, 1, 3, 2, 4, 5]).reshape(n, h // bs, w // bs, bs * bs * c)
    n, c, h, w = x.shape
    return H.permute_copy(x.reshape(n, c, h // bs, bs, w // bs, bs), [0, 3, 5, 1, 2, 4]).reshape(n, c * bs * bs, h // bs, w // bs)


def _d2s_ok(ins, i, t, b, d):
    x = ins[0]
    if not (_perm_ok(x) and x.dim() == 4 and i and int(i[0]) > 0):
        return False
    c = x.shape[3] if (len(i) > 1 and i[1] == 1) else x.shape[1]
    return c % (int(i[0]) ** 2) == 0


@platform("depth_to_space", _d2s_ok)
def _depth_to_space(ins, i, t, b, d):
    x, bs, nhwc = ins[0], int(i[0]), (len(i) > 1 and i[1] == 1)
    if nhwc:
        n, h, w, c = x.shape
        v = x.reshape(n, h, w, bs, bs, c // (bs * bs))
        return H.permute_copy(v, [0, 1, 3, 2, 4, 5]).reshape(n, h * bs, w * bs, c // (bs * bs))
    n, c, h, w = x.shape
    return H.permute_copy(x.reshape(n, bs, bs, c // (bs * bs), h, w), [0, 3, 4, 1, 5, 2]).reshape(n, c // (bs * bs), h * bs, w * bs)


def _s2b_args(name, ins, i):
    if name.endswith("_nd"):
        return [int(v) for v in ins[1].tolist()], ins[2].long().reshape(-1, 2).tolist()
    return [int(i[0])] * 2, ins[1].long().reshape(-1, 2).tolist()


def _reg_space_batch(name, to_batch):
    def ok(ins, i, t, b, d):
        if not (_perm_ok(ins[0]) and len(ins) >= (3 if name.endswith("_nd") else 2) and (name.endswith("_nd") or i)):
            return False
        block, pads = _s2b_args(name, ins, i)
        m = len(block)
        x = ins[0]
        if x.dim() < 1 + m or 2 * m + 1 + (x.dim() - 1 - m) > 8 or len(pads) != m or min(block) < 1:
            return False
        if to_batch:
            return all((x.shape[1 + k] + pads[k][0] + pads[k][1]) % block[k] == 0 for k in range(m))
        nb = 1
        for bl in block:
            nb *= bl
        return x.shape[0] % nb == 0

    def run(ins, i, t, b, d):
        block, pads = _s2b_args(name, ins, i)
        return (H.space_to_batch_nd if to_batch else H.batch_to_space_nd)(ins[0], block, pads)
    platform(name, ok)(run)


for _name, _tb in (("space_to_batch", True), ("space_to_batch_nd", True), ("batch_to_space", False), ("batch_to_space_nd", False)):
    _reg_space_batch(_name, _tb)


def _sru_bi_ok(ins, i, t, b, d):
    if len(ins) < 4 or not all(isinstance(v, torch.Tensor) and L.use_hip(v) and v.dtype == torch.float32 for v in ins[:4]) or len(ins) > 4:
        return False
    x, w, bias, c0 = ins[:4]
    return (x.dim() == 3 and x.shape[2] % 2 == 0 and tuple(w.shape) == (x.shape[2], 3 * x.shape[2]) and bias.numel() == 2 * x.shape[2]
            and tuple(c0.shape) == (x.shape[1], x.shape[2]))


@platform("sru_bi", _sru_bi_ok)
def _sru_bi(ins, i, t, b, d):
    return H.sru_bi_fwd(ins[0], ins[1], ins[2].reshape(-1), ins[3])


@platform("barnes_edge_forces", lambda ins, i, t, b, d: len(ins) >= 4 and _idx(ins[0]) and _idx(ins[1]) and ins[0].dtype == ins[1].dtype
          and L.use_hip(ins[3]) and ins[3].dim() == 2 and 1 <= ins[3].shape[1] <= 8 and ins[3].dtype == torch.float32
          and ins[2].is_cuda and (not i or int(i[0]) == ins[3].shape[0]) and ins[0].numel() >= ins[3].shape[0] + 1)
def _barnes_edge_forces(ins, i, t, b, d):
    return H.barnes_edge_forces(ins[0], ins[1], ins[2], ins[3])


# ------------------------------------------------------------------ recurrent: fused GRU cell + native sequence loop (rnn_seq.hip)
def _gru_usable(ins, i, t, b, d):
    from ...ops import lstm as R
    return len(ins) >= 4 and ins[0].dim() == 3 and R.gru_native_ok(ins[0], ins[1].shape[-1]) and all(
        x is None or x.dtype == ins[0].dtype for x in ins[:5]) and (len(ins) < 5 or ins[4] is None or ins[4].dim() == 1)


@platform("gru", _gru_usable, differentiable=True)
def _gru_platform(ins, i, t, b, d):
    """libnd4j gru time loop (helpers/impl/gru.cpp:123-160: gates = hI x Wh, c = tanh(gates_c * r + x x Wxc + bc)). gruCell keeps the
    generic path: its candidate is ((r * hI) x Wch), a second GEMM between the two halves of the gate math."""
    from ...ops import lstm as R
    return R.gru_sequence(ins[0], ins[1], ins[2], ins[3], ins[4] if len(ins) > 4 else None)


# ----------------------------------------------------------------------------- GEMM-shaped ops on the in-tree MFMA kernels
def _xwb_ok(ins, i, t, b, d):
    from ...ops import linear as LIN
    if len(ins) < 2 or (i and any(int(v) for v in i)):          # transposed-operand variants stay generic
        return False
    return ins[0].dim() == 2 and LIN.usable(ins[0], ins[1]) and (len(ins) < 3 or ins[2] is None or ins[2].numel() == ins[1].shape[1])


@platform("xw_plus_b", _xwb_ok, differentiable=True)
def _xw_plus_b_hip(ins, i, t, b, d):
    from ...ops import linear as LIN
    return LIN.linear(ins[0], ins[1], ins[2] if len(ins) > 2 else None)


def _mm_views(ins, i):
    """(a [.., M, K] view, b [.., N, K] view) of matmul's operands under its transposeX / transposeY flags - no copies."""
    ta = bool(i and int(i[0]))
    tb = bool(i and len(i) > 1 and int(i[1]))
    a, bb = ins[0], ins[1]
    a = a.transpose(-1, -2) if ta else a
    bb = bb if tb else bb.transpose(-1, -2)
    return a, bb


def _mm_any_ok(ins, i, t, b, d):
    """matmul in ANY transpose combination, 2-D or equal-rank 3-D (strided batch), alpha free, beta 0, result not transposed, no autograd:
    the operand views go straight to the layout-aware MFMA GEMM (native/csrc/gemm_trans.h)."""
    from ...ops import gemm as G
    if len(ins) != 2 or not (_fx(ins[0]) and _fx(ins[1])) or ins[0].dim() != ins[1].dim() or ins[0].dim() not in (2, 3):
        return False
    if (i and len(i) > 2 and int(i[2])) or (t and len(t) > 1 and float(t[1]) != 0.0):
        return False
    if torch.is_grad_enabled() and (ins[0].requires_grad or ins[1].requires_grad):
        return False
    a, bb = _mm_views(ins, i)
    return G.supported_ex(a, bb)


@platform("matmul", _mm_any_ok)
def _matmul_any_hip(ins, i, t, b, d):
    from ...ops import gemm as G
    a, bb = _mm_views(ins, i)
    return G.gemm(a, bb, alpha=float(t[0]) if t else 1.0, splits=1)


def _bg_ok(ins, i, t, b, d):
    """batched_gemm (alpha, beta, A_0.., B_0..): every pair through the layout-aware MFMA GEMM; beta must be 0 (no C operands in this
    form of the op), operands 2-D of one shape per side."""
    from ...ops import gemm as G
    n = int(i[8]) if i and len(i) > 8 else (len(ins) - 2) // 2
    if n < 1 or len(ins) < 2 + 2 * n or not all(_fx(x) and x.dim() == 2 for x in ins[2:2 + 2 * n]):
        return False
    if ins[1].numel() and float(ins[1].reshape(-1)[0]) != 0.0:
        return False
    if torch.is_grad_enabled() and any(x.requires_grad for x in ins[2:2 + 2 * n]):
        return False
    ta, tb = bool(i and i[0] in (1, 112)), bool(i and len(i) > 1 and i[1] in (1, 112))
    return all(G.supported_ex(ins[2 + k].t() if ta else ins[2 + k], ins[2 + n + k] if tb else ins[2 + n + k].t()) for k in range(n))


@platform("batched_gemm", _bg_ok)
def _batched_gemm_hip(ins, i, t, b, d):
    from ...ops import gemm as G
    n = int(i[8]) if i and len(i) > 8 else (len(ins) - 2) // 2
    ta, tb = bool(i and i[0] in (1, 112)), bool(i and len(i) > 1 and i[1] in (1, 112))
    al = ins[0].reshape(-1)
    same = len({(x.shape, x.stride()) for x in ins[2:2 + n]}) == 1 and len({(x.shape, x.stride()) for x in ins[2 + n:2 + 2 * n]}) == 1
    A, B = ins[2:2 + n], ins[2 + n:2 + 2 * n]
    if same and al.numel() == 1 and n > 1:
        # equally spaced operands (views of one buffer): ONE strided-batch launch
        da, db = A[1].data_ptr() - A[0].data_ptr(), B[1].data_ptr() - B[0].data_ptr()
        es = A[0].element_size()
        if da > 0 and db > 0 and da % es == 0 and db % es == 0 and \
                all(A[k].data_ptr() - A[0].data_ptr() == k * da and B[k].data_ptr() - B[0].data_ptr() == k * db for k in range(n)):
            sa = torch.as_strided(A[0], (n,) + tuple(A[0].shape), (da // es,) + tuple(A[0].stride()))
            sb = torch.as_strided(B[0], (n,) + tuple(B[0].shape), (db // es,) + tuple(B[0].stride()))
            sa = sa.transpose(-1, -2) if ta else sa
            sb = sb if tb else sb.transpose(-1, -2)
            if G.supported_ex(sa, sb):
                C = G.gemm(sa, sb, alpha=float(al[0]), splits=1)
                return tuple(C[k] for k in range(n))
    outs = []
    for k in range(n):
        a = A[k].t() if ta else A[k]
        bb = B[k] if tb else B[k].t()
        outs.append(G.gemm(a, bb, alpha=float(al[0] if al.numel() == 1 else al[k]), splits=1))
    return tuple(outs)


def _mm_ok(ins, i, t, b, d):
    """matmul(x [.., M, K], w [K, N]) with a plain 2-D right operand, no transposes / alpha / beta: the dense-layer shape."""
    from ...ops import linear as LIN
    if len(ins) != 2 or (i and any(int(v) for v in list(i)[:3])) or (t and (float(t[0]) != 1.0 or (len(t) > 1 and float(t[1]) != 0.0))):
        return False
    return ins[1].dim() == 2 and LIN.usable(ins[0], ins[1])


@platform("matmul", _mm_ok, differentiable=True)
def _matmul_hip(ins, i, t, b, d):
    from ...ops import linear as LIN
    return LIN.linear(ins[0], ins[1], None)


# ----------------------------------------------------------------------------- fused_dense (MI355X graph-fusion target, not a reference op)
# inputs [x, W [K, N], (bias), (residual)]; iArgs [act id (ops/linear.ACT_IDS), has_bias, has_residual].
def _fd_split(ins, i):
    ia = [int(v) for v in (i or [])] + [0, 0, 0]
    act, hb, hr = ia[0], ia[1], ia[2]
    bias = ins[2] if hb else None
    res = ins[2 + hb] if hr else None
    return ins[0], ins[1], bias, res, act


@op("fused_dense")
def _fused_dense_generic(ins, i, t, b, d):
    from ...ops import linear as LIN
    x, w, bias, res, act = _fd_split(ins, i)
    return LIN.fused_dense_reference(x, w, bias, res, act)


def _fd_ok(ins, i, t, b, d):
    from ...ops import linear as LIN
    x, w, bias, res, act = _fd_split(ins, i)
    if not LIN.usable(x, w) or (bias is not None and bias.numel() != w.shape[1]):
        return False
    return res is None or (res.dtype == x.dtype and res.numel() == (x.numel() // w.shape[0]) * w.shape[1])


@platform("fused_dense", _fd_ok, differentiable=True)
def _fused_dense_hip(ins, i, t, b, d):
    from ...ops import linear as LIN
    x, w, bias, res, act = _fd_split(ins, i)
    return LIN.fused_dense(x, w, bias, res, act)


# ----------------------------------------------------------------------------- helpers_misc.hip family
# order statistics, generators, TF-exact bicubic / area resize, adjust_contrast, draw_bounding_boxes, t-SNE steps, word2vec batch
from ...ops import misc_kernels as MK  # noqa: E402


@platform("nth_element", lambda ins, i, t, b, d: len(ins) > 1 and _fx(ins[0]) and ins[0].dim() >= 1 and 0 <= int(ins[1]) < ins[0].shape[-1])
def _nth_element(ins, i, t, b, d):
    return MK.nth_element(ins[0], int(ins[1]), bool(i[0]) if i else False)


def _pct_ok(ins, i, t, b, d):
    return _fx(ins[0]) and bool(t) and ins[0].numel() > 0 and (not i or len(set(k % ins[0].dim() for k in i)) == len(i))


@platform("percentile", _pct_ok)
def _percentile(ins, i, t, b, d):
    x = ins[0]
    interp = int(t[1]) if len(t) > 1 else 2
    keep = bool(t[2]) if len(t) > 2 else False
    dims = sorted(k % x.dim() for k in i) if i else list(range(x.dim()))
    others = [k for k in range(x.dim()) if k not in dims]
    xp = x.permute(*others, *dims).reshape([x.shape[k] for k in others] + [-1]).contiguous()
    r = MK.nth_element(xp, MK.percentile_positions(float(t[0]), xp.shape[-1], interp))
    if keep:
        r = r.reshape([1 if k in dims else x.shape[k] for k in range(x.dim())])
    return r


def _range_args(ins, i, t):
    if ins:
        vals = [x.item() for x in ins]
        dt, dev = ins[0].dtype, ins[0].device
    elif t:
        vals, dt, dev = list(t), torch.float32, None
    else:
        vals, dt, dev = list(i), torch.int64, None
    if len(vals) == 1:
        vals = [0, vals[0], 1]
    elif len(vals) == 2:
        vals = vals + [1]
    return vals, dt, dev


def _range_ok(ins, i, t, b, d):
    return bool(ins) and all(x.numel() == 1 for x in ins) and L.use_hip(ins[0]) and L.dtype_code_all(ins[0]) >= 0


@platform("range", _range_ok)
def _range(ins, i, t, b, d):
    import math
    from .impl_core import dtype_arg
    (s, e, st), dt, dev = _range_args(ins, i, t)
    if st == 0:
        raise ValueError("range: step must not be 0")
    n = max(int(math.ceil((e - s) / st)), 0)
    out_dt = dtype_arg(d, None, 0, dt)
    if L.dtype_code_all(torch.empty(0, dtype=out_dt)) < 0:
        return torch.arange(s, e, st, dtype=out_dt, device=dev)
    return MK.fill_range(s, st, n, out_dt, dev)


@platform("lin_space", lambda ins, i, t, b, d: len(ins) >= 3 and L.use_hip(ins[0]) and int(ins[2]) >= 1)
def _lin_space(ins, i, t, b, d):
    from .impl_core import dtype_arg
    s, e, n = float(ins[0]), float(ins[1]), int(ins[2])
    out_dt = dtype_arg(d, None, 0, torch.float32)
    if L.dtype_code_all(torch.empty(0, dtype=out_dt)) < 0:
        return torch.linspace(s, e, n, dtype=out_dt, device=ins[0].device)
    return MK.fill_range(s, (e - s) / (n - 1) if n > 1 else 0.0, n, out_dt, ins[0].device)


def _bytes_ok(x):
    return L.use_hip(x) and x.element_size() in (1, 2, 4, 8) and not x.is_complex()


@platform("diag", lambda ins, i, t, b, d: _bytes_ok(ins[0]) and ins[0].numel() > 0)
def _diag(ins, i, t, b, d):
    x = ins[0]
    m = MK.diag(x)
    return m.reshape(list(x.shape) * 2) if x.dim() > 1 else m


@platform("diag_part", lambda ins, i, t, b, d: _bytes_ok(ins[0]) and ins[0].numel() > 0 and ins[0].dim() % 2 == 0)
def _diag_part(ins, i, t, b, d):
    x = ins[0]
    return MK.diag_part(x).reshape(x.shape[:x.dim() // 2])


@platform("meshgrid", lambda ins, i, t, b, d: len(ins) >= 1 and all(_bytes_ok(x) and x.numel() > 0 for x in ins))
def _meshgrid(ins, i, t, b, d):
    return MK.meshgrid(ins, xy=(not i or bool(i[0])))


def _resize_tf(mode):
    def f(ins, i, t, b, d):
        x = ins[0]
        if len(ins) > 1:
            h, w = [int(v) for v in ins[1].reshape(-1).tolist()]
        else:
            h, w = int(i[0]), int(i[1])
        return MK.resize_tf(x, h, w, mode, bool(b[0]) if b else False, bool(b[1]) if len(b) > 1 else False)
    return f


platform("resize_bicubic", _resize_ok)(_resize_tf("bicubic"))
platform("resize_area", _resize_ok)(_resize_tf("area"))


@platform("adjust_contrast", lambda ins, i, t, b, d: _fx(ins[0]) and ins[0].dim() >= 3 and ins[0].shape[-1] <= 256 and (bool(t) or len(ins) > 1))
def _adjust_contrast(ins, i, t, b, d):
    return MK.adjust_contrast(ins[0], t[0] if t else float(ins[1]))


platform("adjust_contrast_v2", lambda ins, i, t, b, d: _fx(ins[0]) and ins[0].dim() >= 3 and ins[0].shape[-1] <= 256
         and (bool(t) or len(ins) > 1))(_adjust_contrast)


@platform("draw_bounding_boxes", lambda ins, i, t, b, d: len(ins) >= 3 and _fx(ins[0]) and ins[0].dim() == 4 and L.use_hip(ins[1], ins[2])
          and ins[1].dim() == 3 and ins[1].shape[0] == ins[0].shape[0] and ins[2].dim() == 2 and ins[2].shape[1] >= ins[0].shape[3]
          and ins[2].shape[0] > 0)
def _draw_bb(ins, i, t, b, d):
    C = ins[0].shape[3]
    return MK.draw_boxes(ins[0], ins[1], ins[2][:, :C])


@platform("barnes_gains", lambda ins, i, t, b, d: len(ins) >= 3 and all(L.use_hip(x) and x.dtype == torch.float32 and x.is_contiguous()
                                                                         for x in ins[:3]) and ins[0].shape == ins[1].shape == ins[2].shape)
def _barnes_gains(ins, i, t, b, d):
    return MK.barnes_gains(ins[0], ins[1], ins[2])


@platform("barnes_symmetrized", lambda ins, i, t, b, d: len(ins) >= 3 and L.use_hip(ins[0], ins[1], ins[2]) and ins[2].dtype == torch.float32)
def _barnes_sym(ins, i, t, b, d):
    N = int(i[0]) if i else ins[0].numel() - 1
    return MK.barnes_symmetrize(ins[0], ins[1], ins[2], N)


def _sg_tables(syn1, syn1neg):
    return (syn1 if syn1 is not None and syn1.numel() else None), (syn1neg if syn1neg is not None and syn1neg.numel() else None)


def _sg_ok(ins, i, t, b, d):
    if len(ins) < 11 or not MK.sg_cb_supported(ins[4]):
        return False
    if bool(b[0]) if b else False:            # inference: needs the inference vector
        return len(ins) > 11 and ins[11] is not None and ins[11].is_cuda and ins[11].dtype == torch.float32 and ins[11].is_contiguous()
    return True


@platform("skipgram", _sg_ok)
def _skipgram_hip(ins, i, t, b, d):
    target, ngs, indices, codes, syn0, syn1, syn1neg, expT, negT, alpha, rnd = ins[:11]
    s1, s1n = _sg_tables(syn1, syn1neg)
    infer = ins[11] if (b and b[0]) else None
    B = max(int(target.numel()), 1)
    has_hs = s1 is not None and indices is not None and indices.numel() > 0
    MK.sg_cb_batch(syn0, s1, s1n, negT, target.reshape(-1), ngs.reshape(-1), None, indices.reshape(B, -1) if has_hs else None,
                   codes.reshape(B, -1) if has_hs else None, alpha, rnd, int(i[1]) if len(i) > 1 else 0, True, infer)
    return syn0, syn1, syn1neg


platform("skipgram_inference", _sg_ok)(_skipgram_hip)


def _cbow_ok(ins, i, t, b, d):
    if len(ins) < 12 or not MK.sg_cb_supported(ins[5]):
        return False
    if bool(b[1]) if len(b) > 1 else False:
        return len(ins) > 14 and ins[14] is not None and ins[14].is_cuda and ins[14].dtype == torch.float32 and ins[14].is_contiguous()
    return ins[2] is not None and ins[2].numel() > 0


@platform("cbow", _cbow_ok)
def _cbow_hip(ins, i, t, b, d):
    target, ngs, context, indices, codes, syn0, syn1, syn1neg, expT, negT, alpha, rnd = ins[:12]
    s1, s1n = _sg_tables(syn1, syn1neg)
    infer = ins[14] if (len(b) > 1 and b[1]) else None
    B = max(int(target.numel()), 1)
    has_hs = s1 is not None and indices is not None and indices.numel() > 0
    MK.sg_cb_batch(syn0, s1, s1n, negT, target.reshape(-1), ngs.reshape(-1), context.reshape(B, -1), indices.reshape(B, -1) if has_hs else None,
                   codes.reshape(B, -1) if has_hs else None, alpha, rnd, int(i[1]) if len(i) > 1 else 0, bool(b[0]) if b else True, infer)
    return syn0, syn1, syn1neg


platform("cbow_inference", _cbow_ok)(_cbow_hip)


# ------------------------------------------------------------------ lstmLayer on the fused LSTM sequence kernels (lstm_persist.hip)
# Reference counterpart: the cuDNN platform helper of the same op (platform/cudnn/lstmLayer.cu). Same restrictions in spirit: default
# activations, no cell clipping, no per-example sequence lengths; anything else runs the generic implementation.
def _lstm_layer_args(ins, i, b):
    a = list(i) + [0] * 5
    bb = list(b) + [False] * 8
    has_b, has_seq, has_hi, has_ci, has_p = bb[:5]
    k = 3
    bias = ins[k] if has_b else None
    k += has_b
    seq = ins[k] if has_seq else None
    k += has_seq
    hI = ins[k] if has_hi else None
    k += has_hi
    cI = ins[k] if has_ci else None
    k += has_ci
    Wp = ins[k] if has_p else None
    return dict(fmt=a[0], dir=a[1], gact=a[2] if len(i) > 2 else 2, cact=a[3] if len(i) > 3 else 0, oact=a[4] if len(i) > 4 else 0,
                bias=bias, seq=seq, hI=hI, cI=cI, Wp=Wp, ret=bb[5:8])


def _lstm_layer_ok(ins, i, t, b, d):
    from ...ops import lstm as R
    if len(ins) < 3 or ins[0].dim() != 3:
        return False
    a = _lstm_layer_args(ins, i, b)
    if a["gact"] != 2 or a["cact"] != 0 or a["oact"] != 0 or (t and float(t[0]) != 0.0) or a["seq"] is not None or a["fmt"] not in (0, 1, 2, 3):
        return False
    x = ins[0]
    B = x.shape[1] if a["fmt"] in (0, 3) else x.shape[0]
    n = ins[2].shape[-2]
    ts = [v for v in (ins[1], ins[2], a["bias"], a["hI"], a["cI"], a["Wp"]) if v is not None]
    if not all(v.is_cuda and v.dtype == x.dtype for v in ts) or not R.lstm_layer_fused_ok(x, n, B):
        return False
    grad = torch.is_grad_enabled() and any(v.requires_grad for v in [x] + ts)
    if grad and R._PERSIST_MODE != "steps" and (a["ret"][2] or any(v is not None and v.requires_grad for v in (a["hI"], a["cI"]))):
        return False                                  # cell-state gradients come from the per-step launches only
    return True


@platform("lstmLayer", _lstm_layer_ok, differentiable=True)
def _lstm_layer_hip(ins, i, t, b, d):
    from ...ops import lstm as R
    a = _lstm_layer_args(ins, i, b)
    x, Wx, Wr = ins[0], ins[1], ins[2]
    fmt, dm = a["fmt"], a["dir"]
    xs = {0: lambda: x, 1: lambda: x.permute(1, 0, 2), 2: lambda: x.permute(2, 0, 1), 3: lambda: x}[fmt]()
    pick = lambda v, k: None if v is None else v[k]
    if dm < 2:
        hs, hl, cl = R.lstm_layer_direction(xs, Wx, Wr, a["bias"], a["hI"], a["cI"], a["Wp"], dm == 1)
    else:
        f = R.lstm_layer_direction(xs, Wx[0], Wr[0], pick(a["bias"], 0), pick(a["hI"], 0), pick(a["cI"], 0), pick(a["Wp"], 0), False)
        r = R.lstm_layer_direction(xs, Wx[1], Wr[1], pick(a["bias"], 1), pick(a["hI"], 1), pick(a["cI"], 1), pick(a["Wp"], 1), True)
        hs = f[0] + r[0] if dm == 2 else (torch.cat([f[0], r[0]], -1) if dm == 3 else torch.stack([f[0], r[0]], 1))
        hl, cl = torch.stack([f[1], r[1]]), torch.stack([f[2], r[2]])
    outs = []
    if a["ret"][0]:
        outs.append(hs if dm == 4 else {0: lambda: hs, 1: lambda: hs.permute(1, 0, 2), 2: lambda: hs.permute(1, 2, 0), 3: lambda: hs}[fmt]())
    if a["ret"][1]:
        outs.append(hl)
    if a["ret"][2]:
        outs.append(cl)
    return tuple(outs)
