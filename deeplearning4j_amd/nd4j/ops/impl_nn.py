"""Custom-op implementations, part 2: nn, convo, pooling, recurrent, loss, updaters, random.

Reference contracts: libnd4j ops/declarable/headers/{nn,convo,recurrent,loss,updaters,random}.h and
generic/nn/**; platform helpers in the reference are cuDNN (platform/cudnn/*.cu) — here the
``@platform`` entries route to the hand-written gfx950 kernels (implicit-GEMM MFMA conv, two-stage
BatchNorm, NHWC pooling, fused softmax, fused LSTM cell) when their layout/dtype preconditions
hold (isUsable), else the generic ATen composite runs — same dispatch rule as
DeclarableOp.cpp:692-739.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from ...native import lib as L
from .custom import _IMPL, op, platform
from .impl_core import _unbroadcast, bp_of, dtype_arg, first


# ----------------------------------------------------------------------------- softmax & friends
def _sm_dim(i, x):
    return (int(i[0]) if i else -1) % max(x.dim(), 1)


op("softmax")(lambda ins, i, t, b, d: torch.softmax(ins[0], _sm_dim(i, ins[0])))
op("log_softmax")(lambda ins, i, t, b, d: torch.log_softmax(ins[0], _sm_dim(i, ins[0])))


@platform("softmax", lambda ins, i, t, b, d: L.use_hip(ins[0]) and ins[0].dim() == 2 and _sm_dim(i, ins[0]) == 1 and
          ins[0].dtype in (torch.float32, torch.bfloat16) and ins[0].is_contiguous())
def _softmax_hip(ins, i, t, b, d):
    from ...ops import softmax as SM
    return SM.softmax(ins[0])


@op("softmax_bp")
def _softmax_bp(ins, i, t, b, d):
    x, g = ins[0], ins[1]
    dim = _sm_dim(i, x)
    y = torch.softmax(x, dim)
    return y * (g - (g * y).sum(dim, keepdim=True))


@op("log_softmax_bp")
def _log_softmax_bp(ins, i, t, b, d):
    x, g = ins[0], ins[1]
    dim = _sm_dim(i, x)
    return g - torch.softmax(x, dim) * g.sum(dim, keepdim=True)


# ----------------------------------------------------------------------------- bias / dense
def _bias_shape(x, nchw):
    if x.dim() <= 2 or not nchw:
        return [1] * (x.dim() - 1) + [-1]
    return [1, -1] + [1] * (x.dim() - 2)


op("biasadd")(lambda ins, i, t, b, d: ins[0] + ins[1].reshape(_bias_shape(ins[0], bool(b[0]) if b else (bool(i[0]) if i else False))))


@op("biasadd_bp")
def _biasadd_bp(ins, i, t, b, d):
    x, bias, g = ins
    nchw = bool(b[0]) if b else (bool(i[0]) if i else False)
    ch = 1 if (nchw and x.dim() > 2) else x.dim() - 1
    return g.clone(), g.sum([k for k in range(x.dim()) if k != ch])


def _xwb(ins, i):
    x, w, bias = ins[0], ins[1], ins[2]
    if i and i[0]:
        w = w.t()
    return x @ w + bias


op("xw_plus_b")(lambda ins, i, t, b, d: _xwb(ins, i))
op("relu_layer")(lambda ins, i, t, b, d: F.relu(_xwb(ins, i)))


@op("xw_plus_b_bp")
def _xw_plus_b_bp(ins, i, t, b, d):
    x, w, bias, g = ins
    wt = w.t() if (i and i[0]) else w
    dx = g @ wt.t()
    dw = x.t() @ g
    return dx, (dw.t() if (i and i[0]) else dw), g.sum(0)


op("embedding_lookup")(lambda ins, i, t, b, d: F.embedding(ins[1].long(), ins[0]))
op("apply_sgd")(lambda ins, i, t, b, d: ins[0] - (t[0] if t else float(ins[2])) * ins[1])


# ----------------------------------------------------------------------------- normalisation
def _bn_shape(x, axes):
    shape = [1] * x.dim()
    for a in axes:
        shape[a % x.dim()] = x.shape[a % x.dim()]
    return shape


@op("batchnorm")
def _batchnorm(ins, i, t, b, d):
    x, mean, var = ins[0], ins[1], ins[2]
    scale, offset = bool(i[0]) if i else False, bool(i[1]) if len(i) > 1 else False
    axes = list(i[2:]) if len(i) > 2 else [x.dim() - 1]
    eps = t[0] if t else 1e-5
    sh = _bn_shape(x, axes)
    k = 3
    y = (x - mean.reshape(sh)) * torch.rsqrt(var.reshape(sh) + eps)
    if scale:
        y = y * ins[k].reshape(sh)
        k += 1
    if offset:
        y = y + ins[k].reshape(sh)
    return y


@platform("batchnorm", lambda ins, i, t, b, d: L.use_hip(ins[0]) and ins[0].dim() == 4 and len(i) == 3 and int(i[2]) % 4 == 1
          and i[0] and i[1] and ins[0].dtype in (torch.float32, torch.bfloat16)
          and ins[0].is_contiguous(memory_format=torch.channels_last))
def _batchnorm_hip(ins, i, t, b, d):
    from ...ops import batchnorm as BN
    x, mean, var, gamma, beta = ins[:5]
    scale, shift = BN.infer_coeffs(gamma.float(), beta.float(), mean.float(), var.float(), t[0] if t else 1e-5, False)
    N, C, H, W = x.shape
    x2d = x.permute(0, 2, 3, 1).reshape(-1, C)          # channels_last memory -> a free [M, C] view
    return BN.apply(x2d, scale, shift, relu=False).reshape(N, H, W, C).permute(0, 3, 1, 2)


@op("batchnorm_bp")
def _batchnorm_bp(ins, i, t, b, d):
    *fw, g = ins
    leaves = [v.detach().clone().requires_grad_(True) for v in fw]
    with torch.enable_grad():
        y = _batchnorm(leaves, i, t, b, d)
    gs = torch.autograd.grad(y, leaves, g, allow_unused=True)
    return tuple(gg if gg is not None else torch.zeros_like(v) for gg, v in zip(gs, leaves))


@op("fused_batch_norm")
def _fused_batch_norm(ins, i, t, b, d):
    x, scale, offset = ins[0], ins[1], ins[2]
    nchw = bool(i[0]) if i else False
    training = bool(i[1]) if len(i) > 1 else True
    eps = t[0] if t else 1e-3
    ch = 1 if nchw else x.dim() - 1
    red = [k for k in range(x.dim()) if k != ch]
    sh = [1] * x.dim()
    sh[ch] = -1
    if training:
        m, v = x.mean(red), x.var(red, unbiased=False)
    else:
        m, v = ins[3], ins[4]
    y = (x - m.reshape(sh)) * torch.rsqrt(v.reshape(sh) + eps) * scale.reshape(sh) + offset.reshape(sh)
    return y, m, v


@op("layer_norm")
def _layer_norm(ins, i, t, b, d):
    x, gain = ins[0], ins[1]
    bias = ins[2] if len(ins) > 2 else None
    axes = [a % x.dim() for a in i] if i else [x.dim() - 1]
    nchw = bool(b[0]) if b else False
    m = x.mean(axes, keepdim=True)
    v = x.var(axes, unbiased=False, keepdim=True)
    y = (x - m) * torch.rsqrt(v + (float(t[0]) if t else 1e-5))
    sh = [1, -1] + [1] * (x.dim() - 2) if (nchw and x.dim() > 2) else [1] * (x.dim() - gain.dim()) + list(gain.shape)
    y = y * gain.reshape(sh)
    if bias is not None:
        y = y + bias.reshape(sh)
    return y


op("layer_norm_bp")(bp_of("layer_norm"))


@op("lrn")
def _lrn(ins, i, t, b, d):
    x = ins[0]
    bias, alpha, beta = (list(t) + [1.0, 1.0, 0.5])[:3] if t else (1.0, 1.0, 0.5)
    depth = int(i[0]) if i else 5
    sq = (x * x).unsqueeze(1)
    pad = F.pad(sq, [0, 0, 0, 0, depth, depth])
    s = sum(pad[:, :, k:k + x.shape[1]] for k in range(2 * depth + 1)).squeeze(1)
    return x / torch.pow(bias + alpha * s, beta)


op("lrn_bp")(bp_of("lrn", 1))


# ----------------------------------------------------------------------------- attention
def _sdpa_v2(ins, i, t, b, d):
    q, v = ins[0], ins[1]
    k = ins[2] if len(ins) > 2 and ins[2] is not None and ins[2].dim() == q.dim() else v
    masks = [m for m in ins[3:] if m is not None] if len(ins) > 3 else ([ins[2]] if len(ins) > 2 and ins[2] is not None and ins[2].dim() != q.dim() else [])
    scale = t[0] if t and t[0] != 0 else 1.0
    dropout = t[1] if len(t) > 1 else 0.0
    causal = bool(b[0]) if b else False
    training = bool(b[1]) if len(b) > 1 else False
    # iArgs (extension of this framework): [withWeights (default 1), singleMaskIsKeyMask (default 0)]. Without the weights
    # outputs nothing forces the [.., Tq, Tk] matrices to exist, so the op goes through ops/attention.sdpa — on the GPU
    # the fused flash kernels (forward and backward) for bf16, head size 64 / 128.
    with_weights = bool(i[0]) if i else True
    if len(masks) == 1 and len(i) > 1 and i[1]:
        masks = [None, masks[0]]
    if not with_weights and not (training and dropout > 0) and q.dim() == 4 and (not masks or masks[0] is None):
        from ...ops import attention as ATT
        vmask = masks[1] if len(masks) > 1 else None
        return ATT.sdpa(q, k, v, None if vmask is None else vmask.to(torch.float32), scale, causal)
    logits = torch.matmul(q, k.transpose(-1, -2)) * scale
    Tq, Tv = logits.shape[-2], logits.shape[-1]
    neg = torch.finfo(logits.dtype).min if logits.is_floating_point() else -1e9
    if causal:
        cm = torch.ones(Tq, Tv, dtype=torch.bool, device=logits.device).tril()
        logits = torch.where(cm, logits, torch.full_like(logits, neg))
    if len(masks) >= 1 and masks[0] is not None:          # qMask [b, Tq]
        qm = masks[0].bool().reshape(list(masks[0].shape[:-1]) + [1] * (logits.dim() - masks[0].dim() - 1) + [Tq, 1])
        logits = torch.where(qm, logits, torch.full_like(logits, neg))
    if len(masks) >= 2 and masks[1] is not None:          # vMask [b, Tv]
        vm = masks[1].bool().reshape(list(masks[1].shape[:-1]) + [1] * (logits.dim() - masks[1].dim() - 1) + [1, Tv])
        logits = torch.where(vm, logits, torch.full_like(logits, neg))
    scores = torch.softmax(logits, -1)
    outs_scores = scores
    if training and dropout > 0:
        mask = (torch.rand_like(scores) >= dropout).to(scores.dtype)
        scores = scores * mask / (1.0 - dropout)
        return (torch.matmul(scores, v), outs_scores, logits, mask) if with_weights else torch.matmul(scores, v)
    return (torch.matmul(scores, v), outs_scores, logits) if with_weights else torch.matmul(scores, v)


op("dot_product_attention_v2")(_sdpa_v2)
op("dot_product_attention_v2_bp")(lambda ins, i, t, b, d: _sdpa_v2_bp(ins, i, t, b, d))


def _sdpa_v2_bp(ins, i, t, b, d):
    n = 3 if (len(ins) > 3 and ins[2] is not None and ins[2].dim() == ins[0].dim()) else 2
    fw, g = list(ins[:n]), ins[n] if len(ins) > n else ins[-1]
    leaves = [x.detach().clone().requires_grad_(True) for x in fw]
    with torch.enable_grad():
        y = first(_sdpa_v2(leaves + list(ins[n + 1:]), i, t, [b[0] if b else False, False], d))      # (a bare tensor without the weights outputs)
    return torch.autograd.grad(y, leaves, g)


@op("dot_product_attention")
def _dpa(ins, i, t, b, d):
    """Legacy layout: q [b, dk, Tq], k [b, dk, Tk], v [b, dv, Tk]; iArgs [normalize, withWeights]."""
    q, k, v = ins[0], ins[1], ins[2]
    mask = ins[3] if len(ins) > 3 else None
    normalize, with_w = (bool(i[0]) if i else False), (bool(i[1]) if len(i) > 1 else False)
    w = torch.matmul(k.transpose(-1, -2), q)                # [b, Tk, Tq]
    if normalize:
        w = w / math.sqrt(k.shape[-2])
    if mask is not None:
        m = mask.reshape(list(mask.shape) + [1]) if mask.dim() == w.dim() - 1 else mask
        w = w + (m.to(w.dtype) - 1) * 1e9
    w = torch.softmax(w, -2)
    out = torch.matmul(v, w)                                # [b, dv, Tq]
    return (out, w) if with_w else out


op("dot_product_attention_bp")(lambda ins, i, t, b, d: _dpa_bp(ins, i, t, b, d))


def _dpa_bp(ins, i, t, b, d):
    q, k, v, g = ins[0], ins[1], ins[2], ins[3]
    rest = list(ins[4:])
    leaves = [x.detach().clone().requires_grad_(True) for x in (q, k, v)]
    with torch.enable_grad():
        y = first(_dpa(leaves + rest, [i[0] if i else 0, 0], t, b, d))
    return torch.autograd.grad(y, leaves, g)


@op("multi_head_dot_product_attention")
def _mhdpa(ins, i, t, b, d):
    """q [b, nIn, Tq], k,v [b, nIn, Tk]; Wq,Wk,Wv [nHeads, projSize, nIn]; Wo [nHeads*projSize, nOut]."""
    q, k, v, Wq, Wk, Wv, Wo = ins[:7]
    mask = ins[7] if len(ins) > 7 else None
    normalize, with_w = (bool(i[0]) if i else False), (bool(i[1]) if len(i) > 1 else False)
    pq = torch.einsum("hpi,bit->bhpt", Wq, q)
    pk = torch.einsum("hpi,bit->bhpt", Wk, k)
    pv = torch.einsum("hpi,bit->bhpt", Wv, v)
    w = torch.matmul(pk.transpose(-1, -2), pq)              # [b,h,Tk,Tq]
    if normalize:
        w = w / math.sqrt(pk.shape[-2])
    if mask is not None:
        w = w + (mask.reshape(mask.shape[0], 1, -1, 1).to(w.dtype) - 1) * 1e9
    w = torch.softmax(w, -2)
    att = torch.matmul(pv, w)                               # [b,h,p,Tq]
    bsz, h, p, Tq = att.shape
    out = torch.einsum("bft,fo->bot", att.reshape(bsz, h * p, Tq), Wo)
    return (out, w) if with_w else out


op("multi_head_dot_product_attention_bp")(lambda ins, i, t, b, d: _mh_bp(ins, i, t, b, d))


def _mh_bp(ins, i, t, b, d):
    fw, g = list(ins[:7]), ins[7]
    rest = list(ins[8:])
    leaves = [x.detach().clone().requires_grad_(True) for x in fw]
    with torch.enable_grad():
        y = first(_mhdpa(leaves + rest, [i[0] if i else 0, 0], t, b, d))
    return torch.autograd.grad(y, leaves, g)


# ----------------------------------------------------------------------------- convolution family
def _same_pad(n, k, s, dl):
    out = -(-n // s)
    eff = (k - 1) * dl + 1
    tot = max((out - 1) * s + eff - n, 0)
    return tot // 2, tot - tot // 2


def _w_to_oihw(w, wfmt, nd=2):
    if wfmt == 0:      # [k..., iC, oC]
        return w.permute(nd + 1, nd, *range(nd))
    if wfmt == 1:      # [oC, iC, k...]
        return w
    return w.permute(0, nd + 1, *range(1, nd + 1))      # [oC, k..., iC]


def _w_from_oihw(g, wfmt, nd=2):
    if wfmt == 0:
        return g.permute(*range(2, 2 + nd), 1, 0)
    if wfmt == 1:
        return g
    return g.permute(0, *range(2, 2 + nd), 1)


def _conv_nd(x, w, bias, k, s, p, dl, same, nhwc, wfmt, nd, groups=1):
    if nhwc:
        x = x.permute(0, nd + 1, *range(1, nd + 1))
    W = _w_to_oihw(w, wfmt, nd)
    if same:
        pads = [_same_pad(x.shape[2 + a], W.shape[2 + a], s[a], dl[a]) for a in range(nd)]
        if any(lo != hi for lo, hi in pads):
            flat = []
            for lo, hi in reversed(pads):
                flat += [lo, hi]
            x = F.pad(x, flat)
            p = [0] * nd
        else:
            p = [lo for lo, _ in pads]
    fn = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[nd]
    y = fn(x, W.to(x.dtype), None if bias is None else bias.reshape(-1).to(x.dtype), tuple(s), tuple(p), tuple(dl), groups)
    return y.permute(0, *range(2, 2 + nd), 1) if nhwc else y


def _conv2d_args(i):
    a = list(i) + [0] * (11 - len(i))
    return dict(k=a[0:2], s=a[2:4], p=a[4:6], dl=[max(a[6], 1), max(a[7], 1)], same=bool(a[8]), nhwc=bool(a[9]),
                wfmt=a[10] if len(i) > 10 else 0)


@op("conv2d")
def _conv2d(ins, i, t, b, d):
    c = _conv2d_args(i)
    return _conv_nd(ins[0], ins[1], ins[2] if len(ins) > 2 else None, c["k"], c["s"], c["p"], c["dl"], c["same"], c["nhwc"], c["wfmt"], 2)


def _conv2d_hip_ok(ins, i, t, b, d):
    if not L.use_hip(ins[0], ins[1]) or ins[0].dtype != torch.bfloat16:
        return False
    c = _conv2d_args(i)
    if c["same"] or c["wfmt"] != 1 or c["nhwc"]:
        return False
    from ...ops import conv_igemm
    x = ins[0]
    w = ins[1].contiguous(memory_format=torch.channels_last)
    return x.is_contiguous(memory_format=torch.channels_last) and conv_igemm.supported(x, w, c["s"], c["p"], c["dl"])


@platform("conv2d", _conv2d_hip_ok)
def _conv2d_hip(ins, i, t, b, d):
    from ...ops import conv_igemm
    c = _conv2d_args(i)
    w = ins[1].contiguous(memory_format=torch.channels_last)
    return conv_igemm.forward(ins[0], w, ins[2] if len(ins) > 2 else None, c["s"], c["p"], c["dl"])[0]


@op("conv2d_bp")
def _conv2d_bp(ins, i, t, b, d):
    """gradI, gradW[, gradB] straight from the convolution backward kernels (own implicit-GEMM kernels on the GPU, ATen on the CPU):
    no forward re-computation (reference: generic/nn/convo/conv2d.cpp conv2d_bp -> ConvolutionUtils::conv2dBP)."""
    x, w = ins[0], ins[1]
    bias = ins[2] if len(ins) > 3 else None
    g = ins[-1]
    c = _conv2d_args(i)
    nd = 2
    xs = x.permute(0, 3, 1, 2) if c["nhwc"] else x
    gs = g.permute(0, 3, 1, 2) if c["nhwc"] else g
    W = _w_to_oihw(w, c["wfmt"], nd).to(xs.dtype)
    p = list(c["p"])
    padded = None
    if c["same"]:
        pads = [_same_pad(xs.shape[2 + a], W.shape[2 + a], c["s"][a], c["dl"][a]) for a in range(nd)]
        if any(lo != hi for lo, hi in pads):
            flat = []
            for lo, hi in reversed(pads):
                flat += [lo, hi]
            padded = pads
            xs = F.pad(xs, flat)
            p = [0] * nd
        else:
            p = [lo for lo, _ in pads]
    from ...ops import conv as CONV
    if xs.is_cuda:
        xs = xs.contiguous(memory_format=torch.channels_last)
        gs = gs.contiguous(memory_format=torch.channels_last).to(xs.dtype)
        Wc = W.contiguous(memory_format=torch.channels_last)
    else:
        xs, gs, Wc = xs.contiguous(), gs.contiguous().to(xs.dtype), W.contiguous()
    dx, dW, db = CONV.conv2d_backward(gs, xs, Wc, c["s"], p, c["dl"], True, True, bias is not None)
    if padded is not None:
        (t0, t1), (l0, l1) = padded
        dx = dx[:, :, t0:dx.shape[2] - t1, l0:dx.shape[3] - l1]
    if c["nhwc"]:
        dx = dx.permute(0, 2, 3, 1)
    dw = _w_from_oihw(dW.to(w.dtype), c["wfmt"], nd)
    out = [dx.to(x.dtype), dw]
    if bias is not None:
        out.append(db.to(bias.dtype).reshape(bias.shape))
    return tuple(out)


@op("conv2d_input_bp")
def _conv2d_input_bp(ins, i, t, b, d):
    shape, w, g = [int(v) for v in ins[0].tolist()], ins[1], ins[2]
    x = torch.zeros(shape, dtype=g.dtype, device=g.device, requires_grad=True)
    with torch.enable_grad():
        y = _conv2d([x, w], i, t, b, d)
    return torch.autograd.grad(y, x, g)[0]


@op("conv1d")
def _conv1d(ins, i, t, b, d):
    a = list(i) + [0] * (7 - len(i))
    k, s, p, dl, mode, nwc = a[0], a[1], a[2], max(a[3], 1), a[4], bool(a[5])
    wfmt = a[6] if len(i) > 6 else 0
    x = ins[0]
    if mode == 2:       # causal
        xx = x.permute(0, 2, 1) if nwc else x
        xx = F.pad(xx, [(k - 1) * dl, 0])
        y = _conv_nd(xx, ins[1], ins[2] if len(ins) > 2 else None, [k], [s], [0], [dl], False, False, wfmt, 1)
        return y.permute(0, 2, 1) if nwc else y
    return _conv_nd(x, ins[1], ins[2] if len(ins) > 2 else None, [k], [s], [p], [dl], mode == 1, nwc, wfmt, 1)


@op("conv3dnew")
def _conv3d(ins, i, t, b, d):
    a = list(i) + [0] * (15 - len(i))
    k, s, p, dl = a[0:3], a[3:6], a[6:9], [max(v, 1) for v in a[9:12]]
    return _conv_nd(ins[0], ins[1], ins[2] if len(ins) > 2 else None, k, s, p, dl, bool(a[12]), a[13] == 1,
                    a[14] if len(i) > 14 else 0, 3)


def _generic_bp(fwd):
    def f(ins, i, t, b, d):
        *fw, g = ins
        leaves = [x.detach().clone().requires_grad_(True) if x.is_floating_point() else x for x in fw]
        with torch.enable_grad():
            y = first(fwd(leaves, i, t, b, d))
        req = [x for x in leaves if x.requires_grad]
        return torch.autograd.grad(y, req, g, allow_unused=True)
    return f


op("conv1d_bp")(_generic_bp(_conv1d))
op("conv3dnew_bp")(_generic_bp(_conv3d))


@op("depthwise_conv2d")
def _depthwise(ins, i, t, b, d):
    c = _conv2d_args(i)
    x, w = ins[0], ins[1]
    W = _w_to_oihw(w, c["wfmt"], 2) if c["wfmt"] != 0 else w.permute(2, 3, 0, 1)      # [iC, mC, kH, kW]
    if c["wfmt"] == 1:
        W = W.permute(1, 0, 2, 3)      # [mC, iC, kH, kW] -> [iC, mC, ...]
    iC, mC = W.shape[0], W.shape[1]
    Wg = W.reshape(iC * mC, 1, W.shape[2], W.shape[3])
    return _conv_nd(x, Wg, ins[2] if len(ins) > 2 else None, c["k"], c["s"], c["p"], c["dl"], c["same"], c["nhwc"], 1, 2, groups=iC)


op("depthwise_conv2d_bp")(_generic_bp(_depthwise))


@op("sconv2d")
def _sconv2d(ins, i, t, b, d):
    x, wd = ins[0], ins[1]
    wp = ins[2] if len(ins) > 2 and ins[2] is not None and ins[2].dim() == 4 else None
    bias = ins[3] if len(ins) > 3 else (ins[2] if len(ins) > 2 and wp is None else None)
    c = _conv2d_args(i)
    y = _depthwise([x, wd], i, t, b, d)
    if wp is not None:
        y = _conv_nd(y, wp, bias, [1, 1], [1, 1], [0, 0], [1, 1], False, c["nhwc"], c["wfmt"], 2)
    elif bias is not None:
        y = y + bias.reshape([1, 1, 1, -1] if c["nhwc"] else [1, -1, 1, 1])
    return y


op("sconv2d_bp")(_generic_bp(_sconv2d))
op("pointwise_conv2d")(lambda ins, i, t, b, d: _conv_nd(ins[0], ins[1], ins[2] if len(ins) > 2 else None, [1, 1], [1, 1], [0, 0], [1, 1], False,
                                                        bool(i[0]) if i else False, i[1] if len(i) > 1 else 0, 2))


@op("deconv2d")
def _deconv2d(ins, i, t, b, d):
    c = _conv2d_args(i)
    x, w = ins[0], ins[1]
    bias = ins[2] if len(ins) > 2 else None
    if c["nhwc"]:
        x = x.permute(0, 3, 1, 2)
    # deconv weights: wfmt 0 [kH,kW,oC,iC], 1 [iC,oC,kH,kW], 2 [iC,kH,kW,oC] -> torch wants [iC, oC, kH, kW]
    W = {0: lambda: w.permute(3, 2, 0, 1), 1: lambda: w, 2: lambda: w.permute(0, 3, 1, 2)}[c["wfmt"]]()
    p = c["p"]
    opad = [0, 0]
    if c["same"]:
        p = []
        for a in range(2):
            eff = (W.shape[2 + a] - 1) * c["dl"][a] + 1
            tot = max(eff - c["s"][a], 0)
            p.append(tot // 2)
            opad[a] = tot - 2 * (tot // 2)
    y = F.conv_transpose2d(x, W.to(x.dtype), None if bias is None else bias.reshape(-1), tuple(c["s"]), tuple(p), tuple(opad), 1, tuple(c["dl"]))
    if c["same"]:
        y = y[:, :, :x.shape[2] * c["s"][0], :x.shape[3] * c["s"][1]]
    return y.permute(0, 2, 3, 1) if c["nhwc"] else y


op("deconv2d_bp")(_generic_bp(_deconv2d))


@op("deconv2d_tf")
def _deconv2d_tf(ins, i, t, b, d):
    shape, w, g = [int(v) for v in ins[0].tolist()], ins[1], ins[2]
    return _conv2d_input_bp([ins[0], w, g], i, t, b, d)


@op("deconv3d")
def _deconv3d(ins, i, t, b, d):
    a = list(i) + [0] * (15 - len(i))
    k, s, p, dl = a[0:3], a[3:6], a[6:9], [max(v, 1) for v in a[9:12]]
    ndhwc = a[13] == 1
    x, w = ins[0], ins[1]
    if ndhwc:
        x = x.permute(0, 4, 1, 2, 3)
    wf = a[14] if len(i) > 14 else 0
    W = {0: lambda: w.permute(4, 3, 0, 1, 2), 1: lambda: w, 2: lambda: w.permute(0, 4, 1, 2, 3)}[wf]()
    y = F.conv_transpose3d(x, W.to(x.dtype), None if len(ins) < 3 else ins[2].reshape(-1), tuple(s), tuple(p), 0, 1, tuple(dl))
    return y.permute(0, 2, 3, 4, 1) if ndhwc else y


op("deconv3d_bp")(_generic_bp(_deconv3d))


@op("im2col")
def _im2col(ins, i, t, b, d):
    x = ins[0]
    kH, kW, sH, sW, pH, pW, dH, dW = i[:8]
    same = bool(i[8]) if len(i) > 8 else False
    if same:
        ph, pw = _same_pad(x.shape[2], kH, sH, dH), _same_pad(x.shape[3], kW, sW, dW)
        x = F.pad(x, [pw[0], pw[1], ph[0], ph[1]], value=t[0] if t else 0.0)
        pH = pW = 0
    cols = F.unfold(x, (kH, kW), (dH, dW), (pH, pW), (sH, sW))
    oH = (x.shape[2] + 2 * pH - (kH - 1) * dH - 1) // sH + 1
    oW = (x.shape[3] + 2 * pW - (kW - 1) * dW - 1) // sW + 1
    return cols.reshape(x.shape[0], x.shape[1], kH, kW, oH, oW)


@op("col2im")
def _col2im(ins, i, t, b, d):
    c = ins[0]
    sH, sW, pH, pW, iH, iW, dH, dW = i[:8]
    n, ch, kH, kW, oH, oW = c.shape
    return F.fold(c.reshape(n, ch * kH * kW, oH * oW), (iH, iW), (kH, kW), (dH, dW), (pH, pW), (sH, sW))


@op("im2col_bp")
def _im2col_bp(ins, i, t, b, d):
    x, g = ins[0], ins[1]
    xr = x.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        y = _im2col([xr], i, t, b, d)
    return torch.autograd.grad(y, xr, g)[0]


def _upsample(nd):
    def f(ins, i, t, b, d):
        x = ins[0]
        factors = list(i[:nd])
        nchw = bool(i[nd]) if len(i) > nd else True
        if not nchw:
            x = x.permute(0, nd + 1, *range(1, nd + 1))
        y = x
        for a, fct in enumerate(factors):
            y = y.repeat_interleave(int(fct), dim=2 + a)
        return y if nchw else y.permute(0, *range(2, 2 + nd), 1)
    return f


op("upsampling2d")(_upsample(2))
op("upsampling3d")(_upsample(3))


@op("upsampling2d_bp")
def _up2_bp(ins, i, t, b, d):
    x, g = ins[0], ins[1]
    nchw = bool(i[0]) if i else True
    if not nchw:
        x, g = x.permute(0, 3, 1, 2), g.permute(0, 3, 1, 2)
    fh, fw = g.shape[2] // x.shape[2], g.shape[3] // x.shape[3]
    r = g.reshape(x.shape[0], x.shape[1], x.shape[2], fh, x.shape[3], fw).sum((3, 5))
    return r if nchw else r.permute(0, 2, 3, 1)


@op("upsampling3d_bp")
def _up3_bp(ins, i, t, b, d):
    x, g = ins[0], ins[1]
    nchw = bool(i[0]) if i else True
    if not nchw:
        x, g = x.permute(0, 4, 1, 2, 3), g.permute(0, 4, 1, 2, 3)
    f = [g.shape[2 + a] // x.shape[2 + a] for a in range(3)]
    r = g.reshape(x.shape[0], x.shape[1], x.shape[2], f[0], x.shape[3], f[1], x.shape[4], f[2]).sum((3, 5, 7))
    return r if nchw else r.permute(0, 2, 3, 4, 1)


@op("dilation2d")
def _dilation2d(ins, i, t, b, d):
    x, w = ins[0], ins[1]                       # x NHWC, w [kH,kW,C]
    same = bool(i[0])
    rates = list(i[1:5]) if len(i) >= 9 else [1, 1, 1, 1]
    strides = list(i[5:9]) if len(i) >= 9 else [1, 1, 1, 1]
    rH, rW, sH, sW = rates[1], rates[2], strides[1], strides[2]
    kH, kW, C = w.shape
    xc = x.permute(0, 3, 1, 2)
    if same:
        ph, pw = _same_pad(xc.shape[2], kH, sH, rH), _same_pad(xc.shape[3], kW, sW, rW)
        xc = F.pad(xc, [pw[0], pw[1], ph[0], ph[1]], value=float("-inf"))
    cols = F.unfold(xc, (kH, kW), (rH, rW), 0, (sH, sW))
    oH = (xc.shape[2] - (kH - 1) * rH - 1) // sH + 1
    oW = (xc.shape[3] - (kW - 1) * rW - 1) // sW + 1
    cols = cols.reshape(x.shape[0], C, kH * kW, oH, oW) + w.permute(2, 0, 1).reshape(1, C, kH * kW, 1, 1)
    return cols.amax(2).permute(0, 2, 3, 1)


@op("ismax")
def _ismax(ins, i, t, b, d):
    x = ins[0]
    if not i or x.dim() == 1:
        r = torch.zeros_like(x.reshape(-1))
        r[x.reshape(-1).argmax()] = 1
        return r.reshape(x.shape)
    dim = i[0] % x.dim()
    return torch.zeros_like(x).scatter_(dim, x.argmax(dim, keepdim=True), 1)


# ----------------------------------------------------------------------------- pooling
def _pool_args(i):
    a = list(i) + [0] * (11 - len(i))
    return dict(k=a[0:2], s=a[2:4], p=a[4:6], dl=[max(a[6], 1), max(a[7], 1)], same=bool(a[8]), extra=a[9], nhwc=bool(a[10]))


def _pool2d(kind):
    def f(ins, i, t, b, d):
        c = _pool_args(i)
        x = ins[0].permute(0, 3, 1, 2) if c["nhwc"] else ins[0]
        p = c["p"]
        if c["same"]:
            ph, pw = _same_pad(x.shape[2], c["k"][0], c["s"][0], c["dl"][0]), _same_pad(x.shape[3], c["k"][1], c["s"][1], c["dl"][1])
            if ph[0] != ph[1] or pw[0] != pw[1] or kind != "max":
                fillv = float("-inf") if kind == "max" else 0.0
                xp = F.pad(x, [pw[0], pw[1], ph[0], ph[1]], value=fillv)
                if kind == "avg":
                    s = F.avg_pool2d(xp, tuple(c["k"]), tuple(c["s"]), 0, False, True, divisor_override=1)
                    if c["extra"] == 0:
                        y = s / (c["k"][0] * c["k"][1])
                    else:
                        cnt = F.avg_pool2d(F.pad(torch.ones_like(x[:1, :1]), [pw[0], pw[1], ph[0], ph[1]]), tuple(c["k"]), tuple(c["s"]),
                                           0, False, True, divisor_override=1)
                        y = s / cnt
                    return y.permute(0, 2, 3, 1) if c["nhwc"] else y
                x, p = xp, [0, 0]
            else:
                p = [ph[0], pw[0]]
        if kind == "max":
            y = F.max_pool2d(x, tuple(c["k"]), tuple(c["s"]), tuple(p), tuple(c["dl"]))
        elif kind == "avg":
            y = F.avg_pool2d(x, tuple(c["k"]), tuple(c["s"]), tuple(p), False, count_include_pad=(c["extra"] == 0))
        else:
            pn = float(c["extra"])
            y = F.avg_pool2d(x.abs() ** pn, tuple(c["k"]), tuple(c["s"]), tuple(p), False, True, divisor_override=1) ** (1.0 / pn)
        return y.permute(0, 2, 3, 1) if c["nhwc"] else y
    return f


op("maxpool2d")(_pool2d("max"))
op("avgpool2d")(_pool2d("avg"))
op("pnormpool2d")(_pool2d("pnorm"))
op("maxpool2d_bp")(_generic_bp(_pool2d("max")))
op("avgpool2d_bp")(_generic_bp(_pool2d("avg")))
op("pnormpool2d_bp")(_generic_bp(_pool2d("pnorm")))


def _pool_hip_ok(kind):
    def ok(ins, i, t, b, d):
        c = _pool_args(i)
        x = ins[0]
        return (L.use_hip(x) and x.dim() == 4 and not c["nhwc"] and not c["same"] and c["dl"] == [1, 1]
                and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous(memory_format=torch.channels_last)
                and (kind == "max" or c["extra"] == 0))
    return ok


def _pool_hip(kind):
    def f(ins, i, t, b, d):
        from ...ops import pooling as P
        c = _pool_args(i)
        r = P.pool2d_forward(ins[0], kind, tuple(c["k"]), tuple(c["s"]), tuple(c["p"]), train=False)
        return r[0] if isinstance(r, (tuple, list)) else r
    return f


platform("maxpool2d", _pool_hip_ok("max"))(_pool_hip("max"))
platform("avgpool2d", _pool_hip_ok("avg"))(_pool_hip("avg"))


def _pool3d(kind):
    def f(ins, i, t, b, d):
        a = list(i) + [0] * (15 - len(i))
        k, s, p, dl, same, extra, ndhwc = a[0:3], a[3:6], a[6:9], [max(v, 1) for v in a[9:12]], bool(a[12]), a[13], a[14] == 1
        x = ins[0].permute(0, 4, 1, 2, 3) if ndhwc else ins[0]
        if same:
            pads = [_same_pad(x.shape[2 + q], k[q], s[q], dl[q]) for q in range(3)]
            flat = []
            for lo, hi in reversed(pads):
                flat += [lo, hi]
            x = F.pad(x, flat, value=float("-inf") if kind == "max" else 0.0)
            p = [0, 0, 0]
        if kind == "max":
            y = F.max_pool3d(x, tuple(k), tuple(s), tuple(p), tuple(dl))
        else:
            y = F.avg_pool3d(x, tuple(k), tuple(s), tuple(p), False, count_include_pad=(extra == 0))
        return y.permute(0, 2, 3, 4, 1) if ndhwc else y
    return f


op("maxpool3dnew")(_pool3d("max"))
op("avgpool3dnew")(_pool3d("avg"))
op("maxpool3dnew_bp")(_generic_bp(_pool3d("max")))
op("avgpool3dnew_bp")(_generic_bp(_pool3d("avg")))


@op("max_pool_with_argmax")
def _max_pool_argmax(ins, i, t, b, d):
    c = _pool_args(i)
    x = ins[0].permute(0, 3, 1, 2) if c["nhwc"] else ins[0]
    p = c["p"]
    if c["same"]:
        ph, pw = _same_pad(x.shape[2], c["k"][0], c["s"][0], 1), _same_pad(x.shape[3], c["k"][1], c["s"][1], 1)
        p = [ph[0], pw[0]]
    y, idx = F.max_pool2d(x, tuple(c["k"]), tuple(c["s"]), tuple(p), 1, False, True)
    if c["nhwc"]:
        C = x.shape[1]
        ch = torch.arange(C, device=x.device).reshape(1, C, 1, 1)
        idx = (idx * C + ch).permute(0, 2, 3, 1)
        y = y.permute(0, 2, 3, 1)
    return y, idx


# ----------------------------------------------------------------------------- recurrent
_ACT = {0: torch.tanh, 1: F.relu, 2: torch.sigmoid, 3: lambda x: x, 4: F.leaky_relu, 5: lambda x: torch.where(x > 1, x, torch.zeros_like(x)),
        6: lambda x: x, 7: lambda x: torch.clamp(0.2 * x + 0.5, 0, 1), 8: F.elu, 9: lambda x: x / (1 + x.abs()), 10: F.softplus}


def _lstm_cell_math(x, h, c, Wx, Wr, bias, Wp, gate_act, cell_act, out_act, clip):
    """libnd4j gate order [i, f, c', o] (helpers/impl/lstmLayer.cpp:200-249)."""
    z = x @ Wx + h @ Wr
    if bias is not None:
        z = z + bias
    n = h.shape[-1]
    zi, zf, zg, zo = z[..., :n], z[..., n:2 * n], z[..., 2 * n:3 * n], z[..., 3 * n:]
    if Wp is not None:
        zi = zi + c * Wp[:n]
        zf = zf + c * Wp[n:2 * n]
    ig, fg, gg = gate_act(zi), gate_act(zf), cell_act(zg)
    cn = fg * c + ig * gg
    if clip and clip > 0:
        cn = cn.clamp(-clip, clip)
    if Wp is not None:
        zo = zo + cn * Wp[2 * n:]
    og = gate_act(zo)
    return og * out_act(cn), cn


@op("lstmLayer")
def _lstm_layer(ins, i, t, b, d):
    a = list(i) + [0] * 5
    data_fmt, dir_mode, gact, cact, oact = a[0], a[1], a[2] if len(i) > 2 else 2, a[3] if len(i) > 3 else 0, a[4] if len(i) > 4 else 0
    bb = list(b) + [False] * 8
    has_b, has_seq, has_hi, has_ci, has_p, ret_full, ret_hl, ret_cl = bb[:8]
    clip = t[0] if t else 0.0
    x, Wx, Wr = ins[0], ins[1], ins[2]
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
    # to [T, B, nIn]
    xs = {0: lambda: x, 1: lambda: x.permute(1, 0, 2), 2: lambda: x.permute(2, 0, 1), 3: lambda: x}[data_fmt]()
    T, B = xs.shape[0], xs.shape[1]
    bidir = dir_mode >= 2
    n = Wr.shape[-2]

    def run(xseq, Wx_, Wr_, b_, hi, ci, wp, reverse):
        h = hi if hi is not None else xseq.new_zeros(B, n)
        c = ci if ci is not None else xseq.new_zeros(B, n)
        outs = [None] * T
        order = range(T - 1, -1, -1) if reverse else range(T)
        for step, tt in enumerate(order):
            hn, cn = _lstm_cell_math(xseq[tt], h, c, Wx_, Wr_, b_, wp, _ACT[gact], _ACT[cact], _ACT[oact], clip)
            if seq is not None:
                live = (seq.to(xseq.device).long() > tt).reshape(B, 1)
                hn, cn = torch.where(live, hn, h), torch.where(live, cn, c)
                outs[tt] = torch.where(live, hn, torch.zeros_like(hn))
            else:
                outs[tt] = hn
            h, c = hn, cn
        return torch.stack(outs), h, c

    if not bidir:
        hs, hl, cl = run(xs, Wx, Wr, bias, hI, cI, Wp, dir_mode == 1)
    else:
        f = run(xs, Wx[0], Wr[0], None if bias is None else bias[0], None if hI is None else hI[0], None if cI is None else cI[0],
                None if Wp is None else Wp[0], False)
        r = run(xs, Wx[1], Wr[1], None if bias is None else bias[1], None if hI is None else hI[1], None if cI is None else cI[1],
                None if Wp is None else Wp[1], True)
        if dir_mode == 2:
            hs = f[0] + r[0]
        elif dir_mode == 3:
            hs = torch.cat([f[0], r[0]], -1)
        else:
            hs = torch.stack([f[0], r[0]], 1)         # [T, 2, B, n]
        hl, cl = torch.stack([f[1], r[1]]), torch.stack([f[2], r[2]])
    outs = []
    if ret_full:
        if dir_mode == 4:
            outs.append(hs)
        else:
            outs.append({0: lambda: hs, 1: lambda: hs.permute(1, 0, 2), 2: lambda: hs.permute(1, 2, 0), 3: lambda: hs}[data_fmt]())
    if ret_hl:
        outs.append(hl)
    if ret_cl:
        outs.append(cl)
    return tuple(outs)


@op("lstmLayer_bp")
def _lstm_layer_bp(ins, i, t, b, d):
    bb = list(b) + [False] * 8
    n_out = sum(bb[5:8])
    fw, gs = list(ins[:len(ins) - n_out]), list(ins[len(ins) - n_out:])
    leaves = [x.detach().clone().requires_grad_(True) if x.is_floating_point() else x for x in fw]
    with torch.enable_grad():
        # the fused sequence kernels where they apply (same predicate as the forward op's platform route), else the generic math
        from .impl_platform import _lstm_layer_hip, _lstm_layer_ok
        from .custom import HELPER_HIT
        if _lstm_layer_ok(leaves, i, t, b, d):
            HELPER_HIT["lstmLayer_bp"] += 1
            ys = _lstm_layer_hip(leaves, i, t, b, d)
        else:
            ys = _lstm_layer(leaves, i, t, b, d)
    req = [x for x in leaves if x.requires_grad]
    return torch.autograd.grad(ys, req, gs, allow_unused=True)


@op("lstmLayerCell")
def _lstm_layer_cell(ins, i, t, b, d):
    x, Wx, Wr = ins[0], ins[1], ins[2]
    bb = list(b) + [False] * 2
    k = 3
    bias = ins[k] if bb[0] else None
    k += bb[0]
    hI, cI = ins[k], ins[k + 1]
    Wp = ins[k + 2] if bb[1] and len(ins) > k + 2 else None
    a = list(i) + [2, 0, 0]
    return _lstm_cell_math(x, hI, cI, Wx, Wr, bias, Wp, _ACT[a[0]], _ACT[a[1]], _ACT[a[2]], t[0] if t else 0.0)


op("lstmLayerCellBp")(lambda ins, i, t, b, d: _cell_bp(ins, i, t, b, d))


def _cell_bp(ins, i, t, b, d):
    fw, g = list(ins[:-1]), ins[-1]
    leaves = [x.detach().clone().requires_grad_(True) for x in fw]
    with torch.enable_grad():
        h, _ = _lstm_layer_cell(leaves, i, t, b, d)
    return torch.autograd.grad(h, leaves, g, allow_unused=True)


@op("lstmCell", "lstmBlockCell")
def _lstm_cell(ins, i, t, b, d):
    """TF-style block cell: inputs x, cLast, yLast, W [(nIn+n), 4n] (gate order i, c', f, o), Wci, Wcf, Wco, b."""
    if len(ins) >= 8 and ins[3].dim() == 2 and ins[3].shape[0] == ins[0].shape[-1] + ins[2].shape[-1]:
        x, cLast, yLast, W, Wci, Wcf, Wco, bias = ins[:8]
        peephole = bool(i[0]) if i else False
        forget_bias, clip = (t[0] if t else 0.0), (t[1] if len(t) > 1 else 0.0)
        z = torch.cat([x, yLast], -1) @ W + bias
        n = cLast.shape[-1]
        zi, zc, zf, zo = z[..., :n], z[..., n:2 * n], z[..., 2 * n:3 * n], z[..., 3 * n:]
        if peephole:
            zi, zf = zi + cLast * Wci, zf + cLast * Wcf
        ig, fg, cc = torch.sigmoid(zi), torch.sigmoid(zf + forget_bias), torch.tanh(zc)
        c = fg * cLast + ig * cc
        if clip > 0:
            c = c.clamp(-clip, clip)
        if peephole:
            zo = zo + c * Wco
        og = torch.sigmoid(zo)
        h = og * torch.tanh(c)
        return ig, c, fg, og, cc, torch.tanh(c), h          # 4: block input z = tanh(zz) (helpers/cpu/lstm.cpp:196-204)
    # classic lstmCell: x, hLast, cLast, Wx, Wh, Wc(peephole diag 3n), Wp(projection), b
    x, hLast, cLast, Wx, Wh, Wc, Wp, bias = ins[:8]
    n = cLast.shape[-1]
    z = x @ Wx + hLast @ Wh + bias
    zi, zf, zg, zo = z[..., :n], z[..., n:2 * n], z[..., 2 * n:3 * n], z[..., 3 * n:]      # i, f, c, o (helpers/cpu/lstm.cpp:81-84)
    peephole = bool(i[0]) if i else False
    if peephole:
        zi, zf = zi + cLast * Wc[:n], zf + cLast * Wc[n:2 * n]
    c = torch.sigmoid(zf + (t[2] if len(t) > 2 else 0.0)) * cLast + torch.sigmoid(zi) * torch.tanh(zg)
    if t and t[0] > 0:
        c = c.clamp(-t[0], t[0])
    if peephole:
        zo = zo + c * Wc[2 * n:]
    h = torch.sigmoid(zo) * torch.tanh(c)
    if len(i) > 1 and i[1]:
        h = h @ Wp
        if len(t) > 1 and t[1] > 0:
            h = h.clamp(-t[1], t[1])
    return h, c


@op("lstmBlock", "lstm")
def _lstm_block(ins, i, t, b, d):
    if ins[0].dim() == 0 or (ins[0].numel() == 1 and not ins[0].is_floating_point()):
        ins = ins[1:]        # leading maxTSLength
    x = ins[0]
    if not (len(ins) >= 8 and ins[3].dim() == 2 and ins[3].shape[0] == x.shape[-1] + ins[2].shape[-1]):
        # `lstm` (generic/nn/recurrent/lstm.cpp:35-60): x [time, bS, nIn], h0, c0, Wx, Wh, Wc, Wp, b -> h [time, bS, numProj],
        # c [time, bS, numUnits]; one classic lstmCell per step
        h, c = ins[1], ins[2]
        hs, cs = [], []
        for tt in range(x.shape[0]):
            h, c = _lstm_cell([x[tt], h, c] + list(ins[3:]), i, t, b, d)
            hs.append(h)
            cs.append(c)
        return torch.stack(hs), torch.stack(cs)
    fmt = int(i[1]) if len(i) > 1 else 0          # 0 TNS, 1 NST, 2 NTS
    xs = {0: lambda: x, 1: lambda: x.permute(2, 0, 1), 2: lambda: x.permute(1, 0, 2)}[fmt]()
    c, y = ins[1], ins[2]
    outs = []
    for tt in range(xs.shape[0]):
        r = _lstm_cell([xs[tt], c, y] + list(ins[3:]), i, t, b, d)
        c, y = r[1], r[6]
        outs.append(r)
    st = [torch.stack([o[k] for o in outs]) for k in range(7)]
    back = {0: lambda v: v, 1: lambda v: v.permute(1, 2, 0), 2: lambda v: v.permute(1, 0, 2)}[fmt]
    return tuple(back(v) for v in st)


def _gru_cell_math(x, h, Wru, Wc, bru, bc):
    """gruCell: Wru [(nIn+n), 2n] (r,u), Wc [(nIn+n), n]."""
    n = h.shape[-1]
    ru = torch.sigmoid(torch.cat([x, h], -1) @ Wru + bru)
    r, u = ru[..., :n], ru[..., n:]
    c = torch.tanh(torch.cat([x, r * h], -1) @ Wc + bc)
    return r, u, c, u * h + (1 - u) * c


op("gruCell")(lambda ins, i, t, b, d: _gru_cell_math(*ins[:6]))
op("gruCell_bp")(lambda ins, i, t, b, d: _gru_cell_bp(ins))


def _gru_cell_bp(ins):
    fw, g = list(ins[:6]), ins[-1]
    leaves = [x.detach().clone().requires_grad_(True) for x in fw]
    with torch.enable_grad():
        h = _gru_cell_math(*leaves)[3]
    return torch.autograd.grad(h, leaves, g)


@op("gru")
def _gru(ins, i, t, b, d):
    """x [T,B,nIn], hI [B,n], Wx [nIn,3n], Wh [n,3n], b [3n] (gate order r,u,c)."""
    x, h, Wx, Wh = ins[0], ins[1], ins[2], ins[3]
    bias = ins[4] if len(ins) > 4 else None
    n = h.shape[-1]
    outs = []
    xz = x @ Wx + (bias if bias is not None and bias.dim() == 1 else 0)
    for tt in range(x.shape[0]):
        hz = h @ Wh
        r = torch.sigmoid(xz[tt][..., :n] + hz[..., :n])
        u = torch.sigmoid(xz[tt][..., n:2 * n] + hz[..., n:2 * n])
        c = torch.tanh(xz[tt][..., 2 * n:] + r * hz[..., 2 * n:])
        h = u * h + (1 - u) * c
        outs.append(h)
    return torch.stack(outs)


op("gru_bp")(_generic_bp(_gru))


def _sru_math(x, w, bias, c0, mask=None):
    """x [B, K, T], w [3K, K], b [2K], c0 [B, K] (sru.cpp)."""
    B, K, T = x.shape
    if mask is not None:
        x = x * mask.unsqueeze(-1)
    z = torch.einsum("ok,bkt->bot", w, x)
    xt, f, r = z[:, :K], torch.sigmoid(z[:, K:2 * K] + bias[:K].reshape(1, K, 1)), torch.sigmoid(z[:, 2 * K:] + bias[K:].reshape(1, K, 1))
    c = c0
    hs, cs = [], []
    for tt in range(T):
        c = f[:, :, tt] * c + (1 - f[:, :, tt]) * xt[:, :, tt]
        hs.append(r[:, :, tt] * torch.tanh(c) + (1 - r[:, :, tt]) * x[:, :, tt])
        cs.append(c)
    return torch.stack(hs, -1), torch.stack(cs, -1)


op("sru")(lambda ins, i, t, b, d: _sru_math(ins[0], ins[1], ins[2], ins[3], ins[4] if len(ins) > 4 else None))
op("sru_bp")(lambda ins, i, t, b, d: _sru_bp(ins))


def _sru_bp(ins):
    x, w, bias, c0 = ins[:4]
    leaves = [v.detach().clone().requires_grad_(True) for v in (x, w, bias, c0)]
    with torch.enable_grad():
        h, c = _sru_math(*leaves)
    gh = ins[-1] if ins[-1].shape == h.shape else ins[-2]
    return torch.autograd.grad(h, leaves, gh, allow_unused=True)


@op("sruCell")
def _sru_cell(ins, i, t, b, d):
    x, c0, w, bias = ins[:4]
    K = x.shape[-1]
    z = x @ w
    f, r = torch.sigmoid(z[:, K:2 * K] + bias[:K]), torch.sigmoid(z[:, 2 * K:] + bias[K:])
    c = f * c0 + (1 - f) * z[:, :K]
    return r * torch.tanh(c) + (1 - r) * x, c


@op("sru_bi")
def _sru_bi(ins, i, t, b, d):
    """x [T, B, 2K], w [2K, 6K], b [4K], c0 [B, 2K]."""
    x, w, bias, c0 = ins[:4]
    T, B, K2 = x.shape
    K = K2 // 2
    z = (x @ w).reshape(T, B, 2, 3 * K)
    hs, cs = x.new_zeros(T, B, 2, K), x.new_zeros(T, B, 2, K)
    for dirn in range(2):
        c = c0[:, dirn * K:(dirn + 1) * K]
        bf, br = bias[dirn * 2 * K:dirn * 2 * K + K], bias[dirn * 2 * K + K:(dirn + 1) * 2 * K]
        order = range(T) if dirn == 0 else range(T - 1, -1, -1)
        for tt in order:
            zz = z[tt, :, dirn]
            f, r = torch.sigmoid(zz[:, K:2 * K] + bf), torch.sigmoid(zz[:, 2 * K:] + br)
            c = f * c + (1 - f) * zz[:, :K]
            hs[tt, :, dirn] = r * torch.tanh(c) + (1 - r) * x[tt, :, dirn * K:(dirn + 1) * K]
            cs[tt, :, dirn] = c
    return hs.reshape(T, B, K2), cs.reshape(T, B, K2)


op("sru_bi_bp")(lambda ins, i, t, b, d: _sru_bi_bp(ins, i, t, b, d))


def _sru_bi_bp(ins, i, t, b, d):
    leaves = [v.detach().clone().requires_grad_(True) for v in ins[:4]]
    with torch.enable_grad():
        h, c = _sru_bi(leaves, i, t, b, d)
    g = [v for v in ins[4:] if v.shape == h.shape][-1]
    return torch.autograd.grad(h, leaves, g, allow_unused=True)


def _rnn_cell(x, h, Wx, Wh, bias):
    return torch.tanh(x @ Wx + h @ Wh + bias)


def _simple_rnn(time_major, bidir):
    def f(ins, i, t, b, d):
        x = ins[0]
        if not time_major and not (i and i[0]):
            x = x.permute(1, 0, 2)
        if not bidir:
            Wx, Wh, bias = ins[1], ins[2], ins[3]
            h = ins[4] if len(ins) > 4 and ins[4].is_floating_point() else x.new_zeros(x.shape[1], Wh.shape[0])
            bsum = bias if bias.numel() == Wh.shape[0] else bias[:Wh.shape[0]] + bias[Wh.shape[0]:]
            outs = []
            for tt in range(x.shape[0]):
                h = _rnn_cell(x[tt], h, Wx, Wh, bsum)
                outs.append(h)
            hs = torch.stack(outs)
            return (hs if (time_major or (i and i[0])) else hs.permute(1, 0, 2)), h
        WxF, WhF, bF, WxB, WhB, bB = ins[1:7]
        n = WhF.shape[0]
        hf, hb = x.new_zeros(x.shape[1], n), x.new_zeros(x.shape[1], n)
        of, ob = [], [None] * x.shape[0]
        bsF = bF if bF.numel() == n else bF[:n] + bF[n:]
        bsB = bB if bB.numel() == n else bB[:n] + bB[n:]
        for tt in range(x.shape[0]):
            hf = _rnn_cell(x[tt], hf, WxF, WhF, bsF)
            of.append(hf)
        for tt in range(x.shape[0] - 1, -1, -1):
            hb = _rnn_cell(x[tt], hb, WxB, WhB, bsB)
            ob[tt] = hb
        hs = torch.cat([torch.stack(of), torch.stack(ob)], -1)
        return (hs if (time_major or (i and i[0])) else hs.permute(1, 0, 2)), hf, hb
    return f


op("static_rnn")(_simple_rnn(True, False))
op("dynamic_rnn")(_simple_rnn(False, False))
op("static_bidirectional_rnn")(_simple_rnn(True, True))
op("dynamic_bidirectional_rnn")(_simple_rnn(False, True))


# ----------------------------------------------------------------------------- losses
def _reduce_loss(per, w, mode):
    """reductionMode: 0 none, 1 sum, 2 mean by weight, 3 mean by non-zero weight count."""
    if w is None:
        w = torch.ones((), dtype=per.dtype, device=per.device)
    wl = per * w
    if mode == 0:
        return wl
    s = wl.sum()
    if mode == 1:
        return s
    wb = w.expand_as(per) if w.dim() else w.expand(per.shape)
    if mode == 2:
        ws = wb.sum()
        return torch.where(ws == 0, torch.zeros_like(s), s / torch.where(ws == 0, torch.ones_like(ws), ws))
    nz = (wb != 0).sum().to(per.dtype)
    return torch.where(nz == 0, torch.zeros_like(s), s / nz.clamp_min(1))


def _loss(name, per_fn):
    def f(ins, i, t, b, d):
        pred, w, lab = ins[0], ins[1] if len(ins) > 2 else None, ins[-1]
        return _reduce_loss(per_fn(pred, lab, t, i), w, int(i[0]) if i else 0)
    op(name)(f)

    def g(ins, i, t, b, d):
        pred, w, lab = ins[0], ins[1], ins[2]
        leaves = [pred.detach().clone().requires_grad_(True), w.detach().clone().float().requires_grad_(True),
                  lab.detach().clone().requires_grad_(True) if lab.is_floating_point() else lab]
        with torch.enable_grad():
            y = f(leaves, i, t, b, d).sum()          # (the reduction too: the op may be executed under no_grad by a gradient graph)
        req = [v for v in leaves if v.requires_grad]
        gs = torch.autograd.grad(y, req, allow_unused=True)
        return tuple(gg if gg is not None else torch.zeros_like(v) for gg, v in zip(gs, req))
    op(name + "_grad")(g)


_loss("absolute_difference_loss", lambda p, l, t, i: (p - l).abs())
_loss("mean_sqerr_loss", lambda p, l, t, i: (p - l) ** 2)
_loss("hinge_loss", lambda p, l, t, i: F.relu(1 - (2 * l - 1) * p))
_loss("huber_loss", lambda p, l, t, i: torch.where((p - l).abs() <= (t[0] if t else 1.0), 0.5 * (p - l) ** 2,
                                                     (t[0] if t else 1.0) * (p - l).abs() - 0.5 * (t[0] if t else 1.0) ** 2))
_loss("log_loss", lambda p, l, t, i: -(l * torch.log(p + (t[0] if t else 1e-7)) + (1 - l) * torch.log(1 - p + (t[0] if t else 1e-7))))
_loss("log_poisson_loss", lambda p, l, t, i: torch.exp(p) - l * p + ((l * torch.log(l.clamp_min(1e-30)) - l + 0.5 * torch.log(2 * math.pi * l.clamp_min(1e-30)))
                                                                    * (l > 1).to(p.dtype) if (len(i) > 1 and i[1]) else 0))
_loss("cosine_distance_loss", lambda p, l, t, i: 1 - (p * l).sum(i[1] if len(i) > 1 else -1, keepdim=True))
_loss("sigm_cross_entropy_loss", lambda p, l, t, i: F.relu(p) - p * (l * (1 - (t[0] if t else 0.0)) + 0.5 * (t[0] if t else 0.0))
      + torch.log1p(torch.exp(-p.abs())))


def _mpwse(p, l, t, i):
    dlt = (p - l).reshape(p.shape[0], -1)
    n = dlt.shape[1]
    s1 = (dlt * dlt).sum(1, keepdim=True)
    s2 = dlt.sum(1, keepdim=True) ** 2
    return (2 * (n * s1 - s2) / max(n * (n - 1), 1)).reshape([p.shape[0]] + [1] * (p.dim() - 1))


_loss("mean_pairwssqerr_loss", _mpwse)


def _smce(p, l, t, i):
    ls = t[0] if t else 0.0
    if ls:
        l = l * (1 - ls) + ls / l.shape[-1]
    return -(l * torch.log_softmax(p, -1)).sum(-1, keepdim=True)


_loss("softmax_cross_entropy_loss", _smce)
op("l2_loss")(lambda ins, i, t, b, d: 0.5 * (ins[0] * ins[0]).sum())
op("softmax_cross_entropy_loss_with_logits")(lambda ins, i, t, b, d: -(ins[1] * torch.log_softmax(ins[0], i[0] if i else -1)).sum(i[0] if i else -1))
op("softmax_cross_entropy_loss_with_logits_grad")(lambda ins, i, t, b, d: (
    torch.softmax(ins[0], i[0] if i else -1) * ins[1].sum(i[0] if i else -1, keepdim=True) - ins[1],
    -torch.log_softmax(ins[0], i[0] if i else -1)))
op("sparse_softmax_cross_entropy_loss_with_logits")(lambda ins, i, t, b, d: F.cross_entropy(
    ins[1].reshape(-1, ins[1].shape[-1]), ins[0].long().reshape(-1), reduction="none").reshape(ins[0].shape))
op("sparse_softmax_cross_entropy_loss_with_logits_grad")(lambda ins, i, t, b, d: torch.softmax(ins[1], -1) - F.one_hot(
    ins[0].long(), ins[1].shape[-1]).to(ins[1].dtype))
op("weighted_cross_entropy_with_logits")(lambda ins, i, t, b, d: (1 - ins[0]) * ins[1] + (1 + (ins[2] - 1) * ins[0]) * (
    torch.log1p(torch.exp(-ins[1].abs())) + F.relu(-ins[1])))


@op("ctc_loss")
def _ctc_loss(ins, i, t, b, d):
    """targetLabels [B, L], logitInput [B, T, C], targetLabelLengths [B], logitInputLengths [B]; iArgs [blankIndex]."""
    labels, logits, lab_len, in_len = ins[0], ins[1], ins[2], ins[3]
    blank = int(i[0]) if i else 0
    lp = torch.log_softmax(logits.float(), -1).permute(1, 0, 2)
    return F.ctc_loss(lp, labels.long(), in_len.long(), lab_len.long(), blank=blank, reduction="none", zero_infinity=False)


@op("ctc_loss_grad")
def _ctc_loss_grad(ins, i, t, b, d):
    lg = ins[1].detach().clone().float().requires_grad_(True)
    with torch.enable_grad():
        y = _ctc_loss([ins[0], lg, ins[2], ins[3]], i, t, b, d)
    return torch.autograd.grad(y.sum(), lg)[0]


# ----------------------------------------------------------------------------- updaters
def _updater(kind_name, n_state):
    from ...ops import updater as U

    def f(ins, i, t, b, d):
        g = ins[0]
        states = [s.clone() for s in ins[1:1 + n_state]]
        # the Adam family declares its state arrays as (U | V = second moment, M = first moment[, H]) — generic/updaters/
        # adamUpdater.cpp:36-40 — while the fused kernel's flat layout is [m | v | h]
        swap = kind_name in ("Adam", "AdaMax", "Nadam", "AMSGrad", "AdaBelief")
        if swap:
            states[0], states[1] = states[1], states[0]
        extra_in = ins[1 + n_state:]
        tt = list(t) if t else [float(v) for v in extra_in]
        it = int(i[0]) if i else 0
        n = g.numel()
        defaults = {"Sgd": [0.01], "Nesterovs": [0.01, 0.9], "Adam": [1e-3, 0.9, 0.999, 1e-8], "RmsProp": [0.1, 0.95, 1e-8],
                    "AdaGrad": [0.1, 1e-6], "AdaDelta": [0.95, 1e-6], "AdaMax": [1e-3, 0.9, 0.999, 1e-8],
                    "Nadam": [1e-3, 0.9, 0.999, 1e-8], "AMSGrad": [1e-3, 0.9, 0.999, 1e-8], "AdaBelief": [1e-3, 0.9, 0.999, 1e-8]}[kind_name]
        tt = tt + defaults[len(tt):]
        seg = U.Segment(0, n, 0, U.KIND[kind_name], divide_by_batch=False, update_only=True)
        if kind_name == "Sgd":
            seg.lr = tt[0]
        elif kind_name == "Nesterovs":
            seg.lr, seg.a = tt[0], tt[1]
        elif kind_name in ("RmsProp",):
            seg.lr, seg.a, seg.eps = tt[0], tt[1], tt[2]
        elif kind_name == "AdaGrad":
            seg.lr, seg.eps = tt[0], tt[1]
        elif kind_name == "AdaDelta":
            seg.a, seg.eps = tt[0], tt[1]
        else:
            seg.lr, seg.a, seg.b, seg.eps = tt[0], tt[1], tt[2], tt[3]
        flat_state = torch.cat([s.reshape(-1) for s in states]) if states else g.new_zeros(0)
        grads = g.reshape(-1).clone()
        U.apply_segment_reference(seg, grads.new_zeros(n), grads, flat_state, 1.0, float(it + 1))
        new_states = [flat_state[k * n:(k + 1) * n].reshape(g.shape) for k in range(n_state)]
        if swap:
            new_states[0], new_states[1] = new_states[1], new_states[0]
        return tuple([grads.reshape(g.shape)] + new_states)
    return f


for _nm, _kn, _ns in (("sgd_updater", "Sgd", 0), ("nesterovs_updater", "Nesterovs", 1), ("adam_updater", "Adam", 2),
                      ("rms_prop_updater", "RmsProp", 1), ("ada_grad_updater", "AdaGrad", 1), ("ada_delta_updater", "AdaDelta", 2),
                      ("ada_max_updater", "AdaMax", 2), ("nadam_updater", "Nadam", 2), ("ams_grad_updater", "AMSGrad", 3),
                      ("adabelief_updater", "AdaBelief", 2)):
    op(_nm)(_updater(_kn, _ns))


# ----------------------------------------------------------------------------- random
def _rng():
    from .. import rng
    return rng


def _rand_shape(ins):
    return [int(v) for v in ins[0].reshape(-1).tolist()]


def _fill(opname, shape, args, dev, dt=torch.float32, x=None):
    R = _rng()
    z = torch.empty(shape, dtype=dt if dt.is_floating_point else torch.float32, device=dev)
    R.exec_random(opname, None, z, x, None, args)
    return z.to(dt)


op("randomuniform")(lambda ins, i, t, b, d: _fill("UniformDistribution", _rand_shape(ins), [t[0] if t else float(ins[1]), t[1] if len(t) > 1 else float(ins[2])],
                                                  ins[0].device, dtype_arg(d, i)))
op("random_normal")(lambda ins, i, t, b, d: _fill("GaussianDistribution", _rand_shape(ins), [t[0] if t else 0.0, t[1] if len(t) > 1 else 1.0],
                                                  ins[0].device, dtype_arg(d, None)))
op("random_bernoulli")(lambda ins, i, t, b, d: _fill("BernoulliDistribution", _rand_shape(ins), [t[0]], ins[0].device))
op("random_exponential")(lambda ins, i, t, b, d: _fill("ExponentialDistribution", _rand_shape(ins), [t[0]], ins[0].device))


@op("random_gamma")
def _random_gamma(ins, i, t, b, d):
    shape = _rand_shape(ins)
    alpha = ins[1].float()
    beta = ins[2].float() if len(ins) > 2 else torch.ones_like(alpha)
    full = shape + list(torch.broadcast_shapes(alpha.shape, beta.shape))
    a = alpha.expand(full).contiguous()
    z = _fill("GammaDistribution", full, [1.0, 1.0], a.device, torch.float32, x=a)
    return z / beta.expand(full)


@op("random_poisson")
def _random_poisson(ins, i, t, b, d):
    shape = _rand_shape(ins)
    lam = ins[1].float()
    full = shape + list(lam.shape)
    return _fill("PoissonDistribution", full, [1.0], lam.device, torch.float32, x=lam.expand(full).contiguous())


@op("random_multinomial")
def _random_multinomial(ins, i, t, b, d):
    logits, n = ins[0], int(ins[1])
    dim = int(i[0]) if i else -1          # libnd4j random_multinomial: classes along the last dimension unless dimC is given
    lg = logits if dim in (1, -1) or logits.dim() == 1 else logits.t()
    g = torch.Generator(device=lg.device).manual_seed(_rng().getRandom().nextInt())
    r = torch.multinomial(torch.softmax(lg.float(), -1), n, True, generator=g)
    return r if dim in (1, -1) or logits.dim() == 1 else r.t()


@op("random_crop")
def _random_crop(ins, i, t, b, d):
    x, shape = ins[0], [int(v) for v in ins[1].tolist()]
    R = _rng().getRandom()
    key = tuple(slice(o, o + s) for o, s in ((R.nextInt(x.shape[k] - s + 1), s) for k, s in enumerate(shape)))
    return x[key]


@op("random_shuffle")
def _random_shuffle(ins, i, t, b, d):
    x = ins[0]
    g = torch.Generator().manual_seed(_rng().getRandom().nextInt())
    return x[torch.randperm(x.shape[0], generator=g).to(x.device)]


@op("dropout")
def _dropout(ins, i, t, b, d):
    x = ins[0]
    p = t[0] if t else 0.5          # keep probability
    mask = _fill("BernoulliDistribution", list(x.shape), [p], x.device).to(x.dtype)
    return x * mask / p, mask


op("dropout_bp")(lambda ins, i, t, b, d: ins[1] * (ins[0] != 0).to(ins[1].dtype) / (t[0] if t else 0.5) if len(ins) == 2 else ins[2] * ins[1] / (t[0] if t else 0.5))
op("alpha_dropout_bp")(lambda ins, i, t, b, d: ins[1] * (t[1] if len(t) > 1 else 1.0) * (ins[0] != 0).to(ins[1].dtype))
op("set_seed")(lambda ins, i, t, b, d: (_rng().setDefaultSeed(int(i[0]) if i else int(ins[0])), torch.tensor(0))[1])
op("get_seed")(lambda ins, i, t, b, d: torch.tensor(_rng().getRandom().getSeed(), dtype=torch.int64))
