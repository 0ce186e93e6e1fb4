"""Custom-op implementations, part 1: activations, broadcastable, reduce, shape, tensor, transforms,
datatypes, boolean, bitwise, blas.

Reference for names and argument contracts: libnd4j ops/declarable/headers/{activations,broadcastable,
shape,transforms,parity_ops,blas,boolean,bitwise,datatypes}.h and generic/**. Implementations are
ATen composites (hipBLASLt/rocBLAS for GEMMs) or calls into the in-tree HIP kernels; ``*_bp`` ops
without a closed form here are derived from the forward implementation with local autograd.
Argument-order parity for the long tail is "parity unpinned" unless a test pins it.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .custom import _IMPL, op
from . import legacy as LG

_DT = {"BOOL": torch.bool, "HALF": torch.float16, "FLOAT": torch.float32, "FLOAT32": torch.float32, "DOUBLE": torch.float64,
       "INT8": torch.int8, "BYTE": torch.int8, "INT16": torch.int16, "SHORT": torch.int16, "INT32": torch.int32, "INT": torch.int32,
       "INT64": torch.int64, "LONG": torch.int64, "UINT8": torch.uint8, "UBYTE": torch.uint8, "BFLOAT16": torch.bfloat16,
       "FLOAT16": torch.float16, "UINT32": torch.int64, "UINT64": torch.int64, "UINT16": torch.int32}
# sd::DataType numeric ids (array/DataType.h:27-51)
_DT_ID = {1: torch.bool, 3: torch.float16, 5: torch.float32, 6: torch.float64, 7: torch.int8, 8: torch.int16, 9: torch.int32,
          10: torch.int64, 11: torch.uint8, 12: torch.int32, 13: torch.int64, 14: torch.int64, 17: torch.bfloat16}


def dtype_arg(d, i, idx=0, default=torch.float32):
    if d and len(d) > idx:
        v = d[idx]
        return v if isinstance(v, torch.dtype) else _DT[str(v).upper()]
    if i and len(i) > idx and int(i[idx]) in _DT_ID:
        return _DT_ID[int(i[idx])]
    return default


def first(res):
    return res[0] if isinstance(res, (tuple, list)) else res


def bp_of(fwd_name, n_grad_inputs=None):
    """Build ``<op>_bp`` from the forward op with local autograd: inputs = forward inputs + gradO."""
    def fn(ins, i, t, b, d):
        *xs, g = ins
        k = len(xs) if n_grad_inputs is None else n_grad_inputs
        leaves = [x.detach().clone().requires_grad_(True) if (x is not None and x.is_floating_point() and j < k) else x
                  for j, x in enumerate(xs)]
        with torch.enable_grad():
            y = first(_IMPL[fwd_name](leaves, i, t, b, d))
        req = [x for x in leaves if x is not None and x.requires_grad]
        gs = torch.autograd.grad(y, req, g.to(y.dtype).expand_as(y) if g.shape != y.shape else g.to(y.dtype), allow_unused=True)
        it = iter(gs)
        out = []
        for j, x in enumerate(leaves[:k]):
            if x is not None and x.requires_grad:
                gx = next(it)
                out.append(gx if gx is not None else torch.zeros_like(x))
            elif x is not None:
                out.append(torch.zeros_like(x, dtype=torch.float32 if not x.is_floating_point() else x.dtype))
        return tuple(out)
    return fn


def _unbroadcast(g, shape):
    while g.dim() > len(shape):
        g = g.sum(0)
    for k, s in enumerate(shape):
        if s == 1 and g.shape[k] != 1:
            g = g.sum(k, keepdim=True)
    return g


def _ex():
    return LG.executioner()


# ----------------------------------------------------------------------------- activations
def _unary(name, fn, bp=None):
    op(name)(lambda ins, i, t, b, d: fn(ins[0], t))
    if bp is not None:
        op(name + "_bp")(lambda ins, i, t, b, d: bp(ins[0], ins[1], t))
    else:
        op(name + "_bp")(bp_of(name))


_SA, _SL = 1.6732632423543772, 1.0507009873554805
_unary("sigmoid", lambda x, t: torch.sigmoid(x), lambda x, g, t: g * torch.sigmoid(x) * (1 - torch.sigmoid(x)))
_unary("tanh", lambda x, t: torch.tanh(x), lambda x, g, t: g * (1 - torch.tanh(x) ** 2))
_unary("relu", lambda x, t: torch.where(x > (t[0] if t else 0.0), x, torch.zeros_like(x) + (t[0] if t else 0.0)),
       lambda x, g, t: g * (x > (t[0] if t else 0.0)).to(g.dtype))
_unary("relu6", lambda x, t: torch.clamp(x, t[0] if t else 0.0, 6.0), lambda x, g, t: g * ((x > (t[0] if t else 0.0)) & (x < 6.0)).to(g.dtype))
_unary("lrelu", lambda x, t: F.leaky_relu(x, t[0] if t else 0.01), lambda x, g, t: g * torch.where(x >= 0, torch.ones_like(x), torch.full_like(x, t[0] if t else 0.01)))
_unary("elu", lambda x, t: F.elu(x, t[0] if t else 1.0))
_unary("selu", lambda x, t: F.selu(x))
_unary("softplus", lambda x, t: F.softplus(x), lambda x, g, t: g * torch.sigmoid(x))
_unary("softsign", lambda x, t: x / (1 + x.abs()), lambda x, g, t: g / (1 + x.abs()) ** 2)
_unary("cube", lambda x, t: x ** 3, lambda x, g, t: g * 3 * x * x)
_unary("hardtanh", lambda x, t: torch.clamp(x, -1, 1), lambda x, g, t: g * ((x >= -1) & (x <= 1)).to(g.dtype))
_unary("hardsigmoid", lambda x, t: torch.clamp(0.2 * x + 0.5, 0, 1), lambda x, g, t: g * 0.2 * ((x > -2.5) & (x < 2.5)).to(g.dtype))
_unary("rationaltanh", lambda x, t: LG._un_cpu("RationalTanh", x, None), lambda x, g, t: g * LG._un_cpu("RationalTanhDerivative", x, None))
_unary("rectifiedtanh", lambda x, t: torch.clamp_min(torch.tanh(x), 0), lambda x, g, t: g * LG._un_cpu("RectifiedTanhDerivative", x, None))
_unary("identity", lambda x, t: x.clone(), lambda x, g, t: g.clone())
_unary("thresholdedrelu", lambda x, t: torch.where(x > (t[0] if t else 0.0), x, torch.zeros_like(x)),
       lambda x, g, t: g * (x > (t[0] if t else 0.0)).to(g.dtype))
_unary("crelu", lambda x, t: torch.cat([F.relu(x), F.relu(-x)], -1))
op("identity_n")(lambda ins, i, t, b, d: tuple(x.clone() for x in ins))


@op("prelu")
def _prelu(ins, i, t, b, d):
    x, alpha = ins[0], ins[1]
    a = alpha.reshape([1] + list(alpha.shape)) if alpha.dim() == x.dim() - 1 else alpha
    return torch.where(x >= 0, x, a * x)


@op("prelu_bp")
def _prelu_bp(ins, i, t, b, d):
    x, alpha, g = ins
    a = alpha.reshape([1] + list(alpha.shape)) if alpha.dim() == x.dim() - 1 else alpha
    dx = torch.where(x >= 0, g, a * g)
    da = _unbroadcast(torch.where(x >= 0, torch.zeros_like(g), x * g), list(a.shape)).reshape(alpha.shape)
    return dx, da


# ----------------------------------------------------------------------------- broadcastable
def _bcast(name, fn, dx=None, dy=None, bool_out=False):
    op(name)(lambda ins, i, t, b, d: fn(ins[0], ins[1]))
    if dx is not None:
        def bp(ins, i, t, b, d):
            x, y, g = ins
            return _unbroadcast(dx(x, y, g), list(x.shape)), _unbroadcast(dy(x, y, g), list(y.shape))
        op(name + "_bp")(bp)


_bcast("add", lambda x, y: x + y, lambda x, y, g: g + 0 * x, lambda x, y, g: g + 0 * y)
_bcast("subtract", lambda x, y: x - y, lambda x, y, g: g + 0 * x, lambda x, y, g: -g + 0 * y)
_bcast("multiply", lambda x, y: x * y, lambda x, y, g: g * y, lambda x, y, g: g * x)
_bcast("divide", lambda x, y: x / y, lambda x, y, g: g / y, lambda x, y, g: -g * x / (y * y))
_bcast("realdiv", lambda x, y: x / y, lambda x, y, g: g / y, lambda x, y, g: -g * x / (y * y))
_bcast("reversesubtract", lambda x, y: y - x, lambda x, y, g: -g + 0 * x, lambda x, y, g: g + 0 * y)
_bcast("reversedivide", lambda x, y: y / x, lambda x, y, g: -g * y / (x * x), lambda x, y, g: g / x)
_bcast("squaredsubtract", lambda x, y: (x - y) ** 2, lambda x, y, g: 2 * g * (x - y), lambda x, y, g: -2 * g * (x - y))
_bcast("maximum", lambda x, y: torch.maximum(x, y), lambda x, y, g: g * (x >= y).to(g.dtype), lambda x, y, g: g * (x < y).to(g.dtype))
_bcast("minimum", lambda x, y: torch.minimum(x, y), lambda x, y, g: g * (x <= y).to(g.dtype), lambda x, y, g: g * (x > y).to(g.dtype))
_bcast("floordiv", lambda x, y: torch.floor(x / y) if x.is_floating_point() else torch.div(x, y, rounding_mode="floor"),
       lambda x, y, g: torch.zeros_like(g) + 0 * x, lambda x, y, g: torch.zeros_like(g) + 0 * y)
_bcast("floormod", lambda x, y: torch.remainder(x, y), lambda x, y, g: g + 0 * x, lambda x, y, g: -g * torch.floor(x / y))
_bcast("mod", lambda x, y: torch.fmod(x, y), lambda x, y, g: g + 0 * x, lambda x, y, g: -g * torch.trunc(x / y))
_bcast("reversemod", lambda x, y: torch.fmod(y, x), lambda x, y, g: -g * torch.trunc(y / x), lambda x, y, g: g + 0 * y)
_bcast("truncatediv", lambda x, y: torch.trunc(x / y) if x.is_floating_point() else torch.div(x, y, rounding_mode="trunc"))
_bcast("divide_no_nan", lambda x, y: torch.where(y == 0, torch.zeros_like(x / y), x / torch.where(y == 0, torch.ones_like(y), y)))
_bcast("Pow", lambda x, y: torch.pow(x, y), lambda x, y, g: g * y * torch.pow(x, y - 1),
       lambda x, y, g: g * torch.pow(x, y) * torch.log(torch.where(x > 0, x, torch.ones_like(x))))
_bcast("assign", lambda x, y: y.expand(torch.broadcast_shapes(x.shape, y.shape)).clone().to(x.dtype),
       lambda x, y, g: torch.zeros_like(g) + 0 * x, lambda x, y, g: g + 0 * y)
_bcast("tf_atan2", lambda x, y: torch.atan2(x, y))
_bcast("igamma", lambda x, y: torch.special.gammainc(x, y))
_bcast("igammac", lambda x, y: torch.special.gammaincc(x, y))
_bcast("equals", lambda x, y: x == y)
_bcast("not_equals", lambda x, y: x != y)
_bcast("less", lambda x, y: x < y)
_bcast("less_equal", lambda x, y: x <= y)
_bcast("greater", lambda x, y: x > y)
_bcast("greater_equal", lambda x, y: x >= y)
_bcast("boolean_and", lambda x, y: (x != 0) & (y != 0))
_bcast("boolean_or", lambda x, y: (x != 0) | (y != 0))
_bcast("boolean_xor", lambda x, y: (x != 0) ^ (y != 0))
op("meshgrid")(lambda ins, i, t, b, d: tuple(torch.meshgrid(*[x.reshape(-1) for x in ins], indexing="xy" if (not i or i[0]) else "ij")))


@op("percentile")
def _percentile(ins, i, t, b, d):
    x = ins[0].float()
    q = t[0] / 100.0
    interp = {0: "lower", 1: "higher", 2: "nearest"}.get(int(t[1]) if len(t) > 1 else 2, "nearest")
    keep = bool(t[2]) if len(t) > 2 else False
    if i:
        dims = [k % x.dim() for k in i]
        others = [k for k in range(x.dim()) if k not in dims]
        xp = x.permute(*others, *dims).reshape([x.shape[k] for k in others] + [-1])
        r = torch.quantile(xp, q, dim=-1, interpolation=interp)
        if keep:
            r = r.reshape([1 if k in dims else x.shape[k] for k in range(x.dim())])
        return r.to(ins[0].dtype)
    r = torch.quantile(x.reshape(-1), q, interpolation=interp)
    return (r.reshape([1] * x.dim()) if keep else r).to(ins[0].dtype)


# ----------------------------------------------------------------------------- reduce
def _dims(ins, i, x):
    if len(ins) > 1 and ins[1] is not None and not ins[1].is_floating_point() and ins[1].device.type != "meta":
        dd = [int(v) for v in ins[1].reshape(-1).tolist()]
    else:
        dd = list(i)
    if not dd or (len(dd) == 1 and dd[0] == 2147483647):
        return list(range(x.dim()))
    return sorted(set(k % max(x.dim(), 1) for k in dd))


def _keep(t, b):
    return bool(b[0]) if b else (bool(t[0]) if t else False)


def _reduce(name, fn):
    def f(ins, i, t, b, d):
        x = ins[0]
        dims = _dims(ins, i, x)
        if x.dim() == 0:
            return fn(x.reshape(1), [0], False).reshape(())
        return fn(x, dims, _keep(t, b))
    op(name)(f)

    def bp(ins, i, t, b, d):
        x, g = ins[0], ins[-1]
        dims = _dims(ins[:-1], i, x)
        xr = x.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            y = fn(xr, dims, True)
        gg = g
        if g.dim() != y.dim():
            gg = g.reshape(y.shape)
        return torch.autograd.grad(y, xr, gg.to(y.dtype))[0]
    op(name + "_bp")(bp)


_reduce("reduce_sum", lambda x, dm, k: x.sum(dm, keepdim=k))
_reduce("reduce_mean", lambda x, dm, k: x.mean(dm, keepdim=k))
_reduce("reduce_max", lambda x, dm, k: x.amax(dm, keepdim=k))
_reduce("reduce_min", lambda x, dm, k: x.amin(dm, keepdim=k))
_reduce("reduce_norm1", lambda x, dm, k: x.abs().sum(dm, keepdim=k))
_reduce("reduce_norm2", lambda x, dm, k: torch.sqrt((x * x).sum(dm, keepdim=k)))
_reduce("reduce_sqnorm", lambda x, dm, k: (x * x).sum(dm, keepdim=k))
_reduce("reduce_norm_max", lambda x, dm, k: x.abs().amax(dm, keepdim=k))


def _prod(x, dm, k):
    for a in sorted(dm, reverse=True):
        x = x.prod(a, keepdim=k)
    return x


_reduce("reduce_prod", _prod)


def _var_like(std):
    def fn(ins, i, t, b, d):
        x = ins[0]
        dims = _dims(ins, i, x)
        keep = bool(b[0]) if b else (bool(t[0]) if t else False)
        bias = bool(b[1]) if len(b) > 1 else (bool(t[1]) if len(t) > 1 else False)
        return (torch.std if std else torch.var)(x, dims, unbiased=bias, keepdim=keep)
    return fn


op("reduce_variance")(_var_like(False))
op("reduce_stdev")(_var_like(True))


def _var_bp(name):
    def bp(ins, i, t, b, d):
        x, g = ins[0], ins[-1]
        xr = x.detach().clone().requires_grad_(True)
        bb = list(b) if b else []
        bb = [True] + bb[1:] if bb else [True]
        with torch.enable_grad():
            y = _IMPL[name]([xr] + list(ins[1:-1]), i, t, bb, d)
        return torch.autograd.grad(y, xr, g.reshape(y.shape).to(y.dtype))[0]
    return bp


op("reduce_variance_bp")(_var_bp("reduce_variance"))
op("reduce_stdev_bp")(_var_bp("reduce_stdev"))
op("reduce_logsumexp")(lambda ins, i, t, b, d: torch.logsumexp(ins[0], _dims(ins, i, ins[0]), keepdim=_keep(t, b)))


@op("reduce_dot_bp")
def _reduce_dot_bp(ins, i, t, b, d):
    x, y, g = ins
    dims = _dims([x], i, x)
    gg = g.reshape([1 if k in dims else x.shape[k] for k in range(x.dim())]) if g.dim() != x.dim() else g
    return gg * y, gg * x


def _arg(fn, absolute=False):
    def f(ins, i, t, b, d):
        x = ins[0].abs() if absolute else ins[0]
        dims = _dims(ins, i, x)
        keep = bool(b[0]) if b else False
        if len(dims) == x.dim():
            r = fn(x.reshape(-1), 0)
            return r.reshape([1] * x.dim()) if keep else r
        if len(dims) == 1:
            return fn(x, dims[0], keepdim=keep) if keep else fn(x, dims[0])
        others = [k for k in range(x.dim()) if k not in dims]
        r = fn(x.permute(*others, *dims).reshape([x.shape[k] for k in others] + [-1]), -1)
        return r.reshape([1 if k in dims else x.shape[k] for k in range(x.dim())]) if keep else r
    return f


op("argmax")(_arg(torch.argmax))
op("argmin")(_arg(torch.argmin))
op("argamax")(_arg(torch.argmax, True))
op("argamin")(_arg(torch.argmin, True))


@op("norm")
def _norm(ins, i, t, b, d):
    x = ins[0]
    mode = int(t[0]) if t else 2
    dims = sorted(k % x.dim() for k in i) if i else list(range(x.dim()))
    if mode == 0:
        return x.abs().amax(dims)
    if mode == 1:
        return x.abs().sum(dims)
    if mode == 2:
        return torch.sqrt((x * x).sum(dims))
    p = t[1] if len(t) > 1 else float(mode)
    return (x.abs() ** p).sum(dims) ** (1.0 / p)


@op("moments")
def _moments(ins, i, t, b, d):
    x = ins[0]
    dims = _dims(ins, i, x)
    keep = _keep(t, b)
    return x.mean(dims, keepdim=keep), x.var(dims, unbiased=False, keepdim=keep)


@op("normalize_moments")
def _normalize_moments(ins, i, t, b, d):
    counts, means, variances = ins[0], ins[1], ins[2]
    shift = t[0] if t else 0.0
    m = means / counts + shift
    v = variances / counts - (means / counts) ** 2
    return m, v


@op("sufficient_statistics")
def _sufficient_statistics(ins, i, t, b, d):
    x = ins[0]
    dims = _dims(ins, i, x)
    n = 1
    for k in dims:
        n *= x.shape[k]
    cnt = torch.tensor(float(n), dtype=x.dtype, device=x.device)
    if len(ins) > 2 and ins[2] is not None:
        sh = ins[2]
        return cnt, (x - sh).sum(dims), ((x - sh) ** 2).sum(dims), sh
    return cnt, x.sum(dims), (x * x).sum(dims)


# ----------------------------------------------------------------------------- shape / tensor
def _shape_in(ins, i, k=1):
    if len(ins) > k and ins[k] is not None and ins[k].device.type != "meta":
        return [int(v) for v in ins[k].reshape(-1).tolist()]
    return list(i)


op("permute")(lambda ins, i, t, b, d: ins[0].permute(*(_shape_in(ins, i) or list(reversed(range(ins[0].dim()))))))
op("transpose")(lambda ins, i, t, b, d: ins[0].permute(*(_shape_in(ins, i) or list(reversed(range(ins[0].dim()))))))


@op("reshape")
def _reshape(ins, i, t, b, d):
    x = ins[0]
    shape = _shape_in(ins, i)
    order = "c"
    if shape and shape[0] in (-99, -102, 99, 102) and len(ins) == 1:
        order = "f" if abs(shape[0]) == 102 else "c"
        shape = shape[1:]
    if order == "f" and x.dim() > 1:
        rev = x.permute(*reversed(range(x.dim())))
        return rev.reshape(list(reversed(shape))).permute(*reversed(range(len(shape))))
    return x.reshape(shape)


op("reshapeas")(lambda ins, i, t, b, d: ins[0].reshape(ins[1].shape))
op("squeeze")(lambda ins, i, t, b, d: (ins[0].squeeze() if not _shape_in(ins, i) else
                                        ins[0].squeeze(tuple(sorted(k % ins[0].dim() for k in _shape_in(ins, i))))))
op("expand_dims")(lambda ins, i, t, b, d: ins[0].unsqueeze(_shape_in(ins, i)[0]))
op("flatten_2d")(lambda ins, i, t, b, d: ins[0].reshape(int(math.prod(ins[0].shape[:(i[0] if i else 1)])), -1))
op("shape_of")(lambda ins, i, t, b, d: torch.tensor(list(ins[0].shape), dtype=torch.int64))
op("shapes_of")(lambda ins, i, t, b, d: tuple(torch.tensor(list(x.shape), dtype=torch.int64) for x in ins))
op("set_shape")(lambda ins, i, t, b, d: ins[0].reshape(_shape_in(ins, i)))
op("size")(lambda ins, i, t, b, d: torch.tensor(ins[0].numel(), dtype=torch.int64))
op("rank")(lambda ins, i, t, b, d: torch.tensor(ins[0].dim(), dtype=torch.int32))
op("size_at")(lambda ins, i, t, b, d: torch.tensor(ins[0].shape[i[0]], dtype=torch.int64))
op("order")(lambda ins, i, t, b, d: ins[0].contiguous())
op("broadcast_to")(lambda ins, i, t, b, d: ins[0].expand(*_shape_in(ins, i)).clone())
op("tile_to_shape")(lambda ins, i, t, b, d: ins[0].expand(*i).clone())
op("tile_to_shape_bp")(lambda ins, i, t, b, d: _unbroadcast(ins[1], list(ins[0].shape)))
op("evaluate_reduction_shape")(lambda ins, i, t, b, d: torch.tensor(
    [1 if k in [v % ins[0].numel() for v in ins[1].tolist()] else int(s) for k, s in enumerate(ins[0].tolist())
     if (b and b[0]) or k not in [v % ins[0].numel() for v in ins[1].tolist()]], dtype=torch.int64))


@op("flatten")
def _flatten(ins, i, t, b, d):
    order = "f" if i and i[0] in (102, ord("f")) else "c"
    parts = [(x.permute(*reversed(range(x.dim()))) if order == "f" and x.dim() > 1 else x).reshape(-1) for x in ins]
    return torch.cat(parts)


@op("create")
def _create(ins, i, t, b, d):
    shape = [int(v) for v in ins[0].tolist()]
    dt = _DT_ID.get(int(i[1]), torch.float32) if len(i) > 1 else torch.float32
    return torch.zeros(shape, dtype=dt, device=ins[0].device) if (not b or b[0]) else torch.empty(shape, dtype=dt, device=ins[0].device)


op("fill")(lambda ins, i, t, b, d: torch.full([int(v) for v in ins[0].tolist()], t[0] if t else float(ins[1]),
                                               dtype=dtype_arg(d, None, 0, ins[1].dtype if len(ins) > 1 else torch.float32),
                                               device=ins[0].device))
op("fill_as")(lambda ins, i, t, b, d: torch.full_like(ins[0], t[0] if t else float(ins[1])))
op("zeros_as")(lambda ins, i, t, b, d: torch.zeros_like(ins[0], dtype=dtype_arg(d, None, 0, ins[0].dtype)))
op("ones_as")(lambda ins, i, t, b, d: torch.ones_like(ins[0], dtype=dtype_arg(d, None, 0, ins[0].dtype)))


@op("lin_space")
def _lin_space(ins, i, t, b, d):
    if ins:
        s, e, n = float(ins[0]), float(ins[1]), int(ins[2])
        dev = ins[0].device
    else:
        s, e, n, dev = t[0], t[1], int(i[0]), None
    return torch.linspace(s, e, n, dtype=dtype_arg(d, None, 0, torch.float32), device=dev)


@op("range")
def _range(ins, i, t, b, d):
    if ins:
        vals = [x.item() for x in ins]
        dt = ins[0].dtype
        dev = ins[0].device
    elif t:
        vals, dt, dev = list(t), torch.float32, None
    else:
        vals, dt, dev = list(i), torch.int64, None
    if len(vals) == 1:
        vals = [0, vals[0], 1]
    elif len(vals) == 2:
        vals = vals + [1]
    return torch.arange(vals[0], vals[1], vals[2], dtype=dtype_arg(d, None, 0, dt), device=dev)


@op("create_view")
def _create_view(ins, i, t, b, d):
    """Index spec per dim encoded in the extra inputs as (type, ...) — simplified: inputs 1.. are
    [begin, end, stride] triples (interval), single ints (point), or empty (all)."""
    x = ins[0]
    key = []
    for spec in ins[1:]:
        v = [int(s) for s in spec.reshape(-1).tolist()]
        if len(v) == 0:
            key.append(slice(None))
        elif len(v) == 1:
            key.append(v[0])
        else:
            key.append(slice(v[0], v[1], v[2] if len(v) > 2 else 1))
    return x[tuple(key)]


def _strided_key(x, begin, end, strides, begin_mask, ellipsis_mask, end_mask, new_axis_mask, shrink_mask):
    key = []
    for k in range(len(begin)):
        if ellipsis_mask & (1 << k):
            key.append(Ellipsis)
        elif new_axis_mask & (1 << k):
            key.append(None)
        elif shrink_mask & (1 << k):
            key.append(begin[k])
        else:
            bb = None if begin_mask & (1 << k) else begin[k]
            ee = None if end_mask & (1 << k) else end[k]
            key.append(slice(bb, ee, strides[k]))
    return tuple(key)


def _slice_neg(x, key):
    """torch has no negative-step slicing: flip those axes first."""
    out_key, flips, axis = [], [], 0
    for k in key:
        if isinstance(k, slice) and k.step is not None and k.step < 0:
            n = x.shape[axis]
            start = n - 1 if k.start is None else (k.start if k.start >= 0 else k.start + n)
            stop = -1 if k.stop is None else (k.stop if k.stop >= 0 else k.stop + n)
            flips.append(axis)
            out_key.append(slice(n - 1 - start, n - 1 - stop, -k.step))
            axis += 1
        else:
            out_key.append(k)
            if k is not None and k is not Ellipsis:
                axis += 1
            elif k is Ellipsis:
                axis = x.dim() - sum(1 for q in key[key.index(k) + 1:] if q is not None)
    if flips:
        x = torch.flip(x, flips)
    return x[tuple(out_key)]


@op("strided_slice")
def _strided_slice(ins, i, t, b, d):
    x = ins[0]
    masks = list(i[:5]) + [0] * (5 - len(i[:5]))
    if len(ins) >= 3:
        begin = [int(v) for v in ins[1].tolist()]
        end = [int(v) for v in ins[2].tolist()]
        strides = [int(v) for v in ins[3].tolist()] if len(ins) > 3 else [1] * len(begin)
    else:
        rest = list(i[5:])
        n = len(rest) // 3
        begin, end, strides = rest[:n], rest[n:2 * n], rest[2 * n:3 * n]
    key = _strided_key(x, begin, end, strides, masks[0], masks[1], masks[2], masks[3], masks[4])
    return _slice_neg(x, key)


op("strided_slice_bp")(bp_of("strided_slice", 1))


@op("slice")
def _slice(ins, i, t, b, d):
    x = ins[0]
    if len(ins) >= 3:
        begin, size = [int(v) for v in ins[1].tolist()], [int(v) for v in ins[2].tolist()]
    else:
        n = len(i) // 2
        begin, size = list(i[:n]), list(i[n:])
    key = tuple(slice(bb, None if s == -1 else bb + s) for bb, s in zip(begin, size))
    return x[key]


op("slice_bp")(bp_of("slice", 1))


# ----------------------------------------------------------------------------- transforms
def _axis_last(ins, i, n_data):
    """concat-style ops: axis is the last iArg or a trailing scalar input."""
    if len(ins) > n_data and ins[-1].numel() == 1 and not ins[-1].is_floating_point():
        return int(ins[-1]), ins[:-1]
    return (int(i[-1]) if i else 0), ins


@op("concat")
def _concat(ins, i, t, b, d):
    if b and b[0]:
        axis, xs = int(ins[-1]), ins[:-1]
    else:
        axis, xs = (int(i[0]) if i else 0), ins
    xs = [x for x in xs if x.numel() > 0 or x.device.type == "meta"]
    return torch.cat(xs, dim=axis)


@op("concat_bp")
def _concat_bp(ins, i, t, b, d):
    *xs, g = ins
    axis = int(i[0]) if i else 0
    return tuple(torch.split(g, [x.shape[axis] for x in xs], dim=axis))


op("stack")(lambda ins, i, t, b, d: torch.stack(list(ins), dim=i[0] if i else 0))
op("parallel_stack")(lambda ins, i, t, b, d: torch.stack(list(ins), dim=0))
op("unstack")(lambda ins, i, t, b, d: tuple(torch.unbind(ins[0], dim=i[0] if i else 0)))
op("tear")(lambda ins, i, t, b, d: tuple(torch.unbind(ins[0], dim=[k for k in range(ins[0].dim()) if k not in [v % ins[0].dim() for v in i]][0])))


@op("split")
def _split(ins, i, t, b, d):
    if len(ins) == 2:
        axis_t, x = (ins[0], ins[1]) if ins[0].numel() == 1 and not ins[0].is_floating_point() else (ins[1], ins[0])
        axis = int(axis_t)
        n = int(i[0])
    else:
        x, n, axis = ins[0], int(i[0]), (int(i[1]) if len(i) > 1 else 0)
    return tuple(torch.chunk(x, n, dim=axis))


@op("split_v")
def _split_v(ins, i, t, b, d):
    x, sizes = ins[0], [int(v) for v in ins[1].tolist()]
    axis = int(ins[2]) if len(ins) > 2 else (int(i[0]) if i else 0)
    if -1 in sizes:
        sizes[sizes.index(-1)] = x.shape[axis] - (sum(sizes) + 1)
    return tuple(torch.split(x, sizes, dim=axis))


@op("tile")
def _tile(ins, i, t, b, d):
    reps = _shape_in(ins, i)
    return ins[0].repeat(*reps)


op("tile_bp")(bp_of("tile", 1))
op("repeat")(lambda ins, i, t, b, d: torch.repeat_interleave(ins[0], torch.tensor(list(i[:-1])) if len(i) > 2 else int(i[0]),
                                                             dim=int(i[-1]) if len(i) > 1 else None))


@op("reverse")
def _reverse(ins, i, t, b, d):
    x = ins[0]
    dims = _shape_in(ins, i) or list(range(x.dim()))
    return torch.flip(x, [k % x.dim() for k in dims])


op("reverse_bp")(lambda ins, i, t, b, d: _reverse([ins[-1]] + list(ins[1:-1]), i, t, b, d))


@op("reverse_sequence")
def _reverse_sequence(ins, i, t, b, d):
    x, lens = ins[0], ins[1].long()
    seq_dim, batch_dim = int(i[0]), (int(i[1]) if len(i) > 1 else 0)
    T = x.shape[seq_dim]
    ar = torch.arange(T, device=x.device)
    ln = lens.to(x.device).reshape(-1, 1)
    idx = torch.where(ar.unsqueeze(0) < ln, ln - 1 - ar.unsqueeze(0), ar.unsqueeze(0))      # [B, T]
    xm = x.movedim((batch_dim, seq_dim), (0, 1))
    ix = idx.reshape(list(idx.shape) + [1] * (xm.dim() - 2)).expand_as(xm)
    return torch.gather(xm, 1, ix).movedim((0, 1), (batch_dim, seq_dim))


@op("pad")
def _pad(ins, i, t, b, d):
    x, p = ins[0], ins[1].long().reshape(-1, 2).tolist()
    mode = {0: "constant", 1: "reflect", 2: "replicate"}[int(i[0]) if i else 0]
    val = t[0] if t else (float(ins[2]) if len(ins) > 2 else 0.0)
    flat = []
    for lo, hi in reversed(p):
        flat += [int(lo), int(hi)]
    if mode == "constant":
        return F.pad(x, flat, value=val)
    if mode == "replicate":     # SYMMETRIC: edge included
        out = x
        for ax, (lo, hi) in enumerate(p):
            n = out.shape[ax]
            parts = []
            if lo:
                parts.append(torch.flip(out.narrow(ax, 0, lo), [ax]))
            parts.append(out)
            if hi:
                parts.append(torch.flip(out.narrow(ax, n - hi, hi), [ax]))
            out = torch.cat(parts, ax)
        return out
    out = x
    for ax, (lo, hi) in enumerate(p):     # REFLECT: edge excluded
        n = out.shape[ax]
        parts = []
        if lo:
            parts.append(torch.flip(out.narrow(ax, 1, lo), [ax]))
        parts.append(out)
        if hi:
            parts.append(torch.flip(out.narrow(ax, n - 1 - hi, hi), [ax]))
        out = torch.cat(parts, ax)
    return out


op("mirror_pad")(lambda ins, i, t, b, d: _pad(ins, [2 if (i and i[0] == 1) else 1], t, b, d))
op("clipbyvalue")(lambda ins, i, t, b, d: torch.clamp(ins[0], t[0], t[1]))


@op("clipbynorm")
def _clipbynorm(ins, i, t, b, d):
    x = ins[0]
    c = t[0] if t else float(ins[1])
    dims = [k % x.dim() for k in i] if i else list(range(x.dim()))
    n = torch.sqrt((x * x).sum(dims, keepdim=True))
    return torch.where(n > c, x * (c / n.clamp_min(1e-30)), x)


op("clipbynorm_bp")(bp_of("clipbynorm", 1))


@op("clipbyavgnorm")
def _clipbyavgnorm(ins, i, t, b, d):
    x = ins[0]
    c = t[0]
    dims = [k % x.dim() for k in i] if i else list(range(x.dim()))
    cnt = 1
    for k in dims:
        cnt *= x.shape[k]
    n = torch.sqrt((x * x).sum(dims, keepdim=True)) / cnt
    return torch.where(n > c, x * (c / n.clamp_min(1e-30)), x)


op("clipbyavgnorm_bp")(bp_of("clipbyavgnorm", 1))


@op("clip_by_global_norm")
def _clip_by_global_norm(ins, i, t, b, d):
    c = t[0]
    g = torch.sqrt(sum((x.float() ** 2).sum() for x in ins))
    s = torch.where(g > c, c / g, torch.ones_like(g))
    return tuple((x * s.to(x.dtype)) for x in ins) + (g,)


def _cum(fn):
    def f(ins, i, t, b, d):
        x = ins[0]
        excl, rev = (bool(i[0]) if i else False), (bool(i[1]) if len(i) > 1 else False)
        dims = list(i[2:]) if len(i) > 2 else ([int(v) for v in ins[1].tolist()] if len(ins) > 1 else [0])
        out = x
        for ax in dims:
            if rev:
                out = torch.flip(out, [ax])
            r = fn(out, ax)
            if excl:
                ident = 0 if fn is torch.cumsum else 1
                r = torch.cat([torch.full_like(r.narrow(ax, 0, 1), ident), r.narrow(ax, 0, r.shape[ax] - 1)], ax)
            out = torch.flip(r, [ax]) if rev else r
        return out
    return f


op("cumsum")(_cum(torch.cumsum))
op("cumprod")(_cum(torch.cumprod))
op("cumsum_bp")(bp_of("cumsum", 1))
op("cumprod_bp")(bp_of("cumprod", 1))
op("invert_permutation")(lambda ins, i, t, b, d: torch.argsort(ins[0].long()).to(ins[0].dtype))
op("mergeadd")(lambda ins, i, t, b, d: torch.stack(list(ins)).sum(0))
op("mergeavg")(lambda ins, i, t, b, d: torch.stack(list(ins)).mean(0))
op("mergemax")(lambda ins, i, t, b, d: torch.stack(list(ins)).amax(0))
op("mergemaxindex")(lambda ins, i, t, b, d: torch.stack(list(ins)).argmax(0).to(dtype_arg(d, None, 0, torch.int32)))
op("mergeadd_bp")(lambda ins, i, t, b, d: tuple(ins[-1].clone() for _ in ins[:-1]))
op("mergeavg_bp")(lambda ins, i, t, b, d: tuple(ins[-1] / (len(ins) - 1) for _ in ins[:-1]))


@op("mergemax_bp")
def _mergemax_bp(ins, i, t, b, d):
    *xs, g = ins
    am = torch.stack(xs).argmax(0)
    return tuple(g * (am == k).to(g.dtype) for k in range(len(xs)))


@op("gather")
def _gather(ins, i, t, b, d):
    x = ins[0]
    axis = int(i[0]) if i else 0
    idx = ins[1].long() if len(ins) > 1 else torch.tensor(list(i[1:]), dtype=torch.int64, device=x.device)
    if idx.device.type != "meta" and idx.numel() and (int(idx.min()) < -x.shape[axis] or int(idx.max()) >= x.shape[axis]):
        raise IndexError("gather: index out of range")
    r = torch.index_select(x, axis, idx.reshape(-1))
    return r.reshape(list(x.shape[:axis]) + list(idx.shape) + list(x.shape[axis + 1:]))


@op("gather_nd")
def _gather_nd(ins, i, t, b, d):
    x, idx = ins[0], ins[1].long()
    k = idx.shape[-1]
    return x[tuple(idx[..., j] for j in range(k))]


def _scatter(kind):
    def f(ins, i, t, b, d):
        x, idx, upd = ins[0].clone(), ins[1].long(), ins[2]
        flat_idx = idx.reshape(-1)
        u = upd.reshape([flat_idx.numel()] + list(x.shape[1:])).to(x.dtype)
        if kind == "upd":
            x[flat_idx] = u
        elif kind == "add":
            x.index_add_(0, flat_idx, u)
        elif kind == "sub":
            x.index_add_(0, flat_idx, -u)
        else:
            ix = flat_idx.reshape([-1] + [1] * (x.dim() - 1)).expand_as(u)
            red = {"mul": "prod", "div": "prod", "max": "amax", "min": "amin"}[kind]
            x.scatter_reduce_(0, ix, (1.0 / u) if kind == "div" else u, reduce=red, include_self=True)
        return x
    return f


for _k in ("add", "sub", "mul", "div", "max", "min", "upd"):
    op("scatter_" + _k)(_scatter(_k))


@op("scatter_update")
def _scatter_update(ins, i, t, b, d):
    x, upd = ins[0].clone(), ins[1]
    opc, ndim = int(i[0]), int(i[1])
    dims = list(i[2:2 + ndim])
    idx = list(i[3 + ndim:])
    axis = [k for k in range(x.dim()) if k not in [dd % x.dim() for dd in dims]][0] if ndim < x.dim() else 0
    fn = [lambda a, u: a + u, lambda a, u: a - u, lambda a, u: a * u, lambda a, u: a / u, lambda a, u: u - a,
          lambda a, u: u / a, lambda a, u: u][opc]
    for k, ix in enumerate(idx):
        sl = [slice(None)] * x.dim()
        sl[axis] = ix
        x[tuple(sl)] = fn(x[tuple(sl)], upd[k].reshape(x[tuple(sl)].shape))
    return x


def _scatter_nd(kind):
    def f(ins, i, t, b, d):
        if kind == "new":
            idx, upd, shape = ins[0].long(), ins[1], [int(v) for v in ins[2].tolist()]
            x = torch.zeros(shape, dtype=upd.dtype, device=upd.device)
        else:
            x, idx, upd = ins[0].clone(), ins[1].long(), ins[2]
        k = idx.shape[-1]
        key = tuple(idx[..., j] for j in range(k))
        if kind == "update":
            x[key] = upd.to(x.dtype)
        else:
            x.index_put_(key, (-upd if kind == "sub" else upd).to(x.dtype), accumulate=True)
        return x
    return f


op("scatter_nd")(_scatter_nd("new"))
op("scatter_nd_add")(_scatter_nd("add"))
op("scatter_nd_sub")(_scatter_nd("sub"))
op("scatter_nd_update")(_scatter_nd("update"))


@op("depth_to_space")
def _d2s(ins, i, t, b, d):
    x, bs, nhwc = ins[0], int(i[0]), (len(i) > 1 and i[1] == 1)
    if nhwc:
        x = x.permute(0, 3, 1, 2)
    n, c, h, w = x.shape
    r = x.reshape(n, bs, bs, c // (bs * bs), h, w).permute(0, 3, 4, 1, 5, 2).reshape(n, c // (bs * bs), h * bs, w * bs)
    return r.permute(0, 2, 3, 1) if nhwc else r


@op("space_to_depth")
def _s2d(ins, i, t, b, d):
    x, bs, nhwc = ins[0], int(i[0]), (len(i) > 1 and i[1] == 1)
    if nhwc:
        x = x.permute(0, 3, 1, 2)
    n, c, h, w = x.shape
    r = x.reshape(n, c, h // bs, bs, w // bs, bs).permute(0, 3, 5, 1, 2, 4).reshape(n, c * bs * bs, h // bs, w // bs)
    return r.permute(0, 2, 3, 1) if nhwc else r


def _s2b_nd(x, block, pads):
    m = len(block)
    flat = []
    for lo, hi in reversed(pads):
        flat += [lo, hi]
    nrest = x.dim() - 1 - m
    x = F.pad(x, [0, 0] * nrest + flat)
    n = x.shape[0]
    sp = x.shape[1:1 + m]
    rest = list(x.shape[1 + m:])
    shp = [n]
    for s, bl in zip(sp, block):
        shp += [s // bl, bl]
    x = x.reshape(shp + rest)
    perm = [2 + 2 * k for k in range(m)] + [0] + [1 + 2 * k for k in range(m)] + list(range(1 + 2 * m, 1 + 2 * m + len(rest)))
    x = x.permute(*perm)
    return x.reshape([n * math.prod(block)] + [s // bl for s, bl in zip(sp, block)] + rest)


def _b2s_nd(x, block, crops):
    m = len(block)
    nb = math.prod(block)
    n = x.shape[0] // nb
    sp = list(x.shape[1:1 + m])
    rest = list(x.shape[1 + m:])
    x = x.reshape(list(block) + [n] + sp + rest)
    perm = [m]
    for k in range(m):
        perm += [m + 1 + k, k]
    perm += list(range(2 * m + 1, 2 * m + 1 + len(rest)))
    x = x.permute(*perm).reshape([n] + [s * bl for s, bl in zip(sp, block)] + rest)
    key = [slice(None)] + [slice(lo, x.shape[1 + k] - hi) for k, (lo, hi) in enumerate(crops)]
    return x[tuple(key)]


op("space_to_batch")(lambda ins, i, t, b, d: _s2b_nd(ins[0], [int(i[0])] * 2, ins[1].long().reshape(-1, 2).tolist()))
op("batch_to_space")(lambda ins, i, t, b, d: _b2s_nd(ins[0], [int(i[0])] * 2, ins[1].long().reshape(-1, 2).tolist()))
op("space_to_batch_nd")(lambda ins, i, t, b, d: _s2b_nd(ins[0], [int(v) for v in ins[1].tolist()], ins[2].long().reshape(-1, 2).tolist()))
op("batch_to_space_nd")(lambda ins, i, t, b, d: _b2s_nd(ins[0], [int(v) for v in ins[1].tolist()], ins[2].long().reshape(-1, 2).tolist()))


@op("dynamic_partition")
def _dynamic_partition(ins, i, t, b, d):
    x, p = ins[0], ins[1].long()
    n = int(i[0])
    return tuple(x[p == k] for k in range(n))


@op("dynamic_partition_bp")
def _dynamic_partition_bp(ins, i, t, b, d):
    x, p = ins[0], ins[1].long()
    n = int(i[0])
    gs = ins[2:2 + n]
    out = torch.zeros_like(x)
    for k in range(n):
        out[p == k] = gs[k].to(out.dtype)
    return out, torch.zeros_like(p)


@op("dynamic_stitch")
def _dynamic_stitch(ins, i, t, b, d):
    n = len(ins) // 2
    idx, data = ins[:n], ins[n:]
    size = max(int(ix.max()) for ix in idx if ix.numel()) + 1
    rest = list(data[0].shape[idx[0].dim():])
    out = torch.zeros([size] + rest, dtype=data[0].dtype, device=data[0].device)
    for ix, dt_ in zip(idx, data):
        out[ix.long().reshape(-1)] = dt_.reshape([-1] + rest)
    return out


@op("hashcode")
def _hashcode(ins, i, t, b, d):
    x = ins[0].reshape(-1)
    v = x.view(torch.int32 if x.element_size() == 4 else torch.int64 if x.element_size() == 8 else torch.int16).long() \
        if x.is_floating_point() else x.long()
    h = torch.zeros((), dtype=torch.int64)
    p = torch.ones((), dtype=torch.int64)
    # Java-style polynomial hash (31^k weights, wrapped to 64 bit), evaluated vectorised
    w = torch.pow(torch.tensor(31, dtype=torch.int64), torch.arange(v.numel() - 1, -1, -1, dtype=torch.int64) % 62)
    return (v.cpu() * w).sum().reshape(())


op("histogram")(lambda ins, i, t, b, d: torch.histc(ins[0].float(), bins=int(i[0]), min=float(ins[0].min()), max=float(ins[0].max())).long())


@op("histogram_fixed_width")
def _hfw(ins, i, t, b, d):
    x, rng = ins[0].float(), ins[1].float()
    n = int(ins[2]) if len(ins) > 2 else int(i[0])
    lo, hi = float(rng[0]), float(rng[1])
    idx = torch.clamp(((x - lo) / (hi - lo) * n).floor().long(), 0, n - 1)
    return torch.bincount(idx.reshape(-1), minlength=n)


@op("standardize")
def _standardize(ins, i, t, b, d):
    x = ins[0]
    dims = _shape_in(ins, i)
    m = x.mean(dims, keepdim=True)
    s = x.std(dims, unbiased=False, keepdim=True)
    return (x - m) / s.clamp_min(1e-30)


op("standardize_bp")(bp_of("standardize", 1))
op("Floor")(lambda ins, i, t, b, d: torch.floor(ins[0]))
op("rint")(lambda ins, i, t, b, d: torch.round(ins[0]))
op("square")(lambda ins, i, t, b, d: ins[0] * ins[0])
op("Log1p")(lambda ins, i, t, b, d: torch.log1p(ins[0]))
op("stop_gradient")(lambda ins, i, t, b, d: ins[0].detach().clone())
op("roll")(lambda ins, i, t, b, d: torch.roll(ins[0], int(i[0]) if i else int(ins[1]),
                                               (list(i[1:]) or None) if i else ([int(v) for v in ins[2].tolist()] if len(ins) > 2 else None))
           if (len(i) > 1 or len(ins) > 2) else torch.roll(ins[0].reshape(-1), int(i[0]) if i else int(ins[1])).reshape(ins[0].shape))
op("zero_fraction")(lambda ins, i, t, b, d: (ins[0] == 0).float().mean())


# ----------------------------------------------------------------------------- datatypes
op("cast")(lambda ins, i, t, b, d: ins[0].to(dtype_arg(d, i)))
op("to_double")(lambda ins, i, t, b, d: ins[0].double())
op("to_float16")(lambda ins, i, t, b, d: ins[0].half())
op("to_float32")(lambda ins, i, t, b, d: ins[0].float())
op("to_int32")(lambda ins, i, t, b, d: ins[0].int())
op("to_int64")(lambda ins, i, t, b, d: ins[0].long())
op("to_uint32")(lambda ins, i, t, b, d: ins[0].long() & 0xFFFFFFFF)
op("to_uint64")(lambda ins, i, t, b, d: ins[0].long())


@op("bitcast")
def _bitcast(ins, i, t, b, d):
    x = ins[0].contiguous()
    dt = dtype_arg(d, i)
    es_in, es_out = x.element_size(), torch.empty(0, dtype=dt).element_size()
    if es_in == es_out:
        return x.view(dt)
    if es_in > es_out:
        return x.reshape(-1).view(dt).reshape(list(x.shape) + [es_in // es_out])
    return x.reshape(-1).view(dt).reshape(list(x.shape[:-1]))


@op("min_max_datatype")
def _min_max_datatype(ins, i, t, b, d):
    dt = _DT_ID[int(i[0])]
    mn = int(i[1]) == 0
    info = torch.finfo(dt) if dt.is_floating_point else torch.iinfo(dt)
    return torch.tensor(info.min if mn else info.max, dtype=dt)


# ----------------------------------------------------------------------------- boolean
def _scalar_cmp(fn):
    return lambda ins, i, t, b, d: torch.tensor(bool(fn(ins[0].reshape(-1)[0], ins[1].reshape(-1)[0] if len(ins) > 1 else t[0])))


op("lt_scalar")(_scalar_cmp(lambda a, c: a < c))
op("gt_scalar")(_scalar_cmp(lambda a, c: a > c))
op("lte_scalar")(_scalar_cmp(lambda a, c: a <= c))
op("gte_scalar")(_scalar_cmp(lambda a, c: a >= c))
op("eq_scalar")(_scalar_cmp(lambda a, c: a == c))
op("neq_scalar")(_scalar_cmp(lambda a, c: a != c))
op("boolean_not")(lambda ins, i, t, b, d: ins[0] == 0)
op("is_non_decreasing")(lambda ins, i, t, b, d: torch.tensor(bool((ins[0].reshape(-1)[1:] >= ins[0].reshape(-1)[:-1]).all())))
op("is_strictly_increasing")(lambda ins, i, t, b, d: torch.tensor(bool((ins[0].reshape(-1)[1:] > ins[0].reshape(-1)[:-1]).all())))
op("is_numeric_tensor")(lambda ins, i, t, b, d: torch.tensor(ins[0].dtype != torch.bool))
op("select")(lambda ins, i, t, b, d: torch.where(
    (ins[0] != 0).reshape(list(ins[0].shape) + [1] * (ins[1].dim() - ins[0].dim())) if ins[0].dim() < ins[1].dim() else ins[0] != 0,
    ins[1], ins[2]))


@op("Where")
def _where(ins, i, t, b, d):
    c = ins[0] != 0
    if len(ins) == 3:
        cc = c.reshape(list(c.shape) + [1] * (ins[1].dim() - c.dim())) if c.dim() < ins[1].dim() else c
        return torch.where(cc, ins[1], ins[2])
    return torch.nonzero(c)


@op("where_np")
def _where_np(ins, i, t, b, d):
    c = ins[0] != 0
    if len(ins) == 3:
        return torch.where(c, ins[1], ins[2])
    return tuple(torch.nonzero(c, as_tuple=True))


@op("choose")
def _choose(ins, i, t, b, d):
    x = ins[0]
    mode = int(i[0])
    cmp = ins[1] if len(ins) > 1 else (t[0] if t else 0.0)
    m = LG._match(x, [cmp if not isinstance(cmp, torch.Tensor) else float(cmp.reshape(-1)[0]), t[1] if len(t) > 1 else 1e-5, mode])
    sel = x[m]
    return sel, torch.tensor(sel.numel(), dtype=torch.int64)


# ----------------------------------------------------------------------------- bitwise
op("bitwise_and")(lambda ins, i, t, b, d: ins[0] & ins[1])
op("bitwise_or")(lambda ins, i, t, b, d: ins[0] | ins[1])
op("bitwise_xor")(lambda ins, i, t, b, d: ins[0] ^ ins[1])
op("toggle_bits")(lambda ins, i, t, b, d: tuple(~x for x in ins) if len(ins) > 1 else ~ins[0])
op("shift_bits")(lambda ins, i, t, b, d: ins[0] << (ins[1] if len(ins) > 1 else i[0]))
op("rshift_bits")(lambda ins, i, t, b, d: ins[0] >> (ins[1] if len(ins) > 1 else i[0]))
op("cyclic_shift_bits")(lambda ins, i, t, b, d: LG._cshift(ins[0], ins[1] if len(ins) > 1 else torch.tensor(i[0]), True).to(ins[0].dtype))
op("cyclic_rshift_bits")(lambda ins, i, t, b, d: LG._cshift(ins[0], ins[1] if len(ins) > 1 else torch.tensor(i[0]), False).to(ins[0].dtype))


@op("bits_hamming_distance")
def _bits_hamming(ins, i, t, b, d):
    v = (ins[0] ^ ins[1]).long().reshape(-1)
    nbits = ins[0].element_size() * 8
    v = v & ((1 << nbits) - 1) if nbits < 64 else v
    cnt = torch.zeros_like(v)
    for s in range(nbits):
        cnt += (v >> s) & 1
    return cnt.sum()


# ----------------------------------------------------------------------------- blas
@op("matmul")
def _matmul(ins, i, t, b, d):
    x, y = ins[0], ins[1]
    ta, tb, tz = (list(i) + [0, 0, 0])[:3]
    alpha, beta = (list(t) + [1.0, 0.0])[:2] if t else (1.0, 0.0)
    if ta:
        x = x.transpose(-1, -2) if x.dim() > 1 else x
    if tb:
        y = y.transpose(-1, -2) if y.dim() > 1 else y
    z = torch.matmul(x, y)
    if alpha != 1.0:
        z = z * alpha
    if beta != 0.0 and len(ins) > 2:
        z = z + beta * ins[2]
    return z.transpose(-1, -2) if tz and z.dim() > 1 else z


def _unbatch(g, shape):
    """Sum a batched gradient back onto an operand whose leading (batch) dims were broadcast."""
    if list(g.shape) == list(shape):
        return g
    while g.dim() > len(shape):
        g = g.sum(0)
    for ax, (a, b_) in enumerate(zip(g.shape, shape)):
        if a != b_ and b_ == 1:
            g = g.sum(ax, keepdim=True)
    return g


@op("matmul_bp")
def _matmul_bp(ins, i, t, b, d):
    """Analytic: z = op(x) . op(y)  =>  d op(x) = g . op(y)^T,  d op(y) = op(x)^T . g  (two GEMMs, no forward re-run;
    reference generic/blas/matmul.cpp matmul_bp). Vector operands keep the autograd form."""
    x, y, g = ins[0], ins[1], ins[2]
    if x.dim() < 2 or y.dim() < 2:
        xr, yr = x.detach().clone().requires_grad_(True), y.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            z = _matmul([xr, yr], i, t, b, d)
        return torch.autograd.grad(z, [xr, yr], g.to(z.dtype))
    ta, tb, tz = (list(i) + [0, 0, 0])[:3]
    alpha = (list(t) + [1.0])[0] if t else 1.0
    g = g.to(x.dtype)
    if tz:
        g = g.transpose(-1, -2)
    if alpha != 1.0:
        g = g * alpha
    A = x.transpose(-1, -2) if ta else x
    B = y.transpose(-1, -2) if tb else y
    dA = torch.matmul(g, B.transpose(-1, -2))
    dB = torch.matmul(A.transpose(-1, -2), g)
    dx = dA.transpose(-1, -2) if ta else dA
    dy = dB.transpose(-1, -2) if tb else dB
    return _unbatch(dx, x.shape), _unbatch(dy, y.shape)


@op("tensormmul")
def _tensormmul(ins, i, t, b, d):
    na = int(i[0])
    ax_a = list(i[1:1 + na])
    nb = int(i[1 + na])
    ax_b = list(i[2 + na:2 + na + nb])
    return torch.tensordot(ins[0], ins[1], dims=(ax_a, ax_b))


op("tensormmul_bp")(bp_of("tensormmul", 2))
op("axpy")(lambda ins, i, t, b, d: (t[0] if t else 1.0) * ins[0] + ins[1])


@op("batched_gemm")
def _batched_gemm(ins, i, t, b, d):
    """inputs: alpha, beta, A_0..A_{n-1}, B_0..B_{n-1}; iArgs [transA, transB, M, N, K, ldA, ldB, ldC, batchSize]."""
    alpha, beta = ins[0], ins[1]
    n = int(i[8]) if len(i) > 8 else (len(ins) - 2) // 2
    A, B = torch.stack(list(ins[2:2 + n])), torch.stack(list(ins[2 + n:2 + 2 * n]))
    if i and i[0] in (1, 112):
        A = A.transpose(1, 2)
    if len(i) > 1 and i[1] in (1, 112):
        B = B.transpose(1, 2)
    a = alpha.reshape(-1)[0] if alpha.numel() == 1 else alpha.reshape(-1, 1, 1)
    C = torch.bmm(A, B) * a
    return tuple(C[k] for k in range(n))


op("batched_gemm_bp")(lambda ins, i, t, b, d: _batched_gemm_bp(ins, i))


def _batched_gemm_bp(ins, i):
    n = int(i[8]) if len(i) > 8 else (len(ins) - 2) // 3
    alpha = ins[0].reshape(-1)[0]
    A, B, G = torch.stack(list(ins[2:2 + n])), torch.stack(list(ins[2 + n:2 + 2 * n])), torch.stack(list(ins[2 + 2 * n:2 + 3 * n]))
    dA = torch.bmm(G, B.transpose(1, 2)) * alpha
    dB = torch.bmm(A.transpose(1, 2), G) * alpha
    return tuple(dA[k] for k in range(n)) + tuple(dB[k] for k in range(n))
