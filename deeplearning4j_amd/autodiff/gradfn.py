"""Gradient function of a SameDiff graph AS A GRAPH: ``createGradFunction`` appends, for every op on a path from the differentiated
variables to the loss, the ops that compute its input gradients (per-op ``doDiff`` rules), and names the results ``<var>-grad``.

Reference: nd4j-api autodiff/samediff/SameDiff.java:5013-5330 (createGradFunction: reverse topological walk, DifferentialFunction.doDiff
per op, gradient accumulation for variables with several consumers, ``-grad`` naming) and the ``doDiff`` bodies of
linalg/api/ops/impl/** which emit the libnd4j ``*_bp`` ops. Executing the gradient graph runs those ``_bp`` ops — i.e. the
backward platform helpers (HIP kernels) — with no autograd tape and no forward re-computation for the ops that have an analytic
rule here. Ops without a rule get a ``__vjp__`` node: a vector-Jacobian product of the single forward op (local re-execution of
that one op under autograd), so every graph differentiates and the rule table can grow op by op.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

RULES = {}


def rule(*names):
    def deco(fn):
        for n in names:
            RULES[n] = fn
        return fn
    return deco


# rule(sd, op, in_vars, out_vars, out_grads) -> list of gradient SDVariables (or None) aligned with op.inputs
def _args(o):
    return dict(iArgs=o.iArgs, tArgs=o.tArgs, bArgs=o.bArgs, dArgs=o.dArgs)


def _same_args_bp(n_out, n_grad_in=None):
    """``<op>_bp(inputs..., gradOut; same args) -> n_out gradients`` for the first ``n_grad_in`` inputs."""
    def fn(sd, o, ins, outs, gs):
        k = len(ins) if n_grad_in is None else n_grad_in
        res = sd._op(o.opName + "_bp", list(ins) + [gs[0]], nOut=n_out, name=o.name + "_bp", **_args(o))
        return list(res)[:k] + [None] * (len(ins) - min(k, n_out))
    return fn


for _n in ("add", "subtract", "multiply", "divide", "reversesubtract", "reversedivide", "realdiv", "squaredsubtract", "maximum", "minimum",
           "Pow", "floormod", "floordiv", "mod"):
    RULES[_n] = _same_args_bp(2)
for _n in ("relu", "sigmoid", "tanh", "elu", "selu", "softplus", "softsign", "cube", "hardtanh", "hardsigmoid", "lrelu", "relu6",
           "rationaltanh", "rectifiedtanh", "thresholdedrelu", "softmax", "log_softmax", "identity"):
    RULES[_n] = _same_args_bp(1, 1)
for _n in ("reduce_sum", "reduce_mean", "reduce_max", "reduce_min", "reduce_prod", "reduce_norm1", "reduce_norm2", "reduce_sqnorm",
           "reduce_norm_max", "reduce_variance", "reduce_stdev"):
    RULES[_n] = _same_args_bp(1, 1)
RULES["matmul"] = _same_args_bp(2)
RULES["tensormmul"] = _same_args_bp(2)
RULES["maxpool2d"] = _same_args_bp(1, 1)
RULES["avgpool2d"] = _same_args_bp(1, 1)
RULES["pnormpool2d"] = _same_args_bp(1, 1)
RULES["lrn"] = _same_args_bp(1, 1)
RULES["upsampling2d"] = _same_args_bp(1, 1)
RULES["cumsum"] = _same_args_bp(1, 1)
RULES["tile"] = _same_args_bp(1, 1)
RULES["slice"] = _same_args_bp(1, 1)
RULES["strided_slice"] = _same_args_bp(1, 1)


@rule("conv2d", "depthwise_conv2d", "deconv2d", "conv1d", "conv3dnew", "xw_plus_b", "biasadd", "prelu")
def _all_inputs_bp(sd, o, ins, outs, gs):
    res = sd._op(o.opName + "_bp", list(ins) + [gs[0]], nOut=len(ins), name=o.name + "_bp", **_args(o))
    return list(res)


@rule("layer_norm")
def _layer_norm(sd, o, ins, outs, gs):
    res = sd._op("layer_norm_bp", list(ins) + [gs[0]], nOut=len(ins), name=o.name + "_bp", **_args(o))
    return list(res)


@rule("batchnorm")
def _batchnorm(sd, o, ins, outs, gs):
    # batchnorm_bp(x, mean, var, [gamma], [beta], gradO) -> dx, dmean, dvar, [dgamma], [dbeta]
    res = sd._op("batchnorm_bp", list(ins) + [gs[0]], nOut=len(ins), name=o.name + "_bp", **_args(o))
    return list(res)


@rule("reshape", "squeeze", "expand_dims", "flatten_2d")
def _reshape(sd, o, ins, outs, gs):
    return [sd._op("reshapeas", [gs[0], ins[0]], name=o.name + "_bp")[0]] + [None] * (len(ins) - 1)


@rule("permute", "transpose")
def _permute(sd, o, ins, outs, gs):
    if o.opName == "transpose" and not o.iArgs:
        return [sd._op("transpose", [gs[0]], name=o.name + "_bp")[0]] + [None] * (len(ins) - 1)
    if len(ins) > 1 or not o.iArgs:
        return None                               # permutation given as an input array: generic VJP
    perm = [int(p) for p in o.iArgs]
    inv = [0] * len(perm)
    for k, p in enumerate(perm):
        inv[p % len(perm)] = k
    return [sd._op("permute", [gs[0]], iArgs=inv, name=o.name + "_bp")[0]]


@rule("concat")
def _concat(sd, o, ins, outs, gs):
    res = sd._op("concat_bp", list(ins) + [gs[0]], nOut=len(ins), name=o.name + "_bp", **_args(o))
    return list(res)


@rule("mergeadd")
def _mergeadd(sd, o, ins, outs, gs):
    return [gs[0]] * len(ins)


@rule("stop_gradient", "shape_of", "size", "rank", "argmax", "argmin", "onehot", "zeros_as", "ones_as")
def _no_grad(sd, o, ins, outs, gs):
    return [None] * len(ins)


@rule("softmax_cross_entropy_loss_with_logits")
def _sce_logits(sd, o, ins, outs, gs):
    # grad op returns dL/dlogits, dL/dlabels of the (per-example) loss; chain with the incoming gradient
    dl, dlab = sd._op("softmax_cross_entropy_loss_with_logits_grad", list(ins), nOut=2, name=o.name + "_grad", **_args(o))
    g = sd._op("expand_dims", [gs[0]], iArgs=[-1], name=o.name + "_gexp")[0]
    return [sd._op("multiply", [dl, g], name=o.name + "_gl")[0], None]


# ---- losses: <name>_grad(predictions, weights, labels) -> d/d(predictions, weights, labels) of the reduced loss, chained with the
# incoming gradient (a scalar for the reduced forms, one value per example for reduction mode "none")
def _loss_rule(sd, o, ins, outs, gs):
    if len(ins) != 3:
        return None
    res = sd._op(o.opName + "_grad", list(ins), nOut=3, name=o.name + "_grad", **_args(o))
    return [sd._op("multiply", [r, gs[0]], name="%s_g%d" % (o.name, k))[0] for k, r in enumerate(res)]


for _n in ("softmax_cross_entropy_loss", "sigm_cross_entropy_loss", "mean_sqerr_loss", "absolute_difference_loss", "huber_loss", "log_loss",
           "hinge_loss", "log_poisson_loss", "cosine_distance_loss", "mean_pairwssqerr_loss"):
    RULES[_n] = _loss_rule


def _static_shape(sd, v):
    a = sd._arrays.get(v.name)
    if a is not None:
        return list(a.shape)
    sh = getattr(v, "shape", None)
    return list(sh) if sh is not None and all(s is not None and s >= 0 for s in sh) else None


@rule("embedding_lookup", "gather")
def _rows_lookup(sd, o, ins, outs, gs):
    # table [V, D] indexed by integer ids along axis 0: d(table) = segment sum of the incoming rows by id (unsorted_segment_sum has
    # a platform kernel); ids get no gradient
    if len(ins) < 2 or (o.opName == "gather" and o.iArgs and int(o.iArgs[0]) != 0):
        return None
    shp = _static_shape(sd, ins[0])
    if shp is None or len(shp) != 2:
        return None
    g2 = sd._op("reshape", [gs[0]], iArgs=[-1, int(shp[1])], name=o.name + "_g2d")[0]
    i1 = sd._op("reshape", [ins[1]], iArgs=[-1], name=o.name + "_ids")[0]
    dt = sd._op("unsorted_segment_sum", [g2, i1], iArgs=[int(shp[0])], name=o.name + "_bp")[0]
    return [dt] + [None] * (len(ins) - 1)


@rule("dot_product_attention_v2")
def _attention_v2(sd, o, ins, outs, gs):
    # dot_product_attention_v2_bp(q, v, k, dL/dout, masks...) -> dq, dv, dk (the fused flash backward on the GPU)
    if len(ins) < 3:
        return None
    res = sd._op("dot_product_attention_v2_bp", list(ins[:3]) + [gs[0]] + list(ins[3:]), nOut=3, name=o.name + "_bp", **_args(o))
    return list(res) + [None] * (len(ins) - 3)


def _lstm_layer(sd, o, ins, outs, gs):
    # lstmLayer_bp(inputs..., one gradient per returned output) -> gradients of the floating-point inputs in input order; the
    # sequence-length input (bArgs[1]) is integral and gets none
    bb = list(o.bArgs) + [False] * 8
    gl = [g if g is not None else sd._op("zeros_as", [ov], name=ov.name + "-zgrad")[0] for g, ov in zip(gs, outs)]
    pos = [0, 1, 2]
    k = 3
    for flag, differentiable in ((bb[0], True), (bb[1], False), (bb[2], True), (bb[3], True), (bb[4], True)):
        if flag:
            if differentiable:
                pos.append(k)
            k += 1
    if k != len(ins):
        return None
    res = sd._op("lstmLayer_bp", list(ins) + gl, nOut=len(pos), name=o.name + "_bp", **_args(o))
    out = [None] * len(ins)
    for p_, r in zip(pos, res):
        out[p_] = r
    return out


_lstm_layer.multi = True          # takes the gradients of all outputs (absent ones as zeros)
RULES["lstmLayer"] = _lstm_layer


# ---- legacy (opType, opNum) ops: derivative transforms of the same families or closed forms
def _lg(sd, family, op_name, inputs, name, **kw):
    from .samediff import OpType
    from ..nd4j.ops import legacy as LG
    fam_type = {"transform_same": OpType.TRANSFORM_SAME, "transform_strict": OpType.TRANSFORM_STRICT, "transform_float": OpType.TRANSFORM_FLOAT,
                "scalar": OpType.SCALAR}[family]
    return sd._op(op_name.lower(), list(inputs), name=name, opType=fam_type, opNum=LG.op_num(family, op_name), **kw)[0]


def _mul(sd, a, b, name):
    return sd._op("multiply", [a, b], name=name)[0]


_STRICT_DERIVATIVE = {"tanh": "TanhDerivative", "sigmoid": "SigmoidDerivative", "hardtanh": "HardTanhDerivative", "softsign": "SoftSignDerivative",
                      "selu": "SELUDerivative", "hardsigmoid": "HardSigmoidDerivative", "rationaltanh": "RationalTanhDerivative",
                      "rectifiedtanh": "RectifiedTanhDerivative", "swish": "SwishDerivative", "logsigmoid": "LogSigmoidDerivative",
                      "cube": "CubeDerivative", "gelu": "GELUDerivative", "precisegelu": "PreciseGELUDerivative", "mish": "MishDerivative"}


def _legacy_rule(sd, o, ins, outs, gs):
    """Gradient ops of a legacy transform / scalar / pairwise / reduce node, or None (-> generic VJP)."""
    from .samediff import OpType
    from ..nd4j.ops import legacy as LG
    t, nm, g, x = o.opType, o.opName.lower(), gs[0], ins[0]
    if t == OpType.TRANSFORM_STRICT:
        fam = LG.TRANSFORM_STRICT.get(o.opNum, "")
        if fam.lower() != nm:
            return None
        if nm in _STRICT_DERIVATIVE:
            try:
                d = _lg(sd, "transform_strict", _STRICT_DERIVATIVE[nm], [x], o.name + "_d")
            except KeyError:
                return None
            return [_mul(sd, g, d, o.name + "_bp")]
        if nm == "exp":
            return [_mul(sd, g, outs[0], o.name + "_bp")]
        if nm == "log":
            return [sd._op("divide", [g, x], name=o.name + "_bp")[0]]
        if nm == "softplus":
            return [_mul(sd, g, _lg(sd, "transform_strict", "Sigmoid", [x], o.name + "_d"), o.name + "_bp")]
        if nm == "sin":
            return [_mul(sd, g, _lg(sd, "transform_strict", "Cosine", [x], o.name + "_d"), o.name + "_bp")]
        if nm == "cosine":
            return [_lg(sd, "transform_same", "Neg", [_mul(sd, g, _lg(sd, "transform_strict", "Sin", [x], o.name + "_d"), o.name + "_m")], o.name + "_bp")]
        return None
    if t == OpType.TRANSFORM_SAME:
        fam = LG.TRANSFORM_SAME.get(o.opNum, "").lower()
        if fam != nm:
            return None
        if nm == "neg":
            return [_lg(sd, "transform_same", "Neg", [g], o.name + "_bp")]
        if nm == "square":
            return [_lg(sd, "scalar", "Multiply", [_mul(sd, g, x, o.name + "_m")], o.name + "_bp", tArgs=[2.0])]
        if nm == "abs":
            return [_mul(sd, g, _lg(sd, "transform_same", "Sign", [x], o.name + "_d"), o.name + "_bp")]
        if nm in ("identity", "copy"):
            return [g]
        return None
    if t == OpType.TRANSFORM_FLOAT:
        if LG.TRANSFORM_FLOAT.get(o.opNum, "").lower() == nm == "sqrt":
            q = sd._op("divide", [g, outs[0]], name=o.name + "_q")[0]
            return [_lg(sd, "scalar", "Multiply", [q], o.name + "_bp", tArgs=[0.5])]
        return None
    if t == OpType.SCALAR and len(ins) == 1:
        fam = LG.SCALAR.get(o.opNum, "").lower() if hasattr(LG, "SCALAR") else ""
        sc = o.scalar if o.scalar is not None else (o.tArgs[0] if o.tArgs else None)
        if fam != nm or sc is None or isinstance(sc, torch.Tensor):
            return None
        sc = float(sc)
        if nm in ("add", "subtract"):
            return [g]
        if nm == "multiply":
            return [_lg(sd, "scalar", "Multiply", [g], o.name + "_bp", tArgs=[sc])]
        if nm == "divide" and sc != 0.0:
            return [_lg(sd, "scalar", "Multiply", [g], o.name + "_bp", tArgs=[1.0 / sc])]
        if nm == "reversesubtract":
            return [_lg(sd, "transform_same", "Neg", [g], o.name + "_bp")]
        return None
    if t == OpType.PAIRWISE and len(ins) == 2:
        fam = LG.PAIRWISE.get(o.opNum, "").lower()
        if fam == nm and nm in ("add", "subtract", "multiply", "divide"):
            return list(sd._op(nm + "_bp", [ins[0], ins[1], g], nOut=2, name=o.name + "_bp"))
        return None
    if t in (OpType.REDUCE_SAME, OpType.REDUCE_FLOAT) and len(ins) == 1:
        table = LG.REDUCE_SAME if t == OpType.REDUCE_SAME else LG.REDUCE_FLOAT
        fam = table.get(o.opNum, "").lower()
        custom = {"sum": "reduce_sum_bp", "mean": "reduce_mean_bp", "max": "reduce_max_bp", "min": "reduce_min_bp", "norm2": "reduce_norm2_bp",
                  "norm1": "reduce_norm1_bp", "squarednorm": "reduce_sqnorm_bp", "prod": "reduce_prod_bp"}.get(nm)
        if fam != nm or custom is None:
            return None
        return [sd._op(custom, [x, g], iArgs=[int(d_) for d_ in (o.dimensions or [])], name=o.name + "_bp")[0]]
    return None


class GradFunction:
    """The differentiated graph: ``sd`` holds forward + backward ops, ``grads[name]`` is the variable holding dLoss/dname."""

    def __init__(self, sd, grads: Dict[str, str], vjp_ops: List[str], rule_ops: List[str]):
        self.sd, self.grads, self.vjp_ops, self.rule_ops = sd, grads, vjp_ops, rule_ops

    def gradient_names(self):
        return dict(self.grads)

    def execute(self, placeholders, wrt: Sequence[str], listeners=(), at=None, training=False):
        names = [self.grads[n] for n in wrt if n in self.grads]
        with torch.no_grad():
            res = self.sd._run_graph(self.sd._prep_placeholders(placeholders), names + list(self.sd.lossVariables), training, listeners, at)
        out = {}
        for n in wrt:
            g = res.get(self.grads.get(n))
            if g is None:
                a = self.sd._arrays.get(n)
                g = torch.zeros_like(a) if a is not None else None
            out[n] = g
        return [res[l] for l in self.sd.lossVariables], out


def _topo(sd, need):
    order, seen = [], set()

    def visit(on):
        if on in seen:
            return
        seen.add(on)
        for i in sd.ops[on].inputs:
            p = sd._producer.get(i)
            if p is not None and p in need:
                visit(p)
        order.append(on)
    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
    for on in sorted(need):
        visit(on)
    return order


def create_grad_function(src, wrt: Optional[Sequence[str]] = None) -> GradFunction:
    """Build the gradient graph of ``src`` (left untouched) for its loss variables."""
    from .samediff import SameDiff, SameDiffOp, VariableType, OpType, _CONTROL
    if not src.lossVariables:
        raise RuntimeError("no loss variables set: call setLossVariables / markAsLoss")
    sd = SameDiff(device=src.device)
    src.invokeGraphOn(sd)
    sd._arrays = src._arrays                      # shared storage: the gradient graph always sees the current parameter values
    sd.lossVariables = list(src.lossVariables)
    wrt = list(wrt) if wrt is not None else src.trainableVariables()
    need = src._required_ops(sd.lossVariables)
    if any(src.ops[o].opName.lower() in _CONTROL for o in need):
        raise NotImplementedError("gradient graphs of control-flow (Switch / Merge / loops) use the session autograd path")
    # ops that lie on a path wrt -> loss
    reach = set(wrt)
    order = _topo(src, need)
    active = []
    for on in order:
        o = src.ops[on]
        if any(i in reach for i in o.inputs):
            reach.update(o.outputs)
            active.append(on)
    contrib: Dict[str, list] = {}
    for l in sd.lossVariables:
        contrib.setdefault(l, []).append(sd._op("ones_as", [sd.variables[l]], name=l + "-seed")[0])
    vjp_ops, rule_ops = [], []

    def total(name):
        cs = contrib.get(name)
        if not cs:
            return None
        if len(cs) == 1:
            return cs[0]
        v = sd._op("mergeadd", cs, name=name + "-gradsum")[0]
        contrib[name] = [v]
        return v

    for on in reversed(active):
        o = src.ops[on]
        gs = [total(n) for n in o.outputs]
        if all(g is None for g in gs):
            continue
        ins = [sd.variables[n] for n in o.inputs]
        outs = [sd.variables[n] for n in o.outputs]
        res = None
        r = RULES.get(o.opName) if o.opType in (OpType.CUSTOM, OpType.BOOLEAN, OpType.SHAPE) else _legacy_rule
        if r is not None and (getattr(r, "multi", False) or (gs[0] is not None and all(g is None for g in gs[1:]))):
            res = r(sd, o, ins, outs, gs)
            if res is not None:
                res = list(res) + [None] * (len(ins) - len(res))
                rule_ops.append(on)
        if res is None:
            # generic vector-Jacobian product of this ONE op: inputs + one gradient per output (zeros where absent)
            gl = [g if g is not None else sd._op("zeros_as", [ov], name=ov.name + "-zgrad")[0] for g, ov in zip(gs, outs)]
            vo = SameDiffOp(sd._unique(o.name + "_vjp"), "__vjp__", [v.name for v in ins] + [g.name for g in gl],
                            [sd._unique(o.name + "_vjp_d%d" % k) for k in range(len(ins))], opType=OpType.CUSTOM)
            vo.fwd = o
            vo.sArgs = [o.name]
            sd._add_op(vo)
            res = [sd.variables[n] for n in vo.outputs]
            vjp_ops.append(on)
        for iv, g in zip(ins, res):
            if g is not None and iv.name in reach:
                contrib.setdefault(iv.name, []).append(g)
    grads = {}
    for n in wrt:
        g = total(n)
        if g is None:
            continue
        gname = n + "-grad"
        if gname in sd.variables and gname != g.name:
            gname = sd._unique(gname)
        fin = sd._op("identity", [g], name=gname, outNames=[gname])[0]
        grads[n] = fin.name
    return GradFunction(sd, grads, vjp_ops, rule_ops)


def exec_vjp(sd, vo, ins):
    """Runtime of a ``__vjp__`` node: re-run the forward op on detached leaves under autograd, return dL/dinput for every input."""
    from ..nd4j.ops import legacy as LG
    f = vo.fwd
    nin = len(f.inputs)
    xs, gs = ins[:nin], ins[nin:]
    leaves = [x.detach().clone().requires_grad_(True) if (isinstance(x, torch.Tensor) and x.is_floating_point()) else x for x in xs]
    with torch.enable_grad(), LG.force_torch():
        ys = sd._exec_op(f, leaves, allow_helpers=True)
    pairs = [(y, g) for y, g in zip(ys, gs) if isinstance(y, torch.Tensor) and y.requires_grad]
    req = [x for x in leaves if isinstance(x, torch.Tensor) and x.requires_grad]
    out = []
    if pairs and req:
        got = torch.autograd.grad([y for y, _ in pairs], req, [g.to(y.dtype).expand_as(y) for y, g in pairs], allow_unused=True)
        it = iter(got)
        for x in leaves:
            if isinstance(x, torch.Tensor) and x.requires_grad:
                g = next(it)
                out.append(g if g is not None else torch.zeros_like(x))
            else:
                out.append(torch.zeros((), device=sd.device) if not isinstance(x, torch.Tensor) else torch.zeros_like(x, dtype=torch.float32))
    else:
        out = [torch.zeros_like(x, dtype=x.dtype if x.is_floating_point() else torch.float32) for x in xs]
    return out
