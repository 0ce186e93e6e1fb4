"""Graph optimizer (reference: nd4j-api autodiff/samediff/optimize/{GraphOptimizer,OptimizationHelper} and
optimizations/{Constant,Identity,Shape,Unused,CuDNN}FunctionOptimizations).

Passes rewrite the SameDiff graph in place:
* ConstantFolding      – ops whose inputs are all CONSTANT are evaluated once and replaced by a constant
* IdentityRemoval      – identity / stop-gradient-free passthroughs are bypassed
* ShapeFolding         – shape_of / rank / size of statically known shapes become constants
* UnusedRemoval        – ops and constants that no requested output depends on are dropped
* Fusion (MI355X)      – primitive sub-graphs produced by TF/ONNX exporters are pattern-matched into one fused
  op so they hit a single kernel: TF layer-norm decomposition -> ``layer_norm``; erf/tanh GELU -> legacy GELU
  functors; ``softmax(q·kᵀ·s)·v`` -> ``dot_product_attention_v2`` (BERT speed depends on this).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

from .samediff import OpType, SameDiff, SameDiffOp, VariableType, _CONTROL


class GraphOptimizer:
    @staticmethod
    def defaultOptimizations():
        return [ConstantFolding(), ShapeFolding(), IdentityRemoval(), Fusion(), LinearFusion(), UnusedRemoval()]

    @staticmethod
    def optimize(sd: SameDiff, outputs: Optional[Sequence[str]] = None, optimizations=None, maxPasses=10) -> SameDiff:
        outputs = list(outputs) if outputs else (sd.outputs() + list(sd.lossVariables))
        opts = optimizations or GraphOptimizer.defaultOptimizations()
        for _ in range(maxPasses):
            changed = False
            for p in opts:
                changed = p.apply(sd, outputs) or changed
            if not changed:
                break
        return sd


# ----------------------------------------------------------------------------- helpers
def _remove_op(sd: SameDiff, o: SameDiffOp):
    for i in o.inputs:
        if o.name in sd._consumers.get(i, []):
            sd._consumers[i] = [c for c in sd._consumers[i] if c != o.name]
    for out in o.outputs:
        sd._producer.pop(out, None)
    sd.ops.pop(o.name, None)


def _replace_uses(sd: SameDiff, old: str, new: str, outputs: List[str]):
    for cn in list(sd._consumers.get(old, [])):
        c = sd.ops[cn]
        c.inputs = [new if i == old else i for i in c.inputs]
        sd._consumers[new].append(cn)
    sd._consumers[old] = []
    for k, n in enumerate(outputs):
        if n == old:
            outputs[k] = new
    sd.lossVariables = [new if n == old else n for n in sd.lossVariables]


def _to_constant(sd: SameDiff, name: str, value: torch.Tensor):
    v = sd.variables[name]
    v.variableType = VariableType.CONSTANT
    v.dtype, v.shape = value.dtype, list(value.shape)
    sd._arrays[name] = value


def _producer(sd, name) -> Optional[SameDiffOp]:
    p = sd._producer.get(name)
    return sd.ops[p] if p is not None else None


def _is(o: Optional[SameDiffOp], *names) -> bool:
    return o is not None and o.opName.lower() in names


def _const_val(sd, name) -> Optional[torch.Tensor]:
    v = sd.variables.get(name)
    if v is not None and v.variableType == VariableType.CONSTANT:
        return sd._arrays.get(name)
    return None


def _single_use(sd, name, outputs) -> bool:
    return len(sd._consumers.get(name, [])) == 1 and name not in outputs


class Optimization:
    def apply(self, sd, outputs) -> bool:
        raise NotImplementedError


class ConstantFolding(Optimization):
    MAX_ELEMENTS = 1 << 24
    _SKIP = {"randomuniform", "random_normal", "random_bernoulli", "random_exponential", "dropout", "random_shuffle", "random_crop",
             "random_gamma", "random_poisson", "random_multinomial", "print_variable", "assert"}

    def apply(self, sd, outputs):
        changed = False
        for o in list(sd.ops.values()):
            if o.opName.lower() in _CONTROL or o.opName.lower() in self._SKIP or o.opType == OpType.RANDOM or not o.inputs:
                continue
            if not all(_const_val(sd, i) is not None for i in o.inputs):
                continue
            try:
                with torch.no_grad():
                    outs = sd._exec_op(o, [sd._arrays[i] for i in o.inputs])
            except Exception:
                continue
            if not all(isinstance(t, torch.Tensor) and t.numel() <= self.MAX_ELEMENTS for t in outs):
                continue
            for n, t in zip(o.outputs, outs):
                _to_constant(sd, n, t.detach())
            _remove_op(sd, o)
            changed = True
        return changed


class ShapeFolding(Optimization):
    def apply(self, sd, outputs):
        changed = False
        for o in list(sd.ops.values()):
            if o.opName not in ("shape_of", "rank", "size"):
                continue
            shp = sd.variables[o.inputs[0]].getShape()
            if shp is None or (o.opName != "rank" and any(s is None or s < 0 for s in shp)):
                continue
            val = {"shape_of": torch.tensor(shp, dtype=torch.int64), "rank": torch.tensor(len(shp), dtype=torch.int32),
                   "size": torch.tensor(int(math.prod(shp)), dtype=torch.int64)}[o.opName]
            _to_constant(sd, o.outputs[0], val.to(sd.device))
            _remove_op(sd, o)
            changed = True
        return changed


class IdentityRemoval(Optimization):
    def apply(self, sd, outputs):
        changed = False
        for o in list(sd.ops.values()):
            if o.opName == "identity" and o.opType == OpType.CUSTOM and len(o.inputs) == 1 and not o.controlDeps:
                out = o.outputs[0]
                if out in outputs or out in sd.lossVariables:
                    continue                      # keep named graph outputs addressable
                _replace_uses(sd, out, o.inputs[0], outputs)
                _remove_op(sd, o)
                sd.variables.pop(out, None)
                changed = True
        return changed


class UnusedRemoval(Optimization):
    def apply(self, sd, outputs):
        need = sd._required_ops(list(outputs) + list(sd.lossVariables))
        changed = False
        for o in list(sd.ops.values()):
            if o.name not in need:
                for out in o.outputs:
                    sd.variables.pop(out, None)
                _remove_op(sd, o)
                changed = True
        used = set(outputs) | set(sd.lossVariables)
        for o in sd.ops.values():
            used.update(o.inputs)
            used.update(o.controlDeps)
        for n, v in list(sd.variables.items()):
            if v.variableType == VariableType.CONSTANT and n not in used:
                sd.variables.pop(n)
                sd._arrays.pop(n, None)
                changed = True
        return changed


class Fusion(Optimization):
    def apply(self, sd, outputs):
        changed = False
        for fn in (self._layer_norm_tf, self._gelu_erf, self._gelu_tanh, self._sdpa):
            for o in list(sd.ops.values()):
                if o.name in sd.ops:
                    changed = fn(sd, o, outputs) or changed
        return changed

    # y = x*m + (beta - mean*m),  m = rsqrt(var + eps) * gamma,  var = mean(sqdiff(x, mean)), mean = mean(x, axis, keepdims)
    def _layer_norm_tf(self, sd, add1: SameDiffOp, outputs):
        if not _is(add1, "add"):
            return False
        for a, b in (add1.inputs, add1.inputs[::-1]):
            mul1, sub = _producer(sd, a), _producer(sd, b)
            if not (_is(mul1, "multiply") and _is(sub, "subtract")):
                continue
            beta = sub.inputs[0]
            mul2 = _producer(sd, sub.inputs[1])
            if not _is(mul2, "multiply"):
                continue
            for x, m in (mul1.inputs, mul1.inputs[::-1]):
                if m not in mul2.inputs:
                    continue
                mean_name = [i for i in mul2.inputs if i != m]
                if len(mean_name) != 1:
                    continue
                mean = _producer(sd, mean_name[0])
                mm = _producer(sd, m)
                if not (_is(mean, "reduce_mean") and mean.inputs[0] == x and _is(mm, "multiply")):
                    continue
                for r, gamma in (mm.inputs, mm.inputs[::-1]):
                    rs = _producer(sd, r)
                    if not (_is(rs, "rsqrt") and rs.opType == OpType.TRANSFORM_FLOAT):
                        continue
                    addeps = _producer(sd, rs.inputs[0])
                    if not _is(addeps, "add"):
                        continue
                    for vname, ename in (addeps.inputs, addeps.inputs[::-1]):
                        eps = _const_val(sd, ename)
                        var = _producer(sd, vname)
                        if eps is None or eps.numel() != 1 or not _is(var, "reduce_mean"):
                            continue
                        sq = _producer(sd, var.inputs[0])
                        if not (_is(sq, "squaredsubtract") and set(sq.inputs) == {x, mean_name[0]}):
                            continue
                        axes = list(mean.iArgs)
                        xs = sd.variables[x].getShape()
                        if not axes or xs is None or sorted(a_ % len(xs) for a_ in axes) != [len(xs) - 1]:
                            continue
                        new = SameDiffOp(add1.name, "layer_norm", [x, gamma, beta], add1.outputs, iArgs=[-1], tArgs=[float(eps)], bArgs=[False])
                        _remove_op(sd, add1)
                        sd._add_op(new)
                        return True
        return False

    # 0.5 * x * (1 + erf(x / sqrt(2)))   (any association order of the three factors)
    def _gelu_erf(self, sd, top: SameDiffOp, outputs):
        if not _is(top, "multiply"):
            return False
        factors = self._flatten_mul(sd, top, outputs)
        if factors is None or len(factors) != 3:
            return False
        half = [f for f in factors if (c := _const_val(sd, f)) is not None and c.numel() == 1 and abs(float(c) - 0.5) < 1e-6]
        rest = [f for f in factors if f not in half]
        if len(half) != 1 or len(rest) != 2:
            return False
        for x, e in (rest, rest[::-1]):
            add = _producer(sd, e)
            if not _is(add, "add"):
                continue
            for one, erfn in (add.inputs, add.inputs[::-1]):
                c1 = _const_val(sd, one)
                erf = _producer(sd, erfn)
                if c1 is None or c1.numel() != 1 or abs(float(c1) - 1.0) > 1e-6 or not _is(erf, "erf"):
                    continue
                arg = _producer(sd, erf.inputs[0])
                ok = False
                if _is(arg, "divide") and arg.inputs[0] == x:
                    c = _const_val(sd, arg.inputs[1])
                    ok = c is not None and c.numel() == 1 and abs(float(c) - math.sqrt(2)) < 1e-4
                elif _is(arg, "multiply") and x in arg.inputs:
                    c = _const_val(sd, [i for i in arg.inputs if i != x][0]) if len(set(arg.inputs)) == 2 else None
                    ok = c is not None and c.numel() == 1 and abs(float(c) - 1 / math.sqrt(2)) < 1e-4
                if ok:
                    # exact erf GELU = x * Phi(x): expressed with the fused pairwise kernel chain collapsed to one custom op call
                    new = SameDiffOp(top.name, "gelu_erf_fused", [x], top.outputs, opType=OpType.UDF, opNum=0)
                    _ensure_gelu_udf()
                    _remove_op(sd, top)
                    sd._add_op(new)
                    return True
        return False

    # 0.5 * x * (1 + tanh(sqrt(2/pi) * (x + 0.044715 x^3)))
    def _gelu_tanh(self, sd, top: SameDiffOp, outputs):
        if not _is(top, "multiply"):
            return False
        factors = self._flatten_mul(sd, top, outputs)
        if factors is None or len(factors) != 3:
            return False
        half = [f for f in factors if (c := _const_val(sd, f)) is not None and c.numel() == 1 and abs(float(c) - 0.5) < 1e-6]
        rest = [f for f in factors if f not in half]
        if len(half) != 1 or len(rest) != 2:
            return False
        for x, e in (rest, rest[::-1]):
            add = _producer(sd, e)
            if not _is(add, "add"):
                continue
            for one, tn in (add.inputs, add.inputs[::-1]):
                c1 = _const_val(sd, one)
                th = _producer(sd, tn)
                if c1 is None or c1.numel() != 1 or abs(float(c1) - 1.0) > 1e-6 or not _is(th, "tanh"):
                    continue
                # accept any inner polynomial that references only x and constants ~ (0.79788, 0.044715)
                consts = self._collect_consts(sd, th.inputs[0], x, depth=6)
                if consts is not None and any(abs(v - 0.7978845608) < 1e-4 for v in consts) and any(abs(v - 0.044715) < 1e-5 for v in consts):
                    from ..nd4j.ops import legacy as LG
                    new = SameDiffOp(top.name, "precisegelu", [x], top.outputs, opType=OpType.TRANSFORM_STRICT,
                                     opNum=LG.op_num("transform_strict", "PreciseGELU"))
                    _remove_op(sd, top)
                    sd._add_op(new)
                    return True
        return False

    def _collect_consts(self, sd, name, x, depth) -> Optional[List[float]]:
        if name == x:
            return []
        c = _const_val(sd, name)
        if c is not None:
            return [float(c)] if c.numel() == 1 else None
        o = _producer(sd, name)
        if o is None or depth == 0 or not _is(o, "multiply", "add", "pow", "cube", "square"):
            return None
        out = []
        for i in o.inputs:
            r = self._collect_consts(sd, i, x, depth - 1)
            if r is None:
                return None
            out += r
        return out

    def _flatten_mul(self, sd, top, outputs) -> Optional[List[str]]:
        out = []
        for i in top.inputs:
            p = _producer(sd, i)
            if _is(p, "multiply") and _single_use(sd, i, outputs):
                out += p.inputs
            else:
                out.append(i)
        return out

    # matmul(softmax(matmul(q, k^T) [* scale] [+ (1 - mask) * -big]), v)  ->  dot_product_attention_v2 (single-output form: nothing
    # consumes the [.., Tq, Tk] matrices any more, so the GPU runs the fused flash kernels in both directions)
    def _sdpa(self, sd, mm2: SameDiffOp, outputs):
        if not _is(mm2, "matmul") or (mm2.iArgs and any(mm2.iArgs[:3])):
            return False
        sm = _producer(sd, mm2.inputs[0])
        if not (_is(sm, "softmax") and (not sm.iArgs or sm.iArgs[0] == -1) and _single_use(sd, mm2.inputs[0], outputs)):
            return False
        cur = _producer(sd, sm.inputs[0])
        key_mask = None
        if _is(cur, "add"):                       # additive padding mask
            for a_, m_ in (cur.inputs, cur.inputs[::-1]):
                src = self._additive_key_mask(sd, m_)
                if src is not None:
                    key_mask, mask_at_add, cur = src, m_, _producer(sd, a_)
                    break
            else:
                return False
        scale = 1.0
        if _is(cur, "multiply", "divide"):
            c = _const_val(sd, cur.inputs[1])
            if c is None or c.numel() != 1:
                return False
            scale = float(c) if _is(cur, "multiply") else 1.0 / float(c)
            cur = _producer(sd, cur.inputs[0])
        if not _is(cur, "matmul"):
            return False
        ia = (list(cur.iArgs) + [0, 0, 0])[:3]
        if ia[0] or not ia[1] or ia[2]:
            return False                          # need q · kᵀ
        q, k, v = cur.inputs[0], cur.inputs[1], mm2.inputs[1]
        if key_mask is not None:
            # only a PADDING mask qualifies: [B, 1, .., 1, Tk] at the add, from a [B, Tk] array. A [Tq, Tk] additive mask
            # broadcast over batch and heads (causal (1 - tril) * -1e9) must stay unfused: the fused op would read it as a
            # per-batch key mask.
            qs, ks, ms, ss = sd.inferShape(q), sd.inferShape(k), sd.inferShape(mask_at_add), sd.inferShape(key_mask)
            if qs is None or ks is None or ms is None or ss is None or len(qs) != len(ks) or len(qs) < 3 or len(ms) != len(qs):
                return False
            B, Tk = qs[0], ks[-2]            # a dynamic batch (-1) must be dynamic on both sides
            if list(ms) != [B] + [1] * (len(qs) - 2) + [Tk] or list(ss) != [B, Tk]:
                return False
        ins = [q, v, k] + ([key_mask] if key_mask is not None else [])
        new = SameDiffOp(mm2.name, "dot_product_attention_v2", ins, mm2.outputs, iArgs=[0, 1 if key_mask is not None else 0],
                         tArgs=[scale, 0.0], bArgs=[False, False])
        _remove_op(sd, mm2)
        sd._add_op(new)
        return True

    def _additive_key_mask(self, sd, name) -> Optional[str]:
        """``name`` = (1 - M) * c with c <= -1000 and M a reshape / expand_dims / cast image of a [batch, Tk] array: returns that array."""
        mul = _producer(sd, name)
        if not _is(mul, "multiply") or len(mul.inputs) != 2:
            return None
        for s_, c_ in (mul.inputs, mul.inputs[::-1]):
            c = _const_val(sd, c_)
            sub = _producer(sd, s_)
            if c is None or c.numel() != 1 or float(c) > -1000.0 or not _is(sub, "subtract", "rsub"):
                continue
            one = _const_val(sd, sub.inputs[0])
            if one is None or one.numel() != 1 or abs(float(one) - 1.0) > 1e-6:
                continue
            # walk up through pure shape / type changes only; the rank-2 array must be REACHED through at least one of them
            # or be the operand itself with the broadcast done by the subtract (checked against q / k by the caller)
            m = sub.inputs[1]
            for _ in range(6):
                shp = sd.inferShape(m)
                if shp is not None and len(shp) == 2:
                    return m
                pm = _producer(sd, m)
                if not _is(pm, "reshape", "expand_dims", "cast", "identity"):
                    break
                m = pm.inputs[0]
        return None


class LinearFusion(Optimization):
    """MI355X: dense sub-graphs -> one ``fused_dense`` op (ops/linear.py: GEMM with bias / activation / residual epilogues forward,
    activation-derivative + bias-gradient pass, dx GEMM and split-K dW backward):

      matmul(x, W) -> add / biasadd(., b)          => xw_plus_b(x, W, b)
      act(xw_plus_b(x, W, b))                      => fused_dense(x, W, b; act)         act in gelu / precise gelu / erf gelu / tanh / ...
      add(r, fused_dense | xw_plus_b)              => fused_dense(x, W, b, r; act)      residual connection (same shape, no broadcast)

    The inner op must have no other consumer. Reference counterpart: optimizations/CuDNNFunctionOptimizations (backend-specific rewrites)."""
    ACTS = {"gelu": 6, "precisegelu": 2, "gelu_erf_fused": 3, "tanh": 4, "sigmoid": 5, "relu": 1, "swish": 7}

    def apply(self, sd, outputs):
        changed = False
        for fn in (self._matmul_bias, self._act, self._residual):
            for o in list(sd.ops.values()):
                if o.name in sd.ops:
                    changed = fn(sd, o, outputs) or changed
        return changed

    @staticmethod
    def _dense_parts(o):
        """(x, W, bias, residual, act id) of an xw_plus_b / fused_dense op, or None."""
        if _is(o, "xw_plus_b") and not (o.iArgs and any(int(v) for v in o.iArgs)) and len(o.inputs) == 3:
            return o.inputs[0], o.inputs[1], o.inputs[2], None, 0
        if _is(o, "fused_dense"):
            ia = [int(v) for v in (o.iArgs or [])] + [0, 0, 0]
            hb, hr = ia[1], ia[2]
            return o.inputs[0], o.inputs[1], (o.inputs[2] if hb else None), (o.inputs[2 + hb] if hr else None), ia[0]
        return None

    @staticmethod
    def _emit(sd, old, x, w, b, r, act):
        ins = [x, w] + ([b] if b is not None else []) + ([r] if r is not None else [])
        new = SameDiffOp(old.name, "fused_dense", ins, old.outputs, iArgs=[int(act), int(b is not None), int(r is not None)])
        _remove_op(sd, old)
        sd._add_op(new)
        return True

    def _matmul_bias(self, sd, o, outputs):
        if not _is(o, "add", "biasadd") or len(o.inputs) != 2:
            return False
        for a, b in (o.inputs, o.inputs[::-1]):
            mm = _producer(sd, a)
            if not (_is(mm, "matmul") and len(mm.inputs) == 2 and not (mm.iArgs and any(int(v) for v in list(mm.iArgs)[:3]))
                    and not (mm.tArgs and (float(mm.tArgs[0]) != 1.0 or (len(mm.tArgs) > 1 and float(mm.tArgs[1]) != 0.0)))
                    and _single_use(sd, a, outputs)):
                continue
            ws, bs = sd.inferShape(mm.inputs[1]), sd.inferShape(b)
            if ws is None or bs is None or len(ws) != 2 or len(bs) != 1 or bs[0] != ws[1] or ws[1] < 0:
                continue
            new = SameDiffOp(o.name, "xw_plus_b", [mm.inputs[0], mm.inputs[1], b], o.outputs)
            _remove_op(sd, o)
            sd._add_op(new)
            return True
        return False

    def _act(self, sd, o, outputs):
        act = self.ACTS.get(o.opName.lower())
        if act is None or len(o.inputs) != 1 or (act == 1 and o.tArgs and float(o.tArgs[0]) != 0.0):
            return False
        p = _producer(sd, o.inputs[0])
        parts = self._dense_parts(p) if p is not None else None
        if parts is None or parts[3] is not None or parts[4] != 0 or not _single_use(sd, o.inputs[0], outputs):
            return False
        return self._emit(sd, o, parts[0], parts[1], parts[2], None, act)

    def _residual(self, sd, o, outputs):
        if not _is(o, "add") or len(o.inputs) != 2 or o.inputs[0] == o.inputs[1]:
            return False
        for a, r in (o.inputs, o.inputs[::-1]):
            p = _producer(sd, a)
            parts = self._dense_parts(p) if p is not None else None
            if parts is None or parts[3] is not None or not _single_use(sd, a, outputs):
                continue
            sa, sr = sd.inferShape(a), sd.inferShape(r)
            if sa is None or sr is None or list(sa) != list(sr):
                continue
            return self._emit(sd, o, parts[0], parts[1], parts[2], r, parts[4])
        return False


class MixedPrecision:
    """fp32 graph -> bf16 compute with fp32 master variables (what models/bert.py builds by hand, as a pass for imported graphs):
    every float VARIABLE (and float placeholder that feeds math) is consumed through a ``cast`` to the compute type, loss ops get
    their computed inputs cast back to FLOAT. Attention masks and loss labels / weights stay fp32. Reference counterpart: the network
    dataType(...) switch of DL4J configurations; SameDiff has no such pass, TF-imported BERT runs fp32 there."""

    _LOSS = ("loss", "cross_entropy")
    _MASK_OPS = {"dot_product_attention_v2": 3, "dot_product_attention": 3, "multi_head_dot_product_attention": 6}

    @staticmethod
    def apply(sd: SameDiff, dtype="BFLOAT16") -> SameDiff:
        def is_loss(o):
            n = o.opName.lower()
            return any(k in n for k in MixedPrecision._LOSS)

        def insert_cast(src, dt, consumers):
            out = sd._op("cast", [sd.variables[src]], dArgs=[dt], name=src + "/" + dt.lower())[0]
            cast_op = sd._producer[out.name]
            for cn in consumers:
                c = sd.ops[cn]
                c.inputs = [out.name if i == src else i for i in c.inputs]
                sd._consumers[out.name].append(cn)
            sd._consumers[src] = [x for x in sd._consumers.get(src, []) if x not in consumers or x == cast_op]
            return out.name

        for name, v in list(sd.variables.items()):
            if v.variableType not in (VariableType.VARIABLE, VariableType.PLACEHOLDER):
                continue
            dt = v.dtype if v.dtype is not None else (sd._arrays[name].dtype if name in sd._arrays else None)
            if dt != torch.float32:
                continue
            cons = []
            for cn in list(dict.fromkeys(sd._consumers.get(name, []))):
                c = sd.ops[cn]
                if c.opName == "cast" or is_loss(c):
                    continue
                mi = MixedPrecision._MASK_OPS.get(c.opName)
                if mi is not None and name in c.inputs[mi:] and name not in c.inputs[:mi]:
                    continue
                cons.append(cn)
            if cons:
                insert_cast(name, dtype, cons)
        for o in list(sd.ops.values()):
            if not is_loss(o):
                continue
            for i in list(dict.fromkeys(o.inputs)):
                if sd.variables[i].variableType == VariableType.ARRAY and _producer(sd, i) is not None and not _is(_producer(sd, i), "cast"):
                    insert_cast(i, "FLOAT", [o.name])
        return sd


def _ensure_gelu_udf():
    from ..nd4j.ops.custom import _IMPL
    if "gelu_erf_fused" not in _IMPL:
        _IMPL["gelu_erf_fused"] = lambda ins, i, t, b, d: torch.nn.functional.gelu(ins[0])
