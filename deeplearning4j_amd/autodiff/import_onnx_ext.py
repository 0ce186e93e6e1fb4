"""Second tranche of ONNX import rules: the remaining op types of the reference's onnx-mapping-ruleset.pbtxt
(nd4j/samediff-import/samediff-import-onnx/src/main/resources): inverse hyperbolics, CumSum, Depth/SpaceToDepth, GatherND / ScatterND /
ScatterElements, Mod, NonZero, TopK, random ops, ResizeNearest, RoiAlign, NonMaxSuppression, the Sequence* family (-> list ops) and the
ONNX recurrent ops LSTM / GRU (forward, reverse, bidirectional; ONNX gate orders). Ops whose ONNX semantics have no single libnd4j
counterpart run as small differentiable tensor programs registered under ``onnx_*`` names."""
from __future__ import annotations

import torch

from ..nd4j.ops.custom import _IMPL
from ..nd4j.ops.impl_misc import NDArrayList
from .import_onnx import _DTN, ONNX_DT, _legacy, _simple, onnx_op

for _o, _l in (("Acosh", "ACosh"), ("Asinh", "ASinh"), ("Atanh", "ATanh")):
    onnx_op(_o)(_legacy(_l))
onnx_op("AliasWithName")(_simple("identity"))
onnx_op("GatherND")(_simple("gather_nd"))
onnx_op("ScatterND")(_simple("scatter_nd_update"))


def _udf(name, fn):
    _IMPL[name] = fn
    return name


@onnx_op("Placeholder")
def _placeholder(c, n):
    return c.sd.placeHolder(n.name, "FLOAT")


@onnx_op("CumSum")
def _cumsum(c, n):
    axis = c.ints(n.inputs[1])[0]
    return c.op(n, "cumsum", [c.inp(n.inputs[0])], iArgs=[n.a_int("exclusive"), n.a_int("reverse"), axis])


def _d2s(ins, i, t, b, d):
    x, bs, crd = ins[0], int(i[0]), bool(i[1])
    N, C, H, W = x.shape
    if crd:
        y = x.reshape(N, C // (bs * bs), bs, bs, H, W).permute(0, 1, 4, 2, 5, 3)
    else:                                                   # DCR (default)
        y = x.reshape(N, bs, bs, C // (bs * bs), H, W).permute(0, 3, 4, 1, 5, 2)
    return y.reshape(N, C // (bs * bs), H * bs, W * bs)


def _s2d(ins, i, t, b, d):
    x, bs = ins[0], int(i[0])
    N, C, H, W = x.shape
    return x.reshape(N, C, H // bs, bs, W // bs, bs).permute(0, 3, 5, 1, 2, 4).reshape(N, C * bs * bs, H // bs, W // bs)


_udf("onnx_depth_to_space", _d2s)
_udf("onnx_space_to_depth", _s2d)
onnx_op("DepthToSpace")(lambda c, n: c.op(n, "onnx_depth_to_space", [c.inp(n.inputs[0])], iArgs=[n.a_int("blocksize"), int(n.a_str("mode", "DCR") == "CRD")]))
onnx_op("SpaceToDepth")(lambda c, n: c.op(n, "onnx_space_to_depth", [c.inp(n.inputs[0])], iArgs=[n.a_int("blocksize")]))


@onnx_op("Mod")
def _mod(c, n):
    a, b = c.inp(n.inputs[0]), c.inp(n.inputs[1])
    if n.a_int("fmod"):
        return c.sd.math._legacy("pairwise", "FMod", [a, b], n.name)
    return c.op(n, "floormod", [a, b])


_udf("onnx_nonzero", lambda ins, i, t, b, d: torch.nonzero(ins[0]).t().contiguous())
onnx_op("NonZero")(lambda c, n: c.op(n, "onnx_nonzero", [c.inp(n.inputs[0])]))


def _topk(ins, i, t, b, d):
    k, axis, largest = int(i[0]), int(i[1]), bool(i[2])
    v, ix = torch.topk(ins[0], k, dim=axis, largest=largest, sorted=True)
    return v, ix


_udf("onnx_topk", _topk)


@onnx_op("TopK")
def _topk_rule(c, n):
    k = c.ints(n.inputs[1])[0] if len(n.inputs) > 1 else n.a_int("k", 1)
    return c.op(n, "onnx_topk", [c.inp(n.inputs[0])], nOut=2, iArgs=[k, n.a_int("axis", -1), n.a_int("largest", 1)])


def _scatter_elements(ins, i, t, b, d):
    x, idx, upd = ins[0], ins[1].long(), ins[2]
    axis, red = int(i[0]), int(i[1])
    idx = torch.where(idx < 0, idx + x.shape[axis], idx)
    if red == 0:
        return x.scatter(axis, idx, upd)
    return x.scatter_reduce(axis, idx, upd, {1: "sum", 2: "prod", 3: "amax", 4: "amin"}[red], include_self=True)


_udf("onnx_scatter_elements", _scatter_elements)
onnx_op("ScatterElements", "Scatter")(lambda c, n: c.op(n, "onnx_scatter_elements", [c.inp(k) for k in n.inputs[:3]], iArgs=[
    n.a_int("axis", 0), {"none": 0, "": 0, "add": 1, "mul": 2, "max": 3, "min": 4}[n.a_str("reduction", "none")]]))


def _rand(kind):
    def rule(c, n):
        shape = n.a_ints("shape")
        dt = _DTN.get(ONNX_DT.get(n.a_int("dtype", 1), torch.float32), "FLOAT")
        sv = c.sd.constant(None, torch.tensor(shape, dtype=torch.int64))
        if kind == "normal":
            return c.op(n, "random_normal", [sv], tArgs=[n.a_float("mean", 0.0), n.a_float("scale", 1.0)], dArgs=[dt])
        return c.op(n, "randomuniform", [sv], tArgs=[n.a_float("low", 0.0), n.a_float("high", 1.0)], dArgs=[dt])
    return rule


onnx_op("RandomNormal")(_rand("normal"))
onnx_op("RandomUniform")(_rand("uniform"))


@onnx_op("ResizeNearest")
def _resize_nearest(c, n):
    x = c.inp(n.inputs[0])
    xs = c.shape_of(x)
    hs, ws = n.a_float("height_scale", 1.0), n.a_float("width_scale", 1.0)
    nhwc = c.op(n.name + "/nhwc", "permute", [x], iArgs=[0, 2, 3, 1])
    r = c.op(n.name + "/r", "resize_nearest_neighbor", [nhwc], iArgs=[int(xs[2] * hs), int(xs[3] * ws)], bArgs=[False, False])
    return c.op(n, "permute", [r], iArgs=[0, 3, 1, 2])


def _roi_align(ins, i, t, b, d):
    """ONNX RoiAlign (avg / max, output_half_pixel): x [N, C, H, W], rois [R, 4] (x1, y1, x2, y2), batch_indices [R]."""
    x, rois, bidx = ins[0], ins[1].float(), ins[2].long()
    oh, ow, sr, mode = int(i[0]), int(i[1]), int(i[2]), int(i[3])
    scale = float(t[0]) if t else 1.0
    N, C, H, W = x.shape
    outs = []
    for r in range(rois.shape[0]):
        x1, y1, x2, y2 = [float(v) * scale for v in rois[r]]
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        bw, bh = rw / ow, rh / oh
        gy = sr if sr > 0 else max(int(-(-rh // oh)), 1)
        gx = sr if sr > 0 else max(int(-(-rw // ow)), 1)
        ys = y1 + (torch.arange(oh * gy, device=x.device, dtype=torch.float32) + 0.5) * bh / gy
        xs = x1 + (torch.arange(ow * gx, device=x.device, dtype=torch.float32) + 0.5) * bw / gx
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        valid = (yy >= -1) & (yy <= H) & (xx >= -1) & (xx <= W)
        yc, xc = yy.clamp(0, H - 1), xx.clamp(0, W - 1)
        y0, x0 = yc.floor().long(), xc.floor().long()
        y1i, x1i = (y0 + 1).clamp(max=H - 1), (x0 + 1).clamp(max=W - 1)
        ly, lx = yc - y0, xc - x0
        f = x[bidx[r]].float()
        v = (f[:, y0, x0] * (1 - ly) * (1 - lx) + f[:, y0, x1i] * (1 - ly) * lx + f[:, y1i, x0] * ly * (1 - lx) + f[:, y1i, x1i] * ly * lx)
        v = v * valid
        v = v.reshape(C, oh, gy, ow, gx)
        outs.append(v.amax((2, 4)) if mode == 1 else v.mean((2, 4)))
    return torch.stack(outs).to(x.dtype) if outs else x.new_zeros(0, C, oh, ow)


_udf("onnx_roi_align", _roi_align)
onnx_op("RoiAlign")(lambda c, n: c.op(n, "onnx_roi_align", [c.inp(k) for k in n.inputs[:3]], iArgs=[
    n.a_int("output_height", 1), n.a_int("output_width", 1), n.a_int("sampling_ratio", 0), int(n.a_str("mode", "avg") == "max")],
    tArgs=[n.a_float("spatial_scale", 1.0)]))


def _onnx_nms(ins, i, t, b, d):
    """boxes [B, N, 4], scores [B, C, N] -> selected_indices [K, 3] (batch, class, box)."""
    boxes, scores = ins[0].float(), ins[1].float()
    max_out = int(i[0]) if i else 0
    center = bool(i[1]) if len(i) > 1 else False
    iou_thr, score_thr = (float(t[0]) if t else 0.0), (float(t[1]) if len(t) > 1 else float("-inf"))
    sel = []
    for bi in range(boxes.shape[0]):
        bx = boxes[bi]
        if center:
            cx, cy, w, h = bx.unbind(-1)
            bx = torch.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1)
        y1, x1 = torch.minimum(bx[:, 0], bx[:, 2]), torch.minimum(bx[:, 1], bx[:, 3])
        y2, x2 = torch.maximum(bx[:, 0], bx[:, 2]), torch.maximum(bx[:, 1], bx[:, 3])
        area = (y2 - y1) * (x2 - x1)
        for ci in range(scores.shape[1]):
            sc = scores[bi, ci]
            order = torch.argsort(sc, descending=True).tolist()
            keep = []
            for j in order:
                if float(sc[j]) <= score_thr or (max_out and len(keep) >= max_out):
                    break
                ok = True
                for k in keep:
                    ih = (torch.minimum(y2[j], y2[k]) - torch.maximum(y1[j], y1[k])).clamp(min=0)
                    iw = (torch.minimum(x2[j], x2[k]) - torch.maximum(x1[j], x1[k])).clamp(min=0)
                    inter = ih * iw
                    if float(inter / (area[j] + area[k] - inter + 1e-12)) > iou_thr:
                        ok = False
                        break
                if ok:
                    keep.append(j)
            sel += [[bi, ci, j] for j in keep]
    return torch.tensor(sel, dtype=torch.int64, device=boxes.device).reshape(-1, 3)


_udf("onnx_nms", _onnx_nms)


@onnx_op("NonMaxSuppression")
def _nms_rule(c, n):
    mx = c.ints(n.inputs[2])[0] if len(n.inputs) > 2 and n.inputs[2] else 0
    iou = float(c.const_of(n.inputs[3])) if len(n.inputs) > 3 and n.inputs[3] else 0.0
    sc = float(c.const_of(n.inputs[4])) if len(n.inputs) > 4 and n.inputs[4] else float("-inf")
    return c.op(n, "onnx_nms", [c.inp(n.inputs[0]), c.inp(n.inputs[1])], iArgs=[mx, n.a_int("center_point_box")], tArgs=[iou, sc])


# ----------------------------------------------------------------------------- sequences -> NDArrayList
def _seq_construct(ins, i, t, b, d):
    lst = NDArrayList(len(ins), True)
    for k, x in enumerate(ins):
        lst.write(k, x)
    return (lst,)


def _seq_items(lst):
    return [lst.items[k] for k in sorted(lst.items)]


def _seq_from(items):
    lst = NDArrayList(len(items), True)
    for k, x in enumerate(items):
        lst.write(k, x)
    return lst


def _seq_insert(ins, i, t, b, d):
    items = _seq_items(ins[0])
    pos = int(ins[2]) if len(ins) > 2 else len(items)
    pos = pos + len(items) if pos < 0 else pos
    items.insert(pos, ins[1])
    return (_seq_from(items),)


def _seq_erase(ins, i, t, b, d):
    items = _seq_items(ins[0])
    pos = int(ins[1]) if len(ins) > 1 else len(items) - 1
    items.pop(pos)
    return (_seq_from(items),)


def _seq_at(ins, i, t, b, d):
    items = _seq_items(ins[0])
    return items[int(ins[1])]


_udf("onnx_sequence_construct", _seq_construct)
_udf("onnx_sequence_insert", _seq_insert)
_udf("onnx_sequence_erase", _seq_erase)
_udf("onnx_sequence_at", _seq_at)
onnx_op("SequenceConstruct")(lambda c, n: c.op(n, "onnx_sequence_construct", [c.inp(k) for k in n.inputs if k]))
onnx_op("SequenceEmpty")(lambda c, n: c.op(n, "create_list", [], iArgs=[0, 1]))
onnx_op("SequenceInsert")(lambda c, n: c.op(n, "onnx_sequence_insert", [c.inp(k) for k in n.inputs if k]))
onnx_op("SequenceErase", "SequenceRemove")(lambda c, n: c.op(n, "onnx_sequence_erase", [c.inp(k) for k in n.inputs if k]))
onnx_op("SequenceAt")(lambda c, n: c.op(n, "onnx_sequence_at", [c.inp(k) for k in n.inputs[:2]]))
onnx_op("SequenceLength")(lambda c, n: c.op(n, "size_list", [c.inp(n.inputs[0])]))


# ----------------------------------------------------------------------------- recurrent ops with ONNX semantics
def _act(name):
    return {"sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": torch.relu}[name.lower()]


def _onnx_lstm(ins, i, t, b, d):
    """X [T, B, I], W [D, 4H, I] (gates i, o, f, c), R [D, 4H, H], B [D, 8H]?, seq_lens?, h0 [D, B, H]?, c0?; -> Y [T, D, B, H], Y_h, Y_c.
    iArgs: [hidden, direction (0 fwd, 1 reverse, 2 bidirectional), has_B, has_lens, has_h0, has_c0, layout]."""
    H, direction, hasB, hasL, hasH, hasC = [int(v) for v in i[:6]]
    it = iter(ins)
    X, W, R = next(it), next(it), next(it)
    Bv = next(it) if hasB else None
    lens = next(it) if hasL else None
    h0 = next(it) if hasH else None
    c0 = next(it) if hasC else None
    T, Bsz, _ = X.shape
    D = W.shape[0]
    ys, hs, cs = [], [], []
    for dd in range(D):
        rev = direction == 1 or (direction == 2 and dd == 1)
        h = h0[dd] if h0 is not None else X.new_zeros(Bsz, H)
        cst = c0[dd] if c0 is not None else X.new_zeros(Bsz, H)
        wb = (Bv[dd, :4 * H] + Bv[dd, 4 * H:]) if Bv is not None else 0.0
        out = [None] * T
        for step in (range(T - 1, -1, -1) if rev else range(T)):
            g = X[step] @ W[dd].t() + h @ R[dd].t() + wb
            gi, go, gf, gc = g.split(H, dim=1)
            ig, og, fg, cg = torch.sigmoid(gi), torch.sigmoid(go), torch.sigmoid(gf), torch.tanh(gc)
            c_new = fg * cst + ig * cg
            h_new = og * torch.tanh(c_new)
            if lens is not None:
                m = (step < lens.to(X.device)).to(X.dtype).unsqueeze(1)
                c_new, h_new = m * c_new + (1 - m) * cst, m * h_new + (1 - m) * h
                out[step] = h_new * m
            else:
                out[step] = h_new
            h, cst = h_new, c_new
        ys.append(torch.stack(out))
        hs.append(h)
        cs.append(cst)
    return torch.stack(ys, 1), torch.stack(hs), torch.stack(cs)


def _onnx_gru(ins, i, t, b, d):
    """X [T, B, I], W [D, 3H, I] (gates z, r, h), R [D, 3H, H], B [D, 6H]?, seq_lens?, h0?; iArgs: [hidden, direction, has_B, has_lens,
    has_h0, linear_before_reset]."""
    H, direction, hasB, hasL, hasH, lbr = [int(v) for v in i[:6]]
    it = iter(ins)
    X, W, R = next(it), next(it), next(it)
    Bv = next(it) if hasB else None
    lens = next(it) if hasL else None
    h0 = next(it) if hasH else None
    T, Bsz, _ = X.shape
    D = W.shape[0]
    ys, hs = [], []
    for dd in range(D):
        rev = direction == 1 or (direction == 2 and dd == 1)
        h = h0[dd] if h0 is not None else X.new_zeros(Bsz, H)
        Wz, Wr, Wh = W[dd].split(H, 0)
        Rz, Rr, Rh = R[dd].split(H, 0)
        if Bv is not None:
            wbz, wbr, wbh, rbz, rbr, rbh = Bv[dd].split(H, 0)
        else:
            wbz = wbr = wbh = rbz = rbr = rbh = 0.0
        out = [None] * T
        for step in (range(T - 1, -1, -1) if rev else range(T)):
            x = X[step]
            z = torch.sigmoid(x @ Wz.t() + h @ Rz.t() + wbz + rbz)
            r = torch.sigmoid(x @ Wr.t() + h @ Rr.t() + wbr + rbr)
            if lbr:
                hh = torch.tanh(x @ Wh.t() + r * (h @ Rh.t() + rbh) + wbh)
            else:
                hh = torch.tanh(x @ Wh.t() + (r * h) @ Rh.t() + rbh + wbh)
            h_new = (1 - z) * hh + z * h
            if lens is not None:
                m = (step < lens.to(X.device)).to(X.dtype).unsqueeze(1)
                h_new = m * h_new + (1 - m) * h
                out[step] = h_new * m
            else:
                out[step] = h_new
            h = h_new
        ys.append(torch.stack(out))
        hs.append(h)
    return torch.stack(ys, 1), torch.stack(hs)


_udf("onnx_lstm", _onnx_lstm)
_udf("onnx_gru", _onnx_gru)


def _rnn_rule(opName, n_state):
    def rule(c, n):
        if n.a_int("layout", 0) != 0:
            raise NotImplementedError("%s layout=1 (batch major) is not supported; export with the default layout" % n.op)
        present = [len(n.inputs) > k and bool(n.inputs[k]) for k in range(3, 3 + 2 + n_state)]
        ins = [c.inp(n.inputs[k]) for k in range(3)] + [c.inp(n.inputs[3 + k]) for k, p in enumerate(present) if p]
        direction = {"forward": 0, "reverse": 1, "bidirectional": 2}[n.a_str("direction", "forward")]
        ia = [n.a_int("hidden_size"), direction] + [int(p) for p in present]
        if opName == "onnx_gru":
            ia = ia[:5] + [n.a_int("linear_before_reset", 0)]
        outs = c.op(n, opName, ins, nOut=1 + n_state, iArgs=ia)
        return list(outs)
    return rule


onnx_op("LSTM")(_rnn_rule("onnx_lstm", 2))
onnx_op("GRU")(_rnn_rule("onnx_gru", 1))


@onnx_op("If", "Loop")
def _subgraph_cf(c, n):
    raise NotImplementedError("ONNX %s (node %s): sub-graph control flow is not imported; export the model with the branches / loop "
                              "unrolled, or build the control flow with sd.ifCond / sd.whileLoop around imported sub-graphs" % (n.op, n.name))
