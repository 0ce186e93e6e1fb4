"""Second tranche of TensorFlow import rules: the op names of the reference's tensorflow-mapping-ruleset.pbtxt
(nd4j/samediff-import/samediff-import-tensorflow/src/main/resources) that import_tf.py did not map yet — linear algebra, bitwise,
scatter / segment families, 3-D conv / pooling, image ops, random ops, TensorArray (-> list ops), fake quantisation, RNN block cells,
stateful variables and dataset iterators (-> placeholders). With these every ``inputFrameworkOpName`` of that ruleset has a rule
(``tests/test_import_cpu.py::test_tf_rule_coverage_matches_reference_ruleset``).

Conventions follow the custom ops' own (libnd4j) argument orders; TF attributes become i/t/b args, constant inputs that libnd4j takes
as arguments are read from the graph's Const nodes."""
from __future__ import annotations

import torch

from .import_tf import _DTN, _data, _simple, tf_op


def _ins(c, n, k=None):
    return [c.inp(i) for i in _data(n)][:k]


def _const_var(c, t):
    return c.sd.constant(None, t if isinstance(t, torch.Tensor) else torch.tensor(t))


# ----------------------------------------------------------------------------- straight renames
for _tf, _sd in (("AccumulateNV2", "mergeadd"), ("BatchMatrixDeterminant", "matrix_determinant"), ("BatchMatrixInverse", "matrix_inverse"),
                 ("BatchMatrixSetDiag", "matrix_set_diag"), ("MatrixSetDiag", "matrix_set_diag"), ("BitwiseAnd", "bitwise_and"),
                 ("BitwiseOr", "bitwise_or"), ("BitwiseXor", "bitwise_xor"), ("LeftShift", "shift_bits"), ("RightShift", "rshift_bits"),
                 ("Invert", "toggle_bits"), ("Betainc", "betainc"), ("Igamma", "igamma"), ("Igammac", "igammac"), ("Polygamma", "polygamma"),
                 ("Zeta", "zeta"), ("Cross", "cross"), ("Diag", "diag"), ("DiagPart", "diag_part"), ("DivNoNan", "divide_no_nan"), ("Mod", "mod"),
                 ("TruncateDiv", "truncatediv"), ("Copy", "identity"), ("CopyHost", "identity"), ("DeepCopy", "identity"),
                 ("CheckNumericsV2", "identity"), ("BroadcastArgs", "broadcast_dynamic_shape"), ("BroadcastTo", "broadcast_to"),
                 ("HSVToRGB", "hsv_to_rgb"), ("RGBToHSV", "rgb_to_hsv"), ("SegmentMax", "segment_max"), ("SegmentMean", "segment_mean"),
                 ("SegmentMin", "segment_min"), ("SegmentProd", "segment_prod"), ("SegmentSum", "segment_sum"), ("ScatterAdd", "scatter_add"),
                 ("ScatterDiv", "scatter_div"), ("ScatterMax", "scatter_max"), ("ScatterMin", "scatter_min"), ("ScatterMul", "scatter_mul"),
                 ("ScatterSub", "scatter_sub"), ("ScatterUpdate", "scatter_upd"), ("ScatterNdAdd", "scatter_nd_add"),
                 ("ScatterNdSub", "scatter_nd_sub"), ("ScatterNdUpdate", "scatter_nd_update"), ("TensorScatterAdd", "scatter_nd_add"),
                 ("TensorScatterSub", "scatter_nd_sub"), ("TensorScatterUpdate", "scatter_nd_update"), ("MatrixSolve", "solve"),
                 ("RandomShuffle", "random_shuffle"), ("CompareAndBitpack", "compare_and_bitpack"), ("AdjustContrastv2", "adjust_contrast_v2"),
                 ("AdjustHue", "adjust_hue"), ("AdjustSaturation", "adjust_saturation"), ("DrawBoundingBoxesV2", "draw_bounding_boxes"),
                 ("CropAndResize", "crop_and_resize"), ("LinSpace", "lin_space"), ("NthElement", "nth_element"),
                 ("FakeQuantWithMinMaxVars", "fake_quant_with_min_max_vars"),
                 ("FakeQuantWithMinMaxVarsPerChannel", "fake_quant_with_min_max_vars_per_channel")):
    tf_op(_tf)(_simple(_sd))


@tf_op("ApproximateEqual")
def _approx_equal(c, n):
    x, y = _ins(c, n, 2)
    tol = n.a_float("tolerance", 1e-5)
    return c.sd.math.abs(x.sub(y)).lt(_const_var(c, torch.tensor(tol))).rename(n.name) if False else \
        c.op(n.name, "less", [c.sd.math.abs(x.sub(y)), _const_var(c, torch.tensor(float(tol)))])


@tf_op("Assign")
def _assign(c, n):
    ref, val = _ins(c, n, 2)
    return c.op(n.name, "identity", [val])


@tf_op("ListDiff")
def _listdiff(c, n):
    return c.op(n.name, "listdiff", _ins(c, n, 2), nOut=2)


@tf_op("LogMatrixDeterminant")
def _logdet(c, n):
    return c.op(n.name, "log_matrix_determinant", _ins(c, n, 1), nOut=2)


@tf_op("Lu")
def _lu(c, n):
    return c.op(n.name, "lu", _ins(c, n, 1), nOut=2)


@tf_op("Svd")
def _svd(c, n):
    full, uv = n.a_bool("full_matrices", False), n.a_bool("compute_uv", True)
    return c.op(n.name, "svd", _ins(c, n, 1), nOut=3 if uv else 1, iArgs=[int(full), int(uv), 16])


@tf_op("MatrixBandPart")
def _band(c, n):
    x, lo, hi = _data(n)
    return c.op(n.name, "matrix_band_part", [c.inp(x)], iArgs=[c.ints(lo)[0], c.ints(hi)[0]])


@tf_op("MatrixTriangularSolve")
def _tri_solve(c, n):
    return c.op(n.name, "triangular_solve", _ins(c, n, 2), bArgs=[n.a_bool("lower", True), n.a_bool("adjoint", False)])


@tf_op("Bincount")
def _bincount(c, n):
    arr, size, weights = _data(n)
    ins = [c.inp(arr)]
    w = c.const_of(weights)
    if w is None or w.numel() > 0:
        ins.append(c.inp(weights))
    sz = c.ints(size)[0]
    return c.op(n.name, "bincount", ins, iArgs=[sz, sz])


@tf_op("Bitcast")
def _bitcast(c, n):
    return c.op(n.name, "bitcast", _ins(c, n, 1), dArgs=[_DTN.get(n.a_type("type", torch.float32), "FLOAT")])


@tf_op("SpaceToBatch", "BatchToSpace")
def _s2b(c, n):
    return c.op(n.name, "space_to_batch" if n.op == "SpaceToBatch" else "batch_to_space", _ins(c, n, 2), iArgs=[n.a_int("block_size", 2)])


tf_op("SpaceToBatchND")(_simple("space_to_batch_nd", 3))
tf_op("BatchToSpaceND")(_simple("batch_to_space_nd", 3))


# ----------------------------------------------------------------------------- conv / pooling
def _ndhwc(n):
    return n.a_str("data_format", "NDHWC") == "NDHWC"


def _dhw(v, ndhwc):
    v = list(v) or [1, 1, 1, 1, 1]
    return v[1:4] if ndhwc else v[2:5]


def _pool3d(opName):
    def rule(c, n):
        nd = _ndhwc(n)
        k, s = _dhw(n.a_ints("ksize"), nd), _dhw(n.a_ints("strides"), nd)
        same = n.a_str("padding", "VALID") == "SAME"
        return c.op(n.name, opName, _ins(c, n, 1), iArgs=list(k) + list(s) + [0, 0, 0, 1, 1, 1, int(same), 1 if opName.startswith("avg") else 0, int(nd)])
    return rule


tf_op("MaxPool3D")(_pool3d("maxpool3dnew"))
tf_op("AvgPool3D")(_pool3d("avgpool3dnew"))


@tf_op("MaxPoolV2")
def _maxpool_v2(c, n):
    x, ks, st = _data(n)
    nhwc = n.a_str("data_format", "NHWC") == "NHWC"
    k, s = c.ints(ks), c.ints(st)
    kH, kW = (k[1], k[2]) if nhwc else (k[2], k[3])
    sH, sW = (s[1], s[2]) if nhwc else (s[2], s[3])
    same = n.a_str("padding", "VALID") == "SAME"
    return c.op(n.name, "maxpool2d", [c.inp(x)], iArgs=[kH, kW, sH, sW, 0, 0, 1, 1, int(same), 0, int(nhwc)])


@tf_op("MaxPoolWithArgmax")
def _maxpool_argmax(c, n):
    k, s = n.a_ints("ksize") or [1, 2, 2, 1], n.a_ints("strides") or [1, 2, 2, 1]
    same = n.a_str("padding", "VALID") == "SAME"
    return c.op(n.name, "max_pool_with_argmax", _ins(c, n, 1), nOut=2, iArgs=[k[1], k[2], s[1], s[2], 0, 0, 1, 1, int(same), 1, 1])


@tf_op("Conv3D")
def _conv3d(c, n):
    nd = _ndhwc(n)
    x, w = _ins(c, n, 2)
    s, d = _dhw(n.a_ints("strides"), nd), _dhw(n.a_ints("dilations"), nd)
    same = n.a_str("padding", "VALID") == "SAME"
    ws = c.sd.inferShape(w.name)                              # TF filter [kD, kH, kW, in, out]
    if ws is None:
        raise ValueError("Conv3D %s: the filter shape must be static" % n.name)
    wp = c.op(n.name + "/w_oidhw", "permute", [w], iArgs=[4, 3, 0, 1, 2])
    return c.op(n.name, "conv3dnew", [x, wp], iArgs=list(ws[:3]) + list(s) + [0, 0, 0] + list(d) + [int(same), int(nd), 1])


@tf_op("Conv2DBackpropInput")
def _conv2d_bp_input(c, n):
    sizes, w, g = _ins(c, n, 3)
    nhwc = n.a_str("data_format", "NHWC") == "NHWC"
    st, dl = n.a_ints("strides") or [1, 1, 1, 1], n.a_ints("dilations") or [1, 1, 1, 1]
    sH, sW = (st[1], st[2]) if nhwc else (st[2], st[3])
    dH, dW = (dl[1], dl[2]) if nhwc else (dl[2], dl[3])
    ws = c.sd.inferShape(w.name) or [0, 0]
    same = n.a_str("padding", "VALID") == "SAME"
    return c.op(n.name, "deconv2d_tf", [sizes, w, g], iArgs=[ws[0], ws[1], sH, sW, 0, 0, dH, dW, int(same), int(nhwc), 0])


@tf_op("Dilation2D")
def _dilation2d(c, n):
    r, s = n.a_ints("rates") or [1, 1, 1, 1], n.a_ints("strides") or [1, 1, 1, 1]
    return c.op(n.name, "dilation2d", _ins(c, n, 2), iArgs=[int(n.a_str("padding", "VALID") == "SAME")] + list(r) + list(s))


@tf_op("ExtractImagePatches")
def _patches(c, n):
    k, s, r = n.a_ints("ksizes"), n.a_ints("strides"), n.a_ints("rates")
    return c.op(n.name, "extract_image_patches", _ins(c, n, 1),
                iArgs=[k[1], k[2], s[1], s[2], r[1], r[2], int(n.a_str("padding", "VALID") == "SAME")])


@tf_op("LRN")
def _lrn(c, n):
    return c.op(n.name, "lrn", _ins(c, n, 1), iArgs=[n.a_int("depth_radius", 5)],
                tArgs=[n.a_float("bias", 1.0), n.a_float("alpha", 1.0), n.a_float("beta", 0.5)])


@tf_op("ResizeArea")
def _resize_area(c, n):
    x, size = _data(n)
    h, w = c.ints(size)
    return c.op(n.name, "resize_area", [c.inp(x)], iArgs=[h, w], bArgs=[n.a_bool("align_corners", False)])


@tf_op("NonMaxSuppression", "NonMaxSuppressionV2", "NonMaxSuppressionV3", "NonMaxSuppressionV4")
def _nms(c, n):
    d = _data(n)
    mx = c.ints(d[2])[0]
    iou = n.a_float("iou_threshold", 0.5) if len(d) < 4 else float(c.const_of(d[3]))
    sc = float(c.const_of(d[4])) if len(d) > 4 else float("-inf")
    return c.op(n.name, "non_max_suppression_v3" if len(d) > 4 else "non_max_suppression", [c.inp(d[0]), c.inp(d[1])], iArgs=[mx],
                tArgs=[iou, sc])


@tf_op("NonMaxSuppressionWithOverlaps")
def _nms_ov(c, n):
    d = _data(n)
    return c.op(n.name, "non_max_suppression_overlaps", [c.inp(d[0]), c.inp(d[1])], iArgs=[c.ints(d[2])[0]],
                tArgs=[float(c.const_of(d[3])), float(c.const_of(d[4]))])


@tf_op("RandomCrop")
def _random_crop(c, n):
    return c.op(n.name, "random_crop", _ins(c, n, 2), iArgs=[n.a_int("seed", 0)])


@tf_op("FakeQuantWithMinMaxArgs")
def _fq_args(c, n):
    return c.op(n.name, "fake_quant_with_min_max_vars", _ins(c, n, 1), tArgs=[n.a_float("min", -6.0), n.a_float("max", 6.0)],
                iArgs=[n.a_int("num_bits", 8)], bArgs=[n.a_bool("narrow_range", False)])


# ----------------------------------------------------------------------------- partition / stitch / segment / scatter with attributes
@tf_op("DynamicPartition")
def _dyn_part(c, n):
    k = n.a_int("num_partitions", 1)
    return c.op(n.name, "dynamic_partition", _ins(c, n, 2), nOut=k, iArgs=[k])


@tf_op("DynamicStitch", "ParallelDynamicStitch")
def _dyn_stitch(c, n):
    return c.op(n.name, "dynamic_stitch", _ins(c, n), iArgs=[n.a_int("N", len(_data(n)) // 2)])


@tf_op("ParallelConcat")
def _par_concat(c, n):
    return c.op(n.name, "concat", _ins(c, n), iArgs=[0])


def _unsorted_segment(opName):
    def rule(c, n):
        x, ids, num = _data(n)
        return c.op(n.name, opName, [c.inp(x), c.inp(ids)], iArgs=[c.ints(num)[0]])
    return rule


for _tf, _sd in (("UnsortedSegmentMax", "unsorted_segment_max"), ("UnsortedSegmentMin", "unsorted_segment_min"),
                 ("UnsortedSegmentProd", "unsorted_segment_prod"), ("UnsortedSegmentSum", "unsorted_segment_sum")):
    tf_op(_tf)(_unsorted_segment(_sd))


@tf_op("ScatterNd")
def _scatter_nd(c, n):
    idx, upd, shape = _ins(c, n, 3)
    return c.op(n.name, "scatter_nd", [idx, upd, shape])


def _tensor_scatter_minmax(kind):
    def rule(c, n):
        t, idx, upd = _ins(c, n, 3)
        # max(t, scatter(upd)) restricted to the addressed positions: where(mask, max(t, scattered), t)
        shape = c.op(n.name + "/shape", "shape_of", [t])
        filled = c.op(n.name + "/scattered", "scatter_nd_update", [t, idx, upd])
        return c.op(n.name, "maximum" if kind == "max" else "minimum", [t, filled])
    return rule


tf_op("TensorScatterMax")(_tensor_scatter_minmax("max"))
tf_op("TensorScatterMin")(_tensor_scatter_minmax("min"))


@tf_op("ReverseSequence")
def _rev_seq(c, n):
    return c.op(n.name, "reverse_sequence", _ins(c, n, 2), iArgs=[n.a_int("seq_dim", 1), n.a_int("batch_dim", 0)])


@tf_op("Roll")
def _roll(c, n):
    x, shift, axis = _data(n)
    sh, ax = c.ints(shift), c.ints(axis)
    out = c.inp(x)
    for k, (s_, a_) in enumerate(zip(sh, ax)):
        out = c.op(n.name if k == len(sh) - 1 else "%s/roll%d" % (n.name, k), "roll", [out], iArgs=[s_, a_])
    return out


@tf_op("ShapeN")
def _shape_n(c, n):
    k = len(_data(n))
    return c.op(n.name, "shapes_of", _ins(c, n), nOut=k)


@tf_op("Unique", "UniqueV2")
def _unique(c, n):
    return c.op(n.name, "unique", _ins(c, n, 1), nOut=2)


@tf_op("UniqueWithCounts", "UniqueWithCountsV2")
def _unique_counts(c, n):
    return c.op(n.name, "unique_with_counts", _ins(c, n, 1), nOut=3)


@tf_op("TopK")
def _topk_v1(c, n):
    return c.op(n.name, "top_k", _ins(c, n, 1), nOut=2, iArgs=[n.a_int("k", 1)], bArgs=[n.a_bool("sorted", True)])


@tf_op("InTopK")
def _in_top_k(c, n):
    return c.op(n.name, "in_top_k", _ins(c, n, 2), iArgs=[n.a_int("k", 1)])


@tf_op("InTopKV2")
def _in_top_k2(c, n):
    p, t, k = _data(n)
    return c.op(n.name, "in_top_k", [c.inp(p), c.inp(t)], iArgs=[c.ints(k)[0]])


@tf_op("HistogramFixedWidth")
def _hist(c, n):
    x, rng, nbins = _data(n)
    return c.op(n.name, "histogram_fixed_width", [c.inp(x), c.inp(rng)], iArgs=[c.ints(nbins)[0]])


@tf_op("Empty")
def _empty(c, n):
    return c.op(n.name, "fill", _ins(c, n, 1), tArgs=[0.0], dArgs=[_DTN.get(n.a_type("dtype", torch.float32), "FLOAT")])


@tf_op("CTCLoss")
def _ctc(c, n):
    raise NotImplementedError("CTCLoss (node %s): TF feeds sparse labels (indices / values); convert them to the dense [batch, maxLabelLen] "
                              "form and call sd.loss.ctcLoss / the ctc_loss op (targetLabels, logits, labelLengths, logitLengths)" % n.name)


# ----------------------------------------------------------------------------- random ops (shape input first, libnd4j style)
@tf_op("RandomStandardNormal")
def _rnd_normal(c, n):
    return c.op(n.name, "random_normal", _ins(c, n, 1), tArgs=[0.0, 1.0], dArgs=[_DTN.get(n.a_type("dtype", torch.float32), "FLOAT")])


@tf_op("RandomUniform", "StatelessRandomUniform")
def _rnd_uniform(c, n):
    return c.op(n.name, "randomuniform", _ins(c, n, 1), tArgs=[0.0, 1.0], dArgs=[_DTN.get(n.a_type("dtype", torch.float32), "FLOAT")])


@tf_op("RandomUniformInt")
def _rnd_uniform_int(c, n):
    shape, lo, hi = _data(n)
    return c.op(n.name, "randomuniform", [c.inp(shape)], tArgs=[float(c.ints(lo)[0]), float(c.ints(hi)[0])],
                dArgs=[_DTN.get(n.a_type("Tout", torch.int32), "INT")])


tf_op("RandomGamma")(_simple("random_gamma", 2))
tf_op("RandomPoisson", "RandomPoissonV2")(_simple("random_poisson", 2))


@tf_op("Multinomial")
def _multinomial(c, n):
    return c.op(n.name, "random_multinomial", _ins(c, n, 2), iArgs=[-1])


# ----------------------------------------------------------------------------- recurrent block cells
@tf_op("GRUBlockCell")
def _gru_block(c, n):
    x, h, w_ru, w_c, b_ru, b_c = _ins(c, n, 6)
    return c.op(n.name, "gruCell", [x, h, w_ru, w_c, b_ru, b_c], nOut=4)


@tf_op("LSTMBlockCell")
def _lstm_block_cell(c, n):
    x, cs, h, w, wci, wcf, wco, b = _ins(c, n, 8)
    return c.op(n.name, "lstmBlockCell", [x, cs, h, w, wci, wcf, wco, b], nOut=7, iArgs=[int(n.a_bool("use_peephole", False))],
                tArgs=[n.a_float("forget_bias", 1.0), n.a_float("cell_clip", 3.0)])


@tf_op("BlockLSTM", "BlockLSTMV2")
def _block_lstm(c, n):
    return c.op(n.name, "lstmBlock", _ins(c, n, 9), nOut=7, iArgs=[int(n.a_bool("use_peephole", False)), 0],
                tArgs=[n.a_float("forget_bias", 1.0 if n.op == "BlockLSTM" else 0.0), n.a_float("cell_clip", 3.0)])


# ----------------------------------------------------------------------------- TensorArray -> list ops (the list value rides the "flow" edge)
@tf_op("TensorArrayV3")
def _ta_new(c, n):
    lst = c.op(n.name, "create_list", [], iArgs=[0, 1])
    return [lst, lst]                                     # (handle, flow): both carry the list


def _ta_rule(opName, order, nOut=1, tail_flow=True):
    """order: positions (in TF's input list) of the operands AFTER the list; the list itself is taken from the flow input."""
    def rule(c, n):
        d = _data(n)
        flow = c.inp(d[-1])
        return c.op(n.name, opName, [flow] + [c.inp(d[k]) for k in order], nOut=nOut)
    return rule


# TensorArrayWriteV3(handle, index, value, flow) -> write_list(list, value, index)
tf_op("TensorArrayWriteV3")(_ta_rule("write_list", (2, 1)))
tf_op("TensorArrayRead", "TensorArrayReadV2", "TensorArrayReadV3")(_ta_rule("read_list", (1,)))
tf_op("TensorArrayGather", "TensorArrayGatherV2", "TensorArrayGatherV3")(_ta_rule("gather_list", (1,)))
tf_op("TensorArrayScatter", "TensorArrayScatterV2", "TensorArrayScatterV3")(_ta_rule("scatter_list", (1, 2)))
tf_op("TensorArraySplit", "TensorArraySplitV2", "TensorArraySplitV3")(_ta_rule("split_list", (1, 2)))
tf_op("TensorArraySize", "TensorArraySizeV2", "TensorArraySizeV3")(_ta_rule("size_list", ()))


@tf_op("TensorArrayConcat", "TensorArrayConcatV2", "TensorArrayConcatV3")
def _ta_concat(c, n):
    d = _data(n)
    stacked = c.op(n.name + "/stack", "stack_list", [c.inp(d[-1])])
    shp = c.sd.inferShape(stacked.name)
    if shp is not None and len(shp) >= 2:
        out = c.op(n.name, "reshape", [stacked], iArgs=[-1] + [int(v) for v in shp[2:]])
    else:
        out = c.op(n.name, "identity", [stacked])
    lens = c.op(n.name + "/lengths", "size_list", [c.inp(d[-1])])
    return [out, lens]


# ----------------------------------------------------------------------------- stateful variables / dataset iterators / functional control flow
@tf_op("Variable", "VariableV2")
def _variable(c, n):
    shape = n.a_shape("shape") or []
    dt = n.a_type("dtype", torch.float32)
    return c.sd.var(n.name, torch.zeros([int(s) if s else 1 for s in shape], dtype=dt))


@tf_op("IteratorV2")
def _iterator(c, n):
    return None                                            # resource handle only; IteratorGetNext creates the placeholders


@tf_op("IteratorGetNext")
def _iterator_get_next(c, n):
    types = n.attrs.get("output_types")
    shapes = n.attrs.get("output_shapes")
    tl = types.msg(1).ints(6) if types is not None and types.has(1) else [1]
    sl = shapes.msg(1).msgs(7) if shapes is not None and shapes.has(1) else []
    from .import_tf import TF_DT, parse_shape
    outs = []
    for k, t in enumerate(tl):
        shp = parse_shape(sl[k]) if k < len(sl) else []
        nm = n.name if k == 0 else "%s:%d" % (n.name, k)
        outs.append(c.sd.placeHolder(nm, _DTN.get(TF_DT.get(t, torch.float32), "FLOAT"), *[s if s is not None else -1 for s in (shp or [])]))
    return outs


@tf_op("If", "While", "StatelessIf", "StatelessWhile")
def _functional_cf(c, n):
    raise NotImplementedError("functional control flow (%s, node %s) needs the GraphDef's function library; freeze the graph with "
                              "lower_control_flow=True (Switch / Merge / Enter / Exit / NextIteration are imported)" % (n.op, n.name))
