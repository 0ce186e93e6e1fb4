"""SameDiff op namespaces: sd.math / nn / cnn / rnn / loss / linalg / bitwise / random / image + base ops.

Reference: nd4j-api autodiff/samediff/ops/SD{Base,Math,NN,CNN,RNN,Loss,Linalg,Bitwise,Random,Image}Ops.java
(generated from the codegen Kotlin DSL). Every method appends one graph node naming an op of the
native registry; anything in the registry that has no hand-written method here is still reachable
through attribute access (``sd.math.<opName>(inputs..., iArgs=[...], tArgs=[...])``).
"""
from __future__ import annotations

from typing import Sequence

from ..nd4j.ops import legacy as LG
from ..nd4j.ops.custom import describe
from .generated.sd_base import GenSDBase
from .generated.sd_bitwise import GenSDBitwise
from .generated.sd_cnn import GenSDCNN
from .generated.sd_math import GenSDMath
from .generated.sd_nn import GenSDNN
from .generated.sd_rnn import GenSDRNN
from .generated.sd_image import GenSDImage
from .generated.sd_linalg import GenSDLinalg
from .generated.sd_loss import GenSDLoss
from .generated.sd_random import GenSDRandom


class _NS:
    def __init__(self, sd):
        self.sd = sd

    def _o(self, opName, inputs, name=None, nOut=1, **kw):
        r = self.sd._op(opName, list(inputs), name=name, nOut=nOut, **kw)
        return r[0] if nOut == 1 else r

    def _legacy(self, family, opName, inputs, name=None, **kw):
        from .samediff import OpType
        fam_type = {"transform_same": OpType.TRANSFORM_SAME, "transform_strict": OpType.TRANSFORM_STRICT,
                    "transform_float": OpType.TRANSFORM_FLOAT, "transform_bool": OpType.TRANSFORM_BOOL, "scalar": OpType.SCALAR,
                    "pairwise": OpType.PAIRWISE, "reduce_float": OpType.REDUCE_FLOAT, "reduce_same": OpType.REDUCE_SAME,
                    "reduce_bool": OpType.REDUCE_BOOL, "reduce_long": OpType.REDUCE_LONG, "reduce3": OpType.REDUCE_3,
                    "index_reduce": OpType.INDEX_REDUCE, "summary_stats": OpType.SUMMARYSTATS}[family]
        return self.sd._op(opName.lower(), list(inputs), name=name, opType=fam_type, opNum=LG.op_num(family, opName), **kw)[0]

    def __getattr__(self, opName):
        if opName.startswith("_"):
            raise AttributeError(opName)
        d = describe(opName)
        if d is None:
            raise AttributeError("no op named '%s' in the registry" % opName)

        def call(*inputs, name=None, nOut=None, iArgs=None, tArgs=None, bArgs=None, dArgs=None):
            n = nOut if nOut is not None else max(d.numOutputs, 1)
            return self._o(opName, inputs, name, n, iArgs=iArgs, tArgs=tArgs, bArgs=bArgs, dArgs=dArgs)
        return call


def _dims(d):
    return list(d[0]) if len(d) == 1 and isinstance(d[0], (list, tuple)) else list(d)


class SDBaseOps(GenSDBase, _NS):       # generated surface (codegen/ops/base.py) + the hand-written forms below (they win)
    def identity(self, x, name=None): return self._o("identity", [x], name)
    def concat(self, dim, *xs, name=None): return self._o("concat", xs, name, iArgs=[dim])
    def stack(self, axis, *xs, name=None): return self._o("stack", xs, name, iArgs=[axis])
    def unstack(self, x, axis, num, name=None): return self._o("unstack", [x], name, num, iArgs=[axis])
    def split(self, x, numSplit, axis, name=None): return self._o("split", [x], name, numSplit, iArgs=[numSplit, axis])
    def reshape(self, x, *shape, name=None): return self._o("reshape", [x], name, iArgs=_dims(shape))
    def permute(self, x, *dims, name=None): return self._o("permute", [x], name, iArgs=_dims(dims))
    def transpose(self, x, name=None): return self._o("transpose", [x], name)
    def expandDims(self, x, axis, name=None): return self._o("expand_dims", [x], name, iArgs=[axis])
    def squeeze(self, x, axis, name=None): return self._o("squeeze", [x], name, iArgs=[axis])
    def tile(self, x, *reps, name=None): return self._o("tile", [x], name, iArgs=_dims(reps))
    def gather(self, x, indices, axis=0, name=None): return self._o("gather", [x, indices], name, iArgs=[axis])
    def gatherNd(self, x, indices, name=None): return self._o("gather_nd", [x, indices], name)
    def scatterAdd(self, ref, idx, upd, name=None): return self._o("scatter_add", [ref, idx, upd], name)
    def scatterUpdate(self, ref, idx, upd, name=None): return self._o("scatter_upd", [ref, idx, upd], name)
    def slice(self, x, begin, size, name=None): return self._o("slice", [x], name, iArgs=list(begin) + list(size))
    def stridedSlice(self, x, begin, end, strides, beginMask=0, endMask=0, ellipsisMask=0, newAxisMask=0, shrinkAxisMask=0, name=None):
        return self._o("strided_slice", [x], name, iArgs=[beginMask, ellipsisMask, endMask, newAxisMask, shrinkAxisMask] + list(begin) + list(end) + list(strides))
    def oneHot(self, idx, depth, axis=-1, on=1.0, off=0.0, name=None): return self._o("onehot", [idx], name, iArgs=[axis, depth], tArgs=[on, off])
    def castTo(self, x, dtype, name=None): return self._o("cast", [x], name, dArgs=[str(dtype).upper()])
    def shape(self, x, name=None): return self._o("shape_of", [x], name)
    def size(self, x, name=None): return self._o("size", [x], name)
    def rank(self, x, name=None): return self._o("rank", [x], name)
    def zerosLike(self, x, name=None): return self._o("zeros_as", [x], name)
    def onesLike(self, x, name=None): return self._o("ones_as", [x], name)
    def fill(self, shape, dtype, value, name=None): return self._o("fill", [shape], name, tArgs=[value], dArgs=[str(dtype).upper()])
    def linspace(self, dtype, start, stop, num, name=None): return self._o("lin_space", [], name, tArgs=[start, stop], iArgs=[num], dArgs=[str(dtype).upper()])
    def range(self, frm, to, step, dtype="FLOAT", name=None): return self._o("range", [], name, tArgs=[frm, to, step], dArgs=[str(dtype).upper()])
    def where(self, cond, x=None, y=None, name=None): return self._o("Where", [cond] + ([x, y] if x is not None else []), name)
    def matchCondition(self, x, mode, value=0.0, eps=1e-5, name=None): return self._legacy("transform_bool", "MatchConditionBool", [x], name, tArgs=[value, eps, mode])
    def reverse(self, x, *dims, name=None): return self._o("reverse", [x], name, iArgs=_dims(dims))
    def sequenceMask(self, lengths, maxLen=None, dtype="BOOL", name=None): return self._o("sequence_mask", [lengths], name, iArgs=[maxLen] if maxLen else [], dArgs=[dtype])
    def mmul(self, a, b, transposeA=False, transposeB=False, transposeZ=False, name=None): return self._o("matmul", [a, b], name, iArgs=[int(transposeA), int(transposeB), int(transposeZ)])
    def tensorMmul(self, a, b, dimsA, dimsB, name=None): return self._o("tensormmul", [a, b], name, iArgs=[len(dimsA)] + list(dimsA) + [len(dimsB)] + list(dimsB))
    def batchMmul(self, a, b, name=None): return self._o("matmul", [a, b], name, iArgs=[0, 0, 0])
    def dot(self, a, b, *dims, name=None): return self._o("reduce_sum", [a * b], name, iArgs=_dims(dims))
    def sum(self, x, *dims, keepDims=False, name=None): return self._o("reduce_sum", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def mean(self, x, *dims, keepDims=False, name=None): return self._o("reduce_mean", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def max(self, x, *dims, keepDims=False, name=None): return self._o("reduce_max", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def min(self, x, *dims, keepDims=False, name=None): return self._o("reduce_min", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def prod(self, x, *dims, keepDims=False, name=None): return self._o("reduce_prod", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def norm1(self, x, *dims, keepDims=False, name=None): return self._o("reduce_norm1", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def norm2(self, x, *dims, keepDims=False, name=None): return self._o("reduce_norm2", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def normmax(self, x, *dims, keepDims=False, name=None): return self._o("reduce_norm_max", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def squaredNorm(self, x, *dims, keepDims=False, name=None): return self._o("reduce_sqnorm", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def variance(self, x, biasCorrected, *dims, keepDims=False, name=None): return self._o("reduce_variance", [x], name, iArgs=_dims(dims), bArgs=[keepDims, biasCorrected])
    def standardDeviation(self, x, biasCorrected, *dims, keepDims=False, name=None): return self._o("reduce_stdev", [x], name, iArgs=_dims(dims), bArgs=[keepDims, biasCorrected])
    def argmax(self, x, *dims, keepDims=False, name=None): return self._o("argmax", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def argmin(self, x, *dims, keepDims=False, name=None): return self._o("argmin", [x], name, iArgs=_dims(dims), bArgs=[keepDims])
    def cumsum(self, x, exclusive=False, reverse=False, *axis, name=None): return self._o("cumsum", [x], name, iArgs=[int(exclusive), int(reverse)] + _dims(axis))
    def cumprod(self, x, exclusive=False, reverse=False, *axis, name=None): return self._o("cumprod", [x], name, iArgs=[int(exclusive), int(reverse)] + _dims(axis))
    def eq(self, a, b, name=None): return self._o("equals", [a, b], name)
    def neq(self, a, b, name=None): return self._o("not_equals", [a, b], name)
    def gt(self, a, b, name=None): return self._o("greater", [a, b], name)
    def gte(self, a, b, name=None): return self._o("greater_equal", [a, b], name)
    def lt(self, a, b, name=None): return self._o("less", [a, b], name)
    def lte(self, a, b, name=None): return self._o("less_equal", [a, b], name)
    def segmentSum(self, x, ids, name=None): return self._o("segment_sum", [x, ids], name)
    def unsortedSegmentSum(self, x, ids, n, name=None): return self._o("unsorted_segment_sum", [x, ids], name, iArgs=[n])
    def dynamicPartition(self, x, parts, n, name=None): return self._o("dynamic_partition", [x, parts], name, n, iArgs=[n])
    def dynamicStitch(self, indices, xs, name=None): return self._o("dynamic_stitch", list(indices) + list(xs), name)


class SDMath(GenSDMath, SDBaseOps):       # generated surface (codegen/ops/math.py); the legacy-table loop below only fills names it lacks
    pass


def _unary(fam, opn, py):
    def f(self, x, name=None):
        return self._legacy(fam, opn, [x], name)
    f.__name__ = py
    return f


for _fam, _tbl in (("transform_same", LG.TRANSFORM_SAME), ("transform_strict", LG.TRANSFORM_STRICT), ("transform_float", LG.TRANSFORM_FLOAT),
                   ("transform_bool", LG.TRANSFORM_BOOL)):
    for _n in _tbl.values():
        _py = _n[0].lower() + _n[1:]
        if not hasattr(SDMath, _py):
            setattr(SDMath, _py, _unary(_fam, _n, _py))
SDMath.cos = SDMath.cosine
SDMath.rsqrt = SDMath.rSqrt


def _math_extra():
    def add(name, fn):
        setattr(SDMath, name, fn)
    add("pow", lambda self, x, p, name=None: self._o("Pow", [x, p], name))
    add("clipByValue", lambda self, x, lo, hi, name=None: self._o("clipbyvalue", [x], name, tArgs=[lo, hi]))
    add("clipByNorm", lambda self, x, c, *dims, name=None: self._o("clipbynorm", [x], name, tArgs=[c], iArgs=_dims(dims)))
    add("atan2", lambda self, y, x, name=None: self._o("tf_atan2", [y, x], name))
    add("max", lambda self, a, b=None, *dims, name=None, keepDims=False: self._o("maximum", [a, b], name) if hasattr(b, "name") else
        self._o("reduce_max", [a], name, iArgs=([b] if b is not None else []) + _dims(dims), bArgs=[keepDims]))
    add("min", lambda self, a, b=None, *dims, name=None, keepDims=False: self._o("minimum", [a, b], name) if hasattr(b, "name") else
        self._o("reduce_min", [a], name, iArgs=([b] if b is not None else []) + _dims(dims), bArgs=[keepDims]))
    add("isNaN", SDMath.isNan)
    add("isInfinite", SDMath.isInf)
    add("logSumExp", lambda self, x, *dims, name=None: self._o("reduce_logsumexp", [x], name, iArgs=_dims(dims)))
    add("cosineSimilarity", lambda self, a, b, *dims, name=None: self._legacy("reduce3", "CosineSimilarity", [a, b], name, dimensions=_dims(dims)))
    add("cosineDistance", lambda self, a, b, *dims, name=None: self._legacy("reduce3", "CosineDistance", [a, b], name, dimensions=_dims(dims)))
    add("euclideanDistance", lambda self, a, b, *dims, name=None: self._legacy("reduce3", "EuclideanDistance", [a, b], name, dimensions=_dims(dims)))
    add("manhattanDistance", lambda self, a, b, *dims, name=None: self._legacy("reduce3", "ManhattanDistance", [a, b], name, dimensions=_dims(dims)))
    add("countNonZero", lambda self, x, *dims, name=None: self._legacy("reduce_long", "CountNonZero", [x], name, dimensions=_dims(dims)))
    add("entropy", lambda self, x, *dims, name=None: self._legacy("reduce_float", "Entropy", [x], name, dimensions=_dims(dims)))
    add("iamax", lambda self, x, *dims, name=None: self._legacy("index_reduce", "IndexAbsoluteMax", [x], name, dimensions=_dims(dims)))
    add("standardize", lambda self, x, *dims, name=None: self._o("standardize", [x], name, iArgs=_dims(dims)))
    add("diag", lambda self, x, name=None: self._o("diag", [x], name))
    add("trace", lambda self, x, name=None: self._o("trace", [x], name))
    add("eye", lambda self, n, name=None: self._o("eye", [], name, iArgs=[0, n]))
    add("mergeAdd", lambda self, *xs, name=None: self._o("mergeadd", xs, name))
    add("mergeAvg", lambda self, *xs, name=None: self._o("mergeavg", xs, name))
    add("mergeMax", lambda self, *xs, name=None: self._o("mergemax", xs, name))
    add("erf", SDMath.erf)
    add("confusionMatrix", lambda self, labels, pred, n, name=None: self._o("confusion_matrix", [labels, pred], name, iArgs=[n]))
    add("moments", lambda self, x, *dims, name=None: self._o("moments", [x], name, 2, iArgs=_dims(dims)))


_math_extra()


class SDNN(GenSDNN, _NS):       # generated surface (codegen/ops/nn.py) + overrides
    def linear(self, x, w, b=None, name=None): return self._o("xw_plus_b", [x, w, b], name) if b is not None else self._o("matmul", [x, w], name, iArgs=[0, 0, 0])
    def reluLayer(self, x, w, b, name=None): return self._o("relu_layer", [x, w, b], name)
    def biasAdd(self, x, b, nchw=False, name=None): return self._o("biasadd", [x, b], name, bArgs=[nchw])
    def relu(self, x, cutoff=0.0, name=None): return self._o("relu", [x], name, tArgs=[cutoff])
    def relu6(self, x, cutoff=0.0, name=None): return self._o("relu6", [x], name, tArgs=[cutoff])
    def leakyRelu(self, x, alpha=0.01, name=None): return self._o("lrelu", [x], name, tArgs=[alpha])
    def elu(self, x, name=None): return self._o("elu", [x], name)
    def selu(self, x, name=None): return self._o("selu", [x], name)
    def gelu(self, x, name=None): return self.sd.math.gELU(x, name=name)
    def preciseGelu(self, x, name=None): return self.sd.math.preciseGELU(x, name=name)
    def swish(self, x, name=None): return self.sd.math.swish(x, name=name)
    def sigmoid(self, x, name=None): return self._o("sigmoid", [x], name)
    def tanh(self, x, name=None): return self._o("tanh", [x], name)
    def hardTanh(self, x, name=None): return self._o("hardtanh", [x], name)
    def hardSigmoid(self, x, name=None): return self._o("hardsigmoid", [x], name)
    def softplus(self, x, name=None): return self._o("softplus", [x], name)
    def softsign(self, x, name=None): return self._o("softsign", [x], name)
    def prelu(self, x, alpha, *sharedAxes, name=None): return self._o("prelu", [x, alpha], name, iArgs=list(sharedAxes))
    def softmax(self, x, dim=-1, name=None): return self._o("softmax", [x], name, iArgs=[dim])
    def logSoftmax(self, x, dim=-1, name=None): return self._o("log_softmax", [x], name, iArgs=[dim])
    def dropout(self, x, p, name=None): return self._o("dropout", [x], name, 2, tArgs=[p])[0]
    def layerNorm(self, x, gain, bias=None, channelsFirst=False, *dims, eps=None, name=None):
        """``eps``: added to the variance (tArgs[0] of this framework's layer_norm; default 1e-5 — the reference op standardizes
        without one). BERT checkpoints are trained with 1e-12: models/bert.py passes its configured value."""
        return self._o("layer_norm", [x, gain] + ([bias] if bias is not None else []), name, iArgs=_dims(dims) or [-1], bArgs=[channelsFirst],
                       tArgs=None if eps is None else [float(eps)])
    def batchNorm(self, x, mean, var, gamma, beta, eps=1e-5, *axis, name=None):
        return self._o("batchnorm", [x, mean, var, gamma, beta], name, iArgs=[1, 1] + (_dims(axis) or [1]), tArgs=[eps])
    def lrn(self, x, depth=5, bias=1.0, alpha=1.0, beta=0.5, name=None): return self._o("lrn", [x], name, iArgs=[depth], tArgs=[bias, alpha, beta])
    def pad(self, x, padding, mode=0, value=0.0, name=None): return self._o("pad", [x, padding], name, iArgs=[mode], tArgs=[value])
    def embeddingLookup(self, table, idx, name=None): return self._o("embedding_lookup", [table, idx], name)
    def dotProductAttention(self, q, k, v, mask=None, scaled=True, withWeights=False, name=None):
        return self._o("dot_product_attention", [q, k, v] + ([mask] if mask is not None else []), name, 2 if withWeights else 1,
                       iArgs=[int(scaled), int(withWeights)])
    def dotProductAttentionV2(self, q, v, k=None, qMask=None, vMask=None, scale=1.0, dropout=0.0, causal=False, training=False,
                              withWeights=True, name=None):
        """withWeights=False returns only the attention output (single SDVariable): nothing then needs the score matrices and the
        op runs through the fused flash kernels on the GPU."""
        ins = [q, v] + ([k] if k is not None else []) + ([qMask] if qMask is not None else []) + ([vMask] if vMask is not None else [])
        key_only = int(qMask is None and vMask is not None)
        return self._o("dot_product_attention_v2", ins, name, 3 if withWeights else 1, iArgs=[int(withWeights), key_only],
                       tArgs=[scale, dropout], bArgs=[causal, training])
    def multiHeadDotProductAttention(self, q, k, v, Wq, Wk, Wv, Wo, mask=None, scaled=True, withWeights=False, name=None):
        return self._o("multi_head_dot_product_attention", [q, k, v, Wq, Wk, Wv, Wo] + ([mask] if mask is not None else []), name,
                       2 if withWeights else 1, iArgs=[int(scaled), int(withWeights)])


class SDCNN(GenSDCNN, _NS):       # generated surface (codegen/ops/cnn.py) + overrides
    @staticmethod
    def _c2(kH, kW, sH, sW, pH, pW, dH, dW, same, nhwc, wfmt=None):
        a = [kH, kW, sH, sW, pH, pW, dH, dW, int(same), int(nhwc)]
        return a + ([wfmt] if wfmt is not None else [])

    def conv2d(self, x, w, b=None, kH=1, kW=1, sH=1, sW=1, pH=0, pW=0, dH=1, dW=1, isSameMode=False, dataFormat="NCHW", weightsFormat=0, name=None):
        return self._o("conv2d", [x, w] + ([b] if b is not None else []), name,
                       iArgs=self._c2(kH, kW, sH, sW, pH, pW, dH, dW, isSameMode, dataFormat == "NHWC", weightsFormat))
    def depthWiseConv2d(self, x, w, b=None, kH=1, kW=1, sH=1, sW=1, pH=0, pW=0, dH=1, dW=1, isSameMode=False, dataFormat="NCHW", name=None):
        return self._o("depthwise_conv2d", [x, w] + ([b] if b is not None else []), name, iArgs=self._c2(kH, kW, sH, sW, pH, pW, dH, dW, isSameMode, dataFormat == "NHWC"))
    def separableConv2d(self, x, wd, wp, b=None, kH=1, kW=1, sH=1, sW=1, pH=0, pW=0, dH=1, dW=1, isSameMode=False, dataFormat="NCHW", name=None):
        return self._o("sconv2d", [x, wd, wp] + ([b] if b is not None else []), name, iArgs=self._c2(kH, kW, sH, sW, pH, pW, dH, dW, isSameMode, dataFormat == "NHWC"))
    def deconv2d(self, x, w, b=None, kH=1, kW=1, sH=1, sW=1, pH=0, pW=0, dH=1, dW=1, isSameMode=False, dataFormat="NCHW", name=None):
        return self._o("deconv2d", [x, w] + ([b] if b is not None else []), name, iArgs=self._c2(kH, kW, sH, sW, pH, pW, dH, dW, isSameMode, dataFormat == "NHWC"))
    def conv1d(self, x, w, b=None, k=1, s=1, p=0, d=1, paddingMode=0, dataFormat="NCW", name=None):
        return self._o("conv1d", [x, w] + ([b] if b is not None else []), name, iArgs=[k, s, p, d, paddingMode, int(dataFormat == "NWC")])
    def conv3d(self, x, w, b=None, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), d=(1, 1, 1), isSameMode=False, dataFormat="NCDHW", name=None):
        return self._o("conv3dnew", [x, w] + ([b] if b is not None else []), name, iArgs=list(k) + list(s) + list(p) + list(d) + [int(isSameMode), int(dataFormat == "NDHWC")])
    def maxPooling2d(self, x, kH, kW, sH=1, sW=1, pH=0, pW=0, dH=1, dW=1, isSameMode=False, dataFormat="NCHW", name=None):
        return self._o("maxpool2d", [x], name, iArgs=[kH, kW, sH, sW, pH, pW, dH, dW, int(isSameMode), 0, int(dataFormat == "NHWC")])
    def avgPooling2d(self, x, kH, kW, sH=1, sW=1, pH=0, pW=0, dH=1, dW=1, isSameMode=False, dataFormat="NCHW", includePad=True, name=None):
        return self._o("avgpool2d", [x], name, iArgs=[kH, kW, sH, sW, pH, pW, dH, dW, int(isSameMode), 0 if includePad else 1, int(dataFormat == "NHWC")])
    def maxPooling3d(self, x, k, s, p=(0, 0, 0), d=(1, 1, 1), isSameMode=False, dataFormat="NCDHW", name=None):
        return self._o("maxpool3dnew", [x], name, iArgs=list(k) + list(s) + list(p) + list(d) + [int(isSameMode), 0, int(dataFormat == "NDHWC")])
    def avgPooling3d(self, x, k, s, p=(0, 0, 0), d=(1, 1, 1), isSameMode=False, dataFormat="NCDHW", name=None):
        return self._o("avgpool3dnew", [x], name, iArgs=list(k) + list(s) + list(p) + list(d) + [int(isSameMode), 0, int(dataFormat == "NDHWC")])
    def upsampling2d(self, x, scaleH, scaleW=None, nchw=True, name=None): return self._o("upsampling2d", [x], name, iArgs=[scaleH, scaleW or scaleH, int(nchw)])
    def im2Col(self, x, kH, kW, sH=1, sW=1, pH=0, pW=0, dH=1, dW=1, isSameMode=False, name=None):
        return self._o("im2col", [x], name, iArgs=[kH, kW, sH, sW, pH, pW, dH, dW, int(isSameMode)])
    def col2Im(self, x, sH, sW, pH, pW, iH, iW, dH=1, dW=1, name=None): return self._o("col2im", [x], name, iArgs=[sH, sW, pH, pW, iH, iW, dH, dW])
    def spaceToDepth(self, x, block, dataFormat="NCHW", name=None): return self._o("space_to_depth", [x], name, iArgs=[block, int(dataFormat == "NHWC")])
    def depthToSpace(self, x, block, dataFormat="NCHW", name=None): return self._o("depth_to_space", [x], name, iArgs=[block, int(dataFormat == "NHWC")])
    def spaceToBatch(self, x, block, padding, name=None): return self._o("space_to_batch", [x, padding], name, iArgs=[block])
    def batchToSpace(self, x, block, crops, name=None): return self._o("batch_to_space", [x, crops], name, iArgs=[block])
    def localResponseNormalization(self, x, depth=5, bias=1.0, alpha=1.0, beta=0.5, name=None): return self._o("lrn", [x], name, iArgs=[depth], tArgs=[bias, alpha, beta])


class SDRNN(GenSDRNN, _NS):       # generated surface (codegen/ops/rnn.py) + the configurable lstmLayer front end
    def lstmLayer(self, x, Wx, Wr, b=None, seqLen=None, hI=None, cI=None, Wp=None, dataFormat=0, directionMode=0, gateAct=2, cellAct=0, outAct=0,
                  cellClip=0.0, retFullSequence=True, retLastH=False, retLastC=False, name=None):
        ins = [x, Wx, Wr] + [v for v in (b, seqLen, hI, cI, Wp) if v is not None]
        flags = [b is not None, seqLen is not None, hI is not None, cI is not None, Wp is not None, retFullSequence, retLastH, retLastC]
        n = int(retFullSequence) + int(retLastH) + int(retLastC)
        r = self._o("lstmLayer", ins, name, n, iArgs=[dataFormat, directionMode, gateAct, cellAct, outAct], tArgs=[cellClip], bArgs=flags)
        return r
    def lstmCell(self, x, cLast, yLast, W, Wci, Wcf, Wco, b, peephole=False, forgetBias=0.0, clip=0.0, name=None):
        return self._o("lstmBlockCell", [x, cLast, yLast, W, Wci, Wcf, Wco, b], name, 7, iArgs=[int(peephole)], tArgs=[forgetBias, clip])
    def gruCell(self, x, hLast, Wru, Wc, bru, bc, name=None): return self._o("gruCell", [x, hLast, Wru, Wc, bru, bc], name, 4)
    def gru(self, x, hI, Wx, Wh, b, name=None): return self._o("gru", [x, hI, Wx, Wh, b], name)
    def sru(self, x, w, b, c0, mask=None, name=None): return self._o("sru", [x, w, b, c0] + ([mask] if mask is not None else []), name, 2)
    def sruCell(self, x, c0, w, b, name=None): return self._o("sruCell", [x, c0, w, b], name, 2)


class SDLoss(GenSDLoss, _NS):       # the whole surface is generated (codegen/ops/loss.py)
    pass


class SDLinalg(GenSDLinalg, _NS):       # generated surface (codegen/ops/linalg.py) + the overrides below
    def cholesky(self, x, name=None): return self._o("cholesky", [x], name)
    def lstsq(self, a, b, l2=0.0, fast=True, name=None): return self._o("lstsq", [a, b], name, tArgs=[l2], bArgs=[fast])
    def solve(self, a, b, adjoint=False, name=None): return self._o("solve", [a, b], name, bArgs=[adjoint])
    def triangularSolve(self, a, b, lower=True, adjoint=False, name=None): return self._o("triangular_solve", [a, b], name, bArgs=[lower, adjoint])
    def lu(self, x, name=None): return self._o("lu", [x], name, 2)
    def qr(self, x, full=False, name=None): return self._o("qr", [x], name, 2, bArgs=[full])
    def svd(self, x, fullUV=False, computeUV=True, switchNum=16, name=None): return self._o("svd", [x], name, 3 if computeUV else 1, iArgs=[int(fullUV), int(computeUV), switchNum])
    def matrixInverse(self, x, name=None): return self._o("matrix_inverse", [x], name)
    def matrixDeterminant(self, x, name=None): return self._o("matrix_determinant", [x], name)
    def logdet(self, x, name=None): return self._o("logdet", [x], name)
    def cross(self, a, b, name=None): return self._o("cross", [a, b], name)
    def diag(self, x, name=None): return self._o("diag", [x], name)
    def diagPart(self, x, name=None): return self._o("diag_part", [x], name)
    def matrixBandPart(self, x, lower, upper, name=None): return self._o("matrix_band_part", [x], name, iArgs=[lower, upper])
    def tri(self, rows, cols=None, diag=0, name=None): return self._o("tri", [], name, iArgs=[rows, cols or rows, diag])
    def triu(self, x, diag=0, name=None): return self._o("triu", [x], name, iArgs=[diag])
    def mmul(self, a, b, transposeA=False, transposeB=False, name=None): return self._o("matmul", [a, b], name, iArgs=[int(transposeA), int(transposeB), 0])


class SDBitwise(GenSDBitwise, _NS):       # generated surface (codegen/ops/bitwise.py) + the overrides below
    def bitsAnd(self, a, b, name=None): return self._o("bitwise_and", [a, b], name)
    and_ = bitsAnd
    def bitsOr(self, a, b, name=None): return self._o("bitwise_or", [a, b], name)
    or_ = bitsOr
    def bitsXor(self, a, b, name=None): return self._o("bitwise_xor", [a, b], name)
    xor = bitsXor
    def leftShift(self, a, b, name=None): return self._o("shift_bits", [a, b], name)
    def rightShift(self, a, b, name=None): return self._o("rshift_bits", [a, b], name)
    def leftShiftCyclic(self, a, b, name=None): return self._o("cyclic_shift_bits", [a, b], name)
    def rightShiftCyclic(self, a, b, name=None): return self._o("cyclic_rshift_bits", [a, b], name)
    def bitsHammingDistance(self, a, b, name=None): return self._o("bits_hamming_distance", [a, b], name)


class SDRandom(GenSDRandom, _NS):       # generated surface (codegen/ops/random.py) + the overrides below
    def _shape(self, shape):
        import torch
        return self.sd.constant(None, torch.tensor(list(shape), dtype=torch.int64))

    def uniform(self, lo, hi, dtype="FLOAT", *shape, name=None): return self._o("randomuniform", [self._shape(_dims(shape))], name, tArgs=[lo, hi], dArgs=[dtype])
    def normal(self, mean, std, dtype="FLOAT", *shape, name=None): return self._o("random_normal", [self._shape(_dims(shape))], name, tArgs=[mean, std], dArgs=[dtype])
    def bernoulli(self, p, dtype="FLOAT", *shape, name=None): return self._o("random_bernoulli", [self._shape(_dims(shape))], name, tArgs=[p])
    def exponential(self, lam, dtype="FLOAT", *shape, name=None): return self._o("random_exponential", [self._shape(_dims(shape))], name, tArgs=[lam])


class SDImage(GenSDImage, _NS):       # generated surface (codegen/ops/image.py) + the overrides below
    def resizeBiLinear(self, x, h, w, alignCorners=False, halfPixelCenters=False, name=None):
        return self._o("resize_bilinear", [x], name, iArgs=[h, w], bArgs=[alignCorners, halfPixelCenters])
    def resizeNearestNeighbor(self, x, h, w, alignCorners=False, halfPixelCenters=False, name=None):
        return self._o("resize_nearest_neighbor", [x], name, iArgs=[h, w], bArgs=[alignCorners, halfPixelCenters])
    def resizeBiCubic(self, x, h, w, alignCorners=False, halfPixelCenters=False, name=None):
        return self._o("resize_bicubic", [x], name, iArgs=[h, w], bArgs=[alignCorners, halfPixelCenters])
    def cropAndResize(self, image, boxes, boxIdx, cropSize, method=0, extrapolation=0.0, name=None):
        return self._o("crop_and_resize", [image, boxes, boxIdx, cropSize], name, iArgs=[method], tArgs=[extrapolation])
    def nonMaxSuppression(self, boxes, scores, maxOut, iouThreshold=0.5, scoreThreshold=float("-inf"), name=None):
        return self._o("non_max_suppression", [boxes, scores], name, iArgs=[maxOut], tArgs=[iouThreshold, scoreThreshold])
    def rgbToHsv(self, x, name=None): return self._o("rgb_to_hsv", [x], name)
    def hsvToRgb(self, x, name=None): return self._o("hsv_to_rgb", [x], name)
    def rgbToYuv(self, x, name=None): return self._o("rgb_to_yuv", [x], name)
    def rgbToYiq(self, x, name=None): return self._o("rgb_to_yiq", [x], name)
    def adjustContrast(self, x, f, name=None): return self._o("adjust_contrast", [x], name, tArgs=[f])
    def adjustHue(self, x, d, name=None): return self._o("adjust_hue", [x], name, tArgs=[d])
    def adjustSaturation(self, x, f, name=None): return self._o("adjust_saturation", [x], name, tArgs=[f])
    def extractImagePatches(self, x, kH, kW, sH, sW, rH, rW, same=False, name=None): return self._o("extract_image_patches", [x], name, iArgs=[kH, kW, sH, sW, rH, rW, int(same)])


# ----------------------------------------------------------------------------- remaining generated-namespace surface
# (reference: nd4j-api autodiff/samediff/ops/SD{BaseOps,Math,NN,CNN,RNN,Bitwise,Image,Linalg,Random}.java, generated from the
# codegen op definitions). Every method below maps onto an op that is already registered and implemented.
def _install_namespace_extras():
    import torch
    from .samediff import OpType

    def B(name, fn):
        setattr(SDBaseOps, name, fn)

    def M(name, fn):
        setattr(SDMath, name, fn)

    def scalar_legacy(self, opn, x, value, name=None, family="scalar"):
        return self.sd._op(opn.lower(), [x], name=name, opType=OpType.SCALAR if family == "scalar" else OpType.SCALAR_BOOL,
                           opNum=LG.op_num(family, opn), tArgs=[float(value)])[0]

    # -- SDBaseOps
    B("all", lambda self, x, *dims, name=None: self._legacy("reduce_bool", "All", [x], name, dimensions=_dims(dims)))
    B("any", lambda self, x, *dims, name=None: self._legacy("reduce_bool", "Any", [x], name, dimensions=_dims(dims)))
    B("assign", lambda self, x, y, name=None: self._o("assign", [x, y if hasattr(y, "name") else self.sd.constant(None, torch.tensor(float(y)))], name))
    B("clipByValue", lambda self, x, lo, hi, name=None: self._o("clipbyvalue", [x], name, tArgs=[lo, hi]))
    B("clipByNorm", lambda self, x, c, *dims, name=None: self._o("clipbynorm", [x], name, tArgs=[c], iArgs=_dims(dims)))
    B("create", lambda self, shape, dtype="FLOAT", order="c", initialize=True, name=None:
      self._o("create", [shape], name, iArgs=[ord(order), 0], bArgs=[initialize], dArgs=[str(dtype).upper()]))
    B("createView", lambda self, x, *indices, name=None: self._o("create_view", [x] + list(indices), name))
    B("flatten", lambda self, xs, order="c", name=None: self._o("flatten", list(xs) if isinstance(xs, (list, tuple)) else [xs], name, iArgs=[ord(order)]))
    B("invertPermutation", lambda self, x, name=None: self._o("invert_permutation", [x], name))
    B("isNumericTensor", lambda self, x, name=None: self._o("is_numeric_tensor", [x], name))
    B("matchConditionCount", lambda self, x, mode, value=0.0, *dims, keepDims=False, eps=1e-5, name=None:
      self._legacy("reduce_long", "MatchCondition", [x], name, dimensions=_dims(dims), tArgs=[value, eps, mode], bArgs=[keepDims]))
    B("merge", lambda self, a, b, name=None: self._o("merge", [a, b], name))
    B("minMax", lambda self, dataType, minOrMax, name=None: self.sd.constant(name, torch.tensor(
        (torch.finfo if _TORCH_DT(dataType).is_floating_point else torch.iinfo)(_TORCH_DT(dataType)).max if minOrMax else
        (torch.finfo if _TORCH_DT(dataType).is_floating_point else torch.iinfo)(_TORCH_DT(dataType)).min, dtype=_TORCH_DT(dataType))))
    B("repeat", lambda self, x, repeats, axis=0, name=None: self._o("repeat", [x], name, iArgs=(list(repeats) if isinstance(repeats, (list, tuple)) else [repeats]) + [axis]))
    B("replaceWhere", lambda self, update, frm, condMode, value=0.0, name=None:
      self._legacy("pairwise", "CompareAndReplace", [update, frm], name, tArgs=[value, 1e-5, condMode]) if hasattr(frm, "name") else
      self._legacy("scalar", "CompareAndSet" if False else "CompareAndSet", [update], name, tArgs=[float(frm), value, 1e-5, condMode]) if False else
      self._o("Where", [_cond_var(self, update, condMode, value), update * 0 + frm, update], name))
    B("reverseSequence", lambda self, x, seqLengths, seqDim=-1, batchDim=0, name=None: self._o("reverse_sequence", [x, seqLengths], name, iArgs=[seqDim, batchDim]))
    B("scalarFloorMod", lambda self, x, v, name=None: scalar_legacy(self, "FloorMod", x, v, name))
    B("scalarMax", lambda self, x, v, name=None: scalar_legacy(self, "MaxPairwise", x, v, name))
    B("scalarMin", lambda self, x, v, name=None: scalar_legacy(self, "MinPairwise", x, v, name))
    B("scalarSet", lambda self, x, v, name=None: scalar_legacy(self, "CopyPws", x, v, name))
    for jn, opn in (("scatterDiv", "scatter_div"), ("scatterMax", "scatter_max"), ("scatterMin", "scatter_min"), ("scatterMul", "scatter_mul"),
                    ("scatterSub", "scatter_sub"), ("scatterNdAdd", "scatter_nd_add"), ("scatterNdSub", "scatter_nd_sub"),
                    ("scatterNdUpdate", "scatter_nd_update")):
        B(jn, (lambda o: (lambda self, ref, idx, upd, name=None: self._o(o, [ref, idx, upd], name)))(opn))
    for jn, opn in (("segmentMax", "segment_max"), ("segmentMean", "segment_mean"), ("segmentMin", "segment_min"), ("segmentProd", "segment_prod")):
        B(jn, (lambda o: (lambda self, x, ids, name=None: self._o(o, [x, ids], name)))(opn))
    for jn, opn in (("unsortedSegmentMax", "unsorted_segment_max"), ("unsortedSegmentMean", "unsorted_segment_mean"),
                    ("unsortedSegmentMin", "unsorted_segment_min"), ("unsortedSegmentProd", "unsorted_segment_prod"),
                    ("unsortedSegmentSqrtN", "unsorted_segment_sqrt_n")):
        B(jn, (lambda o: (lambda self, x, ids, n, name=None: self._o(o, [x, ids], name, iArgs=[n])))(opn))
    B("setShape", lambda self, x, shape, name=None: self._o("set_shape", [x, shape], name))
    B("sizeAt", lambda self, x, dim, name=None: self._o("size_at", [x], name, iArgs=[dim]))
    B("sparseToDense", lambda self, indices, shape, values, default=0.0, name=None:
      self._o("scatter_nd_update", [self.fill(shape, "FLOAT", default), indices, values], name))
    B("splitV", lambda self, x, sizes, numSplit, axis=0, name=None: self._o("split_v", [x, sizes], name, numSplit, iArgs=[axis]))
    B("switchOp", lambda self, x, pred, name=None: self._o("Switch", [x, pred], name, 2))
    B("whereNumpy", lambda self, cond, x, y, name=None: self._o("where_np", [cond, x, y], name))

    # -- SDMath (inherits the base ops)
    for jn, opn in (("acos", "ACos"), ("acosh", "ACosh"), ("asin", "ASin"), ("asinh", "ASinh"), ("atan", "ATan"), ("atanh", "ATanh")):
        fam = next(f for f in ("transform_strict", "transform_same", "transform_float") if opn in LG.FAMILIES[f].values())
        M(jn, _unary(fam, opn, jn))
    if not hasattr(SDMath, "ceil"):
        M("ceil", SDMath.ceiling)
    for jn, opn in (("add", "add"), ("sub", "subtract"), ("mul", "multiply"), ("div", "divide"), ("rsub", "reversesubtract"), ("rdiv", "reversedivide"),
                    ("floorDiv", "floordiv"), ("floorMod", "floormod"), ("mod", "mod"), ("divNoNan", "divide_no_nan"), ("squaredDifference", "squaredsubtract")):
        M(jn, (lambda o: (lambda self, x, y, name=None:
                          self._o(o, [x, y if hasattr(y, "name") else self.sd.constant(None, torch.tensor(float(y)))], name)))(opn))
    for jn, opn in (("and", "boolean_and"), ("or", "boolean_or"), ("xor", "boolean_xor")):
        M(jn, (lambda o: (lambda self, x, y, name=None: self._o(o, [x, y], name)))(opn))
    for jn, (fam, opn) in (("asum", ("reduce_same", "ASum")), ("reduceAMax", ("reduce_same", "AMax")), ("reduceAmin", ("reduce_same", "AMin")),
                           ("reduceAmean", ("reduce_float", "AMean")), ("reduceMax", ("reduce_same", "Max")), ("reduceMin", ("reduce_same", "Min")),
                           ("normMax", ("reduce_float", "NormMax")), ("logEntropy", ("reduce_float", "LogEntropy")),
                           ("shannonEntropy", ("reduce_float", "ShannonEntropy")), ("countZero", ("reduce_long", "CountZero"))):
        M(jn, (lambda f, o: (lambda self, x, *dims, keepDims=False, name=None:
                             self._legacy(f, o, [x], name, dimensions=_dims(dims), bArgs=[keepDims])))(fam, opn))
    for jn, opn in (("hammingDistance", "SimpleHammingDistance"), ("jaccardDistance", "JaccardDistance")):
        M(jn, (lambda o: (lambda self, a, b, *dims, name=None: self._legacy("reduce3", o, [a, b], name, dimensions=_dims(dims))))(opn))
    for jn, opn in (("firstIndex", "FirstIndex"), ("lastIndex", "LastIndex")):
        M(jn, (lambda o: (lambda self, x, mode, value=0.0, *dims, keepDims=False, name=None:
                          self._legacy("index_reduce", o, [x], name, dimensions=_dims(dims), tArgs=[value, 1e-5, mode], bArgs=[keepDims])))(opn))
    M("iamin", lambda self, x, *dims, name=None: self._legacy("index_reduce", "IndexAbsoluteMin", [x], name, dimensions=_dims(dims)))
    for jn, opn in (("bitShift", "shift_bits"), ("bitShiftRight", "rshift_bits"), ("bitShiftRotl", "cyclic_shift_bits"), ("bitShiftRotr", "cyclic_rshift_bits")):
        M(jn, (lambda o: (lambda self, x, shift, name=None: self._o(o, [x, shift], name)))(opn))
    M("clipByAvgNorm", lambda self, x, c, *dims, name=None: self._o("clipbyavgnorm", [x], name, tArgs=[c], iArgs=_dims(dims)))
    M("cross", lambda self, a, b, name=None: self._o("cross", [a, b], name))
    M("diagPart", lambda self, x, name=None: self._o("diag_part", [x], name))
    M("embeddingLookup", lambda self, x, indices, partitionMode=0, name=None: self._o("embedding_lookup", [x, indices], name, iArgs=[partitionMode]))
    M("isMax", lambda self, x, name=None: self._o("ismax", [x], name))
    M("isNonDecreasing", lambda self, x, name=None: self._o("is_non_decreasing", [x], name))
    M("isStrictlyIncreasing", lambda self, x, name=None: self._o("is_strictly_increasing", [x], name))
    M("lgamma", lambda self, x, name=None: self._o("lgamma", [x], name))
    M("listDiff", lambda self, x, y, name=None: self._o("listdiff", [x, y], name, 2))
    M("matrixDeterminant", lambda self, x, name=None: self._o("matrix_determinant", [x], name))
    M("matrixInverse", lambda self, x, name=None: self._o("matrix_inverse", [x], name))
    M("mergeMaxIndex", lambda self, *xs, dataType="INT", name=None: self._o("mergemaxindex", list(xs[0]) if len(xs) == 1 and isinstance(xs[0], (list, tuple)) else list(xs), name))
    M("meshgrid", lambda self, inputs, cartesian=True, name=None: self._o("meshgrid", list(inputs), name, len(inputs), iArgs=[int(cartesian)]))
    M("normalizeMoments", lambda self, counts, means, variances, shift=0.0, name=None: self._o("normalize_moments", [counts, means, variances], name, 2, tArgs=[shift]))
    M("setDiag", lambda self, x, diag, name=None: self._o("matrix_set_diag", [x, diag], name))
    M("step", lambda self, x, value=0.0, name=None: scalar_legacy(self, "Step", x, value, name))
    M("zeroFraction", lambda self, x, name=None: self._o("zero_fraction", [x], name))

    # -- SDNN
    N = lambda name, fn: setattr(SDNN, name, fn)   # noqa: E731
    N("cReLU", lambda self, x, name=None: self._o("crelu", [x], name))
    N("hardTanhDerivative", lambda self, x, name=None: self.sd.math._legacy("transform_strict", "HardTanhDerivative", [x], name)
      if "HardTanhDerivative" in LG.FAMILIES["transform_strict"].values() else self._o("hardtanh_bp", [x, x * 0 + 1], name))
    N("leakyReluDerivative", lambda self, x, alpha=0.01, name=None: self._o("lrelu_bp", [x, x * 0 + 1], name, tArgs=[alpha]))
    N("logSigmoid", lambda self, x, name=None: self.sd.math._legacy("transform_strict", "LogSigmoid", [x], name))
    N("sigmoidDerivative", lambda self, x, wrt=None, name=None: self._o("sigmoid_bp", [x, wrt if wrt is not None else x * 0 + 1], name))
    N("softsignDerivative", lambda self, x, name=None: self._o("softsign_bp", [x, x * 0 + 1], name))
    N("topK", lambda self, x, k, sorted=True, name=None: self._o("top_k", [x], name, 2, iArgs=[k], bArgs=[sorted]))

    # -- SDCNN
    C = lambda name, fn: setattr(SDCNN, name, fn)   # noqa: E731
    C("deconv3d", lambda self, x, w, b=None, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), d=(1, 1, 1), sameMode=False, ncdhw=True, name=None:
      self._o("deconv3d", [x, w] + ([b] if b is not None else []), name, iArgs=list(k) + list(s) + list(p) + list(d) + [int(sameMode), 0 if ncdhw else 1]))
    C("dilation2D", lambda self, x, w, rates, strides, sameMode=False, name=None:
      self._o("dilation2d", [x, w], name, iArgs=[int(sameMode)] + list(rates) + list(strides)))
    C("extractImagePatches", lambda self, x, kH, kW, sH=1, sW=1, rH=1, rW=1, sameMode=False, name=None:
      self._o("extract_image_patches", [x], name, iArgs=[kH, kW, sH, sW, rH, rW, int(sameMode)]))
    C("maxPoolWithArgmax", lambda self, x, k=(2, 2), s=(2, 2), p=(0, 0), d=(1, 1), sameMode=False, nchw=True, name=None:
      self._o("max_pool_with_argmax", [x], name, 2, iArgs=list(k) + list(s) + list(p) + list(d) + [int(sameMode), 1, 0 if nchw else 1]))
    C("upsampling3d", lambda self, x, ncdhw=True, scaleD=2, scaleH=2, scaleW=2, name=None:
      self._o("upsampling3d", [x], name, iArgs=[scaleD, scaleH, scaleW, int(ncdhw)]))

    # -- SDRNN
    setattr(SDRNN, "lstmblock", lambda self, maxTSLength, x, cLast, yLast, W, Wci, Wcf, Wco, b, peephole=False, forgetBias=0.0, clip=0.0, name=None:
            self._o("lstmBlock", [maxTSLength, x, cLast, yLast, W, Wci, Wcf, Wco, b], name, 7, iArgs=[int(peephole), 0], tArgs=[forgetBias, clip]))

    # -- SDBitwise (Java names; `and` / `or` are Python keywords, reachable through getattr as in the reference's generated class)
    for jn, src in (("and", "bitsAnd"), ("or", "bitsOr"), ("bitShift", "leftShift"), ("bitShiftRight", "rightShift"), ("bitRotl", "leftShiftCyclic"),
                    ("bitRotr", "rightShiftCyclic")):
        setattr(SDBitwise, jn, getattr(SDBitwise, src))

    # -- SDImage
    I = lambda name, fn: setattr(SDImage, name, fn)   # noqa: E731
    I("imageResize", lambda self, x, size, preserveAspectRatio=False, antialias=False, method=0, name=None:
      self._o("image_resize", [x, size], name, iArgs=[method], bArgs=[preserveAspectRatio, antialias]))
    I("pad", lambda self, x, padding, mode="CONSTANT", value=0.0, name=None:
      self._o("pad", [x, padding], name, iArgs=[{"CONSTANT": 0, "REFLECT": 1, "SYMMETRIC": 2}[str(mode).upper()]], tArgs=[value]))
    I("randomCrop", lambda self, x, shape, seed=0, name=None: self._o("random_crop", [x, shape], name, iArgs=[seed]))
    I("yiqToRgb", lambda self, x, name=None: self._o("yiq_to_rgb", [x], name))
    I("yuvToRgb", lambda self, x, name=None: self._o("yuv_to_rgb", [x], name))

    # -- SDLinalg
    Lg = lambda name, fn: setattr(SDLinalg, name, fn)   # noqa: E731
    Lg("diag_part", lambda self, x, name=None: self._o("diag_part", [x], name))
    Lg("eig", lambda self, x, name=None: self._o("eig", [x], name, 2))
    Lg("matmul", lambda self, a, b, alpha=1.0, beta=0.0, transposeA=False, transposeB=False, name=None:
       self._o("matmul", [a, b], name, iArgs=[int(transposeA), int(transposeB), 0], tArgs=[alpha, beta]))

    # -- SDRandom (legacy random ops: shape input, distribution parameters as tArgs)
    def rnd(opn, nparams):
        def f(self, *args, name=None):
            params, rest = list(args[:nparams]), list(args[nparams:])
            dtype = rest.pop(0) if rest and isinstance(rest[0], str) else "FLOAT"
            return self.sd._op(opn.lower(), [self._shape(_dims(rest))], name=name, opType=OpType.RANDOM, opNum=LG.op_num("random", opn),
                               tArgs=[float(p) for p in params], dArgs=[dtype])[0]
        return f
    setattr(SDRandom, "binomial", rnd("BinomialDistribution", 2))
    setattr(SDRandom, "logNormal", rnd("LogNormalDistribution", 2))
    setattr(SDRandom, "normalTruncated", rnd("TruncatedNormalDistribution", 2))


def _TORCH_DT(dt):
    from ..nd4j.ndarray import DataType
    return DataType.to_torch(dt)


def _cond_var(ns, x, mode, value):
    return ns.matchCondition(x, mode, value)


_install_namespace_extras()
