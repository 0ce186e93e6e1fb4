"""Math namespace (reference definition: codegen/op-codegen/src/main/ops/org/nd4j/codegen/ops/Math.kt): element-wise transforms of the
legacy families, pairwise / scalar arithmetic, distance and statistics reductions, merge ops."""
from ..dsl import Count, DataType as D, Namespace

ns = Namespace("math", doc="Element-wise math, distances, statistics reductions.", className="Math")

_DIMS = "Dimensions to reduce over; none = all dimensions"


def _transform(name, family, legacy_name, doc, out=D.NUMERIC):
    return (ns.op(name, doc=doc).legacyOp(family, legacy_name).input("x", D.NUMERIC, "Input array").output("output", out, "Result, same shape as x"))


_SAME = (("abs", "Abs", "Absolute value"), ("ceil", "Ceiling", "Round up to the next integer"), ("floor", "Floor", "Round down to the previous integer"),
         ("round", "Round", "Round half away from zero"), ("neg", "Neg", "Negation"), ("sign", "Sign", "Sign (-1, 0, 1)"), ("square", "Square", "x^2"),
         ("cube", "Cube", "x^3"), ("reciprocal", "Reciprocal", "1 / x"), ("oneMinus", "OneMinus", "1 - x"), ("timesOneMinus", "TimesOneMinus", "x * (1 - x)"))
_STRICT = (("acos", "ACos", "Arc cosine"), ("acosh", "ACosh", "Inverse hyperbolic cosine"), ("asin", "ASin", "Arc sine"), ("asinh", "ASinh", "Inverse hyperbolic sine"),
           ("atan", "ATan", "Arc tangent"), ("atanh", "ATanh", "Inverse hyperbolic tangent"), ("cos", "Cosine", "Cosine"), ("cosh", "Cosh", "Hyperbolic cosine"),
           ("sin", "Sin", "Sine"), ("sinh", "Sinh", "Hyperbolic sine"), ("tan", "Tan", "Tangent"), ("tanh", "Tanh", "Hyperbolic tangent"),
           ("exp", "Exp", "e^x"), ("expm1", "Expm1", "e^x - 1"), ("log", "Log", "Natural logarithm"), ("log1p", "Log1p", "log(1 + x)"),
           ("erf", "Erf", "Error function"), ("erfc", "Erfc", "Complementary error function"), ("rint", "Rint", "Round half to even"),
           ("sigmoid", "Sigmoid", "Logistic sigmoid"), ("logSigmoid", "LogSigmoid", "log(sigmoid(x))"), ("softplus", "SoftPlus", "log(1 + e^x)"),
           ("softsign", "SoftSign", "x / (1 + |x|)"), ("hardTanh", "HardTanh", "Clipped linear tanh"), ("hardSigmoid", "HardSigmoid", "Piecewise-linear sigmoid"),
           ("rationalTanh", "RationalTanh", "Rational approximation of tanh"), ("rectifiedTanh", "RectifiedTanh", "max(0, tanh(x))"),
           ("selu", "SELU", "Scaled exponential linear unit"), ("swish", "Swish", "x * sigmoid(x)"), ("mish", "Mish", "x * tanh(softplus(x))"),
           ("gelu", "GELU", "Gaussian error linear unit (sigmoid approximation)"), ("preciseGelu", "PreciseGELU", "Gaussian error linear unit (tanh form)"))
for _py, _lg, _doc in _SAME:
    _transform(_py, "transform_same", _lg, _doc + ", element-wise.")
for _py, _lg, _doc in _STRICT:
    _transform(_py, "transform_strict", _lg, _doc + ", element-wise.", D.FLOATING_POINT)
_transform("sqrt", "transform_float", "Sqrt", "Square root, element-wise.", D.FLOATING_POINT)
_transform("rsqrt", "transform_float", "RSqrt", "Reciprocal square root, element-wise.", D.FLOATING_POINT)
for _py, _lg, _doc in (("isFinite", "IsFinite", "finite"), ("isInfinite", "IsInf", "infinite"), ("isNaN", "IsNan", "NaN")):
    _transform(_py, "transform_bool", _lg, "True where the element is %s." % _doc, D.BOOL)

# ---- pairwise / scalar
(ns.op("pow", libnd4j="Pow", doc="Element-wise power x^y (broadcastable).").input("x", D.NUMERIC, "Base").input("y", D.NUMERIC, "Exponent").output("output", D.NUMERIC, "x^y"))
(ns.op("atan2", libnd4j="tf_atan2", doc="Element-wise arc tangent of y / x using the signs of both to select the quadrant.")
   .input("y", D.NUMERIC, "Numerator").input("x", D.NUMERIC, "Denominator").output("output", D.FLOATING_POINT, "Angle in radians"))
for _n, _lib, _d in (("floorDiv", "floordiv", "floor(x / y)"), ("floorMod", "floormod", "x - floor(x / y) * y"), ("squaredDifference", "squaredsubtract", "(x - y)^2"),
                     ("mod", "mod", "remainder of x / y")):
    (ns.op(_n, libnd4j=_lib, doc="Element-wise %s (broadcastable)." % _d).input("x", D.NUMERIC, "First operand").input("y", D.NUMERIC, "Second operand")
       .output("output", D.NUMERIC, _d))
# ---- reductions of the legacy families
for _n, _lg, _d in (("cosineSimilarity", "CosineSimilarity", "Cosine similarity"), ("cosineDistance", "CosineDistance", "Cosine distance (1 - similarity)"),
                    ("euclideanDistance", "EuclideanDistance", "Euclidean distance"), ("manhattanDistance", "ManhattanDistance", "Manhattan (L1) distance"),
                    ("jaccardDistance", "JaccardDistance", "Jaccard distance"), ("hammingDistance", "SimpleHammingDistance", "Hamming distance (count of differing elements)")):
    (ns.op(_n, doc=_d + " between x and y over the dimensions.").legacyOp("reduce3", _lg).input("x", D.NUMERIC, "First array").input("y", D.NUMERIC, "Second array")
       .arg("dimensions", D.INT, _DIMS, count=Count.AtLeast(0), slot="dims").output("output", D.FLOATING_POINT, "Reduced array"))
for _n, _fam, _lg, _d, _t in (("countNonZero", "reduce_long", "CountNonZero", "Number of non-zero elements", D.LONG), ("countZero", "reduce_long", "CountZero", "Number of zero elements", D.LONG),
                              ("entropy", "reduce_float", "Entropy", "Entropy -sum(x log x)", D.FLOATING_POINT), ("logEntropy", "reduce_float", "LogEntropy", "Log of the entropy", D.FLOATING_POINT),
                              ("shannonEntropy", "reduce_float", "ShannonEntropy", "Shannon entropy -sum(x^2 log2 x^2)", D.FLOATING_POINT), ("amean", "reduce_float", "AMean", "Mean of absolute values", D.FLOATING_POINT),
                              ("asum", "reduce_same", "ASum", "Sum of absolute values", D.NUMERIC), ("amax", "reduce_same", "AMax", "Largest absolute value", D.NUMERIC),
                              ("amin", "reduce_same", "AMin", "Smallest absolute value", D.NUMERIC), ("iamax", "index_reduce", "IndexAbsoluteMax", "Index of the largest absolute value", D.LONG),
                              ("iamin", "index_reduce", "IndexAbsoluteMin", "Index of the smallest absolute value", D.LONG)):
    (ns.op(_n, doc=_d + " over the dimensions.").legacyOp(_fam, _lg).input("in_", D.NUMERIC, "Input array")
       .arg("dimensions", D.INT, _DIMS, count=Count.AtLeast(0), slot="dims").output("output", _t, "Reduced array"))
(ns.op("logSumExp", libnd4j="reduce_logsumexp", doc="log(sum(exp(x))) over the dimensions, computed stably.").input("input", D.NUMERIC, "Input array")
   .arg("dimensions", D.INT, _DIMS, count=Count.AtLeast(0)).output("output", D.FLOATING_POINT, "Reduced array"))
(ns.op("standardize", doc="Subtract the mean and divide by the standard deviation over the dimensions.").input("x", D.NUMERIC, "Input array")
   .arg("dimensions", D.INT, "Dimensions the statistics are taken over", count=Count.AtLeast(1)).output("output", D.FLOATING_POINT, "Standardized array"))
(ns.op("moments", doc="Mean and variance over the axes.").input("input", D.NUMERIC, "Input array").arg("axes", D.INT, "Axes to reduce over", count=Count.AtLeast(0))
   .output("mean", D.FLOATING_POINT, "Mean").output("variance", D.FLOATING_POINT, "Variance"))
# ---- matrices / merges
ns.op("diag", doc="Diagonal tensor with the input on its diagonal.").input("x", D.NUMERIC, "Input array").output("output", D.NUMERIC, "Tensor of rank 2r")
ns.op("diagPart", libnd4j="diag_part", doc="Diagonal part of a tensor.").input("x", D.NUMERIC, "Input of even rank").output("output", D.NUMERIC, "Diagonal")
ns.op("trace", doc="Sum of the main diagonal.").input("in_", D.NUMERIC, "Input matrix").output("output", D.NUMERIC, "Scalar trace")
for _n, _lib, _d in (("mergeAdd", "mergeadd", "sum"), ("mergeAvg", "mergeavg", "average"), ("mergeMax", "mergemax", "maximum")):
    (ns.op(_n, libnd4j=_lib, doc="Element-wise %s of any number of equally shaped arrays." % _d).input("inputs", D.NUMERIC, "Input arrays", count=Count.AtLeast(1))
       .output("output", D.NUMERIC, "Merged array"))
(ns.op("confusionMatrix", libnd4j="confusion_matrix", doc="Confusion matrix of integer labels versus predictions.").input("labels", D.NUMERIC, "True class indices")
   .input("pred", D.NUMERIC, "Predicted class indices").arg("numClasses", D.INT, "Number of classes").output("output", D.NUMERIC, "[numClasses, numClasses] counts"))
(ns.op("isNonDecreasing", libnd4j="is_non_decreasing", doc="Scalar: the flattened array never decreases.").input("x", D.NUMERIC, "Input array").output("output", D.BOOL, "Scalar flag"))
(ns.op("isStrictlyIncreasing", libnd4j="is_strictly_increasing", doc="Scalar: the flattened array strictly increases.").input("x", D.NUMERIC, "Input array").output("output", D.BOOL, "Scalar flag"))
(ns.op("isNumericTensor", libnd4j="is_numeric_tensor", doc="Scalar: the array has a numeric data type.").input("x", D.NUMERIC, "Input array").output("output", D.BOOL, "Scalar flag"))
(ns.op("listDiff", libnd4j="listdiff", doc="Values of x that are not in y, with their indices in x.").input("x", D.NUMERIC, "Values").input("y", D.NUMERIC, "Values to remove")
   .output("output1", D.NUMERIC, "Remaining values").output("output2", D.INT, "Their indices in x"))
(ns.op("zeroFraction", libnd4j="zero_fraction", doc="Fraction of elements that are zero.").input("input", D.NUMERIC, "Input array").output("output", D.FLOATING_POINT, "Scalar fraction"))
