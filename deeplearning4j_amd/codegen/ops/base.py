"""Base-ops namespace: shape manipulation, indexing, reductions, comparisons, segment ops (reference definition:
codegen/op-codegen/src/main/ops/org/nd4j/codegen/ops/SDBaseOps.kt). ``sd.<op>`` / ``sd.base.<op>`` and ``NDBase().<op>``."""
from ..dsl import Count, DataType as D, Namespace

ns = Namespace("base", doc="Core array operations: shape manipulation, indexing, reductions, comparisons, segments.", className="Base")

_DIMS = "Dimensions to reduce over; none = all dimensions"


def _reduce(name, lib, doc, extra_b=()):
    o = (ns.op(name, libnd4j=lib, doc=doc).input("x", D.NUMERIC, "Input array")
         .arg("keepDims", D.BOOL, "Keep the reduced dimensions with size 1", default=False))
    for b, bdoc in extra_b:
        o.arg(b, D.BOOL, bdoc, default=False)
    o.arg("dimensions", D.INT, _DIMS, count=Count.AtLeast(0)).output("output", D.NUMERIC, "Reduced array")
    return o


# ---- shape manipulation
ns.op("identity", doc="Pass the input through unchanged.").input("input", D.NUMERIC, "Input array").output("output", D.NUMERIC, "Same values")
(ns.op("concat", doc="Concatenate arrays along one dimension.").arg("dimension", D.INT, "Dimension to concatenate along")
   .input("inputs", D.NUMERIC, "Arrays of equal shape except along the concatenation dimension", count=Count.AtLeast(1))
   .output("output", D.NUMERIC, "Concatenated array"))
(ns.op("stack", doc="Stack arrays of equal shape along a new axis.").arg("axis", D.INT, "Position of the new axis")
   .input("values", D.NUMERIC, "Arrays to stack", count=Count.AtLeast(1)).output("output", D.NUMERIC, "Array of rank r + 1"))
(ns.op("unstack", doc="Split an array into ``num`` arrays along an axis, removing that axis.").input("value", D.NUMERIC, "Input array")
   .arg("axis", D.INT, "Axis to unstack on").arg("num", D.INT, "Number of outputs (size of the axis)", slot="i")
   .output("output", D.NUMERIC, "``num`` arrays of rank r - 1").numOutputs("num"))
(ns.op("reshape", doc="Reshape to the given shape (one -1 entry is inferred).").input("x", D.NUMERIC, "Input array")
   .arg("shape", D.LONG, "New shape", count=Count.AtLeast(0)).output("output", D.NUMERIC, "Reshaped array"))
(ns.op("permute", doc="Permute the dimensions.").input("x", D.NUMERIC, "Input array")
   .arg("dimensions", D.INT, "New order of the dimensions", count=Count.AtLeast(0)).output("output", D.NUMERIC, "Permuted array"))
ns.op("transpose", doc="Reverse the order of the dimensions (matrix transpose for rank 2).").input("x", D.NUMERIC, "Input array").output("output", D.NUMERIC, "Transposed array")
(ns.op("expandDims", libnd4j="expand_dims", doc="Insert a size-1 dimension.").input("x", D.NUMERIC, "Input array").arg("axis", D.INT, "Position of the new dimension")
   .output("output", D.NUMERIC, "Array of rank r + 1"))
(ns.op("squeeze", doc="Remove a size-1 dimension.").input("x", D.NUMERIC, "Input array").arg("axis", D.INT, "Dimension to remove").output("output", D.NUMERIC, "Array of rank r - 1"))
(ns.op("tile", doc="Repeat the array along each dimension.").input("x", D.NUMERIC, "Input array")
   .arg("repeat", D.INT, "Repetitions per dimension", count=Count.AtLeast(1)).output("output", D.NUMERIC, "Tiled array"))
(ns.op("reverse", doc="Reverse the order of the elements along the given dimensions.").input("x", D.NUMERIC, "Input array")
   .arg("dimensions", D.INT, "Dimensions to reverse", count=Count.AtLeast(0)).output("output", D.NUMERIC, "Reversed array"))
(ns.op("reverseSequence", libnd4j="reverse_sequence", doc="Reverse each sequence of a batch up to its own length.")
   .input("x", D.NUMERIC, "Input array").input("seq_lengths", D.INT, "Length of every sequence")
   .arg("seqDim", D.INT, "Sequence (time) dimension", default=-1).arg("batchDim", D.INT, "Batch dimension", default=0).output("output", D.NUMERIC, "Array with reversed prefixes"))
# ---- indexing
(ns.op("gather", doc="Gather slices along an axis at the given indices.").input("df", D.NUMERIC, "Input array").input("indices", D.INT, "Indices to take")
   .arg("axis", D.INT, "Axis to gather along", default=0).output("output", D.NUMERIC, "Gathered slices"))
(ns.op("gatherNd", libnd4j="gather_nd", doc="Gather slices addressed by index tuples.").input("df", D.NUMERIC, "Input array").input("indices", D.INT, "Index tuples in the last dimension")
   .output("output", D.NUMERIC, "Gathered slices"))
for _n, _lib, _d in (("scatterAdd", "scatter_add", "adds the updates to"), ("scatterSub", "scatter_sub", "subtracts the updates from"),
                     ("scatterMul", "scatter_mul", "multiplies the updates into"), ("scatterDiv", "scatter_div", "divides by the updates at"),
                     ("scatterMax", "scatter_max", "takes the maximum of the updates and"), ("scatterMin", "scatter_min", "takes the minimum of the updates and"),
                     ("scatterUpdate", "scatter_upd", "writes the updates over")):
    (ns.op(_n, libnd4j=_lib, doc="Scatter: %s the indexed slices of the reference array." % _d).input("ref", D.NUMERIC, "Reference array")
       .input("indices", D.INT, "Indices of the first dimension").input("updates", D.NUMERIC, "Update slices").output("output", D.NUMERIC, "Updated array"))
(ns.op("oneHot", libnd4j="onehot", doc="One-hot encode integer indices.").input("indices", D.NUMERIC, "Class indices")
   .arg("depth", D.INT, "Number of classes", pos=1).arg("axis", D.INT, "Position of the new class dimension", default=-1, pos=0)
   .arg("on", D.FLOATING_POINT, "Value at the index", default=1.0).arg("off", D.FLOATING_POINT, "Value elsewhere", default=0.0).output("output", D.NUMERIC, "One-hot array"))
(ns.op("castTo", libnd4j="cast", doc="Convert to another data type.").input("arg", D.NUMERIC, "Input array").arg("datatype", D.DATA_TYPE, "Target data type")
   .output("output", D.NUMERIC, "Converted array"))
ns.op("shape", libnd4j="shape_of", doc="Shape of the array as a 1-D LONG array.").input("input", D.NUMERIC, "Input array").output("output", D.LONG, "Shape vector")
ns.op("size", doc="Total number of elements.").input("in_", D.NUMERIC, "Input array").output("output", D.LONG, "Scalar element count")
ns.op("rank", doc="Number of dimensions.").input("in_", D.NUMERIC, "Input array").output("output", D.INT, "Scalar rank")
ns.op("zerosLike", libnd4j="zeros_as", doc="Array of zeros with the shape and type of the input.").input("input", D.NUMERIC, "Template").output("output", D.NUMERIC, "Zeros")
ns.op("onesLike", libnd4j="ones_as", doc="Array of ones with the shape and type of the input.").input("input", D.NUMERIC, "Template").output("output", D.NUMERIC, "Ones")
(ns.op("fill", doc="Array of the given shape filled with a value.").input("shape", D.INT, "Shape as a 1-D integer array")
   .arg("dataType", D.DATA_TYPE, "Data type of the result").arg("value", D.FLOATING_POINT, "Fill value").output("output", D.NUMERIC, "Filled array"))
(ns.op("sequenceMask", libnd4j="sequence_mask", doc="Boolean / numeric mask [batch, maxLen] with ones before each sequence's length.")
   .input("lengths", D.NUMERIC, "Sequence lengths").arg("maxLen", D.INT, "Mask width").arg("dataType", D.DATA_TYPE, "Result type", default="BOOL")
   .output("output", D.NUMERIC, "Mask"))
(ns.op("clipByValue", libnd4j="clipbyvalue", doc="Clip every element to [clipValueMin, clipValueMax].").input("x", D.NUMERIC, "Input array")
   .arg("clipValueMin", D.FLOATING_POINT, "Lower bound").arg("clipValueMax", D.FLOATING_POINT, "Upper bound").output("output", D.NUMERIC, "Clipped array"))
(ns.op("clipByNorm", libnd4j="clipbynorm", doc="Scale the array (or each sub-tensor over the dimensions) down to the given L2 norm when it exceeds it.")
   .input("x", D.NUMERIC, "Input array").arg("clipValue", D.FLOATING_POINT, "Maximum norm")
   .arg("dimensions", D.INT, "Dimensions the norm is taken over; none = whole array", count=Count.AtLeast(0)).output("output", D.NUMERIC, "Clipped array"))
# ---- matrix products
(ns.op("mmul", libnd4j="matmul", doc="Matrix product with optional transposes.").input("x", D.NUMERIC, "First operand").input("y", D.NUMERIC, "Second operand")
   .arg("transposeX", D.BOOL, "Transpose x", default=False, slot="i", convert="int").arg("transposeY", D.BOOL, "Transpose y", default=False, slot="i", convert="int")
   .arg("transposeZ", D.BOOL, "Transpose the result", default=False, slot="i", convert="int").output("output", D.NUMERIC, "x . y"))
# ---- reductions
_reduce("sum", "reduce_sum", "Sum over the dimensions.")
_reduce("mean", "reduce_mean", "Arithmetic mean over the dimensions.")
_reduce("max", "reduce_max", "Maximum over the dimensions.")
_reduce("min", "reduce_min", "Minimum over the dimensions.")
_reduce("prod", "reduce_prod", "Product over the dimensions.")
_reduce("norm1", "reduce_norm1", "L1 norm (sum of absolute values) over the dimensions.")
_reduce("norm2", "reduce_norm2", "L2 (Euclidean) norm over the dimensions.")
_reduce("normmax", "reduce_norm_max", "Max norm (largest absolute value) over the dimensions.")
_reduce("squaredNorm", "reduce_sqnorm", "Sum of squares over the dimensions.")
_reduce("variance", "reduce_variance", "Variance over the dimensions.", (("biasCorrected", "Divide by n - 1 instead of n"),))
_reduce("standardDeviation", "reduce_stdev", "Standard deviation over the dimensions.", (("biasCorrected", "Divide by n - 1 instead of n"),))
_reduce("argmax", "argmax", "Index of the maximum along the dimensions.")
_reduce("argmin", "argmin", "Index of the minimum along the dimensions.")
for _n, _lib in (("cumsum", "cumsum"), ("cumprod", "cumprod")):
    (ns.op(_n, libnd4j=_lib, doc="Cumulative %s along the given axes." % ("sum" if _n == "cumsum" else "product")).input("in_", D.NUMERIC, "Input array")
       .arg("exclusive", D.BOOL, "Exclude the current element from each partial result", default=False, slot="i", convert="int")
       .arg("reverse", D.BOOL, "Accumulate from the end", default=False, slot="i", convert="int")
       .arg("axis", D.INT, "Axes to accumulate along", count=Count.AtLeast(0)).output("output", D.NUMERIC, "Cumulative result"))
# ---- comparisons (broadcastable, boolean result)
for _n, _lib, _sym in (("eq", "equals", "=="), ("neq", "not_equals", "!="), ("gt", "greater", ">"), ("gte", "greater_equal", ">="), ("lt", "less", "<"),
                       ("lte", "less_equal", "<=")):
    (ns.op(_n, libnd4j=_lib, doc="Element-wise x %s y." % _sym).input("x", D.NUMERIC, "First operand").input("y", D.NUMERIC, "Second operand")
       .output("output", D.BOOL, "Boolean array"))
# ---- segments
for _kind in ("Sum", "Max", "Min", "Mean", "Prod"):
    (ns.op("segment" + _kind, libnd4j="segment_" + _kind.lower(), doc="%s of the slices that share a (sorted) segment id." % _kind)
       .input("data", D.NUMERIC, "Input array").input("segmentIds", D.INT, "Sorted segment id of every slice").output("output", D.NUMERIC, "One slice per segment"))
    (ns.op("unsortedSegment" + _kind, libnd4j="unsorted_segment_" + _kind.lower(), doc="%s of the slices that share a segment id (ids in any order)." % _kind)
       .input("data", D.NUMERIC, "Input array").input("segmentIds", D.INT, "Segment id of every slice").arg("numSegments", D.INT, "Number of segments")
       .output("output", D.NUMERIC, "One slice per segment"))
(ns.op("dynamicPartition", libnd4j="dynamic_partition", doc="Split the input into ``numPartitions`` arrays according to a partition index per element.")
   .input("x", D.NUMERIC, "Input array").input("partitions", D.INT, "Partition index of every element").arg("numPartitions", D.INT, "Number of outputs")
   .output("output", D.NUMERIC, "One array per partition").numOutputs("numPartitions"))
