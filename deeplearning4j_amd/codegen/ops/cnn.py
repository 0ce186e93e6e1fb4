"""Convolution namespace (reference definition: codegen/op-codegen/src/main/ops/org/nd4j/codegen/ops/CNN.kt). The reference passes
Conv2DConfig-style objects; here the same fields are flat keyword arguments in the order of the native op's integer argument vector."""
from ..dsl import DataType as D, Namespace

ns = Namespace("cnn", doc="Convolution, pooling and spatial rearrangement ops.", className="CNN")


def _geom2d(o, kernel_default=None):
    for n, d, dv in (("kH", "Kernel height", kernel_default), ("kW", "Kernel width", kernel_default)):
        if dv is None:
            o.arg(n, D.INT, d)
        else:
            o.arg(n, D.INT, d, default=dv)
    for n, d, dv in (("sH", "Stride height", 1), ("sW", "Stride width", 1), ("pH", "Padding height", 0), ("pW", "Padding width", 0),
                     ("dH", "Dilation height", 1), ("dW", "Dilation width", 1)):
        o.arg(n, D.INT, d, default=dv)
    o.arg("isSameMode", D.BOOL, "SAME padding (output = ceil(input / stride)) instead of the explicit padding", default=False, slot="i", convert="int")
    return o


def _conv(name, lib, doc, inputs):
    o = ns.op(name, libnd4j=lib, doc=doc)
    for n, d in inputs:
        o.input(n, D.NUMERIC, d)
    o.input("bias", D.NUMERIC, "Bias [outputChannels]", optional=True)
    _geom2d(o, 1)
    o.arg("isNHWC", D.BOOL, "Activations are NHWC instead of NCHW", default=False, slot="i", convert="int")
    return o


(_conv("conv2d", "conv2d", "2-D convolution.", (("layerInput", "Activations [mb, c, h, w]"), ("weights", "Filters")))
 .arg("weightsFormat", D.ENUM, "Filter layout", default="YXIO", possibleValues=["YXIO", "OIYX", "OYXI"]).output("output", D.NUMERIC, "Convolved activations"))
_conv("depthWiseConv2d", "depthwise_conv2d", "Depth-wise 2-D convolution.", (("layerInput", "Activations"), ("depthWeights", "Depth-wise filters"))).output("output", D.NUMERIC, "Activations")
(_conv("separableConv2d", "sconv2d", "Depth-wise convolution followed by a 1x1 point-wise convolution.",
       (("layerInput", "Activations"), ("depthWeights", "Depth-wise filters"), ("pointWeights", "Point-wise filters"))).output("output", D.NUMERIC, "Activations"))
_conv("deconv2d", "deconv2d", "Transposed 2-D convolution.", (("layerInput", "Activations"), ("weights", "Filters"))).output("output", D.NUMERIC, "Activations")
(ns.op("conv1d", doc="1-D convolution.").input("input", D.NUMERIC, "Activations [mb, c, t]").input("weights", D.NUMERIC, "Filters").input("bias", D.NUMERIC, "Bias", optional=True)
   .arg("k", D.INT, "Kernel size", default=1).arg("s", D.INT, "Stride", default=1).arg("p", D.INT, "Padding", default=0).arg("d", D.INT, "Dilation", default=1)
   .arg("paddingMode", D.ENUM, "Padding mode", default="VALID", possibleValues=["VALID", "SAME", "CAUSAL"])
   .arg("isNWC", D.BOOL, "Activations are [mb, t, c]", default=False, slot="i", convert="int").output("output", D.NUMERIC, "Activations"))
for _n, _lib, _d in (("maxPooling2d", "maxpool2d", "Max"), ("avgPooling2d", "avgpool2d", "Average")):
    o = ns.op(_n, libnd4j=_lib, doc="%s pooling over 2-D windows." % _d).input("input", D.NUMERIC, "Activations")
    _geom2d(o)
    (o.arg("extraParam0", D.INT, "avg: 0 = divide by the window size including padding, 1 = excluding; unused for max", default=0)
      .arg("isNHWC", D.BOOL, "Activations are NHWC", default=False, slot="i", convert="int").output("output", D.NUMERIC, "Pooled activations"))
(ns.op("upsampling2d", doc="Nearest-neighbour upsampling by integer factors.").input("input", D.NUMERIC, "Activations")
   .arg("scaleH", D.INT, "Height factor").arg("scaleW", D.INT, "Width factor").arg("nchw", D.BOOL, "Activations are NCHW", default=True, slot="i", convert="int")
   .output("output", D.NUMERIC, "Upsampled activations"))
o = ns.op("im2Col", libnd4j="im2col", doc="Unfold convolution windows into columns.").input("in_", D.NUMERIC, "Activations [mb, c, h, w]")
_geom2d(o).output("output", D.NUMERIC, "[mb, c, kH, kW, oH, oW]")
(ns.op("col2Im", libnd4j="col2im", doc="Fold columns back into an image (adjoint of im2Col).").input("in_", D.NUMERIC, "Columns [mb, c, kH, kW, oH, oW]")
   .arg("sH", D.INT, "Stride height").arg("sW", D.INT, "Stride width").arg("pH", D.INT, "Padding height").arg("pW", D.INT, "Padding width")
   .arg("iH", D.INT, "Image height").arg("iW", D.INT, "Image width").arg("dH", D.INT, "Dilation height", default=1).arg("dW", D.INT, "Dilation width", default=1)
   .output("output", D.NUMERIC, "[mb, c, iH, iW]"))
for _n, _lib, _d in (("spaceToDepth", "space_to_depth", "Move spatial blocks into the channel dimension"), ("depthToSpace", "depth_to_space", "Move channel groups into spatial blocks")):
    (ns.op(_n, libnd4j=_lib, doc=_d + ".").input("x", D.NUMERIC, "Activations").arg("blockSize", D.INT, "Block edge")
       .arg("isNHWC", D.BOOL, "Activations are NHWC", default=False, slot="i", convert="int").output("output", D.NUMERIC, "Rearranged activations"))
(ns.op("spaceToBatch", libnd4j="space_to_batch", doc="Move spatial blocks into the batch dimension (with padding).").input("x", D.NUMERIC, "Activations")
   .input("padding", D.NUMERIC, "[2, 2] paddings").arg("blockSize", D.INT, "Block edge").output("output", D.NUMERIC, "Rearranged activations"))
(ns.op("batchToSpace", libnd4j="batch_to_space", doc="Inverse of spaceToBatch (with cropping).").input("x", D.NUMERIC, "Activations")
   .input("crops", D.NUMERIC, "[2, 2] crops").arg("blockSize", D.INT, "Block edge").output("output", D.NUMERIC, "Rearranged activations"))
