"""Image namespace (reference definition: codegen/op-codegen/src/main/ops/org/nd4j/codegen/ops/Image.kt)."""
from ..dsl import DataType as D, Namespace

ns = Namespace("image", doc="Image operations on NHWC tensors.", className="Image")


def _resize(name, lib, what):
    (ns.op(name, libnd4j=lib, doc="Resize images with %s interpolation (TensorFlow semantics)." % what)
       .input("input", D.NUMERIC, "Images [N, H, W, C]").arg("height", D.INT, "Output height").arg("width", D.INT, "Output width")
       .arg("alignCorners", D.BOOL, "Map the corner pixels of input and output onto each other", default=False)
       .arg("halfPixelCenters", D.BOOL, "Sample at pixel centres", default=False).output("output", D.NUMERIC, "Resized images [N, height, width, C]"))


_resize("resizeBiLinear", "resize_bilinear", "bilinear")
_resize("resizeNearestNeighbor", "resize_nearest_neighbor", "nearest-neighbour")
_resize("resizeBiCubic", "resize_bicubic", "bicubic")
_resize("resizeArea", "resize_area", "area")
(ns.op("cropAndResize", libnd4j="crop_and_resize", doc="Crop boxes out of images and resize the crops to a common size.")
   .input("image", D.NUMERIC, "Images [N, H, W, C]").input("cropBoxes", D.NUMERIC, "Boxes [B, 4] as normalised y1, x1, y2, x2")
   .input("boxIndices", D.INT, "Image index of every box [B]").input("cropOutSize", D.INT, "Output size [2]")
   .arg("method", D.ENUM, "Interpolation", default=0, possibleValues=["BILINEAR", "NEAREST"]).arg("extrapolationValue", D.FLOATING_POINT, "Value outside the image", default=0.0)
   .output("output", D.NUMERIC, "Crops [B, cropH, cropW, C]"))
(ns.op("nonMaxSuppression", libnd4j="non_max_suppression", doc="Greedy non-maximum suppression of scored boxes.")
   .input("boxes", D.NUMERIC, "Boxes [N, 4]").input("scores", D.NUMERIC, "Scores [N]").arg("maxOutSize", D.INT, "Maximum number of boxes to keep")
   .arg("iouThreshold", D.FLOATING_POINT, "Overlap above which a box is suppressed", default=0.5)
   .arg("scoreThreshold", D.FLOATING_POINT, "Boxes scoring below are dropped", default=float("-inf")).output("output", D.INT, "Indices of the kept boxes"))
for _n, _lib, _doc in (("rgbToHsv", "rgb_to_hsv", "RGB -> HSV"), ("hsvToRgb", "hsv_to_rgb", "HSV -> RGB"), ("rgbToYuv", "rgb_to_yuv", "RGB -> YUV"),
                       ("yuvToRgb", "yuv_to_rgb", "YUV -> RGB"), ("rgbToYiq", "rgb_to_yiq", "RGB -> YIQ"), ("yiqToRgb", "yiq_to_rgb", "YIQ -> RGB")):
    (ns.op(_n, libnd4j=_lib, doc="Colour-space conversion %s on the last axis." % _doc)
       .input("input", D.NUMERIC, "Images [..., 3]").output("output", D.NUMERIC, "Converted images"))
(ns.op("adjustContrast", libnd4j="adjust_contrast", doc="Scale the deviation of every pixel from its channel mean.")
   .input("in", D.NUMERIC, "Images [..., H, W, C]").arg("factor", D.FLOATING_POINT, "Contrast factor").output("output", D.NUMERIC, "Adjusted images"))
(ns.op("adjustHue", libnd4j="adjust_hue", doc="Rotate the hue channel.")
   .input("in", D.NUMERIC, "RGB images [..., 3]").arg("delta", D.FLOATING_POINT, "Hue shift in [-1, 1]").output("output", D.NUMERIC, "Adjusted images"))
(ns.op("adjustSaturation", libnd4j="adjust_saturation", doc="Scale the saturation channel.")
   .input("in", D.NUMERIC, "RGB images [..., 3]").arg("factor", D.FLOATING_POINT, "Saturation factor").output("output", D.NUMERIC, "Adjusted images"))
(ns.op("extractImagePatches", libnd4j="extract_image_patches", doc="Sliding-window patches flattened into the channel axis.")
   .input("image", D.NUMERIC, "Images [N, H, W, C]").arg("kH", D.INT, "Kernel height").arg("kW", D.INT, "Kernel width").arg("sH", D.INT, "Stride height")
   .arg("sW", D.INT, "Stride width").arg("rH", D.INT, "Dilation height").arg("rW", D.INT, "Dilation width")
   .arg("sameMode", D.BOOL, "SAME padding instead of VALID", default=False, slot="i", convert="int").output("output", D.NUMERIC, "Patches [N, oH, oW, kH*kW*C]"))
(ns.op("imageResize", libnd4j="image_resize", doc="Resize with a selectable method.")
   .input("input", D.NUMERIC, "Images [N, H, W, C]").input("size", D.INT, "Output size [2]")
   .arg("method", D.ENUM, "Interpolation method", default=0, possibleValues=["ResizeBilinear", "ResizeNearest", "ResizeBicubic", "ResizeArea"])
   .arg("preserveAspectRatio", D.BOOL, "Keep the aspect ratio", default=False).arg("antialias", D.BOOL, "Antialias when down-sampling", default=False)
   .output("output", D.NUMERIC, "Resized images"))
