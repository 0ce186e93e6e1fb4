"""Loss namespace (reference definition: codegen/op-codegen/src/main/ops/org/nd4j/codegen/ops/SDLoss.kt; native ops in
libnd4j/include/ops/declarable/generic/loss/*.cpp take (predictions, weights, labels) and the reduction mode as the first integer argument)."""
from ..dsl import DataType as D, Namespace

ns = Namespace("loss", doc="Loss functions. Every weighted loss takes optional broadcastable weights (default 1.0) and a reduction mode.",
               className="Loss")

REDUCE = ["NONE", "SUM", "MEAN_BY_WEIGHT", "MEAN_BY_NONZERO_WEIGHT_COUNT"]


def _weighted(name, native, doc, pred="predictions", pred_doc="Predictions"):
    return (ns.op(name, libnd4j=native, doc=doc)
            .input("label", D.NUMERIC, "Label array").input(pred, D.NUMERIC, pred_doc)
            .input("weights", D.NUMERIC, "Weights, broadcastable to the loss shape", optional=True, default=1.0)
            .arg("lossReduce", D.ENUM, "Reduction of the per-example losses", default="MEAN_BY_NONZERO_WEIGHT_COUNT", possibleValues=REDUCE, slot="i")
            .nativeOrder(pred, "weights", "label"))


_weighted("absoluteDifference", "absolute_difference_loss", "Absolute difference loss: sum_i |label_i - predictions_i|.").output("output", D.NUMERIC, "Loss")
(_weighted("cosineDistance", "cosine_distance_loss", "Cosine distance loss: 1 - sum(label * predictions) along a dimension (inputs are expected normalised).")
 .arg("dimension", D.INT, "Dimension along which the cosine distance is taken", default=-1).output("output", D.NUMERIC, "Loss"))
_weighted("hingeLoss", "hinge_loss", "Hinge loss: max(0, 1 - t * predictions) with t = 2 * label - 1.").output("output", D.NUMERIC, "Loss")
(_weighted("huberLoss", "huber_loss", "Huber loss: quadratic below delta, linear above.")
 .arg("delta", D.FLOATING_POINT, "Switch point between the quadratic and the linear part", default=1.0).output("output", D.NUMERIC, "Loss"))
(_weighted("logLoss", "log_loss", "Binary cross entropy on probabilities.")
 .arg("epsilon", D.FLOATING_POINT, "Added inside the logarithms", default=1e-7).output("output", D.NUMERIC, "Loss"))
(_weighted("logPoisson", "log_poisson_loss", "Log-Poisson loss: exp(predictions) - label * predictions (+ Stirling term when full).")
 .arg("full", D.BOOL, "Add the Stirling approximation term", default=False, slot="i", convert="int").output("output", D.NUMERIC, "Loss"))
_weighted("meanPairwiseSquaredError", "mean_pairwssqerr_loss", "Mean pairwise squared error over all pairs of elements of an example.").output("output", D.NUMERIC, "Loss")
_weighted("meanSquaredError", "mean_sqerr_loss", "Mean squared error.").output("output", D.NUMERIC, "Loss")
(_weighted("sigmoidCrossEntropy", "sigm_cross_entropy_loss", "Sigmoid cross entropy on logits, with optional label smoothing.", "predictionLogits", "Logits")
 .arg("labelSmoothing", D.FLOATING_POINT, "Label smoothing factor", default=0.0).output("output", D.NUMERIC, "Loss"))
(_weighted("softmaxCrossEntropy", "softmax_cross_entropy_loss", "Softmax cross entropy on logits against one-hot labels.", "logitPredictions", "Logits")
 .arg("labelSmoothing", D.FLOATING_POINT, "Label smoothing factor", default=0.0).output("output", D.NUMERIC, "Loss"))
(ns.op("sparseSoftmaxCrossEntropy", libnd4j="sparse_softmax_cross_entropy_loss_with_logits", doc="Softmax cross entropy against integer class labels.")
   .input("logits", D.NUMERIC, "Logits [..., nClasses]").input("labels", D.INT, "Class indices [...]").nativeOrder("labels", "logits")
   .output("output", D.NUMERIC, "Per-example loss"))
(ns.op("l2Loss", libnd4j="l2_loss", doc="Half the sum of squares: sum(x^2) / 2.").input("var", D.NUMERIC, "Input").output("output", D.NUMERIC, "Scalar loss"))
(ns.op("ctcLoss", libnd4j="ctc_loss", doc="Connectionist temporal classification loss.")
   .input("targetLabels", D.INT, "Label sequences [batch, maxLabelLength]").input("logitInput", D.NUMERIC, "Logits [batch, time, nClasses]")
   .input("targetLabelLengths", D.INT, "Label lengths [batch]").input("logitInputLengths", D.INT, "Logit sequence lengths [batch]")
   .arg("blankIndex", D.INT, "Index of the blank class", default=0).output("output", D.NUMERIC, "Per-example loss"))
(ns.op("weightedCrossEntropyWithLogits", libnd4j="weighted_cross_entropy_with_logits", doc="Sigmoid cross entropy with a weight on the positive term.")
   .input("targets", D.NUMERIC, "Targets").input("inputs", D.NUMERIC, "Logits").input("weights", D.NUMERIC, "Positive-class weights")
   .output("output", D.NUMERIC, "Loss"))
