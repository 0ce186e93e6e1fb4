"""deeplearning4j_amd — an MI355X-native (gfx950 / CDNA4) deep-learning framework with the
capabilities of Eclipse Deeplearning4j: ND4J-style tensors and op families, SameDiff autodiff,
MultiLayerNetwork / ComputationGraph, zoo models, DataVec ETL, ParallelWrapper / ParallelInference
data parallelism over RCCL/xGMI — with the hot path in hand-written HIP kernels.
"""
import os as _os

# MIOpen NHWC kernels for channels-last activations (library conv path)
_os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
_os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM", "1")

__version__ = "0.1.0"
