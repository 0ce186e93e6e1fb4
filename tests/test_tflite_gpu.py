"""TFLite runner on the MI355X: same model as the CPU test, executed on cuda tensors (channels_last convolutions)."""
import pytest
import torch

from test_tflite_cpu import _cnn
from deeplearning4j_amd.tflite import TensorFlowLiteRunner

pytestmark = pytest.mark.gpu


def test_tflite_cnn_on_gpu_matches_cpu_reference():
    buf, ref = _cnn()
    r = TensorFlowLiteRunner(buf, device="cuda")
    x = torch.randn(8, 16, 16, 3, generator=torch.Generator().manual_seed(3))
    out = r.exec({"input": x.cuda()})["probs"]
    assert out.is_cuda and out.shape == (8, 10)
    assert torch.allclose(out.cpu(), ref(x), atol=2e-4)
