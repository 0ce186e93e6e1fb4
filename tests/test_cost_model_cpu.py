"""Measured cost model of the platform helpers (nd4j/ops/cost_model.py): which arguments are declined and that the decision is
recorded instead of silent."""
import torch

from deeplearning4j_amd.nd4j.ops import cost_model as CM


def test_rules_decline_the_measured_slow_shapes_only():
    img = torch.empty(64, 64, 56, 56)
    wide = torch.empty(64, 4096)
    pads = torch.tensor([[0, 0], [0, 0], [1, 1], [1, 1]])
    assert "innermost row of 56" in CM.prefer_generic("pad", [img, pads], [0], [0.0], [], [])
    assert CM.prefer_generic("pad", [wide, torch.tensor([[0, 0], [1, 1]])], [0], [0.0], [], []) is None
    assert CM.prefer_generic("reverse", [img], [3], [], [], []) is not None and CM.prefer_generic("reverse", [wide], [1], [], [], []) is None
    a, b = torch.empty(128, 64), torch.empty(64, 256)
    assert "plain 2-D GEMM 128x256x64" in CM.prefer_generic("matmul", [a, b], [], [], [], [])
    assert "8192x4096x4096" in CM.prefer_generic("matmul", [torch.empty(8192, 4096, device="meta"), torch.empty(4096, 4096, device="meta")], [0, 1], [], [], [])
    level = [torch.empty(4096, 768, device="meta"), torch.empty(768, 768, device="meta")]
    assert CM.prefer_generic("matmul", level, [], [], [], []) is None and CM.prefer_generic("matmul", level, [0, 1], [], [], []) is None
    assert CM.prefer_generic("matmul", [torch.empty(6, 40, 64), torch.empty(6, 24, 64)], [0, 1], [], [], []) is None      # batched: in-tree
    assert CM.prefer_generic("xw_plus_b", [a, b, torch.empty(256)], [], [], [], []) is not None
    assert CM.prefer_generic("biasadd", [img, torch.empty(64)], [], [], [True], []) is not None
    assert CM.prefer_generic("biasadd", [wide, torch.empty(4096)], [], [], [], []) is None
    for always in ("tile", "concat", "space_to_depth", "im2col", "depthwise_conv2d"):
        assert CM.prefer_generic(always, [img], [2, 2], [], [], []) is not None
    assert CM.prefer_generic("softmax", [wide], [1], [], [], []) is None          # no rule: the helper runs
    assert set(CM.rules()) >= {"pad", "reverse", "tile", "concat", "matmul", "xw_plus_b", "space_to_depth"}


def test_switch_and_malformed_arguments():
    img = torch.empty(2, 2, 8, 8)
    CM.ENABLED = False
    try:
        assert CM.prefer_generic("tile", [img], [2, 2], [], [], []) is None
    finally:
        CM.ENABLED = True
    assert CM.prefer_generic("pad", [], None, None, None, None) is None           # an estimate never raises
