"""BERT-as-SameDiff builder (models/bert.py): tiny configuration trains on the CPU; the attention op runs in its single-output mode."""
import torch

from deeplearning4j_amd.autodiff.samediff import SameDiff, TrainingConfig
from deeplearning4j_amd.models.bert import BertConfig, build_bert_classifier
from deeplearning4j_amd.nd4j.learning import Adam


def _data(n, T, vocab, seed=0):
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(5, vocab, (n, T), generator=g)
    cls = torch.randint(0, 2, (n,), generator=g)
    idx[:, 3] = cls + 1                                   # the class is readable from one token
    lens = torch.randint(T // 2, T + 1, (n,), generator=g)
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()
    return idx, mask, torch.eye(2)[cls]


def test_tiny_bert_trains_and_single_output_attention_matches_weights_mode():
    torch.manual_seed(0)
    T = 16
    sd = SameDiff.create("cpu")
    build_bert_classifier(sd, BertConfig.tiny(vocab=50, hidden=32, layers=2, heads=2, intermediate=64), T, 2)
    sd.setTrainingConfig(TrainingConfig.builder().updater(Adam(3e-3)).build())
    idx, mask, lab = _data(64, T, 50)
    ph = {"tokenIdx": idx, "mask": mask, "label": lab}
    l0 = float(sd.output(ph, ["loss"])["loss"])
    for _ in range(60):
        sd.fit(ph)
    l1 = float(sd.output(ph, ["loss"])["loss"])
    assert l1 < 0.5 * l0, (l0, l1)
    probs = sd.output(ph, ["probs"])["probs"]
    assert float((probs.argmax(1) == lab.argmax(1)).float().mean()) > 0.9

    # the single-output attention equals the first output of the weights-returning mode (same op, key mask only)
    sd2 = SameDiff.create("cpu")
    q = sd2.placeHolder("q", "FLOAT", 2, 2, 5, 4)
    k = sd2.placeHolder("k", "FLOAT", 2, 2, 7, 4)
    v = sd2.placeHolder("v", "FLOAT", 2, 2, 7, 4)
    m = sd2.placeHolder("m", "FLOAT", 2, 7)
    one = sd2.nn.dotProductAttentionV2(q, v, k, None, m, scale=0.5, withWeights=False, name="one")
    qm = sd2.constant("qm", torch.ones(2, 5))
    three = sd2.nn.dotProductAttentionV2(q, v, k, qm, m, scale=0.5, withWeights=True)
    feed = {"q": torch.randn(2, 2, 5, 4), "k": torch.randn(2, 2, 7, 4), "v": torch.randn(2, 2, 7, 4),
            "m": torch.tensor([[1, 1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1.0]])}
    out = sd2.output(feed, ["one", three[0].name])
    assert torch.allclose(out["one"], out[three[0].name], atol=1e-5)
