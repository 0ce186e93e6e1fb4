import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.models.resnet50 import ResNet50
from deeplearning4j_amd.nd4j.learning import Sgd
from deeplearning4j_amd.ops import batchnorm as BN
m = ResNet50(numClasses=5, inputShape=(3, 160, 160), computeDataType="BFLOAT16", updater=Sgd(0.0))
a = m.init(device="cuda")
x = torch.rand(8, 3, 160, 160, device="cuda")
y = torch.eye(5)[torch.randint(0, 5, (8,))].cuda()
res = {}
for tag, fused in (("fused", True), ("unfused", False), ("fused2", True)):
    BN._FUSE_STAGE2 = fused
    a.computeGradientAndScore([x], [y])
    torch.cuda.synchronize()
    res[tag] = (a._grads.clone(), float(a._score))
    print(tag, "score", res[tag][1], "|g|", float(res[tag][0].norm()), "cnt nonzero", int(BN.ticket_ws(x.device)[0].abs().sum()) if BN._TICKET else -1)
for t in ("unfused", "fused2"):
    print("fused vs", t, float((res["fused"][0] - res[t][0]).norm() / res[t][0].norm()))
g0, g1 = res["fused"][0], res["unfused"][0]
for e in a._entries:
    u, v = g0[e.off:e.off + e.spec.numel], g1[e.off:e.off + e.spec.numel]
    r = float((u - v).norm() / v.norm().clamp_min(1e-30))
    if r > 1e-2:
        print("  ", a._layer_key(e.layer_idx), e.spec.key, "rel", r, float(u.norm()), float(v.norm()))
