import torch, sys
sys.path.insert(0, ".")
from tests.test_networks_cpu import _onehot
from tests.test_hip_kernels_gpu import _well_conditioned
from deeplearning4j_amd.models.resnet50 import ResNet50
a = ResNet50(numClasses=10, inputShape=(3, 160, 160), weightInit="RELU").init(device="cpu")
a2 = ResNet50(numClasses=10, inputShape=(3, 160, 160), weightInit="RELU").init(device="cpu")
b = ResNet50(numClasses=10, inputShape=(3, 160, 160), weightInit="RELU", computeDataType="BFLOAT16").init(device="cuda")
_well_conditioned(a, b)
p = a.params().clone().view(-1)
# weights rounded to bf16 (what any bf16 kernel sees at best); activations stay fp32
a2.setParams(p.bfloat16().float())
x = torch.rand(8, 3, 160, 160); y = _onehot(8, 10).float()
a.computeGradientAndScore([x], [y]); a2.computeGradientAndScore([x.bfloat16().float()], [y]); b.computeGradientAndScore([x.cuda()], [y.cuda()])
ga, g2, gb = a._grads.double(), a2._grads.double(), b._grads.cpu().double()
rows = []
for e in a._entries:
    if e.spec.key == "W":
        u, v, w = ga[e.off:e.off+e.spec.numel], gb[e.off:e.off+e.spec.numel], g2[e.off:e.off+e.spec.numel]
        c1 = float((u*v).sum()/(u.norm()*v.norm()).clamp_min(1e-30)); c2 = float((u*w).sum()/(u.norm()*w.norm()).clamp_min(1e-30))
        rows.append((str(a._layer_key(e.layer_idx)), c1, c2))
rows.sort(key=lambda r: r[1])
for r in rows[:12]: print("%-28s bf16-gpu %.4f   fp32 with bf16-rounded weights+input %.4f" % r)
print("min floor", min(r[2] for r in rows), "count<0.95", sum(r[1] < 0.95 for r in rows), "of", len(rows))
