import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.models.resnet50 import ResNet50
b = ResNet50(numClasses=5, inputShape=(3, 160, 160), weightInit="RELU", computeDataType="HALF").init(device="cuda")
x = torch.rand(8, 3, 160, 160).cuda()
acts = b.feedForward(x, train=True)
for n in b.topologicalOrder:
    a = acts.get(n)
    if a is None:
        continue
    af = a.float()
    bad = not torch.isfinite(af).all()
    print("%-28s %-12s max %.4g %s" % (n, str(a.dtype).replace("torch.", ""), float(af.abs().max()), "  <-- NON-FINITE" if bad else ""))
    if bad:
        break
