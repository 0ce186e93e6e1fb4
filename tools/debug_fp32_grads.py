"""fp32 GPU network vs fp32 CPU network, per parameter tensor (well-conditioned start), to localise any fp32 kernel that is not fp32-exact.
DL4J_AMD_CONV=library runs the same comparison on the vendor kernels."""
import os, sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from deeplearning4j_amd.models.resnet50 import ResNet50
from tests.test_hip_kernels_gpu import _well_conditioned


def cos(u, v):
    u, v = u.double(), v.double()
    return float((u * v).sum() / (u.norm() * v.norm()).clamp_min(1e-300))


torch.manual_seed(0)
a = ResNet50(numClasses=5, inputShape=(3, 160, 160), weightInit="RELU").init(device="cpu")
b = ResNet50(numClasses=5, inputShape=(3, 160, 160), weightInit="RELU").init(device="cuda")
_well_conditioned(a, b)
x = torch.rand(8, 3, 160, 160)
y = torch.eye(5)[torch.randint(0, 5, (8,))]
a.computeGradientAndScore([x], [y]); b.computeGradientAndScore([x.cuda()], [y.cuda()])
print("conv=%s scores %.7f %.7f total cos %.7f" % (os.environ.get("DL4J_AMD_CONV", "auto"), float(a._score), float(b._score),
                                                  cos(a._grads, b._grads.cpu())))
gb = b._grads.cpu()
for e in a._entries:
    s, n = e.off, e.spec.numel
    u, v = a._grads[s:s + n], gb[s:s + n]
    if float(u.norm()) == 0:
        continue
    print("  %-18s %-6s cos %.6f  relerr %.2e  |g| %.4g" % (a._layer_key(e.layer_idx), e.spec.key, cos(u, v),
                                                          float((u - v).norm() / u.norm()), float(u.norm())))
