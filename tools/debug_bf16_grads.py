"""Is the bf16-vs-fp32 ResNet-50 gradient gap kernel error or conditioning of the random-init network? Compares
(1) fp32 vs bf16, (2) fp32 vs fp32 with parameters perturbed by bf16-sized relative noise, (3) both again with the last BatchNorm gamma of every
residual branch scaled to 0.2 (a well-conditioned start)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.models.resnet50 import ResNet50


def cos(u, v):
    u, v = u.double(), v.double()
    return float((u * v).sum() / (u.norm() * v.norm()))


def build(dt=None):
    kw = dict(numClasses=10, inputShape=(3, 160, 160), weightInit="RELU")
    if dt:
        kw["computeDataType"] = dt
    return ResNet50(**kw).init(device="cuda")


torch.manual_seed(0)
x = torch.rand(8, 3, 160, 160).cuda()
y = torch.eye(10)[torch.randint(0, 10, (8,))].cuda()
for gamma_c in (1.0, 0.2, 0.0):
    a, b, c = build(), build("BFLOAT16"), build()
    p = a.params().clone().view(-1)
    if gamma_c != 1.0:
        for e in a._entries:
            name = a._layer_key(e.layer_idx) if hasattr(a, "_layer_key") else str(e.layer_idx)
            if e.spec.key == "gamma" and str(name).endswith("branch2c"):
                p[e.off:e.off + e.spec.numel] = gamma_c
            if str(name) == "output" and e.spec.key == "W" and gamma_c != 1.0:
                p[e.off:e.off + e.spec.numel] *= 0.05
    a.setParams(p); b.setParams(p)
    noise = (torch.rand_like(p) - 0.5) * 2 ** -8          # uniform in +-2^-9 relative: the rounding of one bf16 store
    c.setParams(p * (1 + noise))
    for n in (a, b, c):
        n.computeGradientAndScore([x], [y])
    print("gamma2c=%.1f scores fp32 %.5f bf16 %.5f fp32-perturbed %.5f" % (gamma_c, float(a._score), float(b._score), float(c._score)))
    print("   cos(fp32, bf16) %.4f   cos(fp32, fp32 perturbed) %.4f   |g| %.4g %.4g %.4g" %
          (cos(a._grads, b._grads), cos(a._grads, c._grads), float(a._grads.norm()), float(b._grads.norm()), float(c._grads.norm())))
    first = [e for e in a._entries if e.spec.key == "W"][:3] + [e for e in a._entries if e.spec.key == "W"][-3:]
    for e in first:
        s, n = e.off, e.spec.numel
        print("      %s cos bf16 %.4f perturbed %.4f" % (a._layer_key(e.layer_idx) if hasattr(a, "_layer_key") else e.layer_idx,
              cos(a._grads[s:s + n], b._grads[s:s + n]), cos(a._grads[s:s + n], c._grads[s:s + n])))
