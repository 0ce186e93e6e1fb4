"""Trace BatchNorm backward calls (inputs / outputs) of the fp32 CPU and fp32 GPU ResNet-50 and compare call by call."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.models.resnet50 import ResNet50
from deeplearning4j_amd.ops import batchnorm as BN
from tests.test_hip_kernels_gpu import _well_conditioned

log = {"cpu": [], "cuda": []}
real = BN.backward


def spy(x2d, dy2d, out2d, mean, invstd, gamma, dgamma_out, dbeta_out, want_dz=False, mask_coeffs=None):
    dx, dz = real(x2d, dy2d, out2d, mean, invstd, gamma, dgamma_out, dbeta_out, want_dz, mask_coeffs)
    log[x2d.device.type].append(dict(shape=tuple(x2d.shape), x=x2d.detach().cpu().clone(), dy=dy2d.detach().cpu().clone(), dx=dx.detach().cpu().clone(),
                                     dbeta=None if dbeta_out is None else dbeta_out.detach().cpu().clone().reshape(-1),
                                     dgamma=None if dgamma_out is None else dgamma_out.detach().cpu().clone().reshape(-1),
                                     has_out=out2d is not None, mc=mask_coeffs is not None, mcv=None if mask_coeffs is None else [t.detach().cpu().clone() for t in mask_coeffs], mean=mean.detach().cpu().clone(), invstd=invstd.detach().cpu().clone(),
                                     out=None if out2d is None else out2d.detach().cpu().clone()))
    return dx, dz


BN.backward = spy
import deeplearning4j_amd.nn.layers.core as CORE
CORE.BN.backward = spy
torch.manual_seed(0)
a = ResNet50(numClasses=5, inputShape=(3, 160, 160), weightInit="RELU").init(device="cpu")
b = ResNet50(numClasses=5, inputShape=(3, 160, 160), weightInit="RELU").init(device="cuda")
_well_conditioned(a, b)
x = torch.rand(8, 3, 160, 160)
y = torch.eye(5)[torch.randint(0, 5, (8,))]
a.computeGradientAndScore([x], [y]); b.computeGradientAndScore([x.cuda()], [y.cuda()])
print(len(log["cpu"]), len(log["cuda"]))


def rel(u, v):
    return float((u.double() - v.double()).norm() / u.double().norm().clamp_min(1e-300))


for i, (c, g) in enumerate(zip(log["cpu"], log["cuda"])):
    line = "%2d %-12s out=%d/%d mc=%d/%d  x %.1e dy %.1e dx %.1e dbeta %.1e dgamma %.1e" % (
        i, c["shape"], c["has_out"], g["has_out"], c["mc"], g["mc"], rel(c["x"], g["x"]), rel(c["dy"], g["dy"]), rel(c["dx"], g["dx"]),
        rel(c["dbeta"], g["dbeta"]), rel(c["dgamma"], g["dgamma"]))
    if c["out"] is not None and g["out"] is not None:
        line += " maskdiff %d" % int(((c["out"] > 0) != (g["out"] > 0)).sum())
    print(line)
    if rel(c["dy"], g["dy"]) > 1e-4 and i < 20:
        d = (g["dy"].double() - c["dy"].double())
        print("     dy err: per-channel mean of err / per-channel rms of dy (first 6):", (d.mean(0) / c["dy"].double().pow(2).mean(0).sqrt())[:6].tolist())
        rows = d.abs().sum(1)
        print("     rows with err > 1e-3 of max:", int((rows > 1e-3 * rows.max()).sum()), "of", rows.numel(), " first rows:",
              torch.nonzero(rows > 1e-3 * rows.max()).reshape(-1)[:12].tolist())

for i in (11, 12):
    c, g = log["cpu"][i], log["cuda"][i]
    print("call", i, "mask coeff rel diff scale %.2e shift %.2e mean %.2e invstd %.2e" % (rel(c["mcv"][0], g["mcv"][0]), rel(c["mcv"][1], g["mcv"][1]),
          rel(c["mean"], g["mean"]), rel(c["invstd"], g["invstd"])))
    for tag, d in (("cpu", c), ("gpu", g)):
        m = (d["x"] * d["mcv"][0] + d["mcv"][1]) > 0
        m2 = torch.addcmul(d["mcv"][1].expand_as(d["x"]), d["x"], d["mcv"][0].expand_as(d["x"])) > 0
        ref = (d["dy"] * m).sum(0)
        print("  ", tag, "dbeta vs host recompute from its own inputs: %.2e ; mask density %.4f ; fma-vs-mul mask diffs %d" % (rel(ref, d["dbeta"]), float(m.float().mean()), int((m != m2).sum())))
    mc_ = (c["x"] * c["mcv"][0] + c["mcv"][1]) > 0
    mg_ = (g["x"] * g["mcv"][0] + g["mcv"][1]) > 0
    print("   cpu-vs-gpu mask diffs:", int((mc_ != mg_).sum()), "of", mc_.numel())
    diff = torch.nonzero(mc_ != mg_)
    for r, ch in diff[:8].tolist():
        print("     row %d ch %d  x cpu %.9g gpu %.9g  y cpu %.3e gpu %.3e  scale %.6g shift %.6g" % (r, ch, float(c["x"][r, ch]), float(g["x"][r, ch]),
              float(c["x"][r, ch] * c["mcv"][0][ch] + c["mcv"][1][ch]), float(g["x"][r, ch] * g["mcv"][0][ch] + g["mcv"][1][ch]), float(c["mcv"][0][ch]), float(c["mcv"][1][ch])))
