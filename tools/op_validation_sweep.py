"""Run the op-validation sweep (autodiff/validation.py + autodiff/op_cases.py) and write the coverage report.
usage: python tools/op_validation_sweep.py [cpu|cuda] [out.md]"""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.autodiff.validation import sweep, write_report

dev = sys.argv[1] if len(sys.argv) > 1 else "cpu"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r2_op_validation_%s.md" % dev
rep = sweep(dev)
s = write_report(rep, out, "Op validation sweep (%s)" % dev)
if dev == "cuda":
    from deeplearning4j_amd.nd4j.ops.custom import OpExecutioner
    st = OpExecutioner.helperStatistics()
    with open(out, "a") as f:
        f.write("\n## Platform-helper accounting during the sweep\n\nhelper ran: %s\n\nhelper not used: %s\n\nlegacy ops on ATen: %s\n" % (
            dict(sorted(st["hit"].items())), {"%s (%s)" % k: v for k, v in sorted(st["fail"].items())}, st["legacy"]))
print(s)
