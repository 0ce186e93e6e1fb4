#!/usr/bin/env python
"""Per-step kernel breakdown from a rocprofv3 kernel_trace.csv: steps are delimited by the fused
updater launch (segmented_updater_kernel); aggregates the last N steady-state steps."""
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short


def main(path, nsteps=3, top=40):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "segmented_updater_kernel" in r[2]]
    if len(marks) < nsteps + 1:
        print("not enough steps", len(marks))
        return
    lo, hi = marks[-nsteps - 1] + 1, marks[-1] + 1
    sel = rows[lo:hi]
    wall = (sel[-1][1] - sel[0][0]) / 1e6 / nsteps
    agg = {}
    for s, e, n in sel:
        k = short(n)
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print("steady-state: %d steps, wall %.2f ms/step, sum of kernel time %.2f ms/step, %d launches/step\n" % (
        nsteps, wall, tot / 1e6 / nsteps, len(sel) // nsteps))
    print("| kernel | calls/step | ms/step | % | avg us |")
    print("|---|---:|---:|---:|---:|")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("| %s | %.1f | %.3f | %.1f | %.1f |" % (n, c / nsteps, t / 1e6 / nsteps, 100 * t / tot, t / c / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 40)
