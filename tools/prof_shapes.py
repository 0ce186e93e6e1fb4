#!/usr/bin/env python
"""Per-launch-shape breakdown from a rocprofv3 kernel_trace.csv: groups the steady-state steps' dispatches by
(kernel, grid, workgroup) so that inefficient layer shapes stand out.  Usage: prof_shapes.py trace.csv [nsteps] [top]"""
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short


def main(path, nsteps=3, top=60):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            g = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            w = r.get("Workgroup_Size_X", "?")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], g, w, r.get("LDS_Block_Size", "?"),
                         r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"), r.get("Scratch_Size", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "segmented_updater_kernel" in r[2]]
    if len(marks) < nsteps + 1:
        print("not enough steps", len(marks))
        return
    sel = rows[marks[-nsteps - 1] + 1:marks[-1] + 1]
    agg = {}
    for s, e, n, g, w, lds, vg, ag, sc in sel:
        a = agg.setdefault((short(n), g, w, lds, vg, ag, sc), [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print("| kernel | grid | wg | lds | vgpr | agpr | scratch | calls/step | ms/step | % | avg us |")
    print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for (n, g, w, lds, vg, ag, sc), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("| %s | %s | %s | %s | %s | %s | %s | %.1f | %.3f | %.1f | %.1f |" % (n[:48], g, w, lds, vg, ag, sc, c / nsteps, t / 1e6 / nsteps,
                                                                                    100 * t / tot, t / c / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 60)
