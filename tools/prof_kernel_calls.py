#!/usr/bin/env python
"""List every call of one kernel inside the last steady-state step of a rocprofv3 kernel_trace.csv (duration, grid, workgroup)."""
import csv
import sys


def main(path, needle):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append(r)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "segmented_updater_kernel" in r["Kernel_Name"]]
    sel = rows[marks[-2] + 1:marks[-1] + 1]
    print("| # | us | grid | wg | gap before us |")
    print("|---:|---:|---|---|---:|")
    prev_end = None
    for k, r in enumerate(sel):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if needle in r["Kernel_Name"]:
            grid = "x".join(r.get(c, "?") for c in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            wg = "x".join(r.get(c, "?") for c in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
            print("| %d | %.1f | %s | %s | %.1f |" % (k, (e - s) / 1e3, grid, wg, (s - prev_end) / 1e3 if prev_end else 0.0))
        prev_end = e


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
