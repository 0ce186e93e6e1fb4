#!/usr/bin/env python
"""kernel_trace.csv of tools/library_gemm_probe.py -> Markdown table of the vendor GEMM solutions (name fields that matter for
co-scheduling: macro tile, stream-K / global split-U, workgroups launched vs CUs, LDS and scratch per workgroup)."""
import csv
import re
import sys


def main(path, cus=256):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if not n.startswith("Cijk_"):
                continue
            wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
            grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            e = rows.setdefault(n, {"calls": 0, "us": 0.0, "wgs": set(), "wg": wg, "lds": r.get("LDS_Block_Size", "?"),
                                    "scratch": r.get("Scratch_Size", "?"), "vgpr": r.get("VGPR_Count", "?")})
            e["calls"] += 1
            e["us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            e["wgs"].add(grid // max(wg, 1))
    print("| solution (fields) | calls | avg us | workgroups | wg size | LDS B | scratch B | stream-K | split-U | full name |")
    print("|---|---:|---:|---|---:|---:|---:|---|---|---|")
    for n, e in sorted(rows.items(), key=lambda kv: -kv[1]["us"]):
        mt = re.search(r"MT(\d+x\d+x\d+)", n)
        sk = re.search(r"_SK(\d+)", n)
        gsu = re.search(r"_GSU(\d+)", n)
        typ = re.match(r"Cijk_([A-Za-z]+_[A-Za-z]+)_([A-Z]+)_", n)
        print("| %s %s MT%s | %d | %.1f | %s | %d | %s | %s | %s | %s | `%s` |" % (
            typ.group(1) if typ else "?", typ.group(2) if typ else "?", mt.group(1) if mt else "?", e["calls"], e["us"] / e["calls"],
            "/".join(str(w) for w in sorted(e["wgs"])), e["wg"], e["lds"], e["scratch"], sk.group(1) if sk else "-",
            gsu.group(1) if gsu else "-", n))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 256)
