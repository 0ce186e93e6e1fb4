#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs of the GEMM lab: mean counter value per kernel dispatch of gemm_nt_kernel."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "gemm_nt_kernel" not in k:
                continue
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(d, k)
        for c, v in sorted(cs.items()):
            print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
