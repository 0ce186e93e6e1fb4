#!/usr/bin/env python
"""Per-kernel shader-engine counters of a training step from one `rocprofv3 --pmc` run (8 SQ counters fit one pass):
SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 (optional) SQ_ACTIVE_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY.  Prints, per kernel, the share of wave cycles spent waiting on any instruction, the matrix-core
busy share of the busy cycles, and LDS bank-conflict cycles relative to LDS-active cycles.

With the per-step kernel-time table of a plain --kernel-trace run (tools/prof_steps.py) the matrix-core busy cycles are also put
against the cycles the chip offers in the kernel's time: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 2.4 GHz x ms/step).

usage: pmc_sq.py <counter_collection.csv> [steps table .md] [top]"""
import csv
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short


def main(path, table=None, top=16):
    ms = {}
    if table and table != "":
        for line in open(table):
            m = re.match(r"\| (\S.*?) \| [\d.]+ \| ([\d.]+) \|", line)
            if m:
                ms[m.group(1)] = float(m.group(2))
    per = {}
    calls = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            d = per.setdefault(k, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            calls.setdefault(k, set()).add(r.get("Dispatch_Id"))
    names = sorted(per, key=lambda k: -per[k].get("SQ_BUSY_CYCLES", 0.0))
    tot = sum(per[k].get("SQ_BUSY_CYCLES", 0.0) for k in names) or 1.0
    steps = max(1, len(calls.get("segmented_updater_kernel", [1])))
    print("steps counted: %d; SIMD clock taken as 2.4 GHz\n" % steps)
    print("| kernel | launches/step | share of SQ busy cycles | MFMA busy / offered SIMD cycles | waiting / wave cycles | LDS conflict / LDS active | VALU insts per wave quad-cycle |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k in names[:top]:
        c = per[k]
        busy, wave = c.get("SQ_BUSY_CYCLES", 0.0) or 1.0, c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        lds = c.get("SQ_ACTIVE_INST_LDS", 0.0)
        offered = ms[k] * 1e-3 * 2.4e9 * 1024 if k in ms else 0.0
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / steps
        print("| %s | %.1f | %.1f %% | %s | %.1f %% | %s | %.3f |" % (
            k, len(calls[k]) / steps, 100 * busy / tot, ("%.1f %%" % (100 * mf / offered)) if (offered and mf) else "-",
            100 * c.get("SQ_WAIT_INST_ANY", 0.0) / wave, ("%.1f %%" % (100 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds)) if lds else "-",
            c.get("SQ_INSTS_VALU", 0.0) / wave))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else 16)
