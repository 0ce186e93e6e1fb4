#!/usr/bin/env python
"""HBM traffic per training step from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need 3 + 2 of the 4 TCC
counter slots, so each gets a run of its own): sums the counter per kernel, divides by the number of steps (= launches of
segmented_updater_kernel) and, given the per-step kernel times of a plain --kernel-trace run (tools/prof_steps.py table),
prints the achieved bandwidth of every kernel.

usage: pmc_hbm.py <fetch counter_collection.csv> <write counter_collection.csv> [steps table .md]"""
import csv
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short


def load(path, counter):
    per, steps = {}, 0
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
            did = r.get("Dispatch_Id")
            if "segmented_updater_kernel" in k and did not in seen:
                seen.add(did)
                steps += 1
    return per, max(steps, 1)


def main(fetch_csv, write_csv, table=None):
    rd, s1 = load(fetch_csv, "FETCH_SIZE")
    wr, s2 = load(write_csv, "WRITE_SIZE")
    ms = {}
    if table:
        for line in open(table):
            m = re.match(r"\| (\S.*?) \| [\d.]+ \| ([\d.]+) \|", line)
            if m:
                ms[m.group(1)] = float(m.group(2))
    names = sorted(set(rd) | set(wr), key=lambda k: -(rd.get(k, 0) / s1 + wr.get(k, 0) / s2))
    tot_r = sum(rd.values()) / s1 * 1024
    tot_w = sum(wr.values()) / s2 * 1024
    print("steps counted: %d (fetch run), %d (write run); FETCH_SIZE / WRITE_SIZE are in KiB\n" % (s1, s2))
    print("HBM traffic per training step: read %.2f GB, written %.2f GB, total %.2f GB" % (tot_r / 1e9, tot_w / 1e9, (tot_r + tot_w) / 1e9))
    if ms:
        t = sum(ms.values())
        print("sum of kernel time %.2f ms/step -> average %.2f TB/s over the whole step\n" % (t, (tot_r + tot_w) / 1e9 / t))
    print("| kernel | read MB/step | written MB/step | total MB/step | ms/step | achieved TB/s |")
    print("|---|---:|---:|---:|---:|---:|")
    for k in names[:30]:
        r, w = rd.get(k, 0) / s1 * 1024 / 1e6, wr.get(k, 0) / s2 * 1024 / 1e6
        t = ms.get(k)
        print("| %s | %.0f | %.0f | %.0f | %s | %s |" % (k, r, w, r + w, "%.3f" % t if t else "", "%.2f" % ((r + w) / 1e3 / t) if t else ""))


if __name__ == "__main__":
    main(*sys.argv[1:4])
