#!/usr/bin/env python
"""Summarise a rocprofv3 kernel_stats.csv (kernel names shortened) into a markdown table."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+)", name)
    base = m.group(1) if m else name
    if "kernel_grouped_conv" in name or "ck::" in name or name.startswith("_ZN2ck"):
        k = re.search(r"kernel_grouped_conv_(\w+?)_(?:multiple|xdl)", name)
        kind = k.group(1) if k else "ck"
        t = re.search(r"PassThrough, 1, (\d+), (\d+), (\d+), (\d+)", name)
        tile = ("_%sx%sx%s" % (t.group(2), t.group(3), t.group(4))) if t else ""
        return "ck_conv_%s%s" % (kind, tile)
    if base.startswith("at::native"):
        f = re.search(r"(\w+Functor\w*|direct_copy\w*|\w+_kernel\w*)", name)
        return "aten:" + (f.group(1) if f else base)[:60]
    return (base if len(base) > 8 else name)[:90]


def main(path, top=40):
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        n = short(r["Name"])
        a = agg.setdefault(n, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print("| kernel | calls | total ms | % | avg us |")
    print("|---|---:|---:|---:|---:|")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("| %s | %d | %.2f | %.1f | %.1f |" % (n, c, t / 1e6, 100 * t / tot, t / c / 1e3))
    print("\ntotal GPU kernel time: %.1f ms" % (tot / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
