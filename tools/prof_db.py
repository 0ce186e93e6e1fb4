#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (<name>_results.db, the default output of ROCm 7.2) into a markdown kernel table.
usage: prof_db.py results.db [steps_to_divide_by] [last_fraction]  — last_fraction (0..1] keeps only the tail of the timeline
(steady-state replays) before aggregating."""
import collections
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short  # noqa: E402


def main(path, steps=1, frac=1.0, top=40):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, start, end from kernels order by start"))
    if frac < 1.0 and rows:
        t0, t1 = rows[0][1], rows[-1][2]
        cut = t1 - (t1 - t0) * frac
        rows = [r for r in rows if r[1] >= cut]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e in rows:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print("| kernel | calls/step | ms/step | % | avg us |")
    print("|---|---:|---:|---:|---:|")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("| %s | %.1f | %.3f | %.1f | %.1f |" % (n, c / steps, t / 1e6 / steps, 100 * t / tot, t / c / 1e3))
    print("\nsum of kernel time: %.3f ms/step, %d launches/step" % (tot / 1e6 / steps, sum(v[0] for v in agg.values()) / steps))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1, float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
