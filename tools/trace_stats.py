#!/usr/bin/env python3
"""Per-kernel totals of the LAST `n` launches pattern from a rocprofv3 kernel trace: prints name, calls, avg us, total share.
python tools/trace_stats.py <kernel_trace.csv> [skip_fraction]  (skips the first fraction of the trace: setup / warm-up)"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[int(len(rows) * skip):]
    acc = defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in acc.values())
    for name, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / tot * 100:6.2f} %  calls {c:6d}  avg {t / c:9.1f} us  {name[:110]}")


if __name__ == "__main__":
    main()
