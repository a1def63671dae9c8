#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 --kernel-trace CSV restricted to the launches AFTER amgx_create: the collapsed coarse
levels are formed by running the sub-cycle on every unit vector (thousands of tiny launches of the same kernels the cycle
uses), which would otherwise dominate the call counts and averages of `--stats`.  Same columns as rocprofv3's kernel_stats.csv.
    python tools/stats_after_setup.py <kernel_trace.csv> <out.csv>"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = -1
    for i, r in enumerate(rows):
        if "dense_unit_kernel" in r["Kernel_Name"] or "dense_transpose_kernel" in r["Kernel_Name"] or "gj_" in r["Kernel_Name"]:
            last = i
    rows = rows[last + 1:]
    acc = defaultdict(list)
    for r in rows:
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in acc.values()) or 1
    w = csv.writer(open(sys.argv[2], "w"))
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, len(v), sum(v), sum(v) / len(v), round(100.0 * sum(v) / tot, 2), min(v), max(v)])


if __name__ == "__main__":
    main()
