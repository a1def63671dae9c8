#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per (kernel, grid size): launches, mean counter value per launch.
The grid size separates the launches of one kernel on different levels (level 0 is the largest grid).
python tools/pmc_summary.py <counter_collection.csv> [out.csv]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for r in rows:
        k = r.get("Kernel_Name") or r.get("Kernel Name")
        g = r.get("Grid_Size") or r.get("Grid Size") or ""
        k = f"{k[:150]} [grid {g}]"
        c = r.get("Counter_Name") or r.get("Counter Name")
        v = float(r.get("Counter_Value") or r.get("Counter Value") or 0)
        a = acc[k][c]
        a[0] += 1
        a[1] += v
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "counter", "launches", "mean_per_launch", "total"])
    for k, cs in sorted(acc.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
        for c, (n, t) in cs.items():
            w.writerow([k[:200], c, n, t / n, t])


if __name__ == "__main__":
    main()
