#!/usr/bin/env python3
"""Timeline of one steady-state cycle from a rocprofv3 --kernel-trace CSV: per kernel start offset, duration and the gap to
the previous kernel's end (all streams merged).
python tools/trace_gaps.py <kernel_trace.csv> [cycles_from_end] [min_us,max_us of a cycle] [min_us of the marker kernel]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
    # a cycle starts with the first big down kernel: find the starts of kernels whose name has 'pre_restrict' and a big grid
    mark_min = float(sys.argv[4]) * 1e3 if len(sys.argv) > 4 else 100000.0
    marks = [i for i, e in enumerate(ev) if "pre_restrict" in e[2] and (e[1] - e[0]) > mark_min]
    lo, hi = (float(v) * 1e3 for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("500", "5000")))
    pairs = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if lo < ev[marks[i + 1]][0] - ev[marks[i]][0] < hi]
    a, b = pairs[-back]
    t0 = ev[a][0]
    prev_end = None
    busy = 0
    for s, e, name, q in ev[a:b]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  q={q}  {name[:90]}")
        prev_end = e if prev_end is None else max(prev_end, e)
        busy += e - s
    print(f"cycle span {(ev[b][0] - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")


if __name__ == "__main__":
    main()
