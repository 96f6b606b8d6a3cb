#!/usr/bin/env python3
"""Idle time between kernels of a rocprofv3 --kernel-trace CSV: total, and by the kernel BEFORE the gap (development aid).
usage: gpu_gaps.py <kernel_trace.csv> [min_gap_us]"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
busy_end, gaps, busy = ev[0][0], collections.Counter(), 0
cnt = collections.Counter()
last = ""
t0 = ev[0][0]
for s, e, n in ev:
    if s > busy_end:
        g = (s - busy_end) / 1e3
        if g >= min_gap and g < 50000:
            gaps[last[:60] + "  ->  " + n[:50]] += g
            cnt[last[:60] + "  ->  " + n[:50]] += 1
        busy += e - s
    else:
        busy += max(0, e - max(s, busy_end))
    if e > busy_end:
        busy_end, last = e, n
span = ev[-1][1] - t0
print(f"span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({busy / span * 100:.1f} %), gaps >= {min_gap} us: {sum(gaps.values()) / 1e3:.1f} ms")
for k, v in gaps.most_common(25):
    print(f"{v / 1e3:8.2f} ms  {cnt[k]:5d} x {v / cnt[k]:7.1f} us   {k}")
