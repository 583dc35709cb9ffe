#!/usr/bin/env python3
"""Idle time in front of every kernel (start - end of the previous kernel on the device) from a rocprofv3 --kernel-trace CSV:
mean gap and mean duration per kernel name, steady-state part of the trace only.  usage: tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[len(rows) // 2:]          # steady state
gap, dur, cnt = collections.Counter(), collections.Counter(), collections.Counter()
prev_end = rows[0][0]
for s, e, n in rows:
    n = n.split("(")[0].replace("(anonymous namespace)::", "")[-60:]
    gap[n] += max(0, s - prev_end); dur[n] += e - s; cnt[n] += 1
    prev_end = max(prev_end, e)
tot = rows[-1][1] - rows[0][0]
print(f"span {tot / 1e6:.2f} ms, kernels {sum(dur.values()) / 1e6:.2f} ms, gaps {sum(gap.values()) / 1e6:.2f} ms")
for n, g in sorted(gap.items(), key=lambda kv: -kv[1])[:14]:
    print(f"{n:62s} n={cnt[n]:5d} gap {g / cnt[n] / 1e3:7.1f} us  dur {dur[n] / cnt[n] / 1e3:7.1f} us")
