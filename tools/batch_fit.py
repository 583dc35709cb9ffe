#!/usr/bin/env python3
"""Fit t(B) = a + b B per kernel from the kernel-stats of bench.py at local batch 1, 2, 4 (tools/batch_fit.sh)."""
import csv, glob, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles"))
from summarize import short

root = sys.argv[1]
data = {}
for B in (1, 2, 4):
    f = glob.glob(os.path.join(root, f"b{B}", "**", "*_kernel_stats.csv"), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n = short(r["Name"])
        data.setdefault(n, {})[B] = (int(r["Calls"]) / 24.0, float(r["AverageNs"]) / 1e3)      # 8 settle + 3 warm-up + 10 timed + 3 roofline_others steps
rows = []
for n, d in data.items():
    if len(d) < 3:
        continue
    # least squares through the three points
    xs, ys = [1, 2, 4], [d[1][1], d[2][1], d[4][1]]
    mx, my = sum(xs) / 3, sum(ys) / 3
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    a = my - b * mx
    rows.append((d[2][0] * d[2][1], n, d[2][0], ys, a, b))
rows.sort(reverse=True)
print(f"{'kernel':58s} {'n/step':>6s} {'B=1':>7s} {'B=2':>7s} {'B=4':>7s} {'a us':>7s} {'b us':>7s} {'fixed ms/step':>13s}")
tot_a = tot = 0.0
for t, n, c, ys, a, b in rows[:40]:
    print(f"{n[:58]:58s} {c:6.1f} {ys[0]:7.1f} {ys[1]:7.1f} {ys[2]:7.1f} {a:7.1f} {b:7.1f} {a * c / 1e3:13.3f}")
for t, n, c, ys, a, b in rows:
    tot_a += a * c / 1e3
    tot += t / 1e3
print(f"sum over all kernels at B=2: {tot:.2f} ms/step, batch-independent part {tot_a:.2f} ms/step")
