#!/usr/bin/env python3
"""Steady-state per-step kernel statistics of a train loop from TWO rocprofv3 kernel_stats.csv taken at different
step counts: everything that is not a step (model construction, optimiser state fills, the capture's warm-up steps and
their state snapshot / restore copies) is the same in both and cancels in the difference.

    kstats_diff.py <stats at n1 steps> <stats at n2 steps> <n1> <n2> [<out.csv>]
"""
import csv
import sys

a = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
b = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
n1, n2 = float(sys.argv[3]), float(sys.argv[4])
rows = []
for name, rb in b.items():
    ra = a.get(name)
    c = (int(rb["Calls"]) - (int(ra["Calls"]) if ra else 0)) / (n2 - n1)
    t = (float(rb["TotalDurationNs"]) - (float(ra["TotalDurationNs"]) if ra else 0.0)) / (n2 - n1)
    if c > 1e-9:
        rows.append((name, c, t))
rows.sort(key=lambda r: -r[2])
tot_c, tot_t = sum(r[1] for r in rows), sum(r[2] for r in rows)
print(f"{tot_t / 1e6:.3f} ms of kernels per step, {tot_c:.1f} launches per step (steady state: {int(n2)} - {int(n1)} steps)")
for name, c, t in rows[:40]:
    print(f"{name[:72]:72s} {c:8.1f} {t / 1e6:8.3f} ms {t / c / 1e3:8.1f} us {t / tot_t * 100:5.1f}%")
if len(sys.argv) > 5:
    with open(sys.argv[5], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "CallsPerStep", "NsPerStep", "AverageNs", "Percentage"])
        for name, c, t in rows:
            w.writerow([name, f"{c:.3f}", f"{t:.1f}", f"{t / c:.1f}", f"{t / tot_t * 100:.3f}"])
