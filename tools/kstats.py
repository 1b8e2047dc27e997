#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats.csv (and per-step figures when --steps is given)."""
import csv
import sys

path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"{tot / 1e6 / steps:.3f} ms per step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches per step")
for r in rows[:40]:
    print(f"{r['Name'][:64]:64s} {int(r['Calls']) / steps:8.1f} {int(r['TotalDurationNs']) / 1e6 / steps:9.3f}ms "
          f"{float(r['AverageNs']) / 1e3:9.1f}us {float(r['Percentage']):6.2f}%")
