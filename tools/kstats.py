#!/usr/bin/env python
"""Print the top rows of a rocprofv3 *_kernel_stats.csv (short kernel names, per-call average, share)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{nm[:100]:100s} {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:9.1f} us = {float(r['TotalDurationNs']) / 1e6:9.1f} ms "
          f"{100 * float(r['TotalDurationNs']) / tot:5.1f} %")
print(f"total {tot / 1e6:.1f} ms")
