#!/usr/bin/env python3
"""Average / min / max duration of the kernels whose name contains argv[2], from a rocprofv3 --kernel-trace --stats directory."""
import csv
import glob
import sys

f = glob.glob(f"{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)
if not f:
    sys.exit(f"no kernel_stats.csv under {sys.argv[1]}")
for r in csv.DictReader(open(f[0])):
    if sys.argv[2] in r["Name"]:
        print(f"{r['Name'].split('(')[0][:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}  max {float(r['MaxNs']) / 1e3:8.1f}")
