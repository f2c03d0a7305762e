#!/usr/bin/env python3
"""Group the mb_verify_kernel dispatches of a rocprofv3 --kernel-trace CSV by grid size (= launch shape) and print the mean
duration of each group: the rocprofv3 side of bench.py's roofline / roofline_by_shape (HIP-event) numbers.

    python tools/verify_by_grid.py <dir with *kernel_trace.csv>
"""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)
if not files:
    raise SystemExit("no kernel_trace.csv")
groups = defaultdict(list)
for row in csv.DictReader(open(files[0])):
    name = row["Kernel_Name"]
    if not any(k in name for k in ("mb_verify_kernel", "mb_step_kernel", "mb_pack_kernel", "argmax_")):
        continue
    short = name.split("(")[0].replace("void ", "")
    grid = int(row.get("Grid_Size_X", row.get("Grid_Size", 0)) or 0)
    wg = int(row.get("Workgroup_Size_X", row.get("Workgroup_Size", 1)) or 1)
    groups[(short, grid // max(wg, 1))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
print(f"{'kernel':58s} {'workgroups':>10s} {'launches':>8s} {'mean us':>9s} {'min us':>8s} {'max us':>8s}")
for (name, blocks), us in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    if len(us) < 3:
        continue
    print(f"{name[:58]:58s} {blocks:10d} {len(us):8d} {sum(us) / len(us):9.1f} {min(us):8.1f} {max(us):8.1f}")
