#!/usr/bin/env python3
"""Per-dispatch rocprofv3 durations of the convergence launch joined to the bytes each launch read (bench.py's JF_DUMP_LAUNCHES):
the rocprofv3 side of `roofline` for ONE window — the headline window alone, no scripted run blended in.

    JF_DUMP_LAUNCHES=launches.json rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --steps K --warmup W \\
        --no-scripted --no-prewarm --no-shapes --no-sections --cpu-baseline-seconds 0
    python tools/verify_per_dispatch.py <dir> launches.json [steps=20]
"""
import csv
import glob
import json
import sys

prof, dump = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
files = glob.glob(f"{prof}/**/*kernel_trace.csv", recursive=True)
if not files:
    raise SystemExit("no kernel_trace.csv")
rows = [r for r in csv.DictReader(open(files[0])) if "mb_verify_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
la = json.load(open(dump))
n = len(la["valid"])
rows = rows[-n:]                                   # the dumped launches are the last n of the process
assert len(rows) == n, (len(rows), n)
print(f"# {n} mb_verify_kernel dispatches of the command; the timed window = the last {steps}")
print("# dispatch  draft-carrying rows  logits rows     MB   rocprofv3 us   GB/s   of 8 TB/s")
tot_b = tot_t = 0.0
for i, (r, v, lr) in enumerate(zip(rows, la["valid"], la["rows"])):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    b = v * la["V"] * la["esz"]
    timed = i >= n - steps
    if timed:
        tot_b += b; tot_t += us
    print(f"  {i:6d}{'*' if timed else ' '}  {v:18d}  {lr:11d}  {b / 1e6:6.1f}  {us:12.1f}  {b / us / 1e3:6.0f}  {b / us / 1e3 / 8000:8.3f}")
print(f"# timed window (*): {tot_b / steps / 1e6:.1f} MB per launch, {tot_t / steps:.1f} us per launch = {tot_b / tot_t / 1e3:.0f} GB/s = {tot_b / tot_t / 1e3 / 8000:.3f} of 8 TB/s (rocprofv3 kernel durations)")
