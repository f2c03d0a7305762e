#!/usr/bin/env python3
"""GPU idle time around the loop body from a rocprofv3 --kernel-trace CSV: for every mb_verify_kernel dispatch, the gap between
its end and the start of the next kernel (descriptor read-back + host control + mb_pack + first forward launch), and the gap before
it (lm_head GEMM end -> verify start).

    python tools/iteration_bubble.py <dir with *kernel_trace.csv>
"""
import csv
import glob
import sys

import numpy as np

files = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(files[0]))), key=lambda x: x[0])
after, before, pack_to_next, step = [], [], [], []
last_verify_start = None
for i, (s, e, name) in enumerate(rows):
    if "mb_verify_kernel" not in name:
        continue
    if i + 1 < len(rows):
        after.append((rows[i + 1][0] - e) / 1e3)
        j = i + 1
        if "mb_pack_kernel" in rows[j][2] and j + 1 < len(rows):
            pack_to_next.append((rows[j + 1][0] - rows[j][1]) / 1e3)
    if i > 0:
        before.append((s - rows[i - 1][1]) / 1e3)
    if last_verify_start is not None:
        step.append((s - last_verify_start) / 1e3)
    last_verify_start = s
f = lambda a: f"median {np.median(a):8.1f} us  mean {np.mean(a):8.1f}  p90 {np.percentile(a, 90):8.1f}  (n={len(a)})" if len(a) else "n/a"
# idle time per iteration: every gap between consecutive kernels from one verify end to the next verify start
vidx = [i for i, r in enumerate(rows) if "mb_verify_kernel" in r[2]]
idle, big = [], {}
for a, b in zip(vidx[:-1], vidx[1:]):
    tot = 0.0
    for i in range(a, b):
        g = (rows[i + 1][0] - rows[i][1]) / 1e3
        if g > 0:
            tot += g
        if g > 15:
            key = (rows[i][2].split("(")[0][:40], rows[i + 1][2].split("(")[0][:40])
            big.setdefault(key, []).append(g)
    idle.append(tot)
print("GPU idle per iteration (sum of gaps) :", f(idle))
for k, v in sorted(big.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"   gap after {k[0]:40s} before {k[1]:40s}: {len(v):4d} x median {np.median(v):7.1f} us")
print("previous kernel end -> verify start :", f(before))
print("verify end -> next kernel start     :", f(after))
print("mb_pack end -> next kernel start    :", f(pack_to_next))
print("verify start -> next verify start   :", f(step))
