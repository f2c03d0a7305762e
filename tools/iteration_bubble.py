#!/usr/bin/env python3
"""GPU idle time around the loop body from a rocprofv3 --kernel-trace CSV.  Per iteration (one convergence launch:
mb_verify_kernel, or mb_step_kernel when the fused launch does not apply):

  idle_after_body   end of the loop body's LAST kernel (verify -> [kv_commit] -> pack, all queued before the host wakes up)
                    to the start of the next forward's first kernel: the mailbox poll + host control + first launch
  body              start of the convergence launch to the end of the last loop-body kernel
  idle_total        every gap between consecutive kernels from one convergence launch to the next (forward glue included)

    python tools/iteration_bubble.py <dir with *kernel_trace.csv>
"""
import csv
import glob
import sys

import numpy as np

files = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(files[0]))),
              key=lambda x: x[0])
BODY = ("mb_verify_kernel", "mb_step_kernel", "mb_pack_kernel", "kv_commit_kernel", "mb_begin_kernel", "mb_read_ret_kernel")
is_conv = lambda n: "mb_verify_kernel" in n or "mb_step_kernel" in n
is_body = lambda n: any(k in n for k in BODY)
vidx = [i for i, r in enumerate(rows) if is_conv(r[2])]
after, body, before, step, idle, big = [], [], [], [], [], {}
for a, b in zip(vidx, vidx[1:] + [None]):
    j = a
    while j + 1 < len(rows) and is_body(rows[j + 1][2]) and not is_conv(rows[j + 1][2]):
        j += 1
    if j + 1 < len(rows) and (b is None or j + 1 < b):
        after.append((rows[j + 1][0] - rows[j][1]) / 1e3)
        body.append((rows[j][1] - rows[a][0]) / 1e3)
    if a > 0:
        before.append((rows[a][0] - rows[a - 1][1]) / 1e3)
    if b is not None:
        step.append((rows[b][0] - rows[a][0]) / 1e3)
        tot = 0.0
        for i in range(a, b):
            g = (rows[i + 1][0] - rows[i][1]) / 1e3
            if g > 0:
                tot += g
            if g > 15:
                key = (rows[i][2].split("(")[0][:40], rows[i + 1][2].split("(")[0][:40])
                big.setdefault(key, []).append(g)
        idle.append(tot)
f = lambda a: f"median {np.median(a):8.1f} us  mean {np.mean(a):8.1f}  p90 {np.percentile(a, 90):8.1f}  (n={len(a)})" if len(a) else "n/a"
print("loop body: convergence launch start -> last body kernel end :", f(body))
print("idle_after_body: last body kernel end -> first forward kernel :", f(after))
print("previous kernel end -> convergence launch start              :", f(before))
print("GPU idle per iteration (sum of all gaps, forward included)   :", f(idle))
for k, v in sorted(big.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"   gap after {k[0]:40s} before {k[1]:40s}: {len(v):4d} x median {np.median(v):7.1f} us")
print("convergence launch -> next convergence launch                :", f(step))
