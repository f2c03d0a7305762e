#!/usr/bin/env python3
"""Group a rocprofv3 --kernel-trace --stats CSV (kernel_stats) into classes: library GEMMs, attention, K/V gathers and other
index kernels, elementwise, this package's kernels.

    python tools/kernel_classes.py <dir with *kernel_stats.csv>
"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(f"{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
cls = defaultdict(lambda: [0, 0.0])
own = defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in csv.DictReader(open(f)):
    n, calls, t = r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])
    tot += t
    if "Cijk" in n: k = "library GEMM"
    elif "attn_fwd" in n: k = "attention (SDPA)"
    elif "index_elementwise" in n or "scatter_gather" in n or "vectorized_gather" in n or "index_select" in n: k = "index / gather / scatter (K/V gathers, embedding, position lists)"
    elif "layer_norm" in n or "rms_norm" in n: k = "norm"
    elif "at::native" in n or "rocclr" in n: k = "elementwise / copies / fills"
    else:
        k = "this package's kernels"
        s = n.split("(")[0].replace("void ", "")[:60]
        own[s][0] += calls; own[s][1] += t
    cls[k][0] += calls; cls[k][1] += t
for k, (c, t) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:70s} calls {c:7d}  {t / 1e6:9.2f} ms  {100 * t / tot:5.1f} %")
print()
for k, (c, t) in sorted(own.items(), key=lambda kv: -kv[1][1]):
    print(f"    {k:66s} calls {c:7d}  {t / 1e6:9.2f} ms  {100 * t / tot:5.2f} %  avg {t / c / 1e3:8.1f} us")
