#!/usr/bin/env python3
"""Stage times of jf_rs_filter from in-kernel stamps (row 0): tools/build_exp.sh flttrace -DJF_EXP_FLT_TRACE, then
JF_LIB=tools/exp/libjf_exp_flttrace.so python tools/rs_filter_trace.py (tools/exp/ must be pushed: see .gpurunignore)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from jacobiforcing_amd import _native as N, ops
V = 152064
lib = N.lib()
for R in (1, 248):
    for k, tp in ((50, 0.0), (0, 0.9), (40, 0.95)):
        x = (torch.randn(R, V, device="cuda") * 3).to(torch.bfloat16)
        dn = torch.randint(0, V, (R,), device="cuda")
        p = torch.zeros(R, device="cuda"); m = torch.zeros(R, device="cuda"); s = torch.zeros(R, device="cuda")
        packed = ops.new_packed(R, "cuda"); ws = torch.zeros(R * 128 + 1024, device="cuda"); out = torch.empty_like(x)
        for _ in range(2):
            N.check(lib.jf_rs_probs(ops._ptr(x), 1, R, V, V, ops._ptr(dn), 0.8, ops._ptr(p), ops._ptr(m), ops._ptr(s), ops._ptr(packed), ops._ptr(ws), ws.numel() * 4, None))
            N.check(lib.jf_rs_filter(ops._ptr(x), 1, R, V, V, ops._ptr(dn), 0.8, k, tp, ops._ptr(out), ops._ptr(p), ops._ptr(m), ops._ptr(s), None))
            torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        lib_raw = ctypes.CDLL(N.LIB_PATH if not __import__("os").environ.get("JF_LIB") else __import__("os").environ["JF_LIB"])
        lib_raw.jf_exp_flt_trace(buf)
        t = [b / 100.0 for b in buf[:8]]
        names = ["s64", "probs+hist", "topk bisect", "topk counts/ties", "topk renorm", "topp search", "topp renorm"]
        segs = [t[1] - t[0], t[2] - t[1], (t[3] - t[2]) if k else 0, (t[4] - t[3]) if k else 0, (t[5] - t[4]) if k else 0,
                ((t[6] - (t[5] if k else t[2])) if tp else 0), (t[7] - t[6]) if tp else 0]
        print(f"R={R} k={k} p={tp}: " + "  ".join(f"{n} {v:.1f}" for n, v in zip(names, segs)), flush=True)
