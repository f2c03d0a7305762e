#!/usr/bin/env python3
"""Phase times of the bf16 jf_rs_filter (rs_filter_hist_kernel) from in-kernel stamps of row 0: tools/build_exp.sh flttrace
-DJF_EXP_FLT_TRACE, then JF_LIB=tools/exp/libjf_exp_flttrace.so python tools/rs_filter_trace.py (tools/exp/ must be pushed: see
.gpurunignore)."""
import ctypes, os, sys, torch
sys.path.insert(0, ".")
from jacobiforcing_amd import _native as N, ops
V = 152064
lib = N.lib()
raw = ctypes.CDLL(os.environ.get("JF_LIB", str(N.LIB_PATH)))
names = ["zero+tab", "count pass", "S", "top-k", "top-p", "tile counts", "tie ids"]
ZONE = os.environ.get("JF_RS_FILTER_ZONE", "1") != "0"
for R in ((1, 64, 256, 768, 1984) if ZONE else (1, 64, 256)):
    for scale, shape in ((3.0, "peaked"), (0.3, "flat")):
        for k, tp in ((50, 0.0), (0, 0.9), (50, 0.9)):
            x = (torch.randn(R, V, device="cuda") * scale).to(torch.bfloat16)
            dn = torch.randint(0, V, (R,), device="cuda")
            p = torch.zeros(R, device="cuda"); m = torch.zeros(R, device="cuda"); s = torch.zeros(R, device="cuda")
            packed = ops.new_packed(R, "cuda"); ws = torch.zeros(int(lib.jf_rs_workspace_bytes(R, V)) // 4 + 4, device="cuda")
            rf = ops.RowFilter(x.device)
            for _ in range(2):
                packed.zero_()
                N.check(lib.jf_rs_probs(ops._ptr(x), 1, R, V, V, ops._ptr(dn), 0.8, ops._ptr(p), ops._ptr(m), ops._ptr(s), ops._ptr(packed), ops._ptr(ws), ws.numel() * 4, None))
                rf.run(x, dn, 0.8, k, tp, p, m, s)
                torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 16)()
            raw.jf_exp_fh_trace(buf)
            if ZONE:          # rs_filter_zone_kernel's workgroup 0: stamps 8-11 in front of fh_solve's 3-7 (a stage that did not run keeps an older stamp)
                z = [buf[i] / 100.0 for i in (8, 9, 10, 11, 3, 4, 5, 6, 7)]
                zn = ["zero", "count pass", "list", "S + probs", "top-k", "top-p", "tile counts", "tie ids"]
                print(f"zone R={R:4d} {shape:6s} k={k:2d} p={tp}: " + "  ".join(f"{n} {z[i + 1] - z[i]:6.1f}" for i, n in enumerate(zn)) + f"   total {max(z) - z[0]:6.1f} us", flush=True)
                continue
            t = [b / 100.0 for b in buf[:8]]
            print(f"R={R:3d} {shape:6s} k={k:2d} p={tp}: " + "  ".join(f"{n} {t[i + 1] - t[i]:6.1f}" for i, n in enumerate(names)) + f"   total {t[7] - t[0]:6.1f} us", flush=True)
