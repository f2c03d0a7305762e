#!/usr/bin/env python3
"""jf_rs_filter (top-k / top-p of the target distribution as a probability tensor): microseconds per call against rows, dtype,
filter and the shape of the distribution (peaked: N(0, 3^2) logits; flat: N(0, 0.3^2) — in bf16 thousands of ids tie at a cut)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import _native as N, ops  # noqa: E402

V = 152064
for dtype in (torch.bfloat16, torch.float32):
    for R in (248, 1984):
        for scale, shape in ((3.0, "peaked"), (0.3, "flat")):
            x = (torch.randn(R, V, device="cuda") * scale).to(dtype)
            dn = torch.randint(0, V, (R,), device="cuda")
            p = torch.zeros(R, device="cuda"); m = torch.zeros(R, device="cuda"); s = torch.zeros(R, device="cuda")
            packed = ops.new_packed(R, "cuda")
            ws = torch.zeros(R * 128, device="cuda")
            out = torch.empty_like(x)
            m0, s0 = None, None
            for k, tp in ((50, 0.0), (0, 0.9), (40, 0.95)):
                def f():
                    N.check(N.lib().jf_rs_filter(ops._ptr(x), ops._dtype_code(x), R, V, V, ops._ptr(dn), 0.8, k, tp, ops._ptr(out), ops._ptr(p),
                                                 ops._ptr(m), ops._ptr(s), ops._stream(x.device)))
                N.check(N.lib().jf_rs_probs(ops._ptr(x), ops._dtype_code(x), R, V, V, ops._ptr(dn), 0.8, ops._ptr(p), ops._ptr(m), ops._ptr(s),
                                            ops._ptr(packed), ops._ptr(ws), ws.numel() * 4, ops._stream(x.device)))
                m0, s0 = m.clone(), s.clone()
                f(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                tot = 0.0
                for _ in range(3):
                    m.copy_(m0); s.copy_(s0)
                    a.record(); f(); b.record(); torch.cuda.synchronize()
                    tot += a.elapsed_time(b)
                kept = float((out[:8] > 0).sum(-1).float().mean())
                print(f"R={R:5d} {str(dtype)[6:]:>9} {shape:6s} top_k={k:3d} top_p={tp:4.2f}  {tot / 3 * 1e3:9.1f} us  ({tot / 3 * 1e3 / R:6.2f} us per row, ~{kept:.0f} ids kept)", flush=True)
