#!/usr/bin/env python3
"""jf_rs_filter (top-k / top-p of the target distribution as one record per row): microseconds per call against rows, dtype, filter
and the shape of the distribution (peaked: N(0, 3^2) logits; flat: N(0, 0.3^2) — in bf16 thousands of ids tie at a cut), and the
fraction of the 8 TB/s roofline the call reaches on the bytes it has to read (the logits once for the pattern counts, once more
for the tie ids: 2 x R x V x 2 B for bf16)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import _native as N, ops  # noqa: E402

V = 152064
dtypes = (torch.bfloat16, torch.float32) if "--f32" in sys.argv else (torch.bfloat16,)
for dtype in dtypes:
    for R in (64, 248, 1984):
        if dtype == torch.float32 and R > 248:
            continue
        for scale, shape in ((3.0, "peaked"), (0.3, "flat")):
            x = (torch.randn(R, V, device="cuda") * scale).to(dtype)
            if shape == "peaked":
                x[torch.arange(R), torch.randint(0, V, (R,))] = 14.0
            dn = torch.randint(0, V, (R,), device="cuda")
            p = torch.zeros(R, device="cuda"); m = torch.zeros(R, device="cuda"); s = torch.zeros(R, device="cuda")
            packed = ops.new_packed(R, "cuda")
            ws = torch.zeros(int(N.lib().jf_rs_workspace_bytes(R, V)) // 4 + 4, device="cuda")
            rf = ops.RowFilter(x.device)
            for k, tp in ((50, 0.0), (0, 0.9), (50, 0.9)):
                N.check(N.lib().jf_rs_probs(ops._ptr(x), ops._dtype_code(x), R, V, V, ops._ptr(dn), 0.8, ops._ptr(p), ops._ptr(m), ops._ptr(s),
                                            ops._ptr(packed), ops._ptr(ws), ws.numel() * 4, ops._stream(x.device)))
                packed.zero_()
                m0, s0 = m.clone(), s.clone()
                rf.run(x, dn, 0.8, k, tp, p, m, s); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                tot = 0.0
                for _ in range(5):
                    m.copy_(m0); s.copy_(s0)
                    a.record(); rf.run(x, dn, 0.8, k, tp, p, m, s); b.record(); torch.cuda.synchronize()
                    tot += a.elapsed_time(b)
                us = tot / 5 * 1e3
                kept = float((rf.expand(x[:8].contiguous(), 0.8) > 0).sum(-1).float().mean()) if R >= 8 else 0.0
                nbytes = 2 * R * V * x.element_size()
                print(f"R={R:5d} {str(dtype)[6:]:>9} {shape:6s} top_k={k:3d} top_p={tp:4.2f}  {us:9.1f} us  ({us / R:6.2f} us per row, "
                      f"{nbytes / us / 1e3:7.0f} GB/s = {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s on two reads, ~{kept:.0f} ids kept)", flush=True)
