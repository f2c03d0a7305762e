#!/usr/bin/env python3
"""One shape of jf_rs_probs + jf_rs_filter for the profiler (tools/pmc_rs_filter.sh): R = 1 984 rows x V = 152 064, bf16, T = 0.8,
top_k = 50 / top_p = 0.9; argv[1] = peaked | flat; 6 calls (the first one warms up)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import _native as N, ops  # noqa: E402

V, R = 152064, 1984
shape = sys.argv[1] if len(sys.argv) > 1 else "peaked"
k, tp = (int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (50, 0.9)
x = (torch.randn(R, V, device="cuda") * (3.0 if shape == "peaked" else 0.3)).to(torch.bfloat16)
if shape == "peaked":
    x[torch.arange(R), torch.randint(0, V, (R,))] = 14.0
dn = torch.randint(0, V, (R,), device="cuda")
p, m, s = (torch.zeros(R, device="cuda") for _ in range(3))
packed = ops.new_packed(R, "cuda")
ws = torch.zeros(int(N.lib().jf_rs_workspace_bytes(R, V)) // 4 + 4, device="cuda")
rf = ops.RowFilter(x.device)
for _ in range(6):
    packed.zero_()
    N.check(N.lib().jf_rs_probs(ops._ptr(x), 1, R, V, V, ops._ptr(dn), 0.8, ops._ptr(p), ops._ptr(m), ops._ptr(s), ops._ptr(packed), ops._ptr(ws),
                                ws.numel() * 4, ops._stream(x.device)))
    rf.run(x, dn, 0.8, k, tp, p, m, s)
    torch.cuda.synchronize()
print("done", shape, k, tp)
