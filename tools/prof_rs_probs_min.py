#!/usr/bin/env python3
"""Minimal rocprofv3 target for jf_rs_probs: a few launches at R = 496 and R = 1984, bf16 and fp32, T = 1 (and bf16 T = 0.7:
the scaled-rounding variant).  Used by tools/pmc_rs_probs.sh."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import _native as N, ops  # noqa: E402

V = 152064
for dtype, T in ((torch.bfloat16, 1.0), (torch.bfloat16, 0.7), (torch.float32, 1.0)):
    for R in (496, 1984):
        x = (torch.randn(R, V, device="cuda") * 3).to(dtype)
        dn = torch.randint(0, V, (R,), device="cuda")
        p = torch.zeros(R, device="cuda"); m = torch.zeros(R, device="cuda"); s = torch.zeros(R, device="cuda")
        packed = ops.new_packed(R, "cuda")
        ws = torch.zeros(R * 128, device="cuda")
        for _ in range(6):
            N.check(N.lib().jf_rs_probs(ops._ptr(x), ops._dtype_code(x), R, V, V, ops._ptr(dn), T, ops._ptr(p), ops._ptr(m), ops._ptr(s),
                                        ops._ptr(packed), ops._ptr(ws), ws.numel() * 4, ops._stream(x.device)))
            packed.zero_()
        torch.cuda.synchronize()
