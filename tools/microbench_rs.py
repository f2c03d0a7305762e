#!/usr/bin/env python3
"""jf_rs_probs (softmax-gather + argmax, logits read once): GB/s vs rows."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import _native as N, ops  # noqa: E402

V = 152064
TEMPS = [float(a) for a in sys.argv[1:]] or [1.0, 0.8, 0.7]      # 0.8: fl(1/T) = 1.25 exactly (bf16 ties everywhere)
for dtype, T in [(d, t) for d in (torch.bfloat16, torch.float32) for t in TEMPS]:
    for R in (31, 248, 496, 1984):
        x = (torch.randn(R, V, device="cuda") * 3).to(dtype)
        dn = torch.randint(0, V, (R,), device="cuda")
        p = torch.zeros(R, device="cuda"); m = torch.zeros(R, device="cuda"); s = torch.zeros(R, device="cuda")
        packed = ops.new_packed(R, "cuda")
        ws = torch.zeros(R * 128, device="cuda")
        f = lambda: N.check(N.lib().jf_rs_probs(ops._ptr(x), ops._dtype_code(x), R, V, V, ops._ptr(dn), T, ops._ptr(p), ops._ptr(m),
                                                ops._ptr(s), ops._ptr(packed), ops._ptr(ws), ws.numel() * 4, ops._stream(x.device)))
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            f()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100
        byts = R * V * x.element_size()
        print(f"R={R:5d} {str(dtype)[6:]:>9} T={T:<4} {byts / 1e6:8.1f} MB {us:8.1f} us {byts / us / 1e3:7.0f} GB/s", flush=True)
