#!/usr/bin/env python3
"""Minimal rocprofv3 target: a handful of jf_argmax_partial launches at the bench shapes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import ops  # noqa: E402

V = 152064
for dtype in (torch.bfloat16, torch.float32):
    for R in (32, 256, 512):
        x = torch.randn(R, V, device="cuda", dtype=torch.float32).to(dtype)
        packed = ops.new_packed(R, "cuda")
        for _ in range(10):
            ops.argmax_partial(x, packed)
            packed.zero_()
        torch.cuda.synchronize()
print("done")
