#!/usr/bin/env python3
"""jf_rope_kv_append (RoPE + query re-layout + KV append of one layer) at the decode step's shape: Qwen2.5-7B heads
(28 q + 4 kv, D = 128), bf16, N tokens = rows x T.  JF_ROPE_SCALAR=1 selects the element-wise kernel (A/B).

    python tools/microbench_rope.py [rows T] ...
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import ops  # noqa: E402

nq, nkv, D = 28, 4, 128
shapes = [(64, 34), (8, 34), (1, 34), (128, 40)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
for R, T in shapes:
    N = R * T
    S_max, T_max = 2048, 64
    qkv = torch.randn(N, (nq + 2 * nkv) * D, device="cuda").to(torch.bfloat16)
    pos = torch.randint(0, 1024, (N,), dtype=torch.int32, device="cuda")
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.outer(torch.arange(2048, dtype=torch.float32), inv)
    cos, sin = fr.cos().cuda(), fr.sin().cuda()
    kc = torch.zeros(R, nkv, S_max, D, dtype=torch.bfloat16, device="cuda"); vc = torch.zeros_like(kc)
    slot = (torch.arange(R, device="cuda").repeat_interleave(T) * S_max + 100 + torch.arange(T, device="cuda").repeat(R)).to(torch.int64)
    f = lambda: ops.rope_kv_append(qkv, T, nq, nkv, D, pos, cos, sin, kc, vc, slot)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        f()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 50
    moved = N * (nq + 2 * nkv) * D * 2 * 2                     # every element read once and written once
    print(f"rows {R:4d} x T {T:3d} = {N:5d} tokens: {us:7.1f} us per layer   {moved / 1e6:7.1f} MB moved   {moved / us / 1e3:7.0f} GB/s")
