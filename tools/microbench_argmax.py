#!/usr/bin/env python3
"""Kernel microbench for jf_argmax_partial: GB/s of algorithmic bytes vs R, dtype, chunk size."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import ops  # noqa: E402

V = 152064


def bench(R, dtype, iters=50, chunk=None):
    if chunk:
        os.environ["JF_ARGMAX_CHUNK"] = str(chunk)
    else:
        os.environ.pop("JF_ARGMAX_CHUNK", None)
    nbuf = max(1, min(8, int(600e6 // (R * V * (4 if dtype == torch.float32 else 2)))))
    xs = [torch.randn(R, V, device="cuda", dtype=torch.float32).to(dtype) for _ in range(nbuf)]
    packed = ops.new_packed(R, "cuda")
    for i in range(5):
        ops.argmax_partial(xs[i % nbuf], packed)
        packed.zero_()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i in range(iters):
        packed.zero_()
        ev[i][0].record()
        ops.argmax_partial(xs[i % nbuf], packed)
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    med, mn = ts[len(ts) // 2], ts[0]
    byts = R * V * (4 if dtype == torch.float32 else 2)
    return med, mn, byts / med / 1e3, byts / mn / 1e3


if __name__ == "__main__":
    chunks = [None] + [int(c) for c in sys.argv[1:]]
    print(f"{'R':>5} {'dtype':>6} {'chunk':>7} {'MB':>8} {'med_us':>8} {'min_us':>8} {'GB/s(med)':>10} {'GB/s(min)':>10}")
    for dtype in (torch.float32, torch.bfloat16):
        for R in (16, 32, 64, 256, 512, 2048):
            for c in chunks:
                med, mn, g1, g2 = bench(R, dtype, chunk=c)
                print(f"{R:5d} {str(dtype)[6:]:>6} {str(c):>7} {R * V * (4 if dtype == torch.float32 else 2) / 1e6:8.1f} "
                      f"{med:8.1f} {mn:8.1f} {g1:10.0f} {g2:10.0f}", flush=True)
