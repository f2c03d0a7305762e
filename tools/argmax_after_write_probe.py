#!/usr/bin/env python3
"""Why is the argmax launch 15-20 % slower inside the decode step than in the microbenchmark?  Time single launches
(HIP events around each) over the same 1313 x 152064 bf16 logits under different predecessors:
  cold      : a different large buffer was streamed before (logits not in any cache)
  rewritten : the logits were just overwritten by an elementwise kernel (dirty lines in L2 / Infinity Cache)
  gemm      : the logits were just produced by the lm_head GEMM (what the decode step does)
  repeat    : the same launch again right after itself (logits partly cached)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import ops  # noqa: E402
from jacobiforcing_amd.tuning import enable_tuned_gemms  # noqa: E402

enable_tuned_gemms()
R, V, H = 1313, 152064, 3584
dev = "cuda"
w = torch.randn(V, H, device=dev, dtype=torch.bfloat16) * 0.02
h = torch.randn(1536, H, device=dev, dtype=torch.bfloat16)
logits = torch.randn(1536, V, device=dev, dtype=torch.bfloat16)
other = torch.randn(1536, V, device=dev, dtype=torch.bfloat16)
src = torch.randn(1536, V, device=dev, dtype=torch.bfloat16)
idx = torch.arange(1536, dtype=torch.int32, device=dev)
idx[R:] = -1
packed = ops.new_packed(1536, dev)


def timed(pre):
    us = []
    for _ in range(12):
        pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.argmax_scatter(logits, idx, packed)
        b.record()
        torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3)
        packed.zero_()
    us.sort()
    return us[len(us) // 2], us[0]


def cold():
    other.add_(1.0)


def rewritten():
    logits.copy_(src)


def gemm():
    torch.mm(h, w.t(), out=logits)


def repeat():
    ops.argmax_scatter(logits, idx, packed)
    packed.zero_()


for name, f in (("cold", cold), ("rewritten", rewritten), ("gemm", gemm), ("repeat", repeat)):
    med, mn = timed(f)
    print(f"{name:10s} median {med:7.1f} us  min {mn:7.1f} us   {R * V * 2 / med / 1e3:7.0f} GB/s", flush=True)
