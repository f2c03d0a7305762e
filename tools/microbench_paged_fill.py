#!/usr/bin/env python3
"""jf_engine_fill (ops.PagedFill: input_ids / positions / slot_mapping / cu_seqlens / cache_seqlens of a whole Jacobi batch in
ONE launch) next to what it replaces: the reference's per-sequence fill of the same buffers (model_runner.py:1204-1265 — a
Python loop of ~8 small device operations per sequence), restated here with torch on the GPU for the timing only.
Batch 64 x block 32, sequences of 100-400 tokens, 256-token KV blocks.  The kernel has no consumer inside this package (its
own forward keeps one contiguous cache row per request): "boundary-only" in COVERAGE.md."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import ops  # noqa: E402

B, L, BS = 64, 32, 256
dev = torch.device("cuda")
g = np.random.default_rng(0)
S = g.integers(100, 400, size=B).tolist()
tables = [list(g.permutation(4096)[: (s - 1 + L + BS - 1) // BS]) for s in S]
draft = torch.from_numpy(g.integers(0, 150000, size=(B, L))).to(dev)
fill = ops.PagedFill(B, L, 16, BS, dev)

# the loop being replaced (per sequence: copy the draft row, positions = arange + S - 1, refresh the block-table row, gather the
# slots of the L draft positions, two running sums, the cache length): buffers allocated once, like jacobi_buffers
buf = dict(input_ids=torch.zeros(B * L, dtype=torch.int64, device=dev), positions=torch.zeros(B * L, dtype=torch.int64, device=dev),
           slot_mapping=torch.zeros(B * L, dtype=torch.int32, device=dev), cu_q=torch.zeros(B + 1, dtype=torch.int32, device=dev),
           cu_k=torch.zeros(B + 1, dtype=torch.int32, device=dev), cache=torch.zeros(B, dtype=torch.int32, device=dev),
           bt=torch.full((B, 16), -1, dtype=torch.int32, device=dev), base=torch.arange(L, dtype=torch.int64, device=dev))
bt_gpu = [torch.tensor(t, device=dev, dtype=torch.int32) for t in tables]        # (the reference caches these per sequence)


def python_loop():
    buf["cu_q"][0] = 0
    buf["cu_k"][0] = 0
    for i in range(B):
        s, o = S[i], i * L
        buf["input_ids"][o:o + L] = draft[i]
        buf["positions"][o:o + L] = buf["base"] + (s - 1)
        nb = len(tables[i])
        buf["bt"][i, :nb] = bt_gpu[i]
        buf["bt"][i, nb:] = -1
        pos = buf["base"] + (s - 1)
        buf["slot_mapping"][o:o + L] = (buf["bt"][i, (pos // BS)] * BS + (pos % BS)).to(torch.int32)
        buf["cu_q"][i + 1] = buf["cu_q"][i] + L
        buf["cu_k"][i + 1] = buf["cu_k"][i] + (s - 1) + L
        buf["cache"][i] = s - 1


def timed(f, n):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


one = timed(lambda: fill.fill(draft, S, tables, check=False), 200)
loop = timed(python_loop, 20)
out = fill.fill(draft, S, tables)
python_loop()
assert torch.equal(out[0], buf["input_ids"]) and torch.equal(out[1], buf["positions"]) and torch.equal(out[2], buf["slot_mapping"])
assert torch.equal(out[3], buf["cu_q"]) and torch.equal(out[4], buf["cu_k"]) and torch.equal(out[5], buf["cache"])
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); fill.fill(draft, S, tables, check=False); b.record(); torch.cuda.synchronize()
print(f"batch {B} x block {L}: ops.PagedFill.fill (jf_engine_fill, one launch + two host-to-device copies of {B} and {B * 16} ints) {one:8.1f} us per call "
      f"(host wall clock, stream drained per batch of calls; GPU time of one call {a.elapsed_time(b) * 1e3:.1f} us)")
print(f"batch {B} x block {L}: the per-sequence loop it replaces (MR:1204-1265, ~10 device operations per sequence)            {loop:8.1f} us per call   -> {loop / one:.0f} x;  identical buffers")
