#!/usr/bin/env python3
"""How much of a decode step is launch overhead?  Eager vs hipGraph replay of the Qwen2.5-7B forward at decode shapes."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd.modeling.qwen2 import Qwen2Config, Qwen2Model, Qwen2Weights, StaticKVCache  # noqa: E402

if __import__("os").environ.get("PROBE_TUNED", "1") != "0":
    from jacobiforcing_amd.tuning import enable_tuned_gemms
    print("tuned GEMM table:", enable_tuned_gemms(), flush=True)
dev = torch.device("cuda")
cfg = Qwen2Config.qwen2_5_coder_7b()
model = Qwen2Model(cfg, Qwen2Weights(cfg, dev, seed=0))
P = int(__import__("os").environ.get("PROBE_P", "8"))
cache = StaticKVCache(cfg, P, 4096, 0, 1, dev)


def run(T, kv):
    ids = torch.randint(0, 1000, (P, T), device=dev)
    kvl = torch.full((P,), kv, dtype=torch.int32, device=dev)
    pos = kvl.view(P, 1) + torch.arange(T, dtype=torch.int32, device=dev).view(1, T)
    rp = torch.arange(P, dtype=torch.int32, device=dev)
    rc = torch.full((P,), -1, dtype=torch.int32, device=dev)
    rl = torch.full((P,), T, dtype=torch.int32, device=dev)
    f = lambda: model.forward(ids, pos, cache, rp, rc, rl, kvl, False, s_cur=kv + T)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10 * 1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 10 * 1e3
    print(f"T={T:3d} kv={kv:4d} rows={P * T:4d}: eager {eager:6.2f} ms   graph {graph:6.2f} ms", flush=True)


import os
shapes = [(16, 300), (20, 300), (32, 300), (48, 300), (64, 300), (32, 1000), (64, 1000)]
if os.environ.get("PROBE_SHAPES"):
    shapes = [tuple(int(x) for x in s.split(":")) for s in os.environ["PROBE_SHAPES"].split(",")]
for T, kv in shapes:
    try:
        run(T, kv)
    except Exception as e:
        print(f"T={T} kv={kv}: {type(e).__name__}: {e}")
