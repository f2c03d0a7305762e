#!/usr/bin/env python3
"""Kernel microbench for jf_argmax_partial: GB/s of algorithmic bytes vs R and dtype, next to the load-only probe.

The library reads its tuning overrides (JF_ARGMAX_CHUNK / ITEMS / WAVE / NT) ONCE per process: sweep them by running this
script once per setting, e.g.  `for c in 8192 16384 65536; do JF_ARGMAX_CHUNK=$c python tools/microbench_argmax.py; done`."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import ops  # noqa: E402

V = 152064


import ctypes
_PROBE = None


def probe_lib():
    global _PROBE
    if _PROBE is None:
        _PROBE = ctypes.CDLL(str(Path(__file__).resolve().parent / "libprobe_stream.so"))
        _PROBE.probe_stream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    return _PROBE


def bench(R, dtype, iters=50, chunk=None, probe=False, batch=1):
    nbuf = max(1, min(8, int(600e6 // (R * V * (4 if dtype == torch.float32 else 2)))))
    xs = [torch.randn(R, V, device="cuda", dtype=torch.float32).to(dtype) for _ in range(nbuf)]
    packed = ops.new_packed(R, "cuda")
    for i in range(5):
        ops.argmax_partial(xs[i % nbuf], packed)
        packed.zero_()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    esz = 4 if dtype == torch.float32 else 2
    pchunk = chunk or (8192 if esz == 4 else 16384)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(iters):
        packed.zero_()
        ev[i][0].record()
        for j in range(batch):
            x = xs[(i * batch + j) % nbuf]
            if probe:
                probe_lib().probe_stream(x.data_ptr(), esz, R, V, V, packed.data_ptr(), pchunk, st)
            else:
                ops.argmax_partial(x, packed)
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 / batch for a, b in ev)
    med, mn = ts[len(ts) // 2], ts[0]
    byts = R * V * (4 if dtype == torch.float32 else 2)
    return med, mn, byts / med / 1e3, byts / mn / 1e3


if __name__ == "__main__":
    env_chunk = os.environ.get("JF_ARGMAX_CHUNK")
    chunks = [int(env_chunk) if env_chunk else None]           # the probe uses the same chunk as the kernel under test
    print(f"{'R':>5} {'dtype':>6} {'chunk':>7} {'MB':>8} {'med_us':>8} {'min_us':>8} {'GB/s(med)':>10} {'GB/s(min)':>10}")
    batch = int(os.environ.get("MB_BATCH", "8"))
    Rs = [int(r) for r in os.environ.get("MB_ROWS", "16,32,64,256,512,2048").split(",")]
    for dtype in (torch.float32, torch.bfloat16):
        for R in Rs:
            for c in chunks:
                for probe in (False, True):
                    med, mn, g1, g2 = bench(R, dtype, chunk=c, probe=probe, batch=batch)
                    tag = "probe" if probe else "argmx"
                    print(f"{R:5d} {str(dtype)[6:]:>6} {str(c):>7} {R * V * (4 if dtype == torch.float32 else 2) / 1e6:8.1f} "
                          f"{med:8.1f} {mn:8.1f} {g1:10.0f} {g2:10.0f} {tag} x{batch}", flush=True)
