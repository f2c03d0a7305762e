#!/usr/bin/env python3
"""What the library GEMMs of one decode forward achieve at the batch-1 shapes (rows M = 1: AR token; M = 64: one Jacobi
iteration of one prompt), with and without the committed TunableOp table: microseconds and weight-streaming GB/s per GEMM.

    python tools/gemm_shapes_probe.py [--rows 1 8 16 32 64 128]
"""
import argparse
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd.modeling.qwen2 import Qwen2Config  # noqa: E402
from jacobiforcing_amd.tuning import enable_tuned_gemms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[1, 8, 16, 32, 64, 128])
    ap.add_argument("--no-tuned", action="store_true")
    a = ap.parse_args()
    tuned = (not a.no_tuned) and enable_tuned_gemms()
    cfg = Qwen2Config.qwen2_5_coder_7b()
    dev = torch.device("cuda")
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    shapes = dict(qkv=((nq + 2 * nkv) * hd, H), o=(H, nq * hd), gate_up=(2 * I, H), down=(H, I), lm_head=(V, H))
    # several copies of every weight so that consecutive calls do not find it in the Infinity Cache (a layer's weights are cold)
    ws = {k: [torch.randn(n, kk, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(6 if k != "lm_head" else 2)]
          for k, (n, kk) in shapes.items()}
    print(f"# tuned table: {tuned}")
    print("# rows  " + "  ".join(f"{k:>22s}" for k in shapes) + "     layer x 28 + lm_head")
    for M in a.rows:
        cells, layer_us = [], 0.0
        for k, (n, kk) in shapes.items():
            x = torch.randn(M, kk, device=dev, dtype=torch.bfloat16)
            for i in range(6):
                F.linear(x, ws[k][i % len(ws[k])])
            torch.cuda.synchronize()
            reps = 24
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                F.linear(x, ws[k][i % len(ws[k])])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            gbs = n * kk * 2 / us / 1e3
            cells.append(f"{us:8.1f} us {gbs:6.0f} GB/s")
            layer_us += us * (28 if k != "lm_head" else 1)
        print(f"{M:6d}  " + "  ".join(f"{c:>22s}" for c in cells) + f"   {layer_us / 1e3:8.2f} ms")


if __name__ == "__main__":
    main()
