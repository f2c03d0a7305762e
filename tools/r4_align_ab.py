#!/usr/bin/env python3
"""A/B of the batch-1 row alignment (tuning.grid_alignment) on the 7B-shaped model: the vs-AR section of bench.py under
JF_GRID_ALIGN = "t_align,logit_align".  One model load, every variant twice.  -> profiles/batch1_align_ab_r04.txt"""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from jacobiforcing_amd import ops  # noqa: E402
from jacobiforcing_amd.modeling.qwen2 import Qwen2Config, Qwen2Model, Qwen2Weights  # noqa: E402
from jacobiforcing_amd.tuning import enable_tuned_gemms  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    tuned = enable_tuned_gemms()
    cfg = Qwen2Config.qwen2_5_coder_7b()
    model = Qwen2Model(cfg, Qwen2Weights(cfg, dev, dtype=torch.bfloat16, seed=0))
    prm = ops.MultiblockParams(n=32, K=2, r=0.85, lookahead_start_ratio=0.0, n_gram_pool_size=4, eos_token_id=None, pad_token_id=cfg.pad_token_id)
    vocab_hi = min(151643, cfg.vocab_size - 2)
    variants = sys.argv[1:] or ["8,8", "8,64", "16,16", "16,64", "32,32", "64,64"]
    for rep in range(2):
        for v in variants:
            os.environ["JF_GRID_ALIGN"] = v
            r = bench.vs_ar_section(model, cfg, prm, tuned, vocab_hi, 82)
            print(f"align={v:6s} rep={rep} vs_ar={r['vs_ar']:.3f} jacobi_ms_per_step={r['jacobi_ms_per_step']:.3f} ar_ms_per_token={r['ar_ms_per_token']:.3f} "
                  f"iteration_cost_in_ar_steps={r['iteration_cost_in_ar_steps']:.3f} implied_at_4.1={r['implied_vs_ar_at_4_1_tokens_per_forward']:.2f} "
                  f"tpf={r['tokens_per_forward']:.2f} verified={r['verified']} split={json.dumps({k: round(x, 1) for k, x in r['split_us'].items()})}", flush=True)


if __name__ == "__main__":
    main()
