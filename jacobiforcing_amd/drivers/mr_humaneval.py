#!/usr/bin/env python3
"""Counterpart of the reference's timing driver ``JacobiForcing/jacobi_forcing_inference_MR_humaneval.py`` ("DRV-MR"):
decode a set of prompts with multiblock Jacobi + rejection recycling, write one CSV row per prompt with the reference's
columns (DRV-MR:259-273) and print the EOS-only means (DRV-MR:329-349).

    python -m jacobiforcing_amd.drivers.mr_humaneval --model /path/to/checkpoint --prompts humaneval.parquet
    python -m jacobiforcing_amd.drivers.mr_humaneval --synthetic 16            # random-init Qwen2.5-7B, synthetic prompts

Prompts shard over ranks when launched with torchrun (one process per GPU, no data-path collective)."""
from __future__ import annotations

import argparse
import csv
import json
from pathlib import Path

import torch

from .. import distributed as jd
from .. import ops
from ..engine.multiblock_decoder import MultiblockJacobiDecoder
from ..modeling.qwen2 import load_model_directory, Qwen2Config, Qwen2Model, Qwen2Weights
from ..synthetic import humaneval_shaped_prompts

COLUMNS = ["index", "task_id", "prompt_tokens", "new_tokens", "calls", "total_iterations", "avg_iter_per_call",
           "avg_iter_per_token", "time_sec", "toks_per_sec", "stop_reason"]

PROMPT_TEMPLATE = ("Please continue to complete the function. You are not allowed to modify the given code and do the "
                   "completion only. Please return all completed function in a codeblock. Here is the given code to do "
                   "completion:\n```\n{}\n```")         # DRV-MR:107-113


def load_prompts(args, cfg):
    if args.synthetic:
        return [(f"synthetic/{i}", p) for i, p in enumerate(humaneval_shaped_prompts(args.synthetic, seed=args.seed,
                                                                                    vocab_hi=min(151643, cfg.vocab_size - 2)))]
    import pandas as pd
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(args.tokenizer or args.model)
    out = []
    for idx, row in enumerate(pd.read_parquet(args.prompts).to_dict(orient="records")):
        text = tok.apply_chat_template([{"role": "user", "content": PROMPT_TEMPLATE.format(row["prompt"].strip())}],
                                       tokenize=False, add_generation_prompt=True)
        out.append((row.get("task_id", f"idx_{idx}"), tok(text)["input_ids"]))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=None)
    ap.add_argument("--allow-random-init", action="store_true", help="a --model directory without *.safetensors runs random-init")
    ap.add_argument("--tokenizer", default=None)
    ap.add_argument("--prompts", default=None, help="HumanEval parquet (column 'prompt')")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--batch", type=int, default=8, help="prompts decoded side by side on one GPU")
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--K", type=int, default=2)
    ap.add_argument("--r", type=float, default=0.85)
    ap.add_argument("--pool", type=int, default=4)
    ap.add_argument("--lookahead", type=float, default=0.0)
    ap.add_argument("--max-new-tokens", type=int, default=1024)
    ap.add_argument("--max-calls", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--csv", default="diffusion_profile_humaneval.csv")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="library-default GEMM selection instead of the committed table")
    ap.add_argument("--device", default=None, help="default: cuda:<local rank>")
    args = ap.parse_args(argv)

    info = jd.init_from_env()
    dev = torch.device(args.device) if args.device else torch.device("cuda", info.local_rank)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    if args.model:
        cfg, w = load_model_directory(args.model, dev, allow_random_init=args.allow_random_init)
    else:
        cfg = Qwen2Config.qwen2_5_coder_7b()
        w = Qwen2Weights(cfg, dev)
    model = Qwen2Model(cfg, w)
    from ..tuning import enable_tuned_gemms, grid_alignment
    tuned = (not args.no_tuned_gemms) and enable_tuned_gemms()       # rows then stay on the tuned M grid (t_align)
    prm = ops.MultiblockParams(n=args.n, K=args.K, r=args.r, lookahead_start_ratio=args.lookahead, n_gram_pool_size=args.pool,
                               eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id)
    items = jd.shard_prompts(list(enumerate(load_prompts(args, cfg))), info)
    rows, tot_tokens, tot_iters, tot_sec = [], 0, 0, 0.0
    for b0 in range(0, len(items), args.batch):
        chunk = items[b0:b0 + args.batch]
        dec = MultiblockJacobiDecoder(model, len(chunk), prm, max_seq_len=max(len(p) for _, (_, p) in chunk) + args.max_new_tokens + 6 * args.n + 128,
                                      t_align=grid_alignment(len(chunk), tuned)[0], logit_align=grid_alignment(len(chunk), tuned)[1])
        stats, gen_s, iters = dec.generate([p for _, (_, p) in chunk], max_new_tokens=args.max_new_tokens, max_calls=args.max_calls,
                                           seed=args.seed + b0)
        for (idx, (task, _)), st in zip(chunk, stats):
            r = st.row(gen_s)
            rows.append(dict(index=idx, task_id=task, **r))
        tot_tokens += sum(s.new_tokens for s in stats)
        tot_iters += sum(s.total_iterations for s in stats)
        tot_sec += gen_s
        del dec
    agg = jd.gather_throughput(tot_tokens, tot_iters, tot_sec, dev)
    path = args.csv if info.world_size == 1 else f"{Path(args.csv).stem}.rank{info.rank}.csv"
    with open(path, "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=COLUMNS)
        wr.writeheader()
        wr.writerows(rows)
    if info.rank == 0:
        eos = [r for r in rows if r["stop_reason"] == "eos"]
        mean = lambda k, rs: sum(r[k] for r in rs) / len(rs) if rs else float("nan")
        print(f"\n=== Jacobi decoding profile (rank 0 rows) — EOS-only: {len(eos)} / {len(rows)} ===")
        for k, label in (("new_tokens", "Avg new tokens / prompt"), ("calls", "Avg calls / prompt"),
                         ("avg_iter_per_call", "Avg iterations / call"), ("avg_iter_per_token", "Avg iterations / token"),
                         ("toks_per_sec", "Avg toks/sec")):
            print(f"{label}: {mean(k, eos):.4f}   (all prompts: {mean(k, rows):.4f})")
        print(json.dumps(dict(job_tokens_per_sec=agg["tokens"] / agg["seconds"] if agg["seconds"] else 0.0,
                              tokens_per_forward=agg["tokens"] / agg["iterations"] if agg["iterations"] else 0.0,
                              world_size=agg["world_size"], csv=path)))


if __name__ == "__main__":
    main()
