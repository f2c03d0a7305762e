#!/usr/bin/env python3
"""Counterpart of the reference's greedy autoregressive baseline ``JacobiForcing/ar_inference_baseline.py:136-152``
("the AR baseline every speed-up is quoted against"): one prompt at a time, one token per forward over the same static KV
cache and the same Qwen2 forward the Jacobi decoders use, timed per prompt.

    python -m jacobiforcing_amd.drivers.ar_baseline --synthetic 4 --max-new-tokens 64
"""
from __future__ import annotations

import argparse
import json
import time
from pathlib import Path

import torch

from .. import ops
from ..modeling.qwen2 import load_model_directory, Qwen2Config, Qwen2Model, Qwen2Weights, StaticKVCache
from ..synthetic import humaneval_shaped_prompts


@torch.inference_mode()
def generate_greedy(model: Qwen2Model, prompt, max_new_tokens: int, eos_id=None):
    """Greedy AR continuation of one prompt; returns (tokens, seconds spent after the prefill)."""
    dev = model.device
    cache = StaticKVCache(model.cfg, 1, len(prompt) + max_new_tokens + 1, 0, 1, dev, dtype=model.dtype)
    z = torch.zeros(1, dtype=torch.int32, device=dev)

    def step(tokens, start):
        T = len(tokens)
        ids = torch.tensor([tokens], dtype=torch.int64, device=dev)
        pos = (start + torch.arange(T, dtype=torch.int32, device=dev)).view(1, T)
        logits = model.forward(ids, pos, cache, row_prompt=z, row_cand=z - 1, row_len=z + T, kv_len_rows=z + start,
                               any_candidates=False, logits_rows=slice(T - 1, T), s_cur=start + T)
        return int(ops.argmax_rows(logits)[0])

    out = [step(list(prompt), 0)]
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    while len(out) < max_new_tokens and (eos_id is None or out[-1] != eos_id):
        out.append(step([out[-1]], len(prompt) + len(out) - 1))
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return out, time.perf_counter() - t0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=None)
    ap.add_argument("--allow-random-init", action="store_true", help="a --model directory without *.safetensors runs random-init")
    ap.add_argument("--synthetic", type=int, default=4)
    ap.add_argument("--max-new-tokens", type=int, default=1024)      # ar_inference_baseline.py:141
    ap.add_argument("--seed", type=int, default=1234)                # ar_inference_baseline.py:35
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    dev = torch.device(args.device)
    if args.model:
        cfg, w = load_model_directory(args.model, dev, allow_random_init=args.allow_random_init)
    else:
        cfg = Qwen2Config.qwen2_5_coder_7b()
        w = Qwen2Weights(cfg, dev)
    model = Qwen2Model(cfg, w)
    rows = []
    for i, prompt in enumerate(humaneval_shaped_prompts(args.synthetic, seed=args.seed, vocab_hi=min(151643, cfg.vocab_size - 2))):
        toks, sec = generate_greedy(model, prompt, args.max_new_tokens, cfg.eos_token_id)
        rows.append(dict(index=i, prompt_tokens=len(prompt), new_tokens=len(toks), time_sec=sec,
                         toks_per_sec=(len(toks) - 1) / sec if sec > 0 else 0.0))
        print(json.dumps(rows[-1]), flush=True)
    if rows:
        print(json.dumps(dict(mean_toks_per_sec=sum(r["toks_per_sec"] for r in rows) / len(rows), prompts=len(rows))))
    return rows


if __name__ == "__main__":
    main()
