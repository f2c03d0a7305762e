#!/usr/bin/env python3
"""Counterpart of the reference's single-block timing driver ``JacobiForcing/jacobi_forcing_inference_MATH500.py``
("DRV-SB"): greedy single-block Jacobi (``jacobi_forward_greedy``) prompt by prompt, batch 1 like the reference, one CSV row
per prompt with the same columns and the same conventions (prefill call counted, ``new_tokens`` minus the prefill token,
tokens/s over the generation-phase calls only, DRV-SB:86-200), EOS-only means at the end.

    python -m jacobiforcing_amd.drivers.sb_math500 --model /path/to/checkpoint --prompts math500.jsonl --n 128
    python -m jacobiforcing_amd.drivers.sb_math500 --synthetic 10 --n 16         # BASELINE config 2 shape, random-init 7B
"""
from __future__ import annotations

import argparse
import csv
import json
import random
import time
import types
from pathlib import Path

import torch

from ..hf_seam import Qwen2Backend, jacobi_forward_greedy
from ..modeling.qwen2 import load_model_directory, Qwen2Config, Qwen2Model, Qwen2Weights
from .mr_humaneval import COLUMNS

SYSTEM = "You are Qwen, created by Alibaba Cloud. You are a helpful assistant."        # DRV-SB:84


def load_prompts(args, cfg):
    if args.synthetic:
        rng = random.Random(args.seed)
        hi = min(151643, cfg.vocab_size - 2)
        return [(f"synthetic/{i}", [rng.randrange(hi) for _ in range(rng.randint(80, 400))]) for i in range(args.synthetic)]
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(args.tokenizer or args.model)
    out = []
    with open(args.prompts) as f:
        for idx, line in enumerate(f):
            row = json.loads(line)
            text = tok.apply_chat_template([{"role": "system", "content": SYSTEM}, {"role": "user", "content": row["problem"]}],
                                           tokenize=False, add_generation_prompt=True)
            out.append((row.get("task_id", f"idx_{idx}"), tok(text)["input_ids"]))
    return out[:args.limit] if args.limit else out


@torch.inference_mode()
def decode_one(me, prompt, n, eos_id, alt_eos_id, max_new_tokens, max_calls, rng):
    dev = me.jf_backend.device
    input_ids = torch.tensor([prompt], dtype=torch.int64, device=dev)
    generated = list(prompt)
    prompt_len = len(prompt)
    iters, total_new, calls, gen_time = [], 0, 0, 0.0
    cache = first = ngram = None
    stop = None
    while True:
        part = generated[prompt_len:]
        if (eos_id is not None and eos_id in part) or (alt_eos_id is not None and alt_eos_id in part):   # DRV-SB:107-117
            stop = "eos"
        elif total_new >= max_new_tokens:
            stop = "max_new_tokens"
        elif calls >= max_calls:
            stop = "max_calls"
        if stop:
            break
        if cache is None:                                                       # prefill with a random draft (DRV-SB:129-152)
            draft = [rng.choice(generated) for _ in range(n)]
            ids = torch.cat((input_ids, torch.tensor([draft], dtype=torch.int64, device=dev)), dim=-1)
            cache, first, ngram, _ = jacobi_forward_greedy(me, input_ids=ids, past_key_values=None, use_cache=True,
                                                           prefill_phase=True, n_token_seq_len=n, eos_token_id=eos_id)
            itr, added = 0, []
        else:
            if calls == 1:
                inp = ngram
            else:
                tail = [rng.choice(generated) for _ in range(n - 1)]
                inp = torch.cat((first.view(1, -1), torch.tensor([tail], dtype=torch.int64, device=dev)), dim=-1)
            t0 = time.perf_counter()
            cache, first, acc, itr = jacobi_forward_greedy(me, input_ids=inp, past_key_values=cache, use_cache=True,
                                                           prefill_phase=False, n_token_seq_len=n, eos_token_id=eos_id)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            gen_time += time.perf_counter() - t0
            added = acc[0].tolist()
        calls += 1
        iters.append(int(itr))
        generated += added
        total_new += len(added)
    total_new -= 1                                                              # "subtract prefill" (DRV-SB:191)
    tot_it = sum(iters)
    return dict(prompt_tokens=prompt_len, new_tokens=total_new, calls=calls, total_iterations=tot_it,
                avg_iter_per_call=tot_it / max(calls, 1), avg_iter_per_token=tot_it / max(total_new, 1), time_sec=gen_time,
                toks_per_sec=total_new / gen_time if gen_time > 0 else 0.0, stop_reason=stop), generated[prompt_len:]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=None)
    ap.add_argument("--allow-random-init", action="store_true", help="a --model directory without *.safetensors runs random-init")
    ap.add_argument("--tokenizer", default=None)
    ap.add_argument("--prompts", default=None, help="jsonl with a 'problem' field (MATH500)")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--limit", type=int, default=10)                             # the reference runs records[:10] (DRV-SB:72)
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--max-new-tokens", type=int, default=512)
    ap.add_argument("--max-calls", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--csv", default="diffusion_profile_math500.csv")
    ap.add_argument("--no-tuned-gemms", action="store_true")
    args = ap.parse_args(argv)
    dev = torch.device(args.device)
    if args.model:
        cfg, w = load_model_directory(args.model, dev, allow_random_init=args.allow_random_init)
    else:
        cfg = Qwen2Config.qwen2_5_coder_7b()
        w = Qwen2Weights(cfg, dev)
    model = Qwen2Model(cfg, w)
    from ..tuning import enable_tuned_gemms, SMALL_ROW_ALIGN
    tuned = dev.type == "cuda" and not args.no_tuned_gemms and enable_tuned_gemms()   # rows then stay on the tuned M grid
    rng = random.Random(args.seed)
    rows = []
    for idx, (task, prompt) in enumerate(load_prompts(args, cfg)):
        me = types.SimpleNamespace(jf_backend=Qwen2Backend(model, max_seq_len=len(prompt) + args.max_new_tokens + 4 * args.n + 72,
                                                           max_rows=1, max_tokens=len(prompt) + args.n + 8,
                                                           t_align=SMALL_ROW_ALIGN if tuned else 1))
        r, _ = decode_one(me, prompt, args.n, cfg.eos_token_id, getattr(cfg, "alt_eos_token_id", None), args.max_new_tokens,
                          args.max_calls, rng)
        rows.append(dict(index=idx, task_id=task, **r))
    with open(args.csv, "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=COLUMNS)
        wr.writeheader()
        wr.writerows(rows)
    eos = [r for r in rows if r["stop_reason"] == "eos"]
    mean = lambda k, rs: sum(r[k] for r in rs) / len(rs) if rs else float("nan")
    print(f"\n=== single-block Jacobi profile — EOS-only: {len(eos)} / {len(rows)} ===")
    for k, label in (("new_tokens", "Avg new tokens / prompt"), ("calls", "Avg calls / prompt"),
                     ("avg_iter_per_call", "Avg iterations / call"), ("avg_iter_per_token", "Avg iterations / token"),
                     ("toks_per_sec", "Avg toks/sec")):
        print(f"{label}: {mean(k, eos):.4f}   (all prompts: {mean(k, rows):.4f})")
    return rows


if __name__ == "__main__":
    main()
