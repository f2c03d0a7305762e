#!/usr/bin/env python3
"""Tokens per forward MEASURED on a trained checkpoint — the toy of tests/golden/toy_periodic/ (tests/golden/train_toy_checkpoint.py: a
2-layer Qwen2 that has learnt token[i] = PERM[token[i - 6]]) — for every decoder behind LLM.generate, with the greedy outputs checked
against greedy AR.  Not Qwen2.5-Coder-7B: it shows the decoders accepting several tokens per forward on real logits and a real KV
cache, where random-init weights accept one and the bench's 3.9 are planted.   python tools/toy_tpf.py   (GPU: the product has no CPU path)"""
import os
import random
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))
os.environ.setdefault("JF_DTYPE", "float32")
from jacobiforcing_amd import LLM, SamplingParams  # noqa: E402
from train_toy_checkpoint import corpus  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
random.seed(0)
llm = LLM(str(ROOT / "tests" / "golden" / "toy_periodic"), tokenizer_path="none", device=dev, max_model_len=512, max_num_batched_tokens=8192, max_num_seqs=16)
lens = (13, 7, 25, 18, 120, 161, 40, 9, 77, 33, 50, 21)
prompts = [row[:n].tolist() for row, n in zip(corpus(np.random.default_rng(5), len(lens), 300), lens)]
N = 96
ar = [o["token_ids"] for o in llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=N, ignore_eos=True), use_tqdm=False)]
print(f"{len(prompts)} prompts x {N} tokens, greedy; AR = 1.00 token per forward by construction")
for L in (8, 16, 32):
    llm.model_runner.jacobi_decoder = None
    out = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=N, ignore_eos=True, decode_strategy="jacobi", jacobi_block_len=L), use_tqdm=False)
    st = llm.model_runner.jacobi_decoder.stats
    same = all(o["token_ids"][:N] == a for o, a in zip(out, ar))
    print(f"engine single block   n = {L:2d}: {st['tokens_accepted'] / st['num_jacobi_iterations'] / len(prompts):5.2f} tokens per forward   == AR: {same}")
for L in (16, 32):
    out = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=N, ignore_eos=True, decode_strategy="jacobi_multiblock_rejection_recycling",
                                               jacobi_block_len=L), use_tqdm=False)
    lm = llm.model_runner.last_multiblock
    tk, it = sum(len(s.token_ids) for s in lm["stats"]), sum(s.total_iterations for s in lm["stats"])
    same = all(o["token_ids"][:N] == a for o, a in zip(out, ar))
    print(f"multiblock K = 2      n = {L:2d}: {tk / it:5.2f} tokens per forward   == AR: {same}")
for T in (0.3, 0.8):
    llm.model_runner.jacobi_decoder = None
    out = llm.generate(prompts, SamplingParams(temperature=T, max_tokens=N, ignore_eos=True, decode_strategy="jacobi", jacobi_block_len=16), use_tqdm=False)
    st = llm.model_runner.jacobi_decoder.stats
    agree = np.mean([np.mean(np.asarray(o["token_ids"][:N]) == np.asarray(a)) for o, a in zip(out, ar)])
    print(f"engine sampling T = {T}  n = 16: {st['tokens_accepted'] / st['num_jacobi_iterations'] / len(prompts):5.2f} tokens per forward   agreement with greedy AR {agree:.3f}")
