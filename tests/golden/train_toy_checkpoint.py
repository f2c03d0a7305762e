#!/usr/bin/env python3
"""Trains the tiny Qwen2 checkpoint under tests/golden/toy_periodic/ (config.json + model.safetensors, ~0.3 MB) — a TRAINED model whose
greedy continuation is predictable several tokens ahead, so that Jacobi decoding accepts more than one token per forward on the REAL
forward and KV cache (random-init weights accept ~1; the bench's 3.9 tokens per forward are planted logits).  The language: every
sequence repeats a random pattern of PERIOD tokens, each repetition mapped through a fixed permutation of the vocabulary
(token[i] = PERM[token[i - PERIOD]]), so a position's token follows from the token PERIOD places back — already committed for the first
PERIOD draft positions of a block.  transformers' Qwen2ForCausalLM + AdamW, CPU, ~1 minute; deterministic (seed 0).

    python tests/golden/train_toy_checkpoint.py          # rewrites tests/golden/toy_periodic/
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

OUT = Path(__file__).resolve().parent / "toy_periodic"
V, PERIOD, SEQ = 64, 6, 72
CFG = dict(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
           max_position_embeddings=512, rms_norm_eps=1e-6, rope_theta=10000.0, tie_word_embeddings=False, eos_token_id=V - 1,
           pad_token_id=V - 2, model_type="qwen2", torch_dtype="float32")


def corpus(rng: np.random.Generator, n: int, length: int = SEQ) -> np.ndarray:
    perm = np.random.default_rng(12345).permutation(V - 2)          # fixed for the language (ids V-2, V-1 = pad / eos never occur)
    x = np.empty((n, length), dtype=np.int64)
    x[:, :PERIOD] = rng.integers(0, V - 2, size=(n, PERIOD))
    for i in range(PERIOD, length):
        x[:, i] = perm[x[:, i - PERIOD]]
    return x


def main():
    from transformers import Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = Qwen2ForCausalLM(Qwen2Config(**{k: v for k, v in CFG.items() if k != "torch_dtype"}))
    opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=0.0)
    rng = np.random.default_rng(0)
    steps = 900
    for step in range(steps):
        for g in opt.param_groups:
            g["lr"] = 3e-3 * min(1.0, (step + 1) / 50) * (0.1 + 0.9 * (1 - step / steps))
        ids = torch.from_numpy(corpus(rng, 64))
        out = model(input_ids=ids, labels=ids)
        opt.zero_grad()
        out.loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        if step % 100 == 0 or step == steps - 1:
            with torch.no_grad():
                t = torch.from_numpy(corpus(np.random.default_rng(999), 128))
                pred = model(input_ids=t).logits[:, PERIOD - 1:-1].argmax(-1)
                acc = (pred == t[:, PERIOD:]).float().mean().item()
            print(f"step {step:4d} loss {out.loss.item():.4f} next-token accuracy behind the first period {acc:.4f}", flush=True)
    OUT.mkdir(parents=True, exist_ok=True)
    from safetensors.torch import save_file
    save_file({k: v.detach().contiguous() for k, v in model.state_dict().items()}, str(OUT / "model.safetensors"))
    (OUT / "config.json").write_text(json.dumps(CFG, indent=1))
    (OUT / "README.txt").write_text("A trained tiny Qwen2 (2 layers, hidden 64, vocabulary 64): token[i] = PERM[token[i - 6]].  Written by "
                                    "tests/golden/train_toy_checkpoint.py (seed 0, 900 AdamW steps on the CPU); used by tests/test_trained_toy.py.\n")
    print("wrote", OUT, f"next-token accuracy {acc:.4f}")
    if acc < 0.995:
        sys.exit("the toy did not learn its language")


if __name__ == "__main__":
    main()
