#!/usr/bin/env python3
"""Throughput of the ENGINE path (LLM.generate -> scheduler -> ModelRunner -> JacobiDecoder / multiblock decoder) on a
random-init Qwen2.5-Coder-7B-shaped model — the reference's `inference_engine` scenario ("800-1000 tok/s on a
single GPU", batch decode).  Random weights accept ~1 token per forward, so this measures the engine's plumbing at scale,
not a checkpoint's acceptance.

    python tools/engine_throughput.py [--batch 64] [--block-len 32] [--max-tokens 128]
"""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import LLM, SamplingParams  # noqa: E402
from jacobiforcing_amd.synthetic import humaneval_shaped_prompts  # noqa: E402
from jacobiforcing_amd.tuning import enable_tuned_gemms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--block-len", type=int, default=32)
    ap.add_argument("--max-tokens", type=int, default=128)
    ap.add_argument("--only", default="", help="substring of the mode name to run")
    ap.add_argument("--keep-eos", action="store_true", help="keep the EOS id (a request that emits it by chance is decoded alone afterwards)")
    args = ap.parse_args()
    cfg = dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
               num_key_value_heads=4, max_position_embeddings=32768, rms_norm_eps=1e-6, rope_theta=1000000.0,
               tie_word_embeddings=False, eos_token_id=151645, pad_token_id=151643, model_type="qwen2")
    if not args.keep_eos:
        cfg["eos_token_id"] = -1        # random weights emit any id: without this a run may or may not contain an EOS straggler
    d = tempfile.mkdtemp()
    (Path(d) / "config.json").write_text(json.dumps(cfg))
    enable_tuned_gemms()
    llm = LLM(d, tokenizer_path="none", max_model_len=2048, max_num_batched_tokens=65536, max_num_seqs=args.batch)
    prompts = humaneval_shaped_prompts(args.batch, seed=1234, vocab_hi=151643)
    out = {}
    for name, sp in [
        ("autoregressive", SamplingParams(temperature=0.0, max_tokens=args.max_tokens, ignore_eos=True)),
        ("jacobi greedy", SamplingParams(temperature=0.0, max_tokens=args.max_tokens, ignore_eos=True, decode_strategy="jacobi",
                                         jacobi_block_len=args.block_len)),
        ("jacobi T=0.8", SamplingParams(temperature=0.8, max_tokens=args.max_tokens, ignore_eos=True, decode_strategy="jacobi",
                                        jacobi_block_len=args.block_len)),
        ("multiblock", SamplingParams(temperature=0.0, max_tokens=args.max_tokens, ignore_eos=True,
                                      decode_strategy="jacobi_multiblock_rejection_recycling", jacobi_block_len=args.block_len)),
    ]:
        if args.only and args.only not in name:
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = llm.generate(prompts, sp, use_tqdm=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        toks = sum(len(r["token_ids"]) for r in res)
        out[name] = dict(tokens=toks, seconds=round(dt, 3), tokens_per_s=round(toks / dt, 1))
        print(f"{name:16s} {toks:6d} tokens in {dt:7.2f} s (prefill included) = {toks / dt:8.1f} tok/s", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
