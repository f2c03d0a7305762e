#!/usr/bin/env python3
"""Where an iteration of the engine's non-greedy decoding (bench.py's `nongreedy` section: batch 64 x block 32) goes: wall time per
iteration, GPU time by kernel (torch.profiler), and the host side's share."""
import json
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from jacobiforcing_amd import LLM, SamplingParams  # noqa: E402
from jacobiforcing_amd.engine.model_runner import ModelRunner  # noqa: E402
from jacobiforcing_amd.modeling.qwen2 import Qwen2Config, Qwen2Weights  # noqa: E402
from jacobiforcing_amd.tuning import enable_tuned_gemms  # noqa: E402

P, L = 64, 32
dev = torch.device("cuda")
enable_tuned_gemms()
cfg = Qwen2Config.qwen2_5_coder_7b()
weights = Qwen2Weights(cfg, dev, dtype=torch.bfloat16, seed=0)
d = tempfile.mkdtemp()
(Path(d) / "config.json").write_text(json.dumps(dict(
    vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
    num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
    max_position_embeddings=cfg.max_position_embeddings, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
    tie_word_embeddings=cfg.tie_word_embeddings, eos_token_id=-1, pad_token_id=cfg.pad_token_id, model_type="qwen2")))
ModelRunner.shared_weights = weights
llm = LLM(d, tokenizer_path="none", max_model_len=2048, max_num_batched_tokens=65536, max_num_seqs=P)
ModelRunner.shared_weights = None
prompts = [p[:400] for p in bench.humaneval_shaped_prompts(P, seed=4242, vocab_hi=min(151643, cfg.vocab_size - 2))]
mk = lambda mt: SamplingParams(temperature=0.8, max_tokens=mt, ignore_eos=True, decode_strategy="jacobi", jacobi_block_len=L)
llm.generate(prompts, mk(4), use_tqdm=False)
torch.cuda.synchronize()
for mt in (8, 24):
    t0 = time.perf_counter(); llm.generate(prompts, mk(mt), use_tqdm=False); torch.cuda.synchronize()
    print(f"max_tokens={mt}: {time.perf_counter() - t0:.3f} s", flush=True)
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    llm.generate(prompts, mk(12), use_tqdm=False)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=18, max_name_column_width=70))
