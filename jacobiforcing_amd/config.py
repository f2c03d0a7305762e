"""Engine configuration — field names and checks of the reference's ``inference_engine/config.py:6-52``.

Differences (documented in DESIGN.md): tensor parallelism is not implemented on this path (prompts replicate over
GPUs instead), so ``tensor_parallel_size`` must be 1; a directory without ``*.safetensors`` gets
random-init weights (there are no checkpoints on the build/GPU boxes)."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Any, Optional


@dataclass
class Config:
    model: str
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 512
    max_model_len: int = 8192
    gpu_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    hf_config: Optional[Any] = None
    eos: int = -1
    pad: int = -1
    kvcache_block_size: int = 256
    num_kvcache_blocks: int = -1

    jacobi_enabled: bool = True

    jacobi_block_len: int = 64
    jacobi_max_blocks: int = 2
    jacobi_spawn_ratio: float = 0.8
    jacobi_lookahead_start_ratio: float = 0.0
    jacobi_n_gram_pool_size: int = 4
    jacobi_max_iterations: int = 128

    def __post_init__(self):
        model_path = self.model
        if os.path.isdir(model_path):                                   # config.py:31-41 (latest checkpoint-N subdir)
            ck = [d for d in os.listdir(model_path)
                  if d.startswith("checkpoint-") and os.path.isdir(os.path.join(model_path, d))]
            if ck:
                model_path = os.path.join(model_path, max(ck, key=lambda x: int(x.split("-")[1])))
                print(f"[CONFIG] Using DeepSpeed checkpoint: {model_path}")
        assert os.path.isdir(model_path), f"Model path does not exist: {model_path}"
        assert self.kvcache_block_size % 256 == 0
        assert 1 <= self.tensor_parallel_size <= 8
        if self.tensor_parallel_size != 1:
            raise NotImplementedError("tensor parallelism is not part of this path: prompts replicate across GPUs "
                                      "(one process per GPU); use tensor_parallel_size=1")
        self.model_path = model_path
        from .modeling.qwen2 import Qwen2Config
        self.hf_config = Qwen2Config.from_json(os.path.join(model_path, "config.json"))
        self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
        assert self.max_num_batched_tokens >= self.max_model_len
