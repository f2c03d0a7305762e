"""Engine configuration with the reference's field names (``inference_engine/config.py:8-28``) and checks
(``kvcache_block_size % 256 == 0``, ``1 <= tensor_parallel_size <= 8``, newest ``checkpoint-N`` sub-directory wins).

Differences: tensor parallelism is not implemented on this path (prompts replicate over GPUs instead,
SURVEY §8e), so ``tensor_parallel_size`` must be 1; the model description is this package's ``Qwen2Config`` read from
``config.json`` (no transformers dependency); a directory without ``*.safetensors`` gets random-init weights."""
from __future__ import annotations

import re
from dataclasses import field, make_dataclass
from pathlib import Path
from typing import Any, Optional

_FIELDS = [
    # scheduling / memory
    ("max_num_batched_tokens", int, 16384), ("max_num_seqs", int, 512), ("max_model_len", int, 8192),
    ("gpu_memory_utilization", float, 0.9), ("tensor_parallel_size", int, 1), ("enforce_eager", bool, False),
    ("hf_config", Optional[Any], None), ("eos", int, -1), ("pad", int, -1),
    ("kvcache_block_size", int, 256), ("num_kvcache_blocks", int, -1),
    # "contiguous": one static cache row per request (this package's layout); "paged": the reference's pool of blocks addressed
    # through block tables and slot mappings (layers/attention.py:10-40, MR:1204-1265) — same decodes, for callers that want its memory model
    ("kv_cache_layout", str, "contiguous"),
    # Jacobi defaults of the engine (requests override them through SamplingParams)
    ("jacobi_enabled", bool, True), ("jacobi_block_len", int, 64), ("jacobi_max_blocks", int, 2),
    ("jacobi_spawn_ratio", float, 0.8), ("jacobi_lookahead_start_ratio", float, 0.0),
    ("jacobi_n_gram_pool_size", int, 4), ("jacobi_max_iterations", int, 128),
]


def _resolve_model_dir(model: str) -> Path:
    """A training output directory holds ``checkpoint-<step>`` folders: serve the newest one."""
    root = Path(model)
    if root.is_dir():
        steps = [(int(m.group(1)), d) for d in root.iterdir() if d.is_dir() and (m := re.fullmatch(r"checkpoint-(\d+).*", d.name))]
        if steps:
            root = max(steps)[1]
            print(f"[CONFIG] Using DeepSpeed checkpoint: {root}")
    assert root.is_dir(), f"Model path does not exist: {root}"
    return root


def _finish_init(self) -> None:
    model_dir = _resolve_model_dir(self.model)
    assert self.kvcache_block_size % 256 == 0
    assert 1 <= self.tensor_parallel_size <= 8
    if self.kv_cache_layout not in ("contiguous", "paged"):
        raise ValueError(f"kv_cache_layout must be 'contiguous' or 'paged', got {self.kv_cache_layout!r}")
    if self.tensor_parallel_size != 1:
        raise NotImplementedError("tensor parallelism is not part of this path: prompts replicate across GPUs "
                                  "(one process per GPU); use tensor_parallel_size=1")
    self.model_path = str(model_dir)
    from .modeling.qwen2 import Qwen2Config
    self.hf_config = Qwen2Config.from_json(model_dir / "config.json")
    self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
    assert self.max_num_batched_tokens >= self.max_model_len


Config = make_dataclass("Config", [("model", str)] + [(n, t, field(default=d)) for n, t, d in _FIELDS],
                        namespace={"__post_init__": _finish_init})
Config.__module__ = __name__
