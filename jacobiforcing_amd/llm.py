"""User-facing ``LLM`` — the reference's ``inference_engine.LLM`` (inference_engine/llm.py:12-149): ``generate(prompts,
sampling_params, *, greedy=None, jacobi_*=None)`` folds the convenience keywords into ``SamplingParams``.

Documented deviation: the reference forwards ``jacobi_enabled`` / ``jacobi_num_blocks`` /
``jacobi_ngram_pool_size`` to ``dataclasses.replace`` although ``SamplingParams`` has no such fields, which raises
TypeError; here they map onto ``decode_strategy`` / ``jacobi_max_blocks`` / ``jacobi_n_gram_pool_size``."""
from __future__ import annotations

from dataclasses import replace
from typing import Any, Optional

from .engine.llm_engine import LLMEngine
from .sampling_params import SamplingParams


class LLM(LLMEngine):
    def generate(self, prompts, sampling_params: Optional[SamplingParams] = None, *, greedy: Optional[bool] = None,
                 jacobi_enabled: Optional[bool] = None, jacobi_block_len: Optional[int] = None,
                 jacobi_num_blocks: Optional[int] = None, jacobi_spawn_ratio: Optional[float] = None,
                 jacobi_lookahead_start_ratio: Optional[float] = None, jacobi_ngram_pool_size: Optional[int] = None,
                 **kwargs: Any):
        if sampling_params is None:
            sampling_params = SamplingParams()

        def fold(sp: SamplingParams) -> SamplingParams:
            if greedy is not None:
                if greedy:
                    if getattr(sp, "jacobi_on_policy", False):
                        raise ValueError("Cannot use greedy=True with jacobi_on_policy=True. "
                                         "On-policy learning requires non-greedy decoding (temperature > 0).")
                    sp = replace(sp, temperature=0.0)
                elif getattr(sp, "temperature", 1.0) == 0.0:
                    sp = replace(sp, temperature=1.0)
            upd = {}
            multi = jacobi_num_blocks is not None or jacobi_ngram_pool_size is not None
            knobs = any(v is not None for v in (jacobi_block_len, jacobi_num_blocks, jacobi_spawn_ratio,
                                                jacobi_lookahead_start_ratio, jacobi_ngram_pool_size))
            if jacobi_enabled is False:
                upd["decode_strategy"] = "autoregressive"
            elif jacobi_enabled or knobs:
                if sp.decode_strategy == "autoregressive":
                    upd["decode_strategy"] = "jacobi_multiblock_rejection_recycling" if multi else "jacobi"
            if jacobi_block_len is not None:
                upd["jacobi_block_len"] = jacobi_block_len
            if jacobi_num_blocks is not None:
                upd["jacobi_max_blocks"] = jacobi_num_blocks
            if jacobi_spawn_ratio is not None:
                upd["jacobi_spawn_ratio"] = jacobi_spawn_ratio
            if jacobi_lookahead_start_ratio is not None:
                upd["jacobi_lookahead_start_ratio"] = jacobi_lookahead_start_ratio
            if jacobi_ngram_pool_size is not None:
                upd["jacobi_n_gram_pool_size"] = jacobi_ngram_pool_size
            return replace(sp, **upd) if upd else sp

        sps = [fold(sp) for sp in sampling_params] if isinstance(sampling_params, list) else fold(sampling_params)
        return super().generate(prompts, sps, **kwargs)
