"""Request knobs — same field names, defaults and validation as the reference's
``inference_engine/sampling_params.py:4-38`` so existing call sites keep working."""
from dataclasses import dataclass


@dataclass
class SamplingParams:
    temperature: float = 1.0
    max_tokens: int = 64
    ignore_eos: bool = False

    # "autoregressive", "jacobi", or "jacobi_multiblock_rejection_recycling" (the name the reference
    # reserved at sampling_params.py:10 and rejects at model_runner.py:1468-1473; implemented here)
    decode_strategy: str = "autoregressive"

    jacobi_block_len: int = 64
    jacobi_max_iterations: int = 128

    jacobi_max_blocks: int = 2
    jacobi_spawn_ratio: float = 0.85
    jacobi_lookahead_start_ratio: float = 0.0
    jacobi_n_gram_pool_size: int = 4

    jacobi_on_policy: bool = False

    def __post_init__(self):
        assert self.temperature >= 0.0, "temperature must be non-negative"
        if self.jacobi_on_policy and self.temperature == 0.0:
            raise ValueError(
                "jacobi_on_policy=True requires temperature > 0 (non-greedy decoding). "
                "On-policy learning is only supported with non-greedy Jacobi decoding.")

    @property
    def use_jacobi(self) -> bool:
        return self.jacobi_block_len is not None
