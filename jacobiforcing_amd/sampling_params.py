"""Per-request knobs.  The attribute names, their order, defaults and the two validation rules are the reference's
request API (``inference_engine/sampling_params.py:4-38``), so ``SamplingParams(temperature=0.0, max_tokens=…,
decode_strategy="jacobi", jacobi_block_len=…)`` call sites, ``dataclasses.replace`` and ``dataclasses.fields`` keep working.
The class is generated from the table below, which is also where each knob is documented."""
from __future__ import annotations

from dataclasses import field, make_dataclass

DECODE_STRATEGIES = ("autoregressive", "jacobi", "jacobi_multiblock_rejection_recycling")

# (name, type, default, meaning)
_KNOBS = [
    ("temperature", float, 1.0, "0 = greedy; > 0 = sampling (Jacobi requests then use rejection-sampling verification)"),
    ("max_tokens", int, 64, "completion budget"),
    ("ignore_eos", bool, False, "keep decoding past EOS"),
    ("decode_strategy", str, "autoregressive",
     "one of DECODE_STRATEGIES; the multiblock name is reserved but rejected by the reference (model_runner.py:1468-1473) "
     "and implemented here"),
    ("jacobi_block_len", int, 64, "n: tokens per Jacobi block (n_token_seq_len of the HF functions)"),
    ("jacobi_max_iterations", int, 128, "iteration cap per call (on-policy: cap on the number of blocks)"),
    ("jacobi_max_blocks", int, 2, "K: blocks in flight (multiblock)"),
    ("jacobi_spawn_ratio", float, 0.85, "r: accepted fraction of the newest block that spawns the next one"),
    ("jacobi_lookahead_start_ratio", float, 0.0, "accepted fraction from which recycled candidates are tried"),
    ("jacobi_n_gram_pool_size", int, 4, "entries of the rejection-recycling pool"),
    ("jacobi_on_policy", bool, False, "return rollout records instead of tokens (needs temperature > 0)"),
]


def _validate(self) -> None:
    assert self.temperature >= 0.0, "temperature must be non-negative"
    if self.jacobi_on_policy and self.temperature == 0.0:
        raise ValueError("jacobi_on_policy=True requires temperature > 0 (non-greedy decoding). "
                         "On-policy learning is only supported with non-greedy Jacobi decoding.")


SamplingParams = make_dataclass(
    "SamplingParams", [(name, typ, field(default=default)) for name, typ, default, _ in _KNOBS],
    namespace={"__post_init__": _validate, "use_jacobi": property(lambda self: self.jacobi_block_len is not None),
               "__doc__": "Request knobs:\n" + "\n".join(f"  {n} ({t.__name__}, default {d!r}): {doc}" for n, t, d, doc in _KNOBS)})
SamplingParams.__module__ = __name__
