"""jacobiforcing_amd — MI355X-native Jacobi (fixed-point) parallel decoding hot path.

Drop-in for the reference's ``from inference_engine import LLM, SamplingParams`` on the Jacobi
decode path; the loop body runs in hand-written HIP behind a C ABI (include/jacobiforcing.h).
"""
from .sampling_params import SamplingParams  # noqa: F401

__all__ = ["LLM", "SamplingParams"]


def __getattr__(name):
    if name == "LLM":  # lazy: pulls in torch/transformers
        from .llm import LLM
        return LLM
    raise AttributeError(name)
