"""One decoding request: its token list plus the counters the scheduler, the Jacobi decoders and the runner read or write.
Attribute and method names are those of the reference's ``Sequence`` (inference_engine/engine/sequence.py:14-156) because
the decoders' callback contract is expressed in them (``token_ids``, ``num_cached_tokens``, ``block_table``,
``draft_tokens[_gpu]``, ``extend_tokens`` …)."""
from __future__ import annotations

import itertools
from dataclasses import fields
from enum import Enum

from ..sampling_params import SamplingParams

SequenceStatus = Enum("SequenceStatus", ["WAITING", "RUNNING", "FINISHED"])


def _as_int(tok) -> int:
    return int(tok.item()) if hasattr(tok, "item") else int(tok)


class Sequence:
    block_size = 256
    counter = itertools.count()

    def __init__(self, token_ids: list, sampling_params: SamplingParams = SamplingParams()):
        self.seq_id = next(Sequence.counter)
        self.status = SequenceStatus.WAITING
        self.token_ids = list(token_ids)
        self.num_prompt_tokens = len(self.token_ids)
        self.last_token = self.token_ids[-1]
        # every request knob is mirrored as an attribute (temperature, max_tokens, decode_strategy, jacobi_* …)
        self.sampling_params = sampling_params
        for f in fields(sampling_params):
            setattr(self, f.name, getattr(sampling_params, f.name))
        # KV bookkeeping
        self.num_cached_tokens = 0
        self.block_table: list = []
        self.block_table_gpu = None
        self.block_table_version = 0
        self.num_permanent_spec_blocks = 0
        self.cache_row = -1          # row of the static KV cache this request owns (this engine's physical placement)
        # Jacobi draft state
        self.draft_tokens = None
        self.draft_tokens_gpu = None
        self._prefill_draft = None

    # ---- sizes ----------------------------------------------------------------------------------
    def __len__(self):
        return len(self.token_ids)

    def __getitem__(self, key):
        return self.token_ids[key]

    num_tokens = property(lambda self: len(self.token_ids))
    num_completion_tokens = property(lambda self: len(self.token_ids) - self.num_prompt_tokens)
    prompt_token_ids = property(lambda self: self.token_ids[:self.num_prompt_tokens])
    completion_token_ids = property(lambda self: self.token_ids[self.num_prompt_tokens:])
    is_finished = property(lambda self: self.status == SequenceStatus.FINISHED)
    num_cached_blocks = property(lambda self: self.num_cached_tokens // self.block_size)
    num_blocks = property(lambda self: -(-len(self.token_ids) // self.block_size))
    last_block_num_tokens = property(lambda self: len(self.token_ids) - (self.num_blocks - 1) * self.block_size)

    def block(self, i: int) -> list:
        assert 0 <= i < self.num_blocks
        return self.token_ids[i * self.block_size:(i + 1) * self.block_size]

    # ---- growth ---------------------------------------------------------------------------------
    def append_token(self, token_id) -> None:
        if isinstance(token_id, (list, tuple)):
            raise ValueError(f"append_token expects a single integer, got {type(token_id)} with {len(token_id)} elements")
        self.extend_tokens([token_id])

    def extend_tokens(self, tokens) -> None:
        new = [_as_int(t) for t in tokens]
        if new:
            self.token_ids += new
            self.last_token = new[-1]

    def has_draft(self) -> bool:
        return bool(self.draft_tokens)

    def clear_draft(self) -> None:
        self.draft_tokens = None
