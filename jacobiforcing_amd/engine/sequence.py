"""Request state — the attributes of the reference's ``Sequence`` (inference_engine/engine/sequence.py:14-156) that the
scheduler, the decoders and the runner read or write, with the same names."""
from __future__ import annotations

from copy import copy
from enum import Enum, auto
from itertools import count

from ..sampling_params import SamplingParams


class SequenceStatus(Enum):
    WAITING = auto()
    RUNNING = auto()
    FINISHED = auto()


class Sequence:
    block_size = 256
    counter = count()

    def __init__(self, token_ids: list, sampling_params: SamplingParams = SamplingParams()):
        self.seq_id = next(Sequence.counter)
        self.status = SequenceStatus.WAITING
        self.token_ids = copy(token_ids)
        self.last_token = token_ids[-1]
        self.num_tokens = len(self.token_ids)
        self.num_prompt_tokens = len(token_ids)
        self.num_cached_tokens = 0
        self.block_table = []
        self.temperature = sampling_params.temperature
        self.max_tokens = sampling_params.max_tokens
        self.ignore_eos = sampling_params.ignore_eos
        self.decode_strategy = sampling_params.decode_strategy
        self.jacobi_block_len = sampling_params.jacobi_block_len
        self.jacobi_max_blocks = sampling_params.jacobi_max_blocks
        self.jacobi_spawn_ratio = sampling_params.jacobi_spawn_ratio
        self.jacobi_lookahead_start_ratio = sampling_params.jacobi_lookahead_start_ratio
        self.jacobi_n_gram_pool_size = sampling_params.jacobi_n_gram_pool_size
        self.jacobi_max_iterations = sampling_params.jacobi_max_iterations
        self.jacobi_on_policy = sampling_params.jacobi_on_policy
        self.sampling_params = sampling_params
        self.draft_tokens = None
        self.draft_tokens_gpu = None
        self._prefill_draft = None
        self.num_permanent_spec_blocks = 0
        self.block_table_gpu = None
        self.block_table_version = 0
        self.cache_row = -1          # row of the static KV cache this request owns (this engine's "physical" placement)

    def __len__(self):
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    @property
    def is_finished(self):
        return self.status == SequenceStatus.FINISHED

    @property
    def num_completion_tokens(self):
        return self.num_tokens - self.num_prompt_tokens

    @property
    def prompt_token_ids(self):
        return self.token_ids[:self.num_prompt_tokens]

    @property
    def completion_token_ids(self):
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_cached_blocks(self):
        return self.num_cached_tokens // self.block_size

    @property
    def num_blocks(self):
        return (self.num_tokens + self.block_size - 1) // self.block_size

    @property
    def last_block_num_tokens(self):
        return self.num_tokens - (self.num_blocks - 1) * self.block_size

    def block(self, i):
        assert 0 <= i < self.num_blocks
        return self.token_ids[i * self.block_size:(i + 1) * self.block_size]

    def append_token(self, token_id):
        if isinstance(token_id, (list, tuple)):
            raise ValueError(f"append_token expects a single integer, got {type(token_id)} with {len(token_id)} elements")
        token_id = int(token_id.item()) if hasattr(token_id, "item") else int(token_id)
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1

    def extend_tokens(self, tokens):
        if not tokens:
            return
        toks = [int(t.item()) if hasattr(t, "item") else int(t) for t in tokens]
        self.token_ids.extend(toks)
        self.last_token = toks[-1]
        self.num_tokens += len(toks)

    def has_draft(self) -> bool:
        return self.draft_tokens is not None and len(self.draft_tokens) > 0

    def clear_draft(self):
        self.draft_tokens = None
