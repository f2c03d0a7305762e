"""Request scheduling with the reference's behaviour (inference_engine/engine/scheduler.py:9-97): waiting requests are
admitted (prefill) before running ones continue (decode); Jacobi requests skip the per-step block append because the
decoder sizes block tables itself (SCH:58-61); a Jacobi request finishes when EOS shows up among its new tokens or its
budget is reached (SCH:80-97)."""
from __future__ import annotations

from collections import deque
from typing import List, Tuple

from ..config import Config
from .block_manager import BlockManager
from .sequence import Sequence, SequenceStatus

JACOBI_STRATEGIES = ("jacobi", "jacobi_multiblock_rejection_recycling")


class Scheduler:
    def __init__(self, config: Config):
        self.max_num_seqs, self.max_num_batched_tokens, self.eos = config.max_num_seqs, config.max_num_batched_tokens, config.eos
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size)
        self.waiting: deque = deque()
        self.running: deque = deque()
        self.row_capacity = None
        self.release_row = None     # callable(seq): hands a request's static cache row back (ModelRunner.release), set by the engine

    def set_kv_cache(self, kv_cache) -> None:
        self.block_manager.kv_cache = kv_cache
        # a running request owns one row of the static cache: admission stops when the rows are taken (the reference's
        # limit is its paged block pool, SCH:36-37; the rest of the queue waits exactly as it does there)
        self.row_capacity = int(getattr(kv_cache, "P", 0)) or None

    def add(self, seq: Sequence) -> None:
        self.waiting.append(seq)

    def is_finished(self) -> bool:
        return not (self.waiting or self.running)

    # ---- one scheduling decision: (batch, is_prefill) -----------------------------------------------
    def schedule(self) -> Tuple[List[Sequence], bool]:
        batch = self._admit_waiting()
        if batch:
            return batch, True
        batch = self._continue_running()
        assert batch
        return batch, False

    def _admit_waiting(self) -> List[Sequence]:
        batch, tokens = [], 0
        bm = self.block_manager
        while self.waiting and len(batch) < self.max_num_seqs:
            seq = self.waiting[0]
            if tokens + len(seq) > self.max_num_batched_tokens or not bm.can_allocate(seq):
                break
            if self.row_capacity is not None and len(self.running) >= self.row_capacity:
                break
            bm.allocate(seq)
            tokens += len(seq) - seq.num_cached_tokens
            seq.status = SequenceStatus.RUNNING
            self.running.append(self.waiting.popleft())
            batch.append(seq)
        return batch

    def _continue_running(self) -> List[Sequence]:
        batch = []
        bm = self.block_manager
        while self.running and len(batch) < self.max_num_seqs:
            seq = self.running.popleft()
            evicted_self = False
            while not bm.can_append(seq):                       # make room: newest running request first, then itself
                if self.running:
                    self.preempt(self.running.pop())
                else:
                    self.preempt(seq)
                    evicted_self = True
                    break
            if evicted_self:
                continue
            if getattr(seq, "decode_strategy", None) not in JACOBI_STRATEGIES:
                bm.may_append(seq)
            batch.append(seq)
        self.running.extendleft(reversed(batch))
        return batch

    def preempt(self, seq: Sequence) -> None:
        """Back to the queue (SCH:70-73).  The request also gives its cache row back — it is re-prefilled from scratch when it
        is admitted again — so that the admission limit above counts rows that are really free."""
        seq.status = SequenceStatus.WAITING
        self.block_manager.deallocate(seq)
        if self.release_row is not None:
            self.release_row(seq)
        self.waiting.appendleft(seq)

    def _finish(self, seq: Sequence) -> None:
        seq.status = SequenceStatus.FINISHED
        self.block_manager.deallocate(seq)
        self.running.remove(seq)

    # ---- results of a step ------------------------------------------------------------------------------
    def postprocess(self, seqs, token_ids) -> None:
        for seq, tok in zip(seqs, token_ids):
            seq.append_token(tok)
            hit_eos = not seq.ignore_eos and tok == self.eos
            if hit_eos or seq.num_completion_tokens == seq.max_tokens:
                self._finish(seq)

    def postprocess_jacobi(self, seqs, token_ids_batch) -> None:
        for seq, new_tokens in zip(seqs, token_ids_batch):
            hit_eos = not seq.ignore_eos and self.eos in new_tokens
            if hit_eos or seq.num_completion_tokens >= seq.max_tokens:
                self._finish(seq)
