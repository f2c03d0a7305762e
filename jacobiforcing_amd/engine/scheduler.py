"""Request scheduling — the behaviour of the reference's ``Scheduler`` (inference_engine/engine/scheduler.py:9-97):
prefill batches first, then decode batches; Jacobi requests skip ``may_append`` (the decoder sizes block tables itself,
SCH:58-61); ``postprocess_jacobi`` finishes a request on EOS-in-new-tokens or max_tokens (SCH:80-97)."""
from __future__ import annotations

from collections import deque

from ..config import Config
from .block_manager import BlockManager
from .sequence import Sequence, SequenceStatus

JACOBI_STRATEGIES = ("jacobi", "jacobi_multiblock_rejection_recycling")


class Scheduler:
    def __init__(self, config: Config):
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.eos = config.eos
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size)
        self.waiting: deque = deque()
        self.running: deque = deque()

    def set_kv_cache(self, kv_cache):
        self.block_manager.kv_cache = kv_cache

    def is_finished(self):
        return not self.waiting and not self.running

    def add(self, seq: Sequence):
        self.waiting.append(seq)

    def schedule(self):
        scheduled, num_seqs, num_batched = [], 0, 0
        while self.waiting and num_seqs < self.max_num_seqs:
            seq = self.waiting[0]
            if num_batched + len(seq) > self.max_num_batched_tokens or not self.block_manager.can_allocate(seq):
                break
            num_seqs += 1
            self.block_manager.allocate(seq)
            num_batched += len(seq) - seq.num_cached_tokens
            seq.status = SequenceStatus.RUNNING
            self.waiting.popleft()
            self.running.append(seq)
            scheduled.append(seq)
        if scheduled:
            return scheduled, True
        while self.running and num_seqs < self.max_num_seqs:
            seq = self.running.popleft()
            while not self.block_manager.can_append(seq):
                if self.running:
                    self.preempt(self.running.pop())
                else:
                    self.preempt(seq)
                    break
            else:
                num_seqs += 1
                if getattr(seq, "decode_strategy", None) not in JACOBI_STRATEGIES:
                    self.block_manager.may_append(seq)
                scheduled.append(seq)
        assert scheduled
        self.running.extendleft(reversed(scheduled))
        return scheduled, False

    def preempt(self, seq: Sequence):
        seq.status = SequenceStatus.WAITING
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)

    def _finish(self, seq: Sequence):
        seq.status = SequenceStatus.FINISHED
        self.block_manager.deallocate(seq)
        self.running.remove(seq)

    def postprocess(self, seqs, token_ids):
        for seq, token_id in zip(seqs, token_ids):
            seq.append_token(token_id)
            if (not seq.ignore_eos and token_id == self.eos) or seq.num_completion_tokens == seq.max_tokens:
                self._finish(seq)

    def postprocess_jacobi(self, seqs, token_ids_batch) -> None:
        for seq, token_ids in zip(seqs, token_ids_batch):
            if not seq.ignore_eos and self.eos in token_ids:
                self._finish(seq)
            elif seq.num_completion_tokens >= seq.max_tokens:
                self._finish(seq)
