"""Block-table bookkeeping for the engine path — the integers the reference's BlockManager maintains
(inference_engine/engine/block_manager.py): free/used block ids, per-sequence block tables, and the Jacobi helpers
``may_append_batch`` (BM:267-276), ``trim_kv_only_fast`` (BM:534-564) and ``_allocate_block_no_clear`` (BM:114-121).

Physical placement differs by design: this engine keeps each request's K/V contiguous in one row of a static cache
(288 GB of HBM makes paging unnecessary for this path), so block ids are bookkeeping and "trim" moves no data — exactly
the property the reference relies on ("FA respects cache_seqlens", BM:535).  The xxhash prefix cache (BM:66-92) is out
of scope."""
from __future__ import annotations

from collections import deque

from .sequence import Sequence


class BlockManager:
    """Free / used block ids and per-sequence block tables, with the reference's Jacobi entry points.

    No prefix cache: blocks are never hashed or shared between sequences, so the hash bookkeeping the reference does on
    append (BM:195-265 un-finalises / re-finalises the last block's hash) has nothing to maintain here and
    ``may_append_batch`` — called by the decoders after every commit, as in the reference — only has to exist.  Block
    tables still grow and shrink exactly as the reference's do (pinned by tests/golden/bm_cases.json).

    State: ``free_block_ids`` (a FIFO: the reference always takes its head, and callers that grow a table for draft tokens
    do the same, MR:1166-1199), ``used_block_ids``, and one owner count per block id (0 or 1 without sharing)."""

    def __init__(self, num_blocks: int, block_size: int, kv_cache=None):
        self.block_size = int(block_size)
        self.kv_cache = kv_cache
        self.free_block_ids = deque(range(num_blocks))
        self.used_block_ids = set()
        self._owners = [0] * num_blocks

    # ---- one block id in / out of use ----------------------------------------------------------------
    def _take(self, block_id: int) -> int:
        if self._owners[block_id]:
            raise AssertionError(f"block {block_id} is already owned")
        self._owners[block_id] = 1
        self.free_block_ids.remove(block_id)
        self.used_block_ids.add(block_id)
        return block_id

    _allocate_block = _take
    _allocate_block_no_clear = _take                  # BM:114-121: nothing to clear in a length-tracked cache

    def _give_back(self, block_id: int) -> None:
        self._owners[block_id] -= 1
        if self._owners[block_id] == 0:
            self.used_block_ids.discard(block_id)
            self.free_block_ids.append(block_id)

    def _grow(self, seq: Sequence) -> None:
        seq.block_table.append(self._take(self.free_block_ids[0]))

    # ---- sequences -----------------------------------------------------------------------------------------
    def can_allocate(self, seq: Sequence) -> bool:
        return seq.num_blocks <= len(self.free_block_ids)

    def allocate(self, seq: Sequence) -> None:
        if seq.block_table:
            raise AssertionError("sequence already holds blocks")
        for _ in range(seq.num_blocks):
            self._grow(seq)

    def deallocate(self, seq: Sequence) -> None:
        while seq.block_table:                            # last block first, like the reference (BM:172-180)
            self._give_back(seq.block_table.pop())
        seq.num_cached_tokens = 0
        seq.num_permanent_spec_blocks = 0

    def can_append(self, seq: Sequence) -> bool:
        opens_block = len(seq) % self.block_size == 1
        return len(self.free_block_ids) >= int(opens_block)

    def may_append(self, seq: Sequence) -> None:
        """BM:195-265 without the hash cache: a token that opens a new block needs one more block id."""
        if len(seq) % self.block_size == 1 and len(seq.block_table) < seq.num_blocks:
            self._grow(seq)

    def may_append_batch(self, seq: Sequence, num_tokens: int) -> None:
        """BM:267-276: un-finalises the last block's hash; with no prefix cache there is nothing to do."""
        return

    def trim_kv_only_fast(self, seq: Sequence, num_tokens: int) -> None:
        """BM:534-564: forget the last ``num_tokens`` cached positions (never below the committed tokens) and hand back the
        blocks behind the new end, except the ones the forward keeps for draft tokens (``num_permanent_spec_blocks``).
        Pure bookkeeping: attention honours the cached length, no K/V moves."""
        if num_tokens <= 0:
            return
        cached = max(len(seq), seq.num_cached_tokens - num_tokens)
        seq.num_cached_tokens = cached
        keep = -(-cached // self.block_size) + getattr(seq, "num_permanent_spec_blocks", 0)
        while len(seq.block_table) > keep:
            self._give_back(seq.block_table.pop())
