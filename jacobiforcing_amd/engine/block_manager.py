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


class Block:
    def __init__(self, block_id):
        self.block_id = block_id
        self.ref_count = 0
        self.hash = -1
        self.token_ids = []

    def reset(self):
        self.ref_count = 1
        self.hash = -1
        self.token_ids = []


class BlockManager:
    """Free / used block ids and per-sequence block tables, with the reference's Jacobi entry points.

    No prefix cache: blocks are never hashed or shared between sequences, so the hash bookkeeping the reference does on
    append (BM:195-265 un-finalises / re-finalises the last block's hash) has nothing to maintain here and
    ``may_append_batch`` — called by the decoders after every commit, as in the reference — only has to exist.  Block
    tables still grow and shrink exactly as the reference's do (pinned by tests/golden/bm_cases.json)."""

    def __init__(self, num_blocks: int, block_size: int, kv_cache=None):
        self.block_size = block_size
        self.blocks = [Block(i) for i in range(num_blocks)]
        self.free_block_ids = deque(range(num_blocks))
        self.used_block_ids = set()
        self.kv_cache = kv_cache

    def _allocate_block(self, block_id: int) -> Block:
        block = self.blocks[block_id]
        assert block.ref_count == 0
        block.reset()
        self.free_block_ids.remove(block_id)
        self.used_block_ids.add(block_id)
        return block

    _allocate_block_no_clear = _allocate_block        # BM:114-121: nothing to clear in a length-tracked cache

    def _deallocate_block(self, block_id: int) -> None:
        assert self.blocks[block_id].ref_count == 0
        self.used_block_ids.remove(block_id)
        self.free_block_ids.append(block_id)

    def can_allocate(self, seq: Sequence) -> bool:
        return len(self.free_block_ids) >= seq.num_blocks

    def allocate(self, seq: Sequence) -> None:
        assert not seq.block_table
        for _ in range(seq.num_blocks):
            bid = self.free_block_ids[0]
            self._allocate_block(bid)
            seq.block_table.append(bid)

    def deallocate(self, seq: Sequence) -> None:
        for bid in reversed(seq.block_table):
            block = self.blocks[bid]
            block.ref_count -= 1
            if block.ref_count == 0:
                self._deallocate_block(bid)
        seq.num_cached_tokens = 0
        seq.block_table.clear()
        seq.num_permanent_spec_blocks = 0

    def can_append(self, seq: Sequence) -> bool:
        return len(self.free_block_ids) >= (len(seq) % self.block_size == 1)

    def may_append(self, seq: Sequence) -> None:
        """BM:195-265 without the hash cache: a token that opens a new block needs one more block id."""
        if len(seq) % self.block_size == 1 and len(seq.block_table) < seq.num_blocks:
            bid = self.free_block_ids[0]
            self._allocate_block(bid)
            seq.block_table.append(bid)

    def may_append_batch(self, seq: Sequence, num_tokens: int) -> None:
        """BM:267-276: un-finalises the last block's hash; with no prefix cache there is nothing to do."""
        return

    def trim_kv_only_fast(self, seq: Sequence, num_tokens: int) -> None:
        """BM:534-564."""
        if num_tokens <= 0:
            return
        new_num_cached = max(len(seq), seq.num_cached_tokens - num_tokens)
        seq.num_cached_tokens = new_num_cached
        blocks_needed = (new_num_cached + self.block_size - 1) // self.block_size if new_num_cached > 0 else 0
        keep = blocks_needed + getattr(seq, "num_permanent_spec_blocks", 0)
        while len(seq.block_table) > keep:
            bid = seq.block_table.pop()
            block = self.blocks[bid]
            block.ref_count -= 1
            if block.ref_count == 0:
                self._deallocate_block(bid)
