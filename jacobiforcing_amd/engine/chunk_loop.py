"""The chunk loop both engine decoders run — iterate forward -> step -> commit until every request of the batch has hit EOS,
its token budget or its iteration cap (JD:447-724, JDN:356-667) — with the per-iteration state on the device
(``ops.EngineLoop`` / jf_engine_loop_commit):

* the draft of a block-length group is ONE [B, L] device tensor that the step's next draft replaces; budgets, cached lengths,
  next positions and the committed tokens (a ring per request) are device arrays the commit launch maintains;
* per iteration the host polls ONE small record (tokens committed / EOS / still-active per row + the random-stream cursors) and
  keeps numpy mirrors of the counters the next forward needs; nothing is done per row in Python;
* ``Sequence.token_ids`` / ``num_cached_tokens`` and the block tables are brought up to date

  - once per chunk when the caller gave the decoder a ``forward_step_loop`` (the model runner does: its forward reads the
    loop's device arrays, MR:1134-1418 without the per-sequence work), or
  - before every callback otherwise (``forward_step`` / ``forward_step_batch`` receive request objects and may read them:
    the reference's contract, JD:30-44) — the same launches, plus one token read-back and the reference's per-row calls.

What differs between the greedy and the rejection-sampling decoder is in five hooks (``_first_draft``, ``_push_cursors``,
``_enqueue_step``, ``_pull_cursors``, ``_commit_row``); everything else — grouping by block length in the reference's order
(JD:513), compaction when a request leaves its group, stats — is here.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import ops
from .sequence import Sequence


class ChunkLoopMixin:
    KIND = 0                      # N.EL_KIND_*: which step's records the commit launch reads
    forward_step_loop = None      # optional: callable(loop) -> logits [B, L-1, V] reading the loop's device arrays (+ .finish(loop))

    # ---- hooks -----------------------------------------------------------------------------------------------
    def _first_draft(self, seq: Sequence, L: int) -> List[int]:
        raise NotImplementedError

    def _push_cursors(self, st) -> None:
        raise NotImplementedError

    def _enqueue_step(self, st, lp: "ops.EngineLoop", logits: torch.Tensor, ctx) -> torch.Tensor:
        """Queue the step + commit on ``lp``; returns the device tensor [B, L] holding the rows' committed tokens."""
        raise NotImplementedError

    def _pull_cursors(self, lp: "ops.EngineLoop") -> None:
        raise NotImplementedError

    def _commit_row(self, seq: Sequence, toks: List[int], fallback: bool, L: int) -> None:
        raise NotImplementedError

    def _chunk_context(self, seqs: List[Sequence]):
        return None

    def _no_groups(self, single: bool, iters: np.ndarray) -> None:
        pass

    # ---- helpers ---------------------------------------------------------------------------------------------
    @staticmethod
    def _groups(block_lens: np.ndarray, act: np.ndarray) -> List[Tuple[int, np.ndarray]]:
        """The active requests by block length, largest group first, ties in order of first appearance (JD:513 sorts the
        dict of groups by size; Python's sort is stable)."""
        Ls = block_lens[act]
        if (Ls == Ls[0]).all():
            return [(int(Ls[0]), act)]
        groups: Dict[int, List[int]] = {}
        for i, L in zip(act.tolist(), Ls.tolist()):
            groups.setdefault(L, []).append(i)
        return [(L, np.asarray(ix, dtype=np.int64)) for L, ix in sorted(groups.items(), key=lambda x: len(x[1]), reverse=True)]

    def _budgets(self, seqs: List[Sequence], single: bool) -> np.ndarray:
        out = []
        for seq in seqs:
            sp = getattr(seq, "sampling_params", None)
            if sp is not None:
                rem = getattr(sp, "max_tokens", 2048) - seq.num_completion_tokens
                out.append(rem if single else max(0, rem))
            else:
                out.append(2048)
        return np.asarray(out, dtype=np.int64)

    def _open_loop(self, seqs: List[Sequence], idxs: np.ndarray, L: int, remaining: np.ndarray) -> "ops.EngineLoop":
        sub = [seqs[i] for i in idxs.tolist()]
        self._ensure(len(sub), L)                                  # (the stepper and its random streams, before the host draws pads)
        lp = ops.EngineLoop(self.KIND, L, self.device, [len(s) for s in sub], remaining.tolist())
        lp.seq_idx = idxs.copy()                                   # slot -> index into the chunk's request list
        lp.seqs = sub                                              # slot -> request
        lp.seq_len_h = np.asarray([len(s) for s in sub], dtype=np.int64)   # slot -> len(seq): host mirror of kv_start + 1
        lp.flushed = np.zeros(len(sub), dtype=np.int64)            # slot -> ring tokens already appended to the request
        first = [self._first_draft(s, L) for s in sub]             # (host pads, in row order: JD:142-171 / 332-347, JDN:222-241)
        lp.set_draft(torch.tensor(first, dtype=torch.int64))
        return lp

    def _materialise(self, lp: "ops.EngineLoop", accepted: List[List[int]]) -> None:
        """Once per chunk (``forward_step_loop`` callers): the committed tokens of every request of the group, from the ring."""
        ring, rl = lp.tokens_host()
        for slot, seq in enumerate(lp.seqs):
            a, b = int(lp.flushed[slot]), int(rl[slot])
            if b > a:
                toks = ring[slot, a:b].tolist()
                seq.token_ids += toks
                seq.last_token = toks[-1]
                accepted[int(lp.seq_idx[slot])].extend(toks)
                lp.flushed[slot] = b
            seq.num_cached_tokens = len(seq)                       # (the reference trims the cache back to len(seq) every iteration)
            seq.clear_draft()
        fin = getattr(self.forward_step_loop, "finish", None)
        if fin is not None:
            fin(lp)

    # ---- the loop ----------------------------------------------------------------------------------------------
    def _run_chunk(self, seqs: List[Sequence], single: bool):
        """Returns (accepted token lists, iters [B], forwards [B], n_iter_call)."""
        B = len(seqs)
        cfg = [self._get_sampling_cfg(s) for s in seqs]
        block_lens = np.asarray([c[0] for c in cfg], dtype=np.int64)
        max_iters = np.asarray([c[1] for c in cfg], dtype=np.int64)
        max_tokens = self._budgets(seqs, single)
        n_acc = np.zeros(B, dtype=np.int64)
        iters = np.zeros(B, dtype=np.int64)
        forwards = np.zeros(B, dtype=np.int64)
        eos = np.zeros(B, dtype=bool)
        accepted: List[List[int]] = [[] for _ in range(B)]
        ctx = self._chunk_context(seqs)
        fast = self.forward_step_loop is not None
        loops: Dict[int, ops.EngineLoop] = {}
        n_iter_call = 0
        prof = getattr(self, "profiler", None)        # ModelRunner's PROFILE=1 section timer (reference names, MR:116-134)
        tick = (lambda name, on: (prof.start(name) if on else prof.stop(name))) if prof is not None else (lambda name, on: None)
        cuda = self.device.type == "cuda"
        hk = ops.ENGINE_LOOP_HOOKS or {}              # bench.py: events / host clock around the iteration body
        h_fwd, h_b0, h_b1, h_seen = hk.get("forward_begin"), hk.get("body_begin"), hk.get("body_end"), hk.get("record_seen")
        try:
            while True:
                active = ~eos & (n_acc < max_tokens) & (iters < max_iters)
                if not active.any():
                    break
                act = np.flatnonzero(active & (block_lens > 1))
                if act.size == 0:
                    self._no_groups(single, iters)
                    break
                n_iter_call += 1
                tokens_this_iter = 0
                for L, idxs in self._groups(block_lens, act):
                    lp = loops.get(L)
                    if lp is None:
                        lp = loops[L] = self._open_loop(seqs, idxs, L, (max_tokens - n_acc)[idxs])
                    elif idxs.size != lp.B:                        # a request left the group: gather the arrays once
                        lp.compact(np.flatnonzero(active[lp.seq_idx[lp.members]]))
                    iters[idxs] += 1
                    h_fwd and h_fwd(lp)
                    if fast:
                        logits = self.forward_step_loop(lp)
                    else:
                        sub = [seqs[i] for i in idxs.tolist()]
                        for s in sub:
                            s.draft_tokens = None
                        if single and self.KIND == 0:
                            sub[0].draft_tokens = lp.draft[0].tolist()                          # JD:351
                        logits = self._forward_batched(sub, lp.draft)
                    forwards[idxs] += 1
                    tick("jacobi.verify", True)
                    st = self._ensure(int(idxs.size), L)
                    self._push_cursors(st)
                    h_b0 and h_b0(lp)
                    toks_dev = self._enqueue_step(st, lp, logits, ctx)
                    h_b1 and h_b1(lp)
                    if not fast:                                   # the callbacks read the request objects: their tokens, now
                        th = st.tok_host.view(-1)[:toks_dev.numel()].view(toks_dev.shape)
                        th.copy_(toks_dev, non_blocking=True)
                        if cuda:
                            torch.cuda.current_stream(self.device).synchronize()
                    n, e, _a, fb = lp.wait()
                    h_seen and h_seen(lp)
                    self._pull_cursors(lp)
                    tick("jacobi.verify", False)
                    tick("jacobi.commit", True)
                    n_acc[idxs] += n
                    eos[idxs] |= e.astype(bool)
                    lp.seq_len_h[lp.members] += n
                    tokens_this_iter += int(n.sum())
                    if not fast:
                        rows = th.tolist()
                        nl, fl = n.tolist(), fb.tolist()
                        for row, (i, slot) in enumerate(zip(idxs.tolist(), lp.members.tolist())):
                            toks = rows[row][:nl[row]]
                            self._commit_row(seqs[i], toks, bool(fl[row]), L)
                            accepted[i].extend(toks)
                            lp.flushed[slot] += nl[row]
                    tick("jacobi.commit", False)
                    if prof is not None:
                        prof.iterations += 1
                        prof.tokens += tokens_this_iter
                if not single:
                    self.stats["tokens_per_iteration"].append(tokens_this_iter)
        finally:
            for lp in loops.values():
                try:
                    if fast:
                        self._materialise(lp, accepted)
                finally:
                    lp.close()
        return accepted, iters, forwards, n_iter_call
