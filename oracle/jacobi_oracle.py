"""CPU oracle: a plain-Python/numpy restatement of the reference's Jacobi decoding loop.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package ``jacobiforcing_amd`` never
does (its HIP path fails loudly when the extension is missing).

Parity status: PINNED.  Every function below is checked bit-for-bit against golden vectors
produced by running the unmodified reference in the build container
(``tests/golden/gen_golden.py`` -> ``tests/golden/*.json``; checked by
``tests/test_oracle_golden.py``).  The reference ships no golden vectors of its own
(SURVEY.md §4).

Citations use the SURVEY abbreviations (paths relative to the reference root):
  MB  = modeling/cllm2_qwen2_modeling_kv_terminate_on_eos_improved_multiblock_lookahead_unified.py
  SB  = modeling/cllm2_qwen2_modeling_kv_terminate_on_eos_improved.py
  JD  = inference_engine/engine/jacobi_decoding.py
  JDN = inference_engine/engine/jacobi_decoding_nongreedy.py
  MR  = inference_engine/engine/model_runner.py
  BM  = inference_engine/engine/block_manager.py
"""
from __future__ import annotations

import math
from collections import deque
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

Rows = List[List[int]]


# ======================================================================================
# a2 — block-local argmax with torch.argmax semantics (MB:476, SB:197, JD:357/567)
# ======================================================================================
def argmax_rows(logits: np.ndarray) -> np.ndarray:
    """Row-wise argmax of a [R, V] float array with torch semantics: the first index of
    the maximum; NaN compares greater than everything (first NaN wins); -0.0 == +0.0."""
    x = np.asarray(logits)
    if x.dtype != np.float32 and x.dtype != np.float64:
        x = x.astype(np.float32)
    R = x.shape[0]
    out = np.empty((R,), dtype=np.int64)
    nan = np.isnan(x)
    has_nan = nan.any(axis=1)
    with np.errstate(invalid="ignore"):
        am = np.argmax(np.where(nan, -np.inf, x), axis=1)
    out[:] = am
    if has_nan.any():
        out[has_nan] = np.argmax(nan[has_nan], axis=1)
    return out


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    """uint16 bf16 payloads -> float32 values."""
    b = np.asarray(bits).astype(np.uint32) << np.uint32(16)
    return b.view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 payload, round-to-nearest-even (torch .to(bfloat16))."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = np.isnan(np.asarray(x, dtype=np.float32))
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)).astype(np.uint16)
    r[nan] = np.uint16(0x7FC0)
    return r


# ======================================================================================
# a3 — token equality + accepted-prefix scan (MB:482-486, SB:199-200, JD:253-293)
# ======================================================================================
def accept_lengths(draft: Rows, greedy: Rows) -> List[int]:
    """accepted[b] = 1 + number of leading positions i with draft[b][i+1] == greedy[b][i].
    ``draft`` may have one row broadcast against B greedy rows (MB:482)."""
    B = max(len(draft), len(greedy))
    if len(draft) not in (1, B) or len(greedy) not in (1, B):
        raise RuntimeError(f"The size of tensor a ({len(draft)}) must match the size of tensor b ({len(greedy)}) at non-singleton dimension 0")
    acc = []
    for b in range(B):
        d = draft[b if len(draft) > 1 else 0]
        g = greedy[b if len(greedy) > 1 else 0]
        k = 0
        L = len(d)
        while k < L - 1 and d[k + 1] == g[k]:
            k += 1
        acc.append(k + 1)
    return acc


def first_max_index(vals: Sequence[int]) -> int:
    """torch.argmax on an int vector: first index of the maximum (MB:489)."""
    best, bi = None, 0
    for i, v in enumerate(vals):
        if best is None or v > best:
            best, bi = v, i
    return bi


# ======================================================================================
# a8 — rejection recycling candidate build (MB:62-91)
# ======================================================================================
def build_candidates(pool: Sequence[List[int]], token_val: int, out_row: List[int]) -> Rows:
    cands: Rows = []
    L_out = len(out_row)
    for seq in reversed(list(pool)[:-1]):          # newest entry skipped (MB:74)
        pos = -1
        for i, t in enumerate(seq):
            if t == token_val:
                pos = i
                break
        if pos < 0:
            continue
        cand = list(seq[pos:])
        if len(cand) > L_out:
            cand = cand[:L_out]
        elif len(cand) < L_out:
            cand = cand + out_row[len(cand):L_out]  # pad with the current draft's tail (MB:84-86)
        cands.append(cand)
    return cands


# ======================================================================================
# a1 — multiblock Jacobi generation call as an explicit state machine (MB:227-740)
# ======================================================================================
class MultiblockOracle:
    """State machine of one ``jacobi_forward_greedy_multiblock`` generation call.

    Usage::

        st = MultiblockOracle(input_ids, kv_tokens, n=..., K=..., ...)
        while (step := st.begin_iteration()) is not None:
            out_rows, spans = step
            greedy = forward(st.kv_rows_before_forward, out_rows)   # [B][T]
            st.end_iteration(greedy)
            if st.done: break
        st.finalize()
        st.ret, st.next_token, st.iters, st.kv_tokens
    """

    def __init__(self, input_ids: List[int], kv_tokens: List[int], *, n: int, K: int = 2, r: float = 0.85,
                 lookahead_start_ratio: float = 0.0, n_gram_pool_size: int = 4,
                 eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None,
                 max_iteration_count: int = 128):
        self.n, self.K, self.r = int(n), int(K), r
        self.lookahead = lookahead_start_ratio
        self.eos_id, self.pad_id = eos_token_id, pad_token_id
        self.eos_enabled = eos_token_id is not None
        self.max_iter = int(max_iteration_count)
        self.pool: deque = deque(maxlen=n_gram_pool_size)            # MB:238
        # block state (MB:249-257)
        self.out_acc: Rows = [[]]
        self.q_draft: List[Rows] = [[list(input_ids)]]
        self.need_reverify: List[bool] = [False]
        self.total_acc: List[int] = [0]
        self.num_blocks = 1
        self.active_blocks = 1
        self.RA = 0
        self.last_next_token: Optional[int] = None                    # MB:259 (empty tensor)
        self.kv_rows: Rows = [list(kv_tokens)]                        # cache rows (token content)
        self.prompt_len = len(kv_tokens)                              # MB:261
        self.spawn_threshold = math.ceil(r * n)                       # MB:262
        self.iters = 0
        self.done = False
        self.returned_early = False
        self.ret: List[int] = []
        self.next_token: Optional[int] = None
        self.banners: List[str] = []
        self._spans: List[Tuple[int, int, int]] = []
        self._out: Rows = []

    # -- helpers -------------------------------------------------------------------
    def kv_len(self) -> int:
        return len(self.kv_rows[0])

    def _kv_trim(self, num_false: int) -> None:                        # MB:36-59
        if num_false <= 0:
            return
        new_len = max(0, self.kv_len() - num_false)
        self.kv_rows = [row[:new_len] for row in self.kv_rows]

    def _kv_narrow(self, best_idx: int) -> None:                       # MB:500-502
        self.kv_rows = [self.kv_rows[best_idx]]

    def _kv_resize(self, new_B: int) -> None:                          # MB:93-127
        cur = len(self.kv_rows)
        if cur == new_B:
            return
        if new_B > cur:
            if cur == 1:
                self.kv_rows = [list(self.kv_rows[0]) for _ in range(new_B)]
            else:
                reps = (new_B + cur - 1) // cur
                self.kv_rows = [list(r) for r in (self.kv_rows * reps)[:new_B]]
        else:
            self.kv_rows = self.kv_rows[:new_B]

    def _committed_len(self, cur_RA: int) -> int:                      # MB:264-271
        c = self.prompt_len
        for b in range(self.num_blocks):
            if b != cur_RA and not self.need_reverify[b]:
                c += len(self.out_acc[b])
        return c + len(self.out_acc[cur_RA])

    @staticmethod
    def _ensure_batch(rows: Rows, B: int) -> Rows:                     # MB:288-312
        if not rows or len(rows[0]) == 0:
            return [[] for _ in range(B)]
        cur = len(rows)
        if cur == B:
            return rows
        if cur == 1:
            return [list(rows[0]) for _ in range(B)]
        reps = (B + cur - 1) // cur
        return [list(r) for r in (rows * reps)[:B]]

    def _concat_all_blocks_seq(self) -> List[int]:                     # MB:387-411
        seq: List[int] = []
        for bb in range(self.num_blocks):
            seq += self.out_acc[bb]
            q = self.q_draft[bb]
            if q and len(q[0]) > 0:
                seq += q[0]
        if self.pad_id is not None:
            seq = [t for t in seq if t != self.pad_id]
        return seq

    def _final_ret(self) -> List[int]:                                 # MB:533-537 / 602-606 / 725-730
        ret: List[int] = []
        for bb in range(self.num_blocks):
            if bb != self.RA and not self.need_reverify[bb] and len(self.out_acc[bb]) > 0:
                ret += self.out_acc[bb]
        return ret + self.out_acc[self.RA]

    def _return(self, next_token: int) -> None:
        self.ret = self._final_ret()
        td = self.kv_len() - (self.prompt_len + len(self.ret))         # MB:540-543
        if td > 0:
            self._kv_trim(td)
        self.next_token = next_token
        self.done = True
        self.returned_early = True

    # -- iteration -------------------------------------------------------------------
    def begin_iteration(self):
        """MB:414-422.  Returns (out rows [B][T], spans) or None when the loop ends."""
        if self.done or self.iters >= self.max_iter:
            return None
        self.iters += 1
        RA = self.RA
        ra_rows = self.q_draft[RA]
        B_ra = len(ra_rows)
        pieces: List[Rows] = []
        spans: List[Tuple[int, int, int]] = []
        cursor = 1
        L_ra = len(ra_rows[0]) if ra_rows else 0
        if L_ra > 0:
            pieces.append(ra_rows)
            spans.append((RA, cursor, L_ra))
            cursor += L_ra
        for b in range(self.num_blocks):
            if b == RA or not self.need_reverify[b]:
                continue
            L_acc = len(self.out_acc[b])
            if L_acc > 0:
                pieces.append(self._ensure_batch([self.out_acc[b]], B_ra))
                cursor += L_acc
            q = self.q_draft[b]
            L_tail = len(q[0]) if q else 0
            if L_tail > 0:
                pieces.append(self._ensure_batch(q, B_ra))
                spans.append((b, cursor, L_tail))
                cursor += L_tail
        out: Rows = [[] for _ in range(B_ra)]
        for p in pieces:
            for i in range(B_ra):
                out[i] = out[i] + list(p[i])
        if B_ra == 0 or len(out[0]) == 0:
            self.iters_break_empty = True
            return None                                                # MB:418-419 (iters already counted)
        self._kv_resize(len(out))                                      # MB:421-422
        self._out, self._spans = out, spans
        return out, spans

    def end_iteration(self, greedy_all: Rows) -> None:
        """MB:428-721 after the forward: ``greedy_all[b][t]`` = argmax of logits[b, t]."""
        n, RA = self.n, self.RA
        out = self._out
        B = len(out)
        assert len(greedy_all) == B and all(len(g) == len(out[0]) for g in greedy_all)
        # the forward appended every row's tokens to its cache row (DynamicCache.update)
        self.kv_rows = [self.kv_rows[b] + list(out[b]) for b in range(B)]

        for (b, start, L) in self._spans:
            greedy = [g[start - 1:start - 1 + L] for g in greedy_all]                   # MB:473-476
            draft = self.q_draft[b]
            accepted = accept_lengths(draft, greedy)                                     # MB:482-486
            best_idx = first_max_index(accepted) if b == self.RA else 0                  # MB:487-491
            acc_len_raw = accepted[best_idx]
            draft_row = list(draft[best_idx])                                            # MB:496 (IndexError if rows<=best)
            g = list(greedy[best_idx])
            self._kv_narrow(best_idx)                                                    # MB:500-502
            L_eff = len(draft_row)
            if L_eff == 0:
                continue
            acc_len = acc_len_raw
            eos_reached = False
            if self.eos_enabled and b == self.RA and acc_len > 0:                        # MB:513-521
                for i in range(acc_len):
                    if draft_row[i] == self.eos_id:
                        acc_len = i + 1
                        eos_reached = True
                        break
            has_rejected = acc_len < L_eff
            if acc_len > 0:                                                              # MB:526-528
                self.out_acc[b] = self.out_acc[b] + draft_row[:acc_len]
                self.total_acc[b] += acc_len
            if eos_reached and b == self.RA:                                             # MB:531-547
                self._return(draft_row[acc_len - 1])
                return
            if has_rejected:                                                             # MB:550-588
                nxt = g[max(acc_len - 1, 0)]
                self.q_draft[b] = [[nxt] + g[acc_len:-1]]
                if b == self.RA:
                    concat_seq = self._concat_all_blocks_seq()
                    if len(concat_seq) > 0:
                        self.pool.append(concat_seq)
                    tail = g[acc_len:-1]
                    if len(tail) > 0:
                        self.pool.append(list(tail))
                    if self.total_acc[b] / n >= self.lookahead:                          # MB:577 (float division)
                        cands = build_candidates(self.pool, nxt, self.q_draft[b][0])
                        if len(cands) > 1:                                               # MB:579 (Q5)
                            self.q_draft[b] = [self.q_draft[b][0]] + cands
                            self._kv_resize(len(self.q_draft[b]))                        # MB:585
            else:                                                                        # MB:590-593
                self.q_draft[b] = [[]]
                nxt = g[-1]
            if b == self.RA:
                self.last_next_token = nxt
            if self.eos_enabled and b == self.RA and self.last_next_token == self.eos_id:  # MB:599-614
                self.out_acc[b] = self.out_acc[b] + [self.last_next_token]
                self._return(self.last_next_token)
                return

        # MB:617-626 keep exactly the committed length
        self._kv_trim(self.kv_len() - self._committed_len(self.RA))

        # MB:629-653 spawn
        newest = self.num_blocks - 1
        if self.total_acc[newest] >= self.spawn_threshold and self.active_blocks < self.K:
            if self.pad_id is None:
                raise ValueError("pad_token_id must be provided when spawning pseudo-active blocks.")
            self.banners.append("spawn")
            ra_rows = self.q_draft[self.RA]
            L_ra = len(ra_rows[0]) if ra_rows else 0
            pad_len = max(0, n - L_ra)
            self.q_draft.append([list(row) + [self.pad_id] * pad_len for row in ra_rows])
            self.out_acc.append([])
            self.total_acc.append(0)
            self.need_reverify.append(True)
            self.num_blocks += 1
            self.active_blocks += 1

        # MB:656-716 promote
        if self.total_acc[self.RA] >= n:
            for b in range(self.num_blocks):
                if self.need_reverify[b] and self.total_acc[b] > 0:
                    self.banners.append("switch")
                    acc_pref = self.out_acc[b]
                    tail_rows = self.q_draft[b]
                    if len(acc_pref) > 0 and len(tail_rows) != 1:
                        raise RuntimeError("cat of [1,a] with multi-row tail")            # torch.cat would raise
                    q_full = acc_pref + list(tail_rows[0]) if len(acc_pref) > 0 else list(tail_rows[0])
                    assert len(q_full) == n, f"draft size mismatch: draft at {len(q_full)} vs. n_token_seq_len {n}"
                    self.out_acc[b] = []
                    self.total_acc[b] = 0
                    lnt = [] if self.last_next_token is None else [self.last_next_token]
                    self.q_draft[b] = [lnt + q_full[1:]]
                    self.need_reverify[b] = False
                    self.RA = b
                    self._kv_trim(self.kv_len() - self._committed_len(self.RA))
                    self.active_blocks -= 1
                    self.num_blocks -= 1                                                 # Q3: lists keep their entries
                    break
                self.active_blocks -= 1                                                  # Q4

        # MB:719-721 early stop
        if all(self.total_acc[b] >= n for b in range(self.num_blocks)):
            self.banners.append("early_stop")
            self.done = True

    def finalize(self) -> None:
        """MB:723-740 (only when the loop was left without an in-loop return)."""
        if self.returned_early:
            return
        ret: List[int] = []
        for b in range(self.num_blocks):
            if b != self.RA and not self.need_reverify[b] and len(self.out_acc[b]) > 0:
                ret += self.out_acc[b]
        if len(self.out_acc[self.RA]) > 0:
            ret += self.out_acc[self.RA]
        self.ret = ret
        td = self.kv_len() - (self.prompt_len + len(ret))
        if td > 0:
            self._kv_trim(td)
        self.next_token = self.last_next_token
        self.done = True

    @property
    def kv_tokens(self) -> List[int]:
        return self.kv_rows[0]


ForwardFn = Callable[[Rows, Rows], Rows]   # (kv_rows [B][S], out_rows [B][T]) -> greedy [B][T]


def mb_generation_call(forward: ForwardFn, input_ids: List[int], kv_tokens: List[int], **kw):
    """One generation call; returns the state object (ret, next_token, iters, kv_tokens)."""
    st = MultiblockOracle(input_ids, kv_tokens, **kw)
    trace = []
    while True:
        step = st.begin_iteration()
        if step is None:
            break
        out, spans = step
        kv_before = [list(r) for r in st.kv_rows]
        greedy = forward(kv_before, out)
        trace.append(dict(kv_len=len(kv_before[0]), out=[list(r) for r in out], greedy=[list(g) for g in greedy],
                          spans=list(spans)))
        try:
            st.end_iteration(greedy)
        except RuntimeError as e:          # MB:482 broadcast failure (the reference dies too): keep the forwards made so far
            e.trace = trace
            raise
        if st.done:
            break
    st.finalize()
    st.trace = trace
    return st


def mb_prefill(forward: ForwardFn, prompt: List[int], draft: List[int]):
    """MB:175-225: forward prompt ⧺ draft, n-gram = argmax(logits[:, -n-1:-1]), KV cut back to the prompt."""
    n = len(draft)
    greedy = forward([[]], [list(prompt) + list(draft)])[0]
    ngram = greedy[-n - 1:-1] if n > 0 else []
    return ngram, list(prompt)


# ======================================================================================
# a14 — HF single-block Jacobi generation call (SB:140-276)
# ======================================================================================
def sb_generation_call(forward: ForwardFn, input_ids: List[int], kv_tokens: List[int], *, n: int,
                       eos_token_id: Optional[int] = None):
    eos_enabled = eos_token_id is not None
    kv = list(kv_tokens)
    out = list(input_ids)
    accepted_n_gram = list(input_ids)          # aliases the input buffer (SB:145); writes past the end are dropped
    total_accepted = 0
    itr = 0
    next_token = None
    trace = []

    def write(pos, toks):
        for j, t in enumerate(toks):
            if pos + j < len(accepted_n_gram):
                accepted_n_gram[pos + j] = t

    while total_accepted < n:
        itr += 1
        greedy_all = forward([kv], [out])[0]
        trace.append(dict(kv_len=len(kv), out=list(out), greedy=list(greedy_all)))
        kv = kv + out
        L = len(out)
        greedy = greedy_all[:-1]                                                        # SB:197
        k = 0
        while k < L - 1 and out[k + 1] == greedy[k]:
            k += 1
        num_accepted_raw = k + 1                                                        # SB:199-202
        num_accepted = num_accepted_raw
        if eos_enabled:
            for i in range(num_accepted_raw):
                if out[i] == eos_token_id:
                    num_accepted = i + 1
                    break
        if num_accepted > 0:
            write(total_accepted, out[:num_accepted])                                   # SB:215
        total_accepted += num_accepted
        if eos_enabled and eos_token_id in out[:num_accepted]:                          # SB:219-227
            to_delete = max(0, len(kv) - total_accepted)
            if to_delete > 0:
                kv = kv[:len(kv) - to_delete]
            return dict(ret=accepted_n_gram[:total_accepted], next_token=eos_token_id, iters=itr, kv_tokens=kv,
                        trace=trace)
        if num_accepted_raw < L:                                                        # SB:231-255
            kv = kv[:len(kv) - (L - num_accepted_raw)]
            next_token = greedy_all[num_accepted_raw - 1]
            if eos_enabled and next_token == eos_token_id:
                write(total_accepted, [next_token])
                total_accepted += 1
                to_delete = max(0, len(kv) - total_accepted)
                if to_delete > 0:
                    kv = kv[:len(kv) - to_delete]
                return dict(ret=accepted_n_gram[:total_accepted], next_token=next_token, iters=itr, kv_tokens=kv,
                            trace=trace)
            out = [next_token] + greedy_all[num_accepted_raw:-1]
        else:                                                                           # SB:258-273
            next_token = greedy_all[-1]
            write(total_accepted, [next_token])
            total_accepted += 1
            if eos_enabled and next_token == eos_token_id:
                to_delete = max(0, len(kv) - total_accepted)
                if to_delete > 0:
                    kv = kv[:len(kv) - to_delete]
                return dict(ret=accepted_n_gram[:total_accepted], next_token=next_token, iters=itr, kv_tokens=kv,
                            trace=trace)
    return dict(ret=accepted_n_gram[:total_accepted], next_token=next_token, iters=itr, kv_tokens=kv, trace=trace)


# ======================================================================================
# a15/a16/a17 — engine single-block decoder (JD:302-724) with the caller side (MR:1157-1199,
# 1407-1408) and block-manager bookkeeping (BM:267-276, 534-564) reduced to their integers
# ======================================================================================
class OracleSeq:
    """The integers of ``Sequence`` + block-table length the Jacobi path touches (SEQ:14-156)."""

    def __init__(self, prompt: List[int], block_len: int, max_tokens: int, max_iters: int = 128,
                 prefill_draft: Optional[List[int]] = None, block_size: int = 256):
        self.token_ids = list(prompt)
        self.num_prompt_tokens = len(prompt)
        self.num_cached_tokens = len(prompt)
        self.block_len, self.max_tokens, self.max_iters = block_len, max_tokens, max_iters
        self.prefill_draft = prefill_draft
        self.block_size = block_size
        self.num_table_blocks = (len(prompt) + block_size - 1) // block_size     # BlockManager.allocate
        self.num_permanent_spec_blocks = 0

    def __len__(self):
        return len(self.token_ids)

    @property
    def num_completion_tokens(self):
        return len(self.token_ids) - self.num_prompt_tokens

    def grow_for_draft(self, L: int) -> None:                                   # MR:1166-1198
        S = len(self)
        need = (S + L - 1 + self.block_size - 1) // self.block_size
        committed = (S + self.block_size - 1) // self.block_size
        self.num_table_blocks = need                                             # truncate or extend
        self.num_permanent_spec_blocks = max(self.num_permanent_spec_blocks, need - committed)

    def trim_kv_only_fast(self, num_tokens: int) -> None:                       # BM:534-564
        if num_tokens <= 0:
            return
        new_cached = max(len(self), self.num_cached_tokens - num_tokens)
        self.num_cached_tokens = new_cached
        blocks_needed = (new_cached + self.block_size - 1) // self.block_size if new_cached > 0 else 0
        keep = blocks_needed + self.num_permanent_spec_blocks
        if self.num_table_blocks > keep:
            self.num_table_blocks = keep


EngineForward = Callable[[List[OracleSeq], Rows], List[Rows]]
# (seqs, draft [B][L]) -> greedy [B][L-1]  (argmax of logits[:, :-1], MR:1413-1416)


def _engine_next_draft(seq: OracleSeq, L: int, acc_len: int, greedy: List[int], pads: Callable[[int], List[int]]):
    """JD:414-436 / JD:673-709."""
    d = [seq.token_ids[-1]]
    if acc_len < L:
        remaining = greedy[1:] if acc_len == 1 else greedy[acc_len - 1:]
        copy_len = min(len(remaining), L - 1)
        d += remaining[:copy_len]
    else:
        d += [greedy[-1]]
        copy_len = 1
    if copy_len < L - 1:
        d += pads(L - 1 - copy_len)
    return d


def _engine_first_draft(seq: OracleSeq, L: int, pads):
    """JD:332-347 / JD:529-546 (prefill draft) and JD:142-171 (random init)."""
    d = [seq.token_ids[-1]]
    if seq.prefill_draft is not None:
        pl = min(len(seq.prefill_draft), L - 1)
        d += list(seq.prefill_draft[:pl])
        if pl < L - 1:
            d += pads(L - 1 - pl)
        seq.prefill_draft = None
    else:
        if L > 1:
            d += pads(L - 1)
    return d


def _engine_commit_row(seq: OracleSeq, L: int, draft: List[int], greedy: List[int], eos_id: Optional[int]):
    """JD:357-399 / JD:589-649: returns (acc_len, new_tokens, eos_reached)."""
    k = 0
    while k < L - 1 and draft[k + 1] == greedy[k]:
        k += 1
    acc_len = max(1, min(k + 1, L))
    eos = False
    if eos_id is not None and acc_len > 1:
        for i in range(1, acc_len):
            if draft[i] == eos_id:
                acc_len = 1 + (i - 1) + 1
                eos = True
                break
    num_spec = acc_len - 1
    new: List[int] = []
    if num_spec > 0:
        new += draft[1:acc_len]
        seq.token_ids += draft[1:acc_len]
    if acc_len == 1:
        nt = greedy[0]
        seq.token_ids.append(nt)
        new.append(nt)
        num_spec = 1
        if eos_id is not None and nt == eos_id:
            eos = True
    seq.trim_kv_only_fast(L - 1 - num_spec)
    if len(seq) != seq.num_cached_tokens:
        raise RuntimeError(f"Invariant violated: len(token_ids)={len(seq)} != num_cached_tokens={seq.num_cached_tokens}")
    return acc_len, new, eos


def engine_generate_single(forward: EngineForward, seq: OracleSeq, eos_id: Optional[int],
                           pads: Callable[[int], List[int]], stats: Optional[dict] = None) -> List[int]:
    """JD:302-445."""
    L = seq.block_len
    max_tokens = seq.max_tokens - seq.num_completion_tokens
    if L <= 1:
        return []
    accepted: List[int] = []
    q: Optional[List[int]] = None
    eos_reached = False
    iters = 0
    while not eos_reached and len(accepted) < max_tokens and iters < seq.max_iters:
        iters += 1
        if q is None:
            q = _engine_first_draft(seq, L, pads)
        else:
            q[0] = seq.token_ids[-1]                                             # JD:197-199
        seq.grow_for_draft(L)
        greedy = forward([seq], [q])[0]
        seq.num_cached_tokens = len(seq) - 1 + L                                 # MR:1407-1408
        acc_len, new, eos = _engine_commit_row(seq, L, q, greedy, eos_id)
        eos_reached = eos_reached or eos
        accepted += new
        if eos_reached or len(accepted) >= max_tokens:
            break
        q = _engine_next_draft(seq, L, acc_len, greedy, pads)
    if stats is not None:
        stats["num_chunk_calls"] += 1
        stats["num_jacobi_iterations"] += iters
        stats["tokens_accepted"] += len(accepted)
        stats["tokens_per_call"].append(len(accepted))
        stats["iterations_per_call"].append(iters)
    return accepted


def new_stats() -> dict:
    return dict(num_chunk_calls=0, num_jacobi_iterations=0, tokens_accepted=0, tokens_per_call=[],
                tokens_per_iteration=[], iterations_per_call=[])


def engine_generate_batch(forward: EngineForward, seqs: List[OracleSeq], eos_id: Optional[int],
                          pads: Callable[[int], List[int]], stats: Optional[dict] = None) -> List[List[int]]:
    """JD:447-724 (group by L, larger groups first, one stats iteration per batch step)."""
    if not seqs:
        return []
    if len(seqs) == 1:
        return [engine_generate_single(forward, seqs[0], eos_id, pads, stats)]
    B = len(seqs)
    accepted: List[List[int]] = [[] for _ in range(B)]
    q: List[Optional[List[int]]] = [None] * B
    eos_reached = [False] * B
    iters = [0] * B
    max_tokens = [max(0, s.max_tokens - s.num_completion_tokens) for s in seqs]
    n_iter_call = 0
    prev_len = [0] * B
    while True:
        active = [i for i in range(B) if not eos_reached[i] and len(accepted[i]) < max_tokens[i]
                  and iters[i] < seqs[i].max_iters]
        if not active:
            break
        groups: Dict[int, List[int]] = {}
        for i in active:
            if seqs[i].block_len > 1:
                groups.setdefault(seqs[i].block_len, []).append(i)
        if not groups:
            break
        n_iter_call += 1
        tokens_this_iter = 0
        for L, idxs in sorted(groups.items(), key=lambda x: len(x[1]), reverse=True):
            drafts = []
            for i in idxs:
                iters[i] += 1
                if q[i] is None:
                    q[i] = _engine_first_draft(seqs[i], L, pads)
                else:
                    q[i][0] = seqs[i].token_ids[-1]
                drafts.append(q[i])
            for i in idxs:
                seqs[i].grow_for_draft(L)
            greedy = forward([seqs[i] for i in idxs], drafts)
            for i in idxs:
                seqs[i].num_cached_tokens = len(seqs[i]) - 1 + L
            for row, i in enumerate(idxs):
                acc_len, new, eos = _engine_commit_row(seqs[i], L, drafts[row], greedy[row], eos_id)
                if eos:
                    eos_reached[i] = True
                accepted[i] += new
                tokens_this_iter += len(accepted[i]) - prev_len[i]
                prev_len[i] = len(accepted[i])
                if eos_reached[i] or len(accepted[i]) >= max_tokens[i]:
                    q[i] = None
                    continue
                q[i] = _engine_next_draft(seqs[i], L, acc_len, greedy[row], pads)
        if stats is not None:
            stats["tokens_per_iteration"].append(tokens_this_iter)
    if stats is not None:
        tot = sum(len(a) for a in accepted)
        stats["num_chunk_calls"] += 1
        stats["num_jacobi_iterations"] += n_iter_call
        stats["tokens_accepted"] += tot
        stats["tokens_per_call"].append(tot)
        stats["iterations_per_call"].append(n_iter_call)
    return accepted


# ======================================================================================
# a19 — non-greedy rejection-sampling verify (JDN:299-354) with injected randomness
# ======================================================================================
# ---- The probability tensor, defined EXACTLY (round 4) -------------------------------------------------------------------
# torch forms  probs = softmax(xs)  in float32 and (bf16 logits) rounds once to bf16; its float32 internals (vectorised exp,
# summation order) are not a specification — torch's CPU and GPU kernels, numpy and any HIP kernel differ from each other in the
# last place, and after the bf16 rounding that is one bf16 ulp on ~0.1-1 % of the entries: enough to move an inverse-CDF draw
# across a token boundary once in ~10^6 draws (profiles/soak_r02.txt, seed 4555).  What every one of them approximates is
#
#     p_i = RN_dtype( exp(xs_i - M) / sum_j exp(xs_j - M) ),   M = max xs,
#
# the exact real quotient rounded ONCE (nearest, ties to even; subnormals included) to the dtype of the logits — bf16 for the
# engine's logits (MR:1382, JDN:64-70 never widens them), float32 for float32 logits.  That IS the definition here, and the HIP
# kernels implement the same definition (float64 evaluation where a decision depends on it, jf_sampling.hip), so token ids are
# bit-exact by construction instead of "equal unless a draw lands within an ulp of a boundary".  It stays within one ulp of the
# dtype of torch's own tensor (tests/golden/softmax_vectors.json, recorded from the reference).
# float64 evaluation errs by a few 1e-16 relative (exp and division one ulp each, a pairwise sum of non-negative terms); an
# element whose exact quotient lies that close to a rounding boundary cannot be decided in float64.  Elements within 1e-14
# (relative) of a boundary are therefore re-decided in 80-bit extended precision, and those within 1e-18 of it in 60-digit
# decimal arithmetic.
NEAR_TIE_REL = 1e-14
NEAR_TIE_REL_LD = 1e-18
NEAR_TIES_RESOLVED = [0, 0]       # how many elements took the extended / the decimal path (diagnostics for the tests)


def _round_to_grid(q: np.ndarray, mant_bits: int, row_ctx=None) -> np.ndarray:
    """Nearest-even rounding of non-negative float64 values onto the grid of a binary format with ``mant_bits`` explicit
    mantissa bits and float32's exponent range (bf16: 7, float32: 23), subnormals included.  Returned as float64 (exact)."""
    q = np.asarray(q, dtype=np.float64)
    _, ex = np.frexp(q)                                   # q = f * 2^ex, f in [0.5, 1)
    g = np.maximum(ex - 1, -126) - mant_bits              # exponent of the grid spacing at q
    k = np.ldexp(q, -g)                                   # exact (power-of-two scaling, no underflow: q * 2^149 at most)
    r = np.rint(k)                                        # nearest even on the exact value
    if row_ctx is not None:
        frac = np.abs(k - np.floor(k) - 0.5)
        near = (frac < NEAR_TIE_REL * np.maximum(k, 1.0)) & (q > 0)
        if near.any():
            for i in np.nonzero(near)[0]:
                r[i] = row_ctx(int(i), int(g[i]))
    return np.ldexp(r, g)


def _decimal_rounder(d_row: np.ndarray):
    """Returns f(i, g) -> the correctly rounded integer k = RN(exp(d_i) / sum_j exp(d_j) / 2^g): first in 80-bit extended
    precision (x86 long double: 64-bit mantissa), and when the quotient lies within NEAR_TIE_REL_LD of the boundary even
    there, in 60-digit decimal arithmetic."""
    import decimal
    ctx = decimal.Context(prec=60)
    state = {}
    ld = np.longdouble
    have_ld = np.finfo(ld).nmant >= 63

    def decide(i: int, g: int):
        if have_ld:
            if "Sld" not in state:
                state["Sld"] = np.exp(d_row.astype(ld)).sum()
            k = np.ldexp(np.exp(ld(d_row[i])) / state["Sld"], -g)
            fl = np.floor(k)
            if abs(k - fl - ld(0.5)) >= ld(NEAR_TIE_REL_LD) * max(k, ld(1.0)):
                NEAR_TIES_RESOLVED[0] += 1
                return float(fl + 1) if k - fl > ld(0.5) else float(fl)
        NEAR_TIES_RESOLVED[1] += 1
        if "S" not in state:
            uniq, cnt = np.unique(d_row[np.isfinite(d_row)], return_counts=True)
            state["S"] = sum((ctx.multiply(ctx.exp(decimal.Decimal(float(v))), decimal.Decimal(int(c))) for v, c in zip(uniq, cnt)),
                             decimal.Decimal(0))
        q = ctx.divide(ctx.exp(decimal.Decimal(float(d_row[i]))), state["S"])
        k = ctx.multiply(q, ctx.power(decimal.Decimal(2), decimal.Decimal(-g)))
        fl = k.to_integral_value(rounding=decimal.ROUND_FLOOR)
        diff = ctx.subtract(k, fl)
        half = decimal.Decimal("0.5")
        if diff > half or (diff == half and int(fl) % 2 == 1):
            return float(int(fl) + 1)
        return float(int(fl))
    return decide


def exact_softmax_rows(xs: np.ndarray, mant_bits: int) -> np.ndarray:
    """Rows of already temperature-scaled logits (float32 values) -> probabilities per the definition above, as float32.
    Rows without a finite maximum, or holding NaN / +inf, keep the plain float32 formula (NaN where torch's softmax is NaN)."""
    xs = np.asarray(xs, dtype=np.float32)
    out = np.empty(xs.shape, dtype=np.float32)
    flat_in, flat_out = xs.reshape(-1, xs.shape[-1]), out.reshape(-1, xs.shape[-1])
    for r in range(flat_in.shape[0]):
        x = flat_in[r]
        m = x.max() if x.size else np.float32(0)
        if not np.isfinite(m) or np.isnan(x).any():
            with np.errstate(invalid="ignore", over="ignore"):
                e = np.exp(x - m, dtype=np.float32)
                p = (e / e.sum(dtype=np.float32)).astype(np.float32)
            flat_out[r] = p if mant_bits == 23 else bf16_round(p)
            continue
        d = x.astype(np.float64) - np.float64(m)          # exact: both are float32 values
        e = np.exp(d)
        S = float(np.sum(e, dtype=np.float64))
        flat_out[r] = _round_to_grid(e / S, mant_bits, row_ctx=_decimal_rounder(d)).astype(np.float32)
    return out


def softmax_rows_f32(logits: np.ndarray, temperature: float) -> np.ndarray:
    """JDN:65-70 on float32 logits: xs = fl32(logits / T) (torch's true division; T<=0 treated as 1), p = the exact softmax of
    xs rounded once to float32."""
    x = np.asarray(logits, dtype=np.float32)
    t = np.float32(1.0 if (temperature is None or temperature <= 0) else temperature)
    if t != np.float32(1.0):
        x = (x / t).astype(np.float32)
    return exact_softmax_rows(x, 23)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32 values (torch ``.to(torch.bfloat16)``)."""
    return bf16_bits_to_f32(f32_to_bf16_bits(np.asarray(x, dtype=np.float32)))


def softmax_rows_bf16(logits: np.ndarray, temperature: float) -> np.ndarray:
    """JDN:64-70 on a bfloat16 logits tensor, which is what the engine hands the verifier (MR:1382 has no cast and
    JDN never calls ``.float()``).  torch keeps the tensor dtype through both ops:
      * ``logits / float(T)``: float32 quotient of the widened operand by float32(T), rounded to bf16
        (ATen div_true_kernel for reduced floating types; skipped when T == 1, JDN:68) — reproduced bit for bit;
      * ``torch.softmax``: a float32 softmax rounded to bf16 — DEFINED here as the exact softmax rounded once to bf16 (see
        the block comment above); torch's tensor lies within one bf16 ulp of it.
    Returned as float32 values that are exactly bf16-representable; every consumer (``u < p``, the float64 inverse-CDF
    walk, the masked argmax) works on these rounded values, as the reference does."""
    x = np.asarray(logits, dtype=np.float32)
    if not np.array_equal(bf16_round(x), x, equal_nan=True):
        raise ValueError("softmax_rows_bf16 expects bf16-representable logits")
    t = np.float32(1.0 if (temperature is None or temperature <= 0) else temperature)
    if t != np.float32(1.0):
        x = bf16_round((x / t).astype(np.float32))
    return exact_softmax_rows(x, 7)


def _round_values(x64: np.ndarray, mant_bits: int) -> np.ndarray:
    """float64 values (sums of a few thousand grid values, or one quotient of two grid values: both far from needing more than
    float64) rounded ONCE to the probability grid: float32 (23) or bfloat16 (7), nearest-even, as float32."""
    f = np.asarray(x64, dtype=np.float64)
    if mant_bits == 23:
        return f.astype(np.float32)
    # float64 -> bf16 in one rounding: round-to-odd into float32 first (a plain float32 cast would round twice)
    shape = f.shape
    f = np.atleast_1d(f)
    f32 = f.astype(np.float32)
    bits = f32.view(np.uint32).copy()
    back = f32.astype(np.float64)
    bits = np.where(back > f, bits - np.uint32(1), bits)            # toward zero (values are non-negative)
    bits = np.where(back != f, bits | np.uint32(1), bits)           # sticky
    return bf16_round(bits.astype(np.uint32).view(np.float32)).reshape(shape)


def _div_in_dtype(num: np.ndarray, den: float, mant_bits: int) -> np.ndarray:
    """``tensor / scalar-tensor`` as torch forms it in the tensor's dtype: a float32 quotient of the float32 images (ATen's
    div_true_kernel, correctly rounded), rounded once more to bf16 for bf16 tensors."""
    q = (np.asarray(num, dtype=np.float32) / np.float32(den)).astype(np.float32)
    return q if mant_bits == 23 else bf16_round(q)


def filter_probs_row(p: np.ndarray, top_k, top_p, mant_bits: int) -> np.ndarray:
    """_apply_top_k then _apply_top_p (JDN:72-107) on ONE probability row ``p`` (values on the dtype's grid), in the arithmetic of
    that dtype, with the two things torch leaves to its kernels DEFINED:

    * **ties**: torch.topk / torch.sort do not say which of several EQUAL probabilities come first (the CPU kernels are
      neither stable nor consistent between sizes: a 7-element row picks [1, 5, 3] of four equal values, a 5 000-element row the
      lowest indices).  Here equal values are ordered by INDEX, lowest first — the order of a stable descending sort.  A row whose
      kept set ends INSIDE a group of equal values is therefore "ours": the reference's choice among those ids is its kernel's.
    * **sums**: ``out.sum()`` and ``cumsum`` accumulate in float32 in a kernel-specific order and round to the dtype.  Here every
      sum is the EXACT sum rounded once to the dtype (float64 holds sums of <= 2^18 grid values exactly enough: 2^-35 relative).
      torch's result differs only when the exact sum lies within its accumulation error of a rounding boundary.

    Everything else is torch's arithmetic to the bit: ``sum.clamp_min(1e-12)`` in the dtype, the quotient ``p / sum`` (float32
    division, rounded again to bf16 for bf16 tensors), ``cdf <= top_p`` with the Python float cast to the tensor's dtype, "at
    least one kept".  Returns float32 values on the dtype's grid."""
    r = np.asarray(p, dtype=np.float32).copy()
    V = r.shape[0]
    grid = (lambda x: np.float32(x)) if mant_bits == 23 else (lambda x: np.float32(bf16_round(np.array([x], dtype=np.float32))[0]))
    floor = grid(1e-12)
    if top_k is not None and 0 < int(top_k) < V:                                    # JDN:73-84
        order = np.lexsort((np.arange(V), -r.astype(np.float64)))                   # value descending, index ascending
        keep = np.zeros(V, dtype=bool)
        keep[order[:int(top_k)]] = True
        s1 = _round_values(np.sum(r[keep].astype(np.float64)), mant_bits)
        s1 = max(np.float32(s1), floor)
        r = np.where(keep, _div_in_dtype(r, s1, mant_bits), np.float32(0)).astype(np.float32)
    if top_p is not None and 0.0 < float(top_p) < 1.0:                              # JDN:91-107
        tp = grid(float(top_p))
        order = np.lexsort((np.arange(V), -r.astype(np.float64)))
        cdf = _round_values(np.cumsum(r[order].astype(np.float64)), mant_bits)
        keep_sorted = cdf <= tp
        keep_sorted[0] = True
        keep = np.zeros(V, dtype=bool)
        keep[order[keep_sorted]] = True
        s2 = _round_values(np.sum(r[keep].astype(np.float64)), mant_bits)
        s2 = max(np.float32(s2), floor)
        r = np.where(keep, _div_in_dtype(r, s2, mant_bits), np.float32(0)).astype(np.float32)
    return r


def filter_boundary_is_tied(p: np.ndarray, top_k, top_p, mant_bits: int) -> bool:
    """True when a kept set of filter_probs_row ends inside a group of equal values (the reference's own choice among the tied
    ids is then its kernels', not the algorithm's: such rows are 'defined here', not pinned)."""
    r = np.asarray(p, dtype=np.float32).copy()
    V = r.shape[0]
    tied = False
    if top_k is not None and 0 < int(top_k) < V:
        srt = np.sort(r.astype(np.float64))[::-1]
        tied |= bool(srt[int(top_k) - 1] == srt[int(top_k)])
        r = filter_probs_row(p, top_k, None, mant_bits)
    if top_p is not None and 0.0 < float(top_p) < 1.0:
        q = filter_probs_row(r, None, top_p, mant_bits)
        kept_min = q[q > 0].size and r[q > 0].min()
        tied |= bool(((q == 0) & (r == kept_min) & (r > 0)).any())
    return tied


def target_probs(logits: np.ndarray, temperature: float, logits_dtype: str = "f32", top_k=None, top_p=None) -> np.ndarray:
    """_build_target_probs (JDN:110-123 / JDO:128-136) for the dtype the forward callback returned.  top_k / top_p are not
    SamplingParams fields (sampling_params.py:4-38): the reference reads them with getattr, so they only exist when a caller
    planted them on the request object (None: the plain softmax)."""
    if logits_dtype == "bf16":
        p, mb = softmax_rows_bf16(logits, temperature), 7
    elif logits_dtype == "f32":
        p, mb = softmax_rows_f32(logits, temperature), 23
    else:
        raise ValueError(f"logits_dtype must be 'f32' or 'bf16', got {logits_dtype!r}")
    k_on = top_k is not None and 0 < int(top_k) < p.shape[-1]
    p_on = top_p is not None and 0.0 < float(top_p) < 1.0
    if not (k_on or p_on):
        return p
    flat = p.reshape(-1, p.shape[-1])
    out = np.stack([filter_probs_row(row, top_k, top_p, mb) for row in flat], 0)
    return out.reshape(p.shape)


def inverse_cdf_sample(probs: np.ndarray, u: float) -> int:
    """The injected stand-in for torch.multinomial used on both sides of the parity test:
    smallest index whose float64 running sum exceeds u * total (clamped to V-1)."""
    c = np.cumsum(probs.astype(np.float64))
    idx = int(np.searchsorted(c, u * float(c[-1]), side="right"))
    return min(idx, probs.shape[0] - 1)


def rs_verify_row(draft_row: List[int], probs: np.ndarray, eos_id: Optional[int],
                  next_uniform: Callable[[], float], next_bonus_uniform: Callable[[], float]):
    """JDN:315-354.  probs: [L-1, V] target distribution.  Returns (committed, num_keep_for_kv, eos)."""
    L = len(draft_row)
    if L <= 1:
        return [], 0, False
    committed: List[int] = []
    eos = False
    for t in range(L - 1):
        proposed = int(draft_row[t + 1])
        p_x = float(probs[t, proposed])
        u = float(next_uniform())
        if u < p_x:
            committed.append(proposed)
            if eos_id is not None and proposed == eos_id:
                eos = True
                break
            continue
        bonus = None
        for _ in range(16):                                                     # JDN:136-146
            y = inverse_cdf_sample(probs[t], next_bonus_uniform())
            if y != proposed:
                bonus = y
                break
        if bonus is None:                                                       # JDN:147-153
            p2 = probs[t].copy()
            p2[proposed] = 0.0
            bonus = proposed if float(p2.sum()) <= 0 else int(argmax_rows(p2[None, :])[0])
        committed.append(int(bonus))
        if eos_id is not None and int(bonus) == eos_id:
            eos = True
        break
    return committed, len(committed), eos


NonGreedyForward = Callable[[List[OracleSeq], Rows], List[np.ndarray]]
# (seqs, draft [B][L]) -> logits per row, each [L-1, V] float32 (MR:1413-1416)


def _ng_next_draft(seq: OracleSeq, L: int, n_committed: int, greedy: List[int], pads):
    """JDN:444-466 / JDN:619-638 (same shape as the greedy decoder's next draft)."""
    return _engine_next_draft(seq, L, 1 + n_committed, greedy, pads)


def _ng_commit(seq: OracleSeq, L: int, committed: List[int], num_keep: int):
    """JDN:417-437 / JDN:592-612."""
    if committed:
        seq.token_ids += committed
    seq.trim_kv_only_fast((L - 1) - num_keep)
    if len(seq) != seq.num_cached_tokens:
        raise RuntimeError(f"Invariant violated: len(token_ids)={len(seq)} != num_cached_tokens={seq.num_cached_tokens}")


def nongreedy_generate_single(forward: NonGreedyForward, seq: OracleSeq, eos_id: Optional[int], temperature: float,
                              pads, next_uniform, next_bonus_uniform, stats: Optional[dict] = None,
                              logits_dtype: str = "f32", top_k=None, top_p=None):
    """JDN:376-482."""
    L = seq.block_len
    max_tokens = seq.max_tokens - seq.num_completion_tokens
    accepted: List[int] = []
    q: Optional[List[int]] = None
    eos_reached = False
    iters = 0
    while not eos_reached and len(accepted) < max_tokens and iters < seq.max_iters:
        iters += 1
        if L <= 1:
            break
        if q is None:
            q = [seq.token_ids[-1]] + pads(L - 1)
        else:
            q[0] = seq.token_ids[-1]
        seq.grow_for_draft(L)
        logits = forward([seq], [q])[0]
        seq.num_cached_tokens = len(seq) - 1 + L
        probs = target_probs(logits, temperature, logits_dtype, top_k, top_p)
        committed, keep, eos = rs_verify_row(q, probs, eos_id, next_uniform, next_bonus_uniform)
        eos_reached = eos_reached or eos
        _ng_commit(seq, L, committed, keep)
        accepted += committed
        if eos_reached or len(accepted) >= max_tokens:
            break
        greedy = argmax_rows(logits).tolist()
        q = _ng_next_draft(seq, L, len(committed), greedy, pads)
    if stats is not None:
        stats["num_chunk_calls"] += 1
        stats["num_jacobi_iterations"] += iters
        stats["tokens_accepted"] += len(accepted)
        stats["tokens_per_call"].append(len(accepted))
        stats["iterations_per_call"].append(iters)
    return accepted


def nongreedy_generate_batch(forward: NonGreedyForward, seqs: List[OracleSeq], eos_id: Optional[int],
                             temperature: float, pads, next_uniform, next_bonus_uniform,
                             stats: Optional[dict] = None, logits_dtype: str = "f32", top_k=None, top_p=None):
    """JDN:485-667."""
    if not seqs:
        return []
    if len(seqs) == 1:
        return [nongreedy_generate_single(forward, seqs[0], eos_id, temperature, pads, next_uniform,
                                          next_bonus_uniform, stats, logits_dtype, top_k, top_p)]
    B = len(seqs)
    accepted: List[List[int]] = [[] for _ in range(B)]
    q: List[Optional[List[int]]] = [None] * B
    eos_reached = [False] * B
    iters = [0] * B
    max_tokens = [max(0, s.max_tokens - s.num_completion_tokens) for s in seqs]
    n_iter_call = 0
    while True:
        active = [i for i in range(B) if not eos_reached[i] and len(accepted[i]) < max_tokens[i]
                  and iters[i] < seqs[i].max_iters]
        if not active:
            break
        groups: Dict[int, List[int]] = {}
        for i in active:
            if seqs[i].block_len > 1:
                groups.setdefault(seqs[i].block_len, []).append(i)
        if not groups:
            break
        n_iter_call += 1
        tokens_this_iter = 0
        for L, idxs in sorted(groups.items(), key=lambda x: len(x[1]), reverse=True):
            drafts = []
            for i in idxs:
                iters[i] += 1
                if q[i] is None:
                    q[i] = [seqs[i].token_ids[-1]] + pads(L - 1)
                else:
                    q[i][0] = seqs[i].token_ids[-1]
                drafts.append(q[i])
            for i in idxs:
                seqs[i].grow_for_draft(L)
            logits = forward([seqs[i] for i in idxs], drafts)
            for i in idxs:
                seqs[i].num_cached_tokens = len(seqs[i]) - 1 + L
            for row, i in enumerate(idxs):
                probs = target_probs(logits[row], temperature, logits_dtype, top_k, top_p)
                committed, keep, eos = rs_verify_row(drafts[row], probs, eos_id, next_uniform, next_bonus_uniform)
                eos_reached[i] = eos_reached[i] or eos
                _ng_commit(seqs[i], L, committed, keep)
                accepted[i] += committed
                tokens_this_iter += len(committed)
                if eos_reached[i] or len(accepted[i]) >= max_tokens[i]:
                    q[i] = None
                    continue
                greedy = argmax_rows(logits[row]).tolist()
                q[i] = _ng_next_draft(seqs[i], L, len(committed), greedy, pads)
        if stats is not None:
            stats["tokens_per_iteration"].append(tokens_this_iter)
    if stats is not None:
        tot = sum(len(a) for a in accepted)
        stats["num_chunk_calls"] += 1
        stats["num_jacobi_iterations"] += n_iter_call
        stats["tokens_accepted"] += tot
        stats["tokens_per_call"].append(tot)
        stats["iterations_per_call"].append(n_iter_call)
    return accepted


class CounterStream:
    """Counter-based injected randomness shared by the golden generator and the tests:
    element k = mix32(seed, k)."""

    def __init__(self, seed: int):
        from oracle.scripted_model import mix32
        self._mix = mix32
        self.seed = seed
        self.k = 0

    def next_u32(self) -> int:
        v = self._mix(self.seed, self.k)
        self.k += 1
        return v

    def pads(self, vocab: int) -> Callable[[int], List[int]]:
        return lambda count: [self.next_u32() % vocab for _ in range(count)]

    def uniform(self) -> float:
        return (self.next_u32() >> 8) / float(1 << 24)


# ------------------------------------------------------------------------------------------------
# On-policy rollout records — restates inference_engine/engine/jacobi_decoding_nongreedy_on_policy.py
# ("JDO").  Randomness is injected in the reference's call order: ``rnd`` (random.choice / randrange,
# JDO:254-266, 474), ``next_uniform`` (torch.rand, JDO:306), ``next_multinomial_uniform`` (every
# torch.multinomial: bonus draws JDO:157-163 and the re-draft samples JDO:476).
# ------------------------------------------------------------------------------------------------
def trim_left_padding(ids: List[int], pad: Optional[int]) -> List[int]:          # JDO:77-87
    if not ids:
        return []
    if pad is None:
        return list(ids)
    for i, t in enumerate(ids):
        if int(t) != int(pad):
            return list(ids[i:])
    return []


def truncate_after_stop(ids: List[int], start_idx: int, stop_ids) -> List[int]:  # JDO:170-182
    stop = set(int(x) for x in stop_ids)
    for i in range(max(0, int(start_idx)), len(ids)):
        if int(ids[i]) in stop:
            return list(ids[:i + 1])
    return list(ids)


def onpolicy_verify(proposed: List[int], probs: np.ndarray, stop_ids, next_uniform, next_multinomial_uniform):
    """JDO:270-327.  Returns (committed, stop_hit)."""
    committed: List[int] = []
    stop = set(int(x) for x in stop_ids)
    stop_hit = False
    for t, x in enumerate(proposed):
        x = int(x)
        if x < 0 or x >= probs.shape[-1]:
            raise ValueError(f"Token index {x} out of bounds for vocab size {probs.shape[-1]}.")
        p_x = float(probs[t, x])
        u = float(next_uniform())
        if u < p_x:
            committed.append(x)
        else:
            bonus = None
            for _ in range(16):                                                  # JDO:157-163
                y = inverse_cdf_sample(probs[t], next_multinomial_uniform())
                if y != x:
                    bonus = y
                    break
            if bonus is None:                                                    # JDO:164-168
                p2 = probs[t].copy()
                p2[x] = 0.0
                bonus = x if float(p2.sum()) <= 0.0 else int(argmax_rows(p2[None, :])[0])
            committed.append(int(bonus))
            break
        if committed[-1] in stop:
            stop_hit = True
            break
    if committed and committed[-1] in stop:
        stop_hit = True
    return committed, stop_hit


def onpolicy_run_one_block(forward: NonGreedyForward, seq: OracleSeq, block_len: int, budget: int, completion_start: int,
                           temperature: float, stop_ids, pad_id: int, vocab: int, rnd, next_uniform,
                           next_multinomial_uniform, logits_dtype: str = "f32", top_k=None, top_p=None):
    """JDO:331-488.  Returns (trajectory, appended_total, forwards_used, stopped)."""
    full_len = int(block_len)
    if full_len <= 0 or budget <= 0:
        return [], 0, 0, True
    gen_len = min(full_len, int(budget))
    choices = [int(t) for t in seq.token_ids if int(t) != int(pad_id)]           # JDO:254-266
    init = [rnd.choice(choices) if choices else rnd.randrange(vocab) for _ in range(gen_len)]
    block = list(init) + [pad_id] * (full_len - gen_len)
    accepted, stopped, fwd_used, appended = 0, False, 0, 0
    traj = [list(block)]
    stop = set(int(x) for x in stop_ids)
    while accepted < gen_len and not stopped:
        remaining = gen_len - accepted
        if not seq.token_ids:
            seq.token_ids = [pad_id]
        proposed = [int(t) for t in block[accepted:gen_len]]
        draft = [int(seq.token_ids[-1])] + proposed
        seq.grow_for_draft(remaining + 1)
        logits = forward([seq], [draft])[0]                                      # [remaining, V]
        seq.num_cached_tokens = len(seq) - 1 + (remaining + 1)
        fwd_used += 1
        probs = target_probs(logits, temperature, logits_dtype, top_k, top_p)
        committed, stop_hit = onpolicy_verify(proposed, probs, stop_ids, next_uniform, next_multinomial_uniform)
        if not committed:
            committed = [proposed[0]]
            stop_hit = committed[0] in stop
        seq.token_ids += committed                                               # JDO:412-416
        appended += len(committed)
        seq.trim_kv_only_fast(remaining - len(committed))                        # JDO:420-426
        if len(seq) != seq.num_cached_tokens:
            raise RuntimeError(f"Invariant violated: len(token_ids)={len(seq)} != num_cached_tokens={seq.num_cached_tokens}")
        prev = accepted
        accepted = min(gen_len, accepted + len(committed))
        block[prev:accepted] = committed[:accepted - prev]
        if stop_hit:                                                             # JDO:439-463
            full = list(seq.token_ids)
            trunc = truncate_after_stop(full, completion_start, stop_ids)
            if len(trunc) != len(full):
                seq.token_ids = trunc
                seq.trim_kv_only_fast(len(full) - len(trunc))
            stopped = True
            pos = next((j for j in range(prev, accepted) if int(block[j]) in stop), None)
            if pos is not None:
                for k in range(pos + 1, full_len):
                    block[k] = pad_id
                accepted = min(accepted, pos + 1)
        if not stopped and accepted < gen_len:                                   # JDO:465-477
            local_start = len(committed)
            new = []
            for jj in range(gen_len - accepted):
                li = local_start + jj
                new.append(rnd.randrange(vocab) if li >= probs.shape[0]
                           else inverse_cdf_sample(probs[li], next_multinomial_uniform()))
            block[accepted:gen_len] = new
        for k in range(gen_len, full_len):
            block[k] = pad_id
        traj.append(list(block))
    return traj, appended, fwd_used, stopped


def onpolicy_rollout_records_batch(forward: NonGreedyForward, seqs: List[OracleSeq], temperature: float, stop_ids,
                                   pad_id: int, vocab: int, rnd, next_uniform, next_multinomial_uniform,
                                   n_token_seq_len: Optional[int] = None, data_ids: Optional[List[str]] = None,
                                   logits_dtype: str = "f32", top_k=None, top_p=None):
    """JDO:494-614: (records per sequence {block index -> record}, metrics per sequence).
    ``seq.max_iters`` is the maximum number of BLOCKS (JDO:232-233)."""
    B = len(seqs)
    if B == 0:
        return [], []
    starts = [len(s.token_ids) for s in seqs]
    block_lens = [int(n_token_seq_len) if n_token_seq_len is not None else int(s.block_len) for s in seqs]
    budgets = [max(0, s.max_tokens - s.num_completion_tokens) for s in seqs]
    stopped = [False] * B
    done_blocks, forwards, generated = [0] * B, [0] * B, [0] * B
    ids = data_ids if data_ids is not None else [f"data_{i}" for i in range(B)]
    out: List[Dict[int, dict]] = [dict() for _ in range(B)]
    while True:
        active = [i for i in range(B) if not stopped[i] and done_blocks[i] < seqs[i].max_iters and budgets[i] > 0]
        if not active:
            break
        for i in active:
            seq, k = seqs[i], done_blocks[i]
            if block_lens[i] <= 0:
                stopped[i] = True
                continue
            prompt_trim = trim_left_padding(list(seq.token_ids), pad_id)
            traj, app, fw, hit = onpolicy_run_one_block(forward, seq, block_lens[i], budgets[i], starts[i], temperature,
                                                        stop_ids, pad_id, vocab, rnd, next_uniform, next_multinomial_uniform,
                                                        logits_dtype, top_k=top_k, top_p=top_p)
            done_blocks[i] += 1
            forwards[i] += fw
            generated[i] += app
            budgets[i] = max(0, budgets[i] - app)
            stopped[i] = bool(hit)
            teacher = trim_left_padding(truncate_after_stop(list(seq.token_ids), starts[i], stop_ids), pad_id)
            out[i][k] = dict(diffusion_itr_id=f"itr_{k}", data_id=str(ids[i]), prompt_ids=prompt_trim,
                             answer_trajectory_ids=traj, teacher_output_ids=teacher,
                             tokens_per_iter=float(generated[i]) / float(max(1, done_blocks[i])),
                             tokens_per_forward=float(generated[i]) / float(max(1, forwards[i])),
                             num_iters=done_blocks[i], num_forwards=forwards[i])
    final = {}
    for i in range(B):
        final[str(ids[i])] = trim_left_padding(truncate_after_stop(list(seqs[i].token_ids), starts[i], stop_ids), pad_id)
    for i in range(B):
        for k in out[i]:
            out[i][k]["teacher_output_ids"] = final.get(str(ids[i]), [])
    metrics = [dict(total_tokens=float(generated[i]), num_iters=float(done_blocks[i]), num_forwards=float(forwards[i]),
                    tokens_per_iter=float(generated[i]) / float(max(1, done_blocks[i])),
                    tokens_per_forward=float(generated[i]) / float(max(1, forwards[i]))) for i in range(B)]
    return out, metrics


class ScriptedRandom:
    """``random.choice`` / ``random.randrange`` over a CounterStream (the golden generator patches the same in)."""

    def __init__(self, stream: "CounterStream"):
        self.stream = stream

    def choice(self, seq):
        return seq[self.stream.next_u32() % len(seq)]

    def randrange(self, n):
        return self.stream.next_u32() % n


# ------------------------------------------------------------------------------------------------
# paged-KV index buffers of one batched Jacobi forward — restates MR:1204-1265 ("jacobi.buffer_fill")
# and MR:965-986 (_get_slot_mapping_pattern).  Test infrastructure only.
# ------------------------------------------------------------------------------------------------
def engine_fill_ref(draft, seq_lens, block_tables, block_size, max_cols):
    """draft [B][L]; seq_lens[i] = len(seq_i) (>= 1); block_tables[i] = list of block ids.
    Returns dict(input_ids, positions, slot_mapping, cu_seqlens_q, cu_seqlens_k, cache_seqlens, block_tables, max_seqlen_k)."""
    B, L = len(draft), len(draft[0])
    input_ids, positions, slots = [], [], []
    cu_q, cu_k, cache = [0], [0], []
    bt = np.full((B, max_cols), -1, dtype=np.int32)                                  # MR:678, 1248-1249
    for i in range(B):
        S = seq_lens[i]
        if S < 1:
            raise ValueError(f"Sequence {i} has invalid length S={S}. Must be >= 1.")    # MR:1222-1223
        if len(block_tables[i]) > max_cols:
            raise RuntimeError(f"Sequence {i} needs {len(block_tables[i])} blocks but buffer only has {max_cols}.")  # MR:1240-1247
        bt[i, :len(block_tables[i])] = block_tables[i]
        input_ids += [int(t) for t in draft[i]]                                      # MR:1227
        for j in range(L):
            pos = S - 1 + j                                                          # MR:1229, 979
            positions.append(pos)
            slots.append(int(bt[i, pos // block_size]) * block_size + pos % block_size)   # MR:981-982, 1252-1254
        cu_q.append(cu_q[-1] + L)                                                    # MR:1256
        cu_k.append(cu_k[-1] + (S - 1) + L)                                          # MR:1257
        cache.append(S - 1)                                                          # MR:1260-1263
    return dict(input_ids=input_ids, positions=positions, slot_mapping=slots, cu_seqlens_q=cu_q, cu_seqlens_k=cu_k,
                cache_seqlens=cache, block_tables=bt, max_seqlen_k=max(S - 1 + L for S in seq_lens))   # MR:1273
