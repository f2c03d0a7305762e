"""Engine single-block greedy Jacobi decoder — same constructor, callbacks, return values and ``stats`` as the
reference's ``JacobiDecoder`` (inference_engine/engine/jacobi_decoding.py:47-724 = "JD"), with the per-iteration body
(argmax, accept scan, EOS cap, commit, AR fallback, next draft incl. random pads) in one HIP launch
(``jf_argmax_partial`` + ``jf_engine_step``) and one read-back per iteration instead of JD's per-row ``.item()`` /
``.tolist()`` syncs.

Callback contract (JD:30-44, MR:1134-1418): ``forward_step_batch(seqs, draft[B, L]) -> logits[B, L-1, V]`` where
``draft[:, 0]`` is the already-cached last token ("seed"); the callback sets ``seq.draft_tokens_gpu`` and
``seq.num_cached_tokens = len(seq) - 1 + L``.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import ops
from .block_manager import BlockManager
from .sequence import Sequence

LogitsForwardFn = Callable[[Sequence, Tensor], Tensor]
LogitsForwardFnBatch = Callable[[List[Sequence], Tensor], Tensor]

_PAD_STREAM_LEN = 1 << 16


class JacobiDecoder:
    def __init__(self, block_manager: BlockManager, forward_step: Optional[LogitsForwardFn] = None,
                 forward_step_batch: Optional[LogitsForwardFnBatch] = None, eos_token_id: Optional[int] = None,
                 pad_token_id: Optional[int] = None, vocab_size: Optional[int] = None,
                 device: Optional[torch.device] = None) -> None:
        if forward_step is None and forward_step_batch is None:
            raise ValueError("Provide at least one of forward_step or forward_step_batch.")          # JD:78-79
        self.block_manager = block_manager
        self.forward_step = forward_step
        self.forward_step_batch = forward_step_batch
        self.eos_token_id = eos_token_id
        self.pad_token_id = pad_token_id
        if vocab_size is None:
            raise ValueError("vocab_size must be provided from model config. Do not use hard-coded values.")  # JD:88-89
        self.vocab_size = vocab_size
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(device)
        self.debug = os.environ.get("JACOBI_DEBUG", "0") == "1"
        self.stats = {"num_chunk_calls": 0, "num_jacobi_iterations": 0, "tokens_accepted": 0, "tokens_per_call": [],
                      "tokens_per_iteration": [], "iterations_per_call": []}                        # JD:101-108
        self._pad_stream_host: Optional[np.ndarray] = None
        self._pad_cursor = 0
        self._stepper: Optional[ops.EngineStepper] = None

    # ----------------------------------------------------------------------------------- random pads
    def set_pad_stream(self, stream) -> None:
        """Inject the random-pad stream (JD draws ``torch.randint(0, vocab_size)`` at JD:169/344/434/542/707; the stream
        is consumed in exactly that order).  By default one is drawn from torch's generator on first use."""
        self._pad_stream_host = np.asarray(stream, dtype=np.int64).copy()
        self._pad_cursor = 0
        self._stepper = None

    def _ensure(self, B: int, L: int) -> ops.EngineStepper:
        if self._pad_stream_host is None:
            self._pad_stream_host = torch.randint(0, self.vocab_size, (_PAD_STREAM_LEN,)).numpy().astype(np.int64)
        st = self._stepper
        if st is None or st.max_rows < B or st.max_L < L:
            cur = self._pad_cursor
            st = ops.EngineStepper(max(B, 8 if st is None else st.max_rows), max(L, 64 if st is None else st.max_L),
                                   self.device, torch.from_numpy(self._pad_stream_host))
            st.pad_cursor.fill_(cur)
            self._stepper = st
        return st

    def _host_pads(self, count: int) -> List[int]:
        s = self._pad_stream_host
        idx = (self._pad_cursor + np.arange(count)) % len(s)
        self._pad_cursor += count
        return s[idx].tolist()

    # ----------------------------------------------------------------------------------- config helpers
    def _get_sampling_cfg(self, seq: Sequence) -> Tuple[int, int]:
        sp = getattr(seq, "sampling_params", None)
        g = lambda name, default: getattr(sp, name, default) if sp is not None else default
        return int(g("jacobi_block_len", 64)), int(g("jacobi_max_iterations", 128))

    def _first_draft(self, seq: Sequence, L: int) -> List[int]:
        """JD:332-347 / JD:529-546 (prefill draft) or JD:142-171 (random init)."""
        self._ensure(1, L)
        d = [seq.token_ids[-1]]
        pd = getattr(seq, "_prefill_draft", None)
        if pd is not None:
            k = min(len(pd), L - 1)
            d += [int(t) for t in pd[:k]]
            if k < L - 1:
                d += self._host_pads(L - 1 - k)
            seq._prefill_draft = None
        elif L > 1:
            d += self._host_pads(L - 1)
        return d

    def _forward_batched(self, seqs: List[Sequence], draft_batch: Tensor) -> Tensor:
        """JD:212-249."""
        if draft_batch.dim() != 2:
            raise ValueError(f"draft_batch must be [B, L], got {tuple(draft_batch.shape)}")
        B, L = int(draft_batch.size(0)), int(draft_batch.size(1))
        if B != len(seqs):
            raise ValueError(f"B mismatch: got draft_batch B={B} but len(seqs)={len(seqs)}")
        if self.forward_step_batch is not None:
            logits = self.forward_step_batch(seqs, draft_batch)
        else:
            logits = torch.cat([self.forward_step(s, draft_batch[i:i + 1, :]) for i, s in enumerate(seqs)], dim=0)
        if logits.ndim != 3 or logits.size(0) != B or logits.size(1) != (L - 1):
            raise ValueError(f"forward must return logits [B, L-1, vocab] for verifying speculative tokens, "
                             f"expected [{B}, {L - 1}, *], got {tuple(logits.shape)}")
        return logits

    @staticmethod
    def _accept_lengths(draft: Tensor, greedy: Tensor) -> Tensor:
        """JD:253-293 (kept for callers that used the static helper): HIP accepted-prefix scan."""
        B, L = draft.shape
        if L == 0:
            return torch.zeros((B,), device=draft.device, dtype=torch.long)
        if L == 1:
            return torch.ones((B,), device=draft.device, dtype=torch.long)
        if greedy.size(1) != L - 1:
            raise ValueError(f"Expected greedy.shape[1]={L - 1}, got {greedy.size(1)}")
        acc, _ = ops.accept_lengths(draft, greedy)
        return acc.long()

    # ----------------------------------------------------------------------------------- public API
    @torch.inference_mode()
    def generate_chunk(self, seq: Sequence) -> List[int]:
        return self._run([seq], single=True)[0]

    @torch.inference_mode()
    def generate_chunk_batch(self, seqs: List[Sequence]) -> List[List[int]]:
        if not seqs:
            return []
        if len(seqs) == 1:
            return [self.generate_chunk(seqs[0])]                                              # JD:453-454
        return self._run(seqs, single=False)

    # ----------------------------------------------------------------------------------- core loop (JD:302-724)
    def _run(self, seqs: List[Sequence], single: bool) -> List[List[int]]:
        B = len(seqs)
        accepted: List[List[int]] = [[] for _ in range(B)]
        q_draft: List[Optional[Tensor]] = [None] * B
        eos_reached = [False] * B
        iters = [0] * B
        cfg = [self._get_sampling_cfg(s) for s in seqs]
        block_lens, max_iters = [c[0] for c in cfg], [c[1] for c in cfg]
        max_tokens = []
        for seq in seqs:
            sp = getattr(seq, "sampling_params", None)
            if sp is not None:
                rem = getattr(sp, "max_tokens", 2048) - seq.num_completion_tokens
                max_tokens.append(rem if single else max(0, rem))
            else:
                max_tokens.append(2048)
        if single and block_lens[0] <= 1:
            return [[]]                                                                        # JD:313-314
        n_iter_call = 0
        prev_len = [0] * B
        dev = self.device
        prof = getattr(self, "profiler", None)        # ModelRunner's PROFILE=1 section timer (reference names, MR:116-134)
        tick = (lambda name, on: (prof.start(name) if on else prof.stop(name))) if prof is not None else (lambda name, on: None)
        while True:
            active = [i for i in range(B) if not eos_reached[i] and len(accepted[i]) < max_tokens[i] and iters[i] < max_iters[i]]
            if not active:
                break
            groups = {}
            for i in active:
                if block_lens[i] > 1:
                    groups.setdefault(block_lens[i], []).append(i)
            if not groups:
                break
            n_iter_call += 1
            tokens_this_iter = 0
            for L, idxs in sorted(groups.items(), key=lambda x: len(x[1]), reverse=True):       # JD:513
                rows_t = []
                for i in idxs:
                    iters[i] += 1
                    if q_draft[i] is None:
                        q_draft[i] = torch.tensor(self._first_draft(seqs[i], L), dtype=torch.int64, device=dev)
                    rows_t.append(q_draft[i])
                    seqs[i].draft_tokens = None
                draft_batch = torch.stack(rows_t, 0)
                sub = [seqs[i] for i in idxs]
                if single:
                    sub[0].draft_tokens = draft_batch[0].tolist()                              # JD:351
                logits = self._forward_batched(sub, draft_batch)
                tick("jacobi.verify", True)
                st = self._ensure(len(idxs), L)
                st.pad_cursor.fill_(self._pad_cursor)
                rows, new_tokens, next_draft = st.step(draft_batch, logits, self.eos_token_id,
                                                       [max_tokens[i] - len(accepted[i]) for i in idxs])
                tick("jacobi.verify", False)
                tick("jacobi.commit", True)
                for row, i in enumerate(idxs):
                    seq = sub[row]
                    acc_len, n_new, eos, active_next, n_pads = (int(x) for x in rows[row][:5])
                    toks = [int(t) for t in new_tokens[row, :n_new]]
                    if acc_len > 1:                                                            # JD:609-614
                        seq.extend_tokens(toks)
                        if self.block_manager is not None:
                            self.block_manager.may_append_batch(seq, acc_len - 1)
                        num_spec = acc_len - 1
                    else:                                                                      # JD:619-631
                        seq.append_token(toks[0])
                        if self.block_manager is not None:
                            self.block_manager.may_append(seq)
                        num_spec = 1
                    accepted[i].extend(toks)
                    if eos:
                        eos_reached[i] = True
                    tokens_this_iter += len(accepted[i]) - prev_len[i]
                    prev_len[i] = len(accepted[i])
                    trim = L - 1 - num_spec                                                    # JD:638-646
                    if trim > 0 and self.block_manager is not None:
                        self.block_manager.trim_kv_only_fast(seq, trim)
                    seq.clear_draft()
                    if len(seq) != seq.num_cached_tokens:                                       # JD:651-654
                        raise RuntimeError(f"Invariant violated: len(token_ids)={len(seq)} != num_cached_tokens={seq.num_cached_tokens}")
                    self._pad_cursor += n_pads
                    q_draft[i] = next_draft[row].clone() if active_next else None
                tick("jacobi.commit", False)
                if prof is not None:
                    prof.iterations += 1; prof.tokens += tokens_this_iter
            if not single:
                self.stats["tokens_per_iteration"].append(tokens_this_iter)
        total = sum(len(a) for a in accepted)
        self.stats["num_chunk_calls"] += 1
        self.stats["num_jacobi_iterations"] += iters[0] if single else n_iter_call
        self.stats["tokens_accepted"] += total
        self.stats["tokens_per_call"].append(total)
        self.stats["iterations_per_call"].append(iters[0] if single else n_iter_call)
        return accepted
