"""Engine single-block greedy Jacobi decoder — same constructor, callbacks, return values and ``stats`` as the
reference's ``JacobiDecoder`` (inference_engine/engine/jacobi_decoding.py:47-724 = "JD"), with the per-iteration body
(argmax, accept scan, EOS cap, commit, AR fallback, next draft incl. random pads) in one HIP launch
(``jf_argmax_partial`` + ``jf_engine_step``), the loop around it on device arrays (``jf_engine_loop_commit``:
engine/chunk_loop.py) and one polled record per iteration instead of JD's per-row ``.item()`` / ``.tolist()`` syncs.

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

from .. import _native as N
from .. import ops
from .block_manager import BlockManager
from .chunk_loop import ChunkLoopMixin
from .sequence import Sequence

LogitsForwardFn = Callable[[Sequence, Tensor], Tensor]
LogitsForwardFnBatch = Callable[[List[Sequence], Tensor], Tensor]

_PAD_STREAM_LEN = 1 << 16


class JacobiDecoder(ChunkLoopMixin):
    KIND = N.EL_KIND_GREEDY

    def __init__(self, block_manager: BlockManager, forward_step: Optional[LogitsForwardFn] = None,
                 forward_step_batch: Optional[LogitsForwardFnBatch] = None, eos_token_id: Optional[int] = None,
                 pad_token_id: Optional[int] = None, vocab_size: Optional[int] = None,
                 device: Optional[torch.device] = None, forward_step_loop=None) -> None:
        """``forward_step_loop`` (new, optional): ``f(loop: ops.EngineLoop) -> logits [B, L-1, V]`` — a forward that takes the
        draft, the positions and the cached lengths from the loop's DEVICE arrays instead of request objects; with it the
        decoder touches no request object between the first and the last iteration of a chunk (engine/chunk_loop.py)."""
        if forward_step is None and forward_step_batch is None and forward_step_loop is None:
            raise ValueError("Provide at least one of forward_step or forward_step_batch.")          # JD:78-79
        self.block_manager = block_manager
        self.forward_step = forward_step
        self.forward_step_batch = forward_step_batch
        self.forward_step_loop = forward_step_loop
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
        self._pad_cursor = 0                 # host mirror of the stepper's device cursor (the iteration record refreshes it)
        self._cursor_dirty = True
        self._stepper: Optional[ops.EngineStepper] = None

    # ----------------------------------------------------------------------------------- random pads
    def set_pad_stream(self, stream) -> None:
        """Inject the random-pad stream (JD draws ``torch.randint(0, vocab_size)`` at JD:169/344/434/542/707; the stream
        is consumed in exactly that order).  By default one is drawn from torch's generator on first use."""
        self._pad_stream_host = np.asarray(stream, dtype=np.int64).copy()
        self._pad_cursor = 0
        self._cursor_dirty = True
        self._stepper = None

    def _ensure(self, B: int, L: int) -> ops.EngineStepper:
        if self._pad_stream_host is None:
            self._pad_stream_host = torch.randint(0, self.vocab_size, (_PAD_STREAM_LEN,)).numpy().astype(np.int64)
        st = self._stepper
        if st is None or st.max_rows < B or st.max_L < L:
            st = ops.EngineStepper(max(B, 8 if st is None else st.max_rows), max(L, 64 if st is None else st.max_L),
                                   self.device, torch.from_numpy(self._pad_stream_host))
            self._cursor_dirty = True
            self._stepper = st
        return st

    def _host_pads(self, count: int) -> List[int]:
        s = self._pad_stream_host
        idx = (self._pad_cursor + np.arange(count)) % len(s)
        self._pad_cursor += count
        self._cursor_dirty = True
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
        if single and self._get_sampling_cfg(seqs[0])[0] <= 1:
            return [[]]                                                                        # JD:313-314
        accepted, iters, _forwards, n_iter_call = self._run_chunk(seqs, single)
        total = sum(len(a) for a in accepted)
        self.stats["num_chunk_calls"] += 1
        self.stats["num_jacobi_iterations"] += int(iters[0]) if single else n_iter_call
        self.stats["tokens_accepted"] += total
        self.stats["tokens_per_call"].append(total)
        self.stats["iterations_per_call"].append(int(iters[0]) if single else n_iter_call)
        return accepted

    # ---- chunk-loop hooks (engine/chunk_loop.py): the iteration body is jf_argmax_partial + jf_engine_step + the commit launch
    def _push_cursors(self, st: ops.EngineStepper) -> None:
        if self._cursor_dirty:                               # the host drew pads for a first draft (or the stepper is new)
            st.pad_cursor.fill_(self._pad_cursor)
            self._cursor_dirty = False

    def _enqueue_step(self, st: ops.EngineStepper, lp: ops.EngineLoop, logits: Tensor, ctx) -> Tensor:
        st.step_loop(lp, logits, self.eos_token_id)
        return st.new_tokens.view(-1)[:lp.B * lp.L].view(lp.B, lp.L)

    def _pull_cursors(self, lp: ops.EngineLoop) -> None:
        self._pad_cursor = lp.cursors_host[0]

    def _commit_row(self, seq: Sequence, toks: List[int], fallback: bool, L: int) -> None:
        """What JD:609-654 does to one request after a step, for callers whose callbacks read the request objects."""
        bm = self.block_manager
        if not fallback:                                                                       # JD:609-614
            seq.extend_tokens(toks)
            if bm is not None:
                bm.may_append_batch(seq, len(toks))
            num_spec = len(toks)
        else:                                                                                  # JD:619-631
            seq.append_token(toks[0])
            if bm is not None:
                bm.may_append(seq)
            num_spec = 1
        trim = L - 1 - num_spec                                                                # JD:638-646
        if trim > 0 and bm is not None:
            bm.trim_kv_only_fast(seq, trim)
        seq.clear_draft()
        if len(seq) != seq.num_cached_tokens:                                                   # JD:651-654
            raise RuntimeError(f"Invariant violated: len(token_ids)={len(seq)} != num_cached_tokens={seq.num_cached_tokens}")
