"""Engine single-block NON-greedy Jacobi decoder: rejection-sampling verification (speculative-decoding style with a
delta proposal) — the API and semantics of the reference's ``JacobiDecoderNonGreedy``
(inference_engine/engine/jacobi_decoding_nongreedy.py:156-667 = "JDN").

Per iteration: ``jf_rs_probs`` reads the logits once (softmax-gather of the drafted ids + argmax), ``jf_rs_step`` runs the
sequential accept/reject of every row, the residual ("bonus") draw and the next draft in one launch, ``jf_engine_loop_commit``
keeps the loop's state on the device (engine/chunk_loop.py); one polled record per iteration.
Randomness (JDN:329 ``torch.rand``, JDN:132 ``torch.multinomial``, JDN:240/463 ``torch.randint``) comes from three
pre-drawn streams consumed in the reference's order; ``set_streams`` injects them for reproducible runs.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import _native as N
from .. import ops
from .block_manager import BlockManager
from .chunk_loop import ChunkLoopMixin
from .jacobi_decoding import LogitsForwardFn, LogitsForwardFnBatch
from .sequence import Sequence

_STREAM_LEN = 1 << 16


class JacobiDecoderNonGreedy(ChunkLoopMixin):
    KIND = N.EL_KIND_SAMPLING

    def __init__(self, block_manager: BlockManager, forward_step: Optional[LogitsForwardFn] = None,
                 forward_step_batch: Optional[LogitsForwardFnBatch] = None, eos_token_id: Optional[int] = None,
                 pad_token_id: Optional[int] = None, vocab_size: Optional[int] = None,
                 device: Optional[torch.device] = None, forward_step_loop=None) -> None:
        """``forward_step_loop`` (new, optional): see JacobiDecoder."""
        if forward_step is None and forward_step_batch is None and forward_step_loop is None:
            raise ValueError("Provide at least one of forward_step or forward_step_batch.")
        self.block_manager = block_manager
        self.forward_step = forward_step
        self.forward_step_batch = forward_step_batch
        self.forward_step_loop = forward_step_loop
        self.eos_token_id = eos_token_id
        self.pad_token_id = pad_token_id
        if vocab_size is None:
            raise ValueError("vocab_size must be provided from model config. Do not use hard-coded values.")
        self.vocab_size = vocab_size
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(device)
        self.debug = os.environ.get("JACOBI_DEBUG", "0") == "1"
        self.stats: Dict[str, object] = {"num_chunk_calls": 0, "num_jacobi_iterations": 0, "tokens_accepted": 0,
                                         "tokens_per_call": [], "tokens_per_iteration": [], "iterations_per_call": []}
        self._pads = self._unis = self._bonus = None
        self._own_streams = False        # the streams are the decoder's own draws (not injected): replaced before they would wrap
        self._cur = [0, 0, 0]            # uniforms, bonus, pads: host mirror of the stepper's device cursors (refreshed by the iteration record)
        self._cursor_dirty = True
        self._stepper: Optional[ops.RsStepper] = None

    def set_streams(self, pads, uniforms, bonus) -> None:
        """Inject the three random streams (consumed cyclically, in the reference's order of torch.randint / torch.rand /
        torch.multinomial calls): reproducible runs.  Without it the decoder draws its own from torch's generator and
        replaces them before a cursor would wrap (a stream that wraps repeats its uniforms every few dozen iterations
        at batch 64 x block 32)."""
        self._own_streams = False
        self._pads = np.asarray(pads, dtype=np.int64).copy()
        self._unis = np.asarray(uniforms, dtype=np.float32).copy()
        self._bonus = np.asarray(bonus, dtype=np.float32).copy()
        self._cur = [0, 0, 0]
        self._cursor_dirty = True
        self._stepper = None

    def _ensure(self, B: int, L: int) -> ops.RsStepper:
        if self._pads is None:
            self._draw_streams(max(_STREAM_LEN, 8 * B * max(L, 16)))
        st = self._stepper
        if st is None or st.max_rows < B or st.max_L < L:
            st = ops.RsStepper(max(B, 8 if st is None else st.max_rows), max(L, 64 if st is None else st.max_L), self.device,
                               self._pads, self._unis, self._bonus)
            self._cursor_dirty = True
            self._stepper = st
        return st

    def _draw_streams(self, n: int) -> None:
        self.set_streams(torch.randint(0, self.vocab_size, (n,)).numpy(), torch.rand(n).numpy(), torch.rand(n).numpy())
        self._own_streams = True

    def _refresh_streams(self, st: ops.RsStepper, B: int, L: int) -> None:
        """Fresh draws in place of the decoder's own streams when the next step could run past their end (a step consumes at
        most B (L-1) uniforms and pads and 16 B bonus draws)."""
        need, n = B * max(L, 16), len(self._pads)
        if not self._own_streams or max(self._cur) + need <= n:
            return
        keep = self._stepper
        self._draw_streams(n)                                 # (the same length: the stepper's device copies are refilled in place)
        self._stepper = keep                                  # (set_streams drops the stepper)
        st.pad_stream.copy_(torch.from_numpy(self._pads))
        st.u_stream.copy_(torch.from_numpy(self._unis))
        st.bonus_stream.copy_(torch.from_numpy(self._bonus))

    def _host_pads(self, count: int) -> List[int]:
        idx = (self._cur[2] + np.arange(count)) % len(self._pads)
        self._cur[2] += count
        self._cursor_dirty = True
        return self._pads[idx].tolist()

    def _get_sampling_cfg(self, seq: Sequence) -> Tuple[int, int]:
        sp = getattr(seq, "sampling_params", None)
        g = lambda name, default: getattr(sp, name, default) if sp is not None else default
        return int(g("jacobi_block_len", 64)), int(g("jacobi_max_iterations", 128))

    def _forward_batched(self, seqs: List[Sequence], draft_batch: Tensor) -> Tensor:
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
            raise ValueError(f"forward must return logits [B, L-1, vocab], expected [{B}, {L - 1}, *], got {tuple(logits.shape)}")
        return logits

    @torch.inference_mode()
    def generate_chunk(self, seq: Sequence, return_metrics: bool = False):
        toks, met = self._run([seq], single=True)
        return (toks[0], met[0]) if return_metrics else toks[0]

    @torch.inference_mode()
    def generate_chunk_batch(self, seqs: List[Sequence], return_metrics: bool = False):
        if not seqs:
            return ([], []) if return_metrics else []
        if len(seqs) == 1:
            toks, met = self._run(seqs, single=True)
            return (toks, met) if return_metrics else toks
        # One distribution setting (temperature, top_k, top_p) per launch: the reference builds the target distribution request
        # by request (JDN:110-123, _verify_block_rejection_sampling(seq, ...)), so a batch may mix settings there.  Here such a
        # batch is decoded setting by setting (each part is an ordinary batch with its own forwards); the random streams are
        # then consumed part after part instead of interleaved — the tokens are samples of the same distributions.
        parts: Dict[Tuple[float, int, float], List[int]] = {}
        for i, seq in enumerate(seqs):
            parts.setdefault(self._setting(seq), []).append(i)
        if len(parts) == 1:
            toks, met = self._run(seqs, single=False)
            return (toks, met) if return_metrics else toks
        toks: List[List[int]] = [[] for _ in seqs]
        met: List[dict] = [{} for _ in seqs]
        for idxs in parts.values():
            t, m = self._run([seqs[i] for i in idxs], single=False)
            for k, i in enumerate(idxs):
                toks[i], met[i] = t[k], m[k]
        return (toks, met) if return_metrics else toks

    def _setting(self, seq: Sequence) -> Tuple[float, int, float]:
        """(temperature, top_k, top_p) of a request as the verify launches take them; top_k / top_p are planted on the request
        object by the caller (JDN:117-118 reads them with getattr)."""
        sp = getattr(seq, "sampling_params", None)
        k, p = ops.active_filters(sp, self.vocab_size)
        return float(getattr(sp, "temperature", 1.0)), k, p

    def _run(self, seqs: List[Sequence], single: bool):
        accepted, iters, forwards, n_iter_call = self._run_chunk(seqs, single)
        total = sum(len(a) for a in accepted)
        self.stats["num_chunk_calls"] = int(self.stats["num_chunk_calls"]) + 1
        self.stats["num_jacobi_iterations"] = int(self.stats["num_jacobi_iterations"]) + (int(iters[0]) if single else n_iter_call)
        self.stats["tokens_accepted"] = int(self.stats["tokens_accepted"]) + total
        self.stats["tokens_per_call"].append(total)
        self.stats["iterations_per_call"].append(int(iters[0]) if single else n_iter_call)
        metrics = []
        for i in range(len(seqs)):
            tok, it, fw = float(len(accepted[i])), float(iters[i]), float(forwards[i])
            m = {"tokens_per_iter": tok / it if it > 0 else 0.0, "tokens_per_forward": tok / fw if fw > 0 else 0.0,
                 "num_iters": it, "num_forwards": fw}
            if not single:
                m["total_tokens"] = tok
            metrics.append(m)
        return accepted, metrics

    # ---- chunk-loop hooks (engine/chunk_loop.py): the iteration body is jf_rs_probs [+ jf_rs_filter] + jf_rs_step + the commit launch
    def _chunk_context(self, seqs: List[Sequence]) -> Tuple[float, int, float]:
        settings = {self._setting(s) for s in seqs}
        if len(settings) > 1:                                   # (generate_chunk_batch splits such batches)
            raise NotImplementedError(f"one (temperature, top_k, top_p) setting per chunk, got {sorted(settings)}")
        return next(iter(settings))

    def _first_draft(self, seq: Sequence, L: int) -> List[int]:
        self._ensure(1, L)
        return [seq.token_ids[-1]] + self._host_pads(L - 1)                                    # JDN:222-241 random init (no prefill draft)

    def _no_groups(self, single: bool, iters: np.ndarray) -> None:
        if single:
            iters[0] += 1                                                                      # JDN:391-396 counts the iteration, then breaks

    def _push_cursors(self, st: ops.RsStepper) -> None:
        self._refresh_streams(st, st.max_rows, st.max_L)
        if self._cursor_dirty:                               # the host drew pads for a first draft (or the stepper is new)
            st.cursors.copy_(torch.tensor(self._cur, dtype=torch.int64), non_blocking=True)
            self._cursor_dirty = False

    def _enqueue_step(self, st: ops.RsStepper, lp: ops.EngineLoop, logits: Tensor, ctx) -> Tensor:
        temperature, top_k, top_p = ctx
        st.step_loop(lp, logits, temperature, self.eos_token_id, top_k, top_p)
        return st.committed.view(-1)[:lp.B * lp.L].view(lp.B, lp.L)

    def _pull_cursors(self, lp: ops.EngineLoop) -> None:
        self._cur = list(lp.cursors_host[:3])

    def _commit_row(self, seq: Sequence, toks: List[int], fallback: bool, L: int) -> None:
        """What JDN:417-429 / 592-605 does to one request after a step, for callers whose callbacks read the request objects."""
        bm = self.block_manager
        if toks:
            seq.extend_tokens(toks)
            if bm is not None:
                bm.may_append_batch(seq, len(toks))
        trim = (L - 1) - len(toks)
        if trim > 0 and bm is not None:
            bm.trim_kv_only_fast(seq, trim)
        seq.clear_draft()
        if len(seq) != seq.num_cached_tokens:
            raise RuntimeError(f"Invariant violated: len(token_ids)={len(seq)} != num_cached_tokens={seq.num_cached_tokens}")
