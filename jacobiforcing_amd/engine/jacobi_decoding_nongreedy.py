"""Engine single-block NON-greedy Jacobi decoder: rejection-sampling verification (speculative-decoding style with a
delta proposal) — the API and semantics of the reference's ``JacobiDecoderNonGreedy``
(inference_engine/engine/jacobi_decoding_nongreedy.py:156-667 = "JDN").

Per iteration: ``jf_rs_probs`` reads the logits once (softmax-gather of the drafted ids + argmax), ``jf_rs_step`` runs the
sequential accept/reject of every row, the residual ("bonus") draw and the next draft in one launch; one read-back.
Randomness (JDN:329 ``torch.rand``, JDN:132 ``torch.multinomial``, JDN:240/463 ``torch.randint``) comes from three
pre-drawn streams consumed in the reference's order; ``set_streams`` injects them for reproducible runs.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import ops
from .block_manager import BlockManager
from .jacobi_decoding import LogitsForwardFn, LogitsForwardFnBatch
from .sequence import Sequence

_STREAM_LEN = 1 << 16


class JacobiDecoderNonGreedy:
    def __init__(self, block_manager: BlockManager, forward_step: Optional[LogitsForwardFn] = None,
                 forward_step_batch: Optional[LogitsForwardFnBatch] = None, eos_token_id: Optional[int] = None,
                 pad_token_id: Optional[int] = None, vocab_size: Optional[int] = None,
                 device: Optional[torch.device] = None) -> None:
        if forward_step is None and forward_step_batch is None:
            raise ValueError("Provide at least one of forward_step or forward_step_batch.")
        self.block_manager = block_manager
        self.forward_step = forward_step
        self.forward_step_batch = forward_step_batch
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
        self._cur = [0, 0, 0]            # uniforms, bonus, pads
        self._stepper: Optional[ops.RsStepper] = None

    def set_streams(self, pads, uniforms, bonus) -> None:
        self._pads = np.asarray(pads, dtype=np.int64).copy()
        self._unis = np.asarray(uniforms, dtype=np.float32).copy()
        self._bonus = np.asarray(bonus, dtype=np.float32).copy()
        self._cur = [0, 0, 0]
        self._stepper = None

    def _ensure(self, B: int, L: int) -> ops.RsStepper:
        if self._pads is None:
            self.set_streams(torch.randint(0, self.vocab_size, (_STREAM_LEN,)).numpy(), torch.rand(_STREAM_LEN).numpy(),
                             torch.rand(_STREAM_LEN).numpy())
        st = self._stepper
        if st is None or st.max_rows < B or st.max_L < L:
            st = ops.RsStepper(max(B, 8 if st is None else st.max_rows), max(L, 64 if st is None else st.max_L), self.device,
                               self._pads, self._unis, self._bonus)
            self._stepper = st
        return st

    def _host_pads(self, count: int) -> List[int]:
        idx = (self._cur[2] + np.arange(count)) % len(self._pads)
        self._cur[2] += count
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
        toks, met = self._run(seqs, single=False)
        return (toks, met) if return_metrics else toks

    def _run(self, seqs: List[Sequence], single: bool):
        B = len(seqs)
        accepted: List[List[int]] = [[] for _ in range(B)]
        q_draft: List[Optional[Tensor]] = [None] * B
        eos_reached = [False] * B
        iters = [0] * B
        forwards = [0] * B
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
        temperature = float(getattr(getattr(seqs[0], "sampling_params", None), "temperature", 1.0))
        # top_k / top_p planted on the request objects (JDN:117-118 reads them with getattr): one setting per call, like the
        # temperature above (the reference builds the distribution sequence by sequence; a batch that mixes settings is split
        # by the caller)
        filters = {ops.active_filters(getattr(seq, "sampling_params", None), self.vocab_size) for seq in seqs}
        if len(filters) > 1:
            raise NotImplementedError(f"one top_k / top_p setting per generate_chunk_batch call, got {sorted(filters)}")
        top_k, top_p = next(iter(filters))
        n_iter_call = 0
        dev = self.device
        prof = getattr(self, "profiler", None)        # ModelRunner's PROFILE=1 section timer (reference names, MR:116-134)
        tick = (lambda name, on: (prof.start(name) if on else prof.stop(name))) if prof is not None else (lambda name, on: None)
        while True:
            active = [i for i in range(B) if not eos_reached[i] and len(accepted[i]) < max_tokens[i] and iters[i] < max_iters[i]]
            if not active:
                break
            groups: Dict[int, List[int]] = {}
            for i in active:
                if block_lens[i] > 1:
                    groups.setdefault(block_lens[i], []).append(i)
            if not groups:
                if single:
                    iters[0] += 1                                                 # JDN:391-396 counts the iteration, then breaks
                break
            n_iter_call += 1
            tokens_this_iter = 0
            for L, idxs in sorted(groups.items(), key=lambda x: len(x[1]), reverse=True):
                self._ensure(len(idxs), L)
                rows_t = []
                for i in idxs:
                    iters[i] += 1
                    if q_draft[i] is None:                                       # JDN:222-241 random init (no prefill draft)
                        q_draft[i] = torch.tensor([seqs[i].token_ids[-1]] + self._host_pads(L - 1), dtype=torch.int64, device=dev)
                    rows_t.append(q_draft[i])
                draft_batch = torch.stack(rows_t, 0)
                sub = [seqs[i] for i in idxs]
                for row, i in enumerate(idxs):
                    sub[row].draft_tokens = None
                logits = self._forward_batched(sub, draft_batch)
                for i in idxs:
                    forwards[i] += 1
                tick("jacobi.verify", True)
                st = self._ensure(len(idxs), L)
                rows, committed, next_draft = st.step(draft_batch, logits, temperature, self.eos_token_id,
                                                      [max_tokens[i] - len(accepted[i]) for i in idxs], self._cur, top_k, top_p)
                tick("jacobi.verify", False)
                tick("jacobi.commit", True)
                for row, i in enumerate(idxs):
                    seq = sub[row]
                    n_c, eos, _rej, n_b, n_u, n_p, act, _ = (int(x) for x in rows[row])
                    toks = [int(t) for t in committed[row, :n_c]]
                    if toks:                                                     # JDN:417-424 / 592-599
                        seq.extend_tokens(toks)
                        if self.block_manager is not None:
                            self.block_manager.may_append_batch(seq, len(toks))
                        accepted[i].extend(toks)
                        tokens_this_iter += len(toks)
                    if eos:
                        eos_reached[i] = True
                    trim = (L - 1) - n_c                                          # JDN:427-429 / 601-605
                    if trim > 0 and self.block_manager is not None:
                        self.block_manager.trim_kv_only_fast(seq, trim)
                    seq.clear_draft()
                    if len(seq) != seq.num_cached_tokens:
                        raise RuntimeError(f"Invariant violated: len(token_ids)={len(seq)} != num_cached_tokens={seq.num_cached_tokens}")
                    self._cur[0] += n_u
                    self._cur[1] += n_b
                    self._cur[2] += n_p
                    q_draft[i] = next_draft[row].clone() if act else None
                tick("jacobi.commit", False)
                if prof is not None:
                    prof.iterations += 1; prof.tokens += tokens_this_iter
            if not single:
                self.stats["tokens_per_iteration"].append(tokens_this_iter)
        total = sum(len(a) for a in accepted)
        self.stats["num_chunk_calls"] = int(self.stats["num_chunk_calls"]) + 1
        self.stats["num_jacobi_iterations"] = int(self.stats["num_jacobi_iterations"]) + (iters[0] if single else n_iter_call)
        self.stats["tokens_accepted"] = int(self.stats["tokens_accepted"]) + total
        self.stats["tokens_per_call"].append(total)
        self.stats["iterations_per_call"].append(iters[0] if single else n_iter_call)
        metrics = []
        for i in range(B):
            tok, it, fw = float(len(accepted[i])), float(iters[i]), float(forwards[i])
            m = {"tokens_per_iter": tok / it if it > 0 else 0.0, "tokens_per_forward": tok / fw if fw > 0 else 0.0,
                 "num_iters": it, "num_forwards": fw}
            if not single:
                m["total_tokens"] = tok
            metrics.append(m)
        return accepted, metrics
