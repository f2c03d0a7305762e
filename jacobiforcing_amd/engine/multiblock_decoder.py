"""Multiblock Jacobi decoding with rejection recycling for a batch of prompts.

Host orchestration of the hot path the north star names: per Jacobi iteration one PyTorch forward
over every prompt's rows, then ``jf_argmax_partial`` + ``jf_mb_step`` (verify, accept, re-draft,
pool/candidates, spawn/promote) + ``jf_kv_commit`` in HIP, and ONE small descriptor read-back.

The per-prompt semantics are those of the reference's ``jacobi_forward_greedy_multiblock``
(MB:140-740) and its driver loop (JacobiForcing/jacobi_forcing_inference_MR_humaneval.py:152-273 =
"DRV"): prefill with a random draft, first call seeded with the prefill n-gram, later calls with
``[first_correct_token] + n-1 tokens drawn from the text so far``; stop on EOS / max_new_tokens /
max_calls; tokens-per-second excludes the prefill and counts ``new_tokens - 1`` (DRV:243).
The reference runs one prompt at a time; here P prompts share a forward, each with its own state
machine and rolling call restarts (BASELINE config 4).
"""
from __future__ import annotations

import os

import random
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from .. import _native as N
from .. import ops
from ..modeling.qwen2 import Qwen2Model, StaticKVCache


@dataclass
class PromptStats:
    """One row of the reference driver's CSV (DRV:259-273)."""
    prompt_tokens: int = 0
    new_tokens: int = 0
    calls: int = 0
    total_iterations: int = 0
    time_sec: float = 0.0
    stop_reason: Optional[str] = None
    token_ids: List[int] = field(default_factory=list)

    @property
    def avg_iter_per_call(self):
        return self.total_iterations / self.calls if self.calls else 0.0

    @property
    def avg_iter_per_token(self):
        return self.total_iterations / self.new_tokens if self.new_tokens else 0.0

    def row(self, gen_time: float) -> dict:
        return dict(prompt_tokens=self.prompt_tokens, new_tokens=self.new_tokens, calls=self.calls,
                    total_iterations=self.total_iterations, avg_iter_per_call=self.avg_iter_per_call,
                    avg_iter_per_token=self.avg_iter_per_token, time_sec=self.time_sec,
                    toks_per_sec=(self.new_tokens / gen_time) if gen_time > 0 else 0.0, stop_reason=self.stop_reason)


LogitsHook = Callable[[torch.Tensor, "MultiblockJacobiDecoder"], torch.Tensor]


class MultiblockJacobiDecoder:
    def __init__(self, model: Qwen2Model, num_prompts: int, params: ops.MultiblockParams, max_seq_len: int = 4096,
                 logits_hook: Optional[LogitsHook] = None, t_align: int = 1, compact_logits: bool = True,
                 logit_align: Optional[int] = None):
        self.model = model
        self.P = int(num_prompts)
        self.params = params
        self.device = model.device
        self.batch = ops.MultiblockBatch(self.P, params, self.device)
        self.cand_rows = self.batch.max_rows - 1
        # rows longer than (K+2)*n only occur when the reference's block counters run away; the forward buffers are sized
        # for the realistic case and a longer row is a capacity error rather than gigabytes of idle scratch
        self.t_cap = min(self.batch.max_tokens, max(128, (params.K + 2) * params.n))
        self.cache = StaticKVCache(model.cfg, self.P, max_seq_len, self.cand_rows, self.t_cap, self.device, dtype=model.dtype)
        self.max_seq_len = max_seq_len
        self.logits_hook = logits_hook
        self.t_align = int(t_align)          # pad the per-iteration row length to a multiple (tuned-GEMM shape grid)
        self.compact_logits = bool(compact_logits)   # lm_head + argmax on draft-carrying positions only (no padding rows)
        self.logit_align = int(logit_align) if logit_align else self.t_align   # lm_head M rounded up to this multiple
        self.kv_len_host = np.zeros(self.P, dtype=np.int64)
        self._kv_len_pin = torch.zeros((self.P,), dtype=torch.int32, pin_memory=self.device.type == "cuda")
        self.forwards = 0
        self.last_logits_rows = 0
        self.last_valid_rows = 0             # sum_p B_p * T_p of the last forward (algorithmic rows, without padding)
        self.profiler = None                 # optional ProfileTimer (PROFILE=1 sections, reference names MR:116-134)
        self._f = {k: N.DESC_FIELDS.index(k) for k in N.DESC_FIELDS}

    # ------------------------------------------------------------------------------ prefill (MB:175-225)
    @torch.inference_mode()
    def prefill(self, prompts: Sequence[Sequence[int]], drafts: Sequence[Sequence[int]]) -> List[List[int]]:
        """Forward prompt ⧺ draft per prompt, first n-gram = argmax(logits[-n-1:-1]); KV is cut back to the prompt.
        lm_head runs on the last n+1 positions only (the reference computes all S+n rows, MB:216)."""
        n = self.params.n
        dev = self.device
        rows = [list(prompt) + list(draft) for prompt, draft in zip(prompts, drafts)]
        for p, row in enumerate(rows):
            if len(row) > self.max_seq_len:
                raise RuntimeError(f"prompt {p}: {len(row)} tokens exceed max_seq_len={self.max_seq_len}")
        ngrams: List[List[int]] = [None] * len(rows)
        budget = int(os.environ.get("JF_PREFILL_TOKENS", "16384"))       # padded tokens per prefill forward
        i = 0
        while i < len(rows):                                             # ragged prompts, several per forward
            j, tmax = i, 0
            while j < len(rows) and (j == i or (j - i + 1) * max(tmax, len(rows[j])) <= budget):
                tmax = max(tmax, len(rows[j]))
                j += 1
            G = j - i
            ids = torch.zeros((G, tmax), dtype=torch.int64)
            for r in range(i, j):
                ids[r - i, :len(rows[r])] = torch.tensor(rows[r], dtype=torch.int64)
            lens = torch.tensor([len(rows[r]) for r in range(i, j)], dtype=torch.int32, device=dev)
            pos = torch.arange(tmax, dtype=torch.int32, device=dev).view(1, tmax).expand(G, tmax).contiguous()
            idx = torch.tensor([(r - i) * tmax + t for r in range(i, j) for t in range(len(rows[r]) - n - 1, len(rows[r]) - 1)],
                               dtype=torch.int32, device=dev)
            logits = self.model.forward(ids.to(dev), pos, self.cache, row_prompt=torch.arange(i, j, dtype=torch.int32, device=dev),
                                        row_cand=torch.full((G,), -1, dtype=torch.int32, device=dev), row_len=lens,
                                        kv_len_rows=torch.zeros(G, dtype=torch.int32, device=dev), any_candidates=False,
                                        logit_index=idx, s_cur=tmax)
            for r in range(i, j):
                lg = logits[(r - i) * n:(r - i + 1) * n]
                if self.logits_hook is not None:
                    lg = self.logits_hook(lg, self, prefill=(r, len(prompts[r])))
                ngrams[r] = ops.argmax_rows(lg).cpu().tolist()
                self.kv_len_host[r] = len(prompts[r])
            i = j
        self._push_kv_len()
        return ngrams

    # ------------------------------------------------------------------------------ one Jacobi iteration
    @torch.inference_mode()
    def iteration(self, d: np.ndarray) -> np.ndarray:
        """forward -> verify/accept/re-draft (HIP) -> KV commit.  ``d`` is the current descriptor table."""
        packed_in = self.batch.pack(d, self.t_align, compact=self.compact_logits, valid_align=self.logit_align)
        if packed_in is None:
            return d
        ids, pos, row_prompt, row_len = packed_in
        if ids.shape[1] > self.t_cap:
            raise RuntimeError(f"a row of {ids.shape[1]} tokens exceeds the forward capacity {self.t_cap} "
                               "(the block counters ran away, see DESIGN.md §3.2)")
        B = d[:, self._f["B"]]
        any_cand = bool((B > 1).any())
        dev = self.device
        R = ids.shape[0]
        if any_cand:
            # candidate index inside a prompt -> scratch row p*cand_rows + (b-1); row 0 writes the main cache
            bidx = np.concatenate([np.arange(b) for b in B if b > 0])
            pidx = np.repeat(np.arange(self.P), B)
            rc = np.where(bidx > 0, pidx * max(self.cand_rows, 1) + bidx - 1, -1).astype(np.int32)
            row_cand = torch.from_numpy(rc).to(dev, non_blocking=True)
        else:
            row_cand = torch.full((R,), -1, dtype=torch.int32, device=dev)
        self.last_valid_rows = int((B * d[:, self._f["T"]]).sum())
        if int(self.kv_len_host[B > 0].max()) + ids.shape[1] > self.max_seq_len:
            raise RuntimeError(f"KV cache rows hold {self.max_seq_len} positions; a prompt at "
                               f"{int(self.kv_len_host[B > 0].max())} cannot take {ids.shape[1]} more (raise max_seq_len)")
        prof = self.profiler
        kv_rows = self.cache.kv_len[row_prompt.long()]
        s_cur = int(self.kv_len_host[B > 0].max()) + ids.shape[1]
        if prof: prof.start("jacobi.forward")
        logits = self.model.forward(ids, pos, self.cache, row_prompt=row_prompt, row_cand=row_cand, row_len=row_len,
                                    kv_len_rows=kv_rows, any_candidates=any_cand, s_cur=s_cur,
                                    logit_index=self.batch.valid_index)
        if self.logits_hook is not None:
            logits = self.logits_hook(logits, self, prefill=None)
        if prof: prof.stop("jacobi.forward")
        self.forwards += 1
        self.last_logits_rows = logits.shape[0]
        if prof: prof.start("jacobi.verify")          # argmax + accept + re-draft + pool + spawn/promote in two launches
        d = self.batch.verify(logits)
        if prof: prof.stop("jacobi.verify")
        if self.cache.committer is not None and any_cand:
            if prof: prof.start("jacobi.commit")
            self.cache.committer.commit(self.batch.desc_dev)
            if prof: prof.stop("jacobi.commit")
        if prof:
            prof.iterations += 1
            prof.tokens += int(d[:, self._f["accepted"]].sum())
        act = B > 0
        self.kv_len_host[act] = d[act, self._f["kv_len"]]
        self._push_kv_len()
        return d

    def _push_kv_len(self) -> None:
        """Committed lengths to the device through a pinned staging buffer (asynchronous: every caller sits behind the
        descriptor read-back's stream sync, so the buffer is never rewritten under a copy in flight)."""
        self._kv_len_pin.copy_(torch.from_numpy(self.kv_len_host.astype(np.int32)))
        self.cache.kv_len.copy_(self._kv_len_pin, non_blocking=True)

    # ------------------------------------------------------------------------------ streaming (applications/)
    def generate_stream(self, prompts, **kw):
        """Generator counterpart of the reference's streaming driver (applications/jacobi_streaming_driver.py:7-193):
        yields ``(prompt_index, new_token_ids)`` the moment a prompt finishes a block-level call; the generator's return
        value (``StopIteration.value``) is ``generate``'s result."""
        return self._generate_events(prompts, **kw)

    def generate(self, prompts, **kw):
        """Decode every prompt to EOS / max_new_tokens / max_calls.  Returns (stats per prompt, gen_seconds, iterations);
        ``gen_seconds`` covers the generation phase only (prefill excluded, DRV:217-230)."""
        it = self._generate_events(prompts, **kw)
        while True:
            try:
                next(it)
            except StopIteration as stop:
                return stop.value

    # ------------------------------------------------------------------------------ driver (DRV:152-273)
    def _generate_events(self, prompts: Sequence[Sequence[int]], max_new_tokens: int = 1024, max_calls: int = 1024,
                 seed: int = 1234, on_iteration: Optional[Callable[[int, np.ndarray], None]] = None,
                 max_iterations: Optional[int] = None, on_generation_start: Optional[Callable[[], None]] = None,
                 on_call_done: Optional[Callable[[int, List[int]], None]] = None):
        assert len(prompts) == self.P
        # one budget per prompt (a scalar applies to all): a prompt stops its calls once it holds that many new tokens
        budgets = ([int(max_new_tokens)] * self.P if np.isscalar(max_new_tokens) else [int(x) for x in max_new_tokens])
        assert len(budgets) == self.P
        n, eos = self.params.n, self.params.eos_token_id
        rngs = [random.Random(seed + p) for p in range(self.P)]          # one stream per prompt (order-independent)
        stats = [PromptStats(prompt_tokens=len(p)) for p in prompts]
        text = [list(p) for p in prompts]                              # generated_ids incl. the prompt (DRV:150)
        t0 = time.perf_counter()
        drafts = [[rngs[p].choice(text[p]) for _ in range(n)] for p in range(self.P)]      # DRV:176-180
        ngrams = self.prefill(prompts, drafts)
        for s in stats:
            s.calls = 1                                                # the prefill call counts (DRV:234)
        active = np.ones(self.P, dtype=bool)
        inputs = np.array(ngrams, dtype=np.int64)                      # call 1 reuses the prefill n-gram (DRV:206-208)
        begin_kv = self.kv_len_host.astype(np.int32).copy()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t_gen = time.perf_counter()
        d = self.batch.begin(torch.from_numpy(inputs), torch.from_numpy(begin_kv))
        if on_generation_start is not None:
            on_generation_start()
        iters_total = 0
        while active.any():
            d = self.iteration(d)
            iters_total += 1
            if on_iteration is not None:
                on_iteration(iters_total, d)
            done = (d[:, self._f["done"]] == 1) & active
            if done.any():
                res = self.batch.results(d)
                restart = np.full(self.P, N.JF_MB_KEEP, dtype=np.int32)
                for p in np.nonzero(done)[0]:
                    r = res[p]
                    st = stats[p]
                    text[p] += r["ret"]
                    st.token_ids += r["ret"]
                    if on_call_done is not None:
                        on_call_done(int(p), r["ret"])
                    yield int(p), list(r["ret"])
                    st.calls += 1
                    st.total_iterations += r["iters"]
                    new_total = len(st.token_ids)
                    self.kv_len_host[p] = r["kv_len"]
                    if eos is not None and eos in st.token_ids:                      # DRV:154-160
                        st.stop_reason = "eos"
                    elif new_total >= budgets[p]:
                        st.stop_reason = "max_new_tokens"
                    elif st.calls >= max_calls:
                        st.stop_reason = "max_calls"
                    elif r["kv_len"] + self.params.n * (self.params.K + 1) + self.t_cap > self.max_seq_len:
                        st.stop_reason = "max_seq_len"         # the next call could outgrow this prompt's cache row
                    if st.stop_reason is not None:
                        active[p] = False
                        restart[p] = N.JF_MB_INACTIVE
                    else:
                        nt = r["next_token"]
                        inputs[p] = [nt] + [rngs[p].choice(text[p]) for _ in range(n - 1)]   # DRV:209-215
                        restart[p] = r["kv_len"]
                if active.any():
                    d = self.batch.begin(torch.from_numpy(inputs), torch.from_numpy(restart))
                    self._push_kv_len()
            if max_iterations is not None and iters_total >= max_iterations:
                break
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        gen_seconds = time.perf_counter() - t_gen
        for st in stats:
            st.new_tokens = max(len(st.token_ids) - 1, 0)              # DRV:243 "subtract prefill"
            st.time_sec = time.perf_counter() - t0
            if st.stop_reason is None:
                st.stop_reason = "interrupted"
        return stats, gen_seconds, iters_total
