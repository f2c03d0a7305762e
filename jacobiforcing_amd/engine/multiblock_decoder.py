"""Multiblock Jacobi decoding with rejection recycling for a batch of prompts.

Host orchestration of the hot path the north star names: per Jacobi iteration one PyTorch forward
over every prompt's rows, then ``jf_mb_loop_iterate`` (argmax + verify, accept, re-draft, pool /
candidates, spawn/promote, committed lengths, call restarts, next forward's inputs: two launches) +
``jf_kv_commit`` in HIP, and a poll of the mailbox the device stamps (ops.MultiblockLoop).

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
                 logit_align: Optional[int] = None, resident: Optional[bool] = None, draw_len: int = 8192):
        self.model = model
        self.P = int(num_prompts)
        self.params = params
        self.device = model.device
        self.batch = ops.MultiblockBatch(self.P, params, self.device)
        self.cand_rows = self.batch.max_rows - 1
        # rows longer than (K+2)*n only occur when the reference's block counters run away (K >= 3 with a small spawn ratio:
        # up to ~77 n tokens, tests/golden/mb_cases_v3.json); the candidate scratch is sized for the realistic case and
        # grows on demand (_forward) instead of reserving gigabytes up front
        self.t_cap = min(self.batch.max_tokens, max(128, (params.K + 2) * params.n))
        self.cache = StaticKVCache(model.cfg, self.P, max_seq_len, self.cand_rows, self.t_cap, self.device, dtype=model.dtype)
        self.max_seq_len = max_seq_len
        self.logits_hook = logits_hook
        self.t_align = int(t_align)          # pad the per-iteration row length to a multiple (tuned-GEMM shape grid)
        self.compact_logits = bool(compact_logits)   # lm_head + argmax on draft-carrying positions only (no padding rows)
        self.logit_align = int(logit_align) if logit_align else self.t_align   # lm_head M rounded up to this multiple
        # resident driver (default): finished calls restart on the device (DRV:206-250 inside the convergence launch);
        # JF_RESIDENT=0 / resident=False keeps the driver loop on the host (one extra round trip per finished call)
        self.resident = (os.environ.get("JF_RESIDENT", "1") != "0") if resident is None else bool(resident)
        self.draw_len = int(draw_len)
        self.text_cap = int(max_seq_len) + self.t_cap + 8
        dev = self.device
        self.drv = torch.zeros((self.P, N.DRV_HDR_INTS + self.text_cap), dtype=torch.int32, device=dev)
        self.draws_dev = torch.zeros((self.P, self.draw_len), dtype=torch.int32, device=dev)
        self.loop = ops.MultiblockLoop(self.batch, kv_len=self.cache.kv_len, t_cap=self.t_cap, t_align=self.t_align,
                                       valid_align=self.logit_align, compact=self.compact_logits, cand_rows=self.cand_rows, order=1,
                                       max_seq_len=max_seq_len, drv=self.drv if self.resident else None,
                                       draws=self.draws_dev if self.resident else None)
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
    def _forward(self, s: ops.LoopSummary) -> torch.Tensor:
        """Queue the forward ``s`` describes (its inputs were written by the pack launch queued behind the launch that
        published ``s``) and return its logits."""
        if s.Tpad > self.cache.T_max and s.Rtot > s.Rmain:
            # a runaway row WITH candidate rows: the scratch doubles (the state machine itself holds rows of max_tokens)
            self.cache.grow_candidates(min(self.batch.max_tokens, max(2 * self.cache.T_max, -(-s.Tpad // 64) * 64)))
        if s.max_kv + s.Tpad > self.max_seq_len:
            raise RuntimeError(f"KV cache rows hold {self.max_seq_len} positions; a prompt at "
                               f"{s.max_kv} cannot take {s.Tpad} more (raise max_seq_len)")
        ids, pos, row_prompt, row_len, row_cand, row_kv = self.loop.inputs()
        self.last_valid_rows = s.Nvalid
        prof = self.profiler
        if prof: prof.start("jacobi.forward")
        if ops.LOOP_HOOKS and "forward_begin" in ops.LOOP_HOOKS:
            ops.LOOP_HOOKS["forward_begin"](self.batch)
        logits = self.model.forward(ids, pos, self.cache, row_prompt=row_prompt, row_cand=row_cand, row_len=row_len,
                                    kv_len_rows=row_kv, any_candidates=s.Rtot > s.Rmain, s_cur=s.max_kv + s.Tpad,
                                    logit_index=self.loop.valid_index(), n_main=s.Rmain)
        if self.logits_hook is not None:
            logits = self.logits_hook(logits, self, prefill=None)
        if prof: prof.stop("jacobi.forward")
        self.forwards += 1
        self.last_logits_rows = logits.shape[0]
        return logits

    @torch.inference_mode()
    def _verify(self, s: ops.LoopSummary, logits: torch.Tensor) -> ops.LoopSummary:
        """Queue the convergence check + loop body (+ restarts) + next pack (one launch + the pack) and the KV commit behind
        the forward, then poll the mailbox: returns the header of the next forward's summary."""
        prof = self.profiler
        if prof: prof.start("jacobi.verify")          # argmax + accept + re-draft + pool + spawn/promote (+ restart): one launch
        self.loop.iterate(logits)
        if self.cache.committer is not None and s.Rtot > s.Rmain:
            self.cache.committer.commit(self.batch.desc_dev)
        s2 = self.loop.wait(snapshot=False)
        if prof:
            prof.stop("jacobi.verify")
            prof.iterations += 1
            prof.tokens += s2.accepted
        return s2

    def _account(self, s_prev: ops.LoopSummary, s: ops.LoopSummary) -> None:
        """Host bookkeeping of the iteration that ran forward ``s_prev`` and published ``s`` (off the critical path)."""
        self.loop.snapshot(s)
        act = s_prev.d[:, self._f["B"]] > 0
        self.kv_len_host[act] = s.d[act, self._f["kv_len"]]

    def iteration(self, s: ops.LoopSummary) -> ops.LoopSummary:
        """forward -> convergence check + loop body -> KV commit -> mailbox, one after the other (the drivers below overlap
        the host's bookkeeping with the next forward instead)."""
        if s.Rtot == 0:
            return s
        self.loop.snapshot(s)
        s2 = self._verify(s, self._forward(s))
        self._account(s, s2)
        return s2

    def _push_kv_len(self) -> None:
        """Committed lengths to the device through a pinned staging buffer (prefill only: inside the loop the step writes
        them itself)."""
        self._kv_len_pin.copy_(torch.from_numpy(self.kv_len_host.astype(np.int32)))
        self.cache.kv_len.copy_(self._kv_len_pin, non_blocking=True)

    # ------------------------------------------------------------------------------ streaming (applications/)
    def generate_stream(self, prompts, **kw):
        """Generator counterpart of the reference's streaming driver (applications/jacobi_streaming_driver.py:7-193):
        yields ``(prompt_index, new_token_ids)`` the moment a prompt finishes a block-level call; the generator's return
        value (``StopIteration.value``) is ``generate``'s result."""
        return self._generate_events(prompts, chunks=True, **kw)

    def generate(self, prompts, **kw):
        """Decode every prompt to EOS / max_new_tokens / max_calls.  Returns (stats per prompt, gen_seconds, iterations);
        ``gen_seconds`` covers the generation phase only (prefill excluded, DRV:217-230)."""
        it = self._generate_events(prompts, chunks=kw.get("on_call_done") is not None, **kw)
        while True:
            try:
                next(it)
            except StopIteration as stop:
                return stop.value

    # ------------------------------------------------------------------------------ driver (DRV:152-273)
    def _generate_events(self, prompts: Sequence[Sequence[int]], max_new_tokens: int = 1024, max_calls: int = 1024,
                 seed: int = 1234, on_iteration: Optional[Callable[[int, np.ndarray], None]] = None,
                 max_iterations: Optional[int] = None, on_generation_start: Optional[Callable[[], None]] = None,
                 on_call_done: Optional[Callable[[int, List[int]], None]] = None, chunks: bool = False,
                 draws: Optional[ops.DrawStreams] = None, fence_iterations=()):
        assert len(prompts) == self.P
        # one budget per prompt (a scalar applies to all): a prompt stops its calls once it holds that many new tokens
        budgets = ([int(max_new_tokens)] * self.P if np.isscalar(max_new_tokens) else [int(x) for x in max_new_tokens])
        assert len(budgets) == self.P
        n = self.params.n
        # random.choice(generated_ids) of DRV:176-180 / 209-215: one pre-drawn stream per prompt (order-independent)
        draws = draws if draws is not None else ops.DrawStreams(self.P, seed=seed, length=self.draw_len)
        rngs = [draws.rng(p) for p in range(self.P)]
        stats = [PromptStats(prompt_tokens=len(p)) for p in prompts]
        text = [list(p) for p in prompts]                              # generated_ids incl. the prompt (DRV:150)
        t0 = time.perf_counter()
        drafts = [[rngs[p].choice(text[p]) for _ in range(n)] for p in range(self.P)]      # DRV:176-180
        ngrams = self.prefill(prompts, drafts)
        inputs = np.array(ngrams, dtype=np.int64)                      # call 1 reuses the prefill n-gram (DRV:206-208)
        begin_kv = self.kv_len_host.astype(np.int32).copy()
        if self.resident:
            self._load_driver(prompts, budgets, max_calls, draws)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t_gen = time.perf_counter()
        s = self.loop.begin(torch.from_numpy(inputs), torch.from_numpy(begin_kv))
        if on_generation_start is not None:
            on_generation_start()
        run = self._run_resident if self.resident else self._run_host_driven
        iters_total = yield from run(s, stats, text, rngs, inputs, budgets, max_calls, on_iteration, max_iterations, on_call_done, chunks,
                                     frozenset(fence_iterations))
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        gen_seconds = time.perf_counter() - t_gen
        if self.resident:
            self._collect_driver(stats, prompts)
        for st in stats:
            st.new_tokens = max(len(st.token_ids) - 1, 0)              # DRV:243 "subtract prefill"
            st.time_sec = time.perf_counter() - t0
            if st.stop_reason is None:
                st.stop_reason = "interrupted"
        return stats, gen_seconds, iters_total

    # -- resident driver: calls end and restart inside the convergence launch -------------------------------------
    def _load_driver(self, prompts, budgets, max_calls, draws: ops.DrawStreams) -> None:
        H = N.DRV_HDR_INTS
        blk = np.zeros((self.P, H + self.text_cap), dtype=np.int32)
        f = N.DRV_FIELDS.index
        for p, prompt in enumerate(prompts):
            blk[p, f("active")] = 1
            blk[p, f("calls")] = 1                                     # the prefill call counts (DRV:234)
            blk[p, f("budget")] = min(int(budgets[p]), (1 << 31) - 1)
            blk[p, f("max_calls")] = min(int(max_calls), (1 << 31) - 1)
            blk[p, f("text_len")] = len(prompt)
            blk[p, f("cursor")] = int(draws.cursor[p] % draws.length)
            blk[p, H:H + len(prompt)] = prompt
        self.drv.copy_(torch.from_numpy(blk))
        self.draws_dev.copy_(torch.from_numpy(draws.words.view(np.int32)))

    def _run_resident(self, s, stats, text, rngs, inputs, budgets, max_calls, on_iteration, max_iterations, on_call_done, chunks,
                      fence_iterations=()):
        """The loop: forward -> one launch -> mailbox.  The host's bookkeeping for iteration i (descriptor copy, callbacks,
        streamed chunks) runs after the forward of iteration i+1 has been queued, i.e. while the GPU is busy — except at
        ``fence_iterations`` (and at the end), where the callback runs before any further GPU work is queued (bench.py's
        timing fences)."""
        iters_total = 0
        ev = self._f["events"]

        def account(s_prev, s_new, i):
            self._account(s_prev, s_new)
            if on_iteration is not None:
                on_iteration(i, s_new.d)
            if chunks and s_new.n_call_end:
                ended = np.nonzero(s_new.d[:, ev] & N.EVT_CALL_END)[0]
                fin = s_new.fin
                off, ln = fin[:, N.FIN_FIELDS.index("text_off")], fin[:, N.FIN_FIELDS.index("ret_len")]
                H = N.DRV_HDR_INTS
                rows = [self.drv[int(p), H + int(off[p]):H + int(off[p]) + int(ln[p])] for p in ended]
                flat = self._fetch_rows(rows) if rows else []
                at = 0
                for p in ended:
                    ret = flat[at:at + int(ln[p])]
                    at += int(ln[p])
                    if on_call_done is not None:
                        on_call_done(int(p), ret)
                    yield int(p), ret

        pending = None                                   # (s_prev, s_new, i): an iteration whose bookkeeping is still due
        while s.Rtot > 0:
            logits = self._forward(s)
            if pending is not None:
                yield from account(*pending)
                pending = None
            s_new = self._verify(s, logits)
            iters_total += 1
            last = (max_iterations is not None and iters_total >= max_iterations) or s_new.Rtot == 0
            if last or iters_total in fence_iterations:
                yield from account(s, s_new, iters_total)
            else:
                pending = (s, s_new, iters_total)
            s = s_new
            if max_iterations is not None and iters_total >= max_iterations:
                break
        return iters_total

    def _fetch_rows(self, rows) -> list:
        """Token slices of the driver blocks to the host WITHOUT waiting for the forward that has been queued since: the
        launch that wrote them has finished (its mailbox stamp has been seen), so a copy on a side stream only waits for
        itself — `torch.cat(rows).cpu()` on the decoding stream waited for the whole next forward."""
        if self.device.type != "cuda":
            return torch.cat(rows).tolist()
        side = getattr(self, "_side_stream", None)
        if side is None:
            side = self._side_stream = torch.cuda.Stream(self.device)
        with torch.cuda.stream(side):
            host = torch.cat(rows).to("cpu", non_blocking=True)
        side.synchronize()
        return host.tolist()

    def _collect_driver(self, stats, prompts) -> None:
        blk = self.drv.cpu().numpy()
        f = N.DRV_FIELDS.index
        H = N.DRV_HDR_INTS
        for p, st in enumerate(stats):
            tl = int(blk[p, f("text_len")])
            st.token_ids = blk[p, H + len(prompts[p]):H + tl].tolist()
            st.calls = int(blk[p, f("calls")])
            st.total_iterations = int(blk[p, f("iters_total")])
            st.stop_reason = N.STOP_REASONS.get(int(blk[p, f("stop")]))

    # -- host-driven restarts (JF_RESIDENT=0): the reference driver's loop on the host ------------------------------
    def _run_host_driven(self, s, stats, text, rngs, inputs, budgets, max_calls, on_iteration, max_iterations, on_call_done, chunks,
                         fence_iterations=()):
        n, eos = self.params.n, self.params.eos_token_id
        for st in stats:
            st.calls = 1                                               # the prefill call counts (DRV:234)
        active = np.ones(self.P, dtype=bool)
        iters_total = 0
        while active.any() and s.Rtot > 0:
            s = self.iteration(s)
            iters_total += 1
            d = s.d
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
                    if chunks:
                        yield int(p), list(r["ret"])
                    st.calls += 1
                    st.total_iterations += r["iters"]
                    new_total = len(st.token_ids)
                    self.kv_len_host[p] = r["kv_len"]
                    if eos is not None and eos in r["ret"]:                          # DRV:154-160
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
                    s = self.loop.begin(torch.from_numpy(inputs), torch.from_numpy(restart))
            if max_iterations is not None and iters_total >= max_iterations:
                break
        return iters_total
