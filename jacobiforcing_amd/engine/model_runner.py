"""Per-GPU runner: owns the PyTorch-ROCm Qwen2 forward + static KV cache and routes a scheduled batch to the decode
strategy — the role of the reference's ``ModelRunner`` (inference_engine/engine/model_runner.py = "MR") for the Jacobi
path: ``run`` (MR:1464-1550), ``_jacobi_prefill_with_drafting`` (MR:777-963), ``_jacobi_forward_step_batch``
(MR:1134-1418), decoder selection by temperature (MR:293-373).  Tensor parallelism, CUDA-graph capture tables and the
shared-memory RPC are out of scope: prompts replicate across GPUs instead.

New relative to the reference: ``decode_strategy="jacobi_multiblock_rejection_recycling"`` is implemented (the reference
raises NotImplementedError at MR:1468-1473) via ``MultiblockJacobiDecoder``.
"""
from __future__ import annotations

import os
import random
import time
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch

from .. import _native, ops
from ..config import Config
from ..modeling.qwen2 import Qwen2Model, Qwen2Weights, StaticKVCache
from .jacobi_decoding import JacobiDecoder
from .jacobi_decoding_nongreedy import JacobiDecoderNonGreedy
from .jacobi_decoding_nongreedy_on_policy import JacobiDecoderNonGreedyOnPolicy
from .multiblock_decoder import MultiblockJacobiDecoder
from .sequence import Sequence


class ProfileTimer:
    """PROFILE=1 section timer with the reference's section names (MR:29-144) so reports are comparable."""

    def __init__(self, device):
        self.enabled = os.environ.get("PROFILE", "0") == "1"
        self.device = device
        self.timings, self.counts, self._t0 = {}, {}, {}
        self.tokens = self.iterations = 0

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def start(self, name):
        if self.enabled:
            self._sync()
            self._t0[name] = time.perf_counter()

    def stop(self, name):
        if self.enabled and name in self._t0:
            self._sync()
            dt = (time.perf_counter() - self._t0.pop(name)) * 1e3
            self.timings[name] = self.timings.get(name, 0.0) + dt
            self.counts[name] = self.counts.get(name, 0) + 1

    def report(self) -> str:
        tot = sum(self.timings.values()) or 1.0
        lines = [f"{k:<32}{v:10.2f} ms {self.counts[k]:6d} calls {100 * v / tot:5.1f}%" for k, v in sorted(self.timings.items())]
        fwd = self.timings.get("jacobi.forward", 0.0) + self.timings.get("jacobi.lm_head", 0.0)
        if fwd and self.tokens:
            lines.append(f"overhead vs forward-only: {100 * (1 - fwd / tot):.1f}%  TPF={self.tokens / max(self.iterations, 1):.2f}")
        return "\n".join(lines)


class LoopForward:
    """``forward_step_loop`` of the engine decoders (engine/chunk_loop.py): MR:1134-1418 for a batch whose draft, positions and
    cached lengths are the DEVICE arrays of an ``ops.EngineLoop`` — block tables grow from the loop's host mirror of the lengths
    (one numpy comparison per iteration; a request is touched only in the iteration it crosses into a new block), the forward
    reads ``loop.draft`` / ``loop.positions`` / ``loop.kv_start`` where the commit launch left them.  No seed read-back, no
    per-sequence index tensors, no request object touched between the first and the last iteration of a chunk."""

    def __init__(self, runner: "ModelRunner"):
        self.r = runner

    def _state(self, lp):
        r = self.r
        st = getattr(lp, "_mr", None)
        if st is None:
            st = lp._mr = dict(version=-1, bt_len=np.asarray([len(s.block_table) for s in lp.seqs], dtype=np.int64),
                               perm=np.asarray([s.num_permanent_spec_blocks for s in lp.seqs], dtype=np.int64))
        if st["version"] != lp.version:
            rows = [r._row(lp.seqs[s]) for s in lp.members.tolist()]
            B, dev = len(rows), r.device
            st.update(version=lp.version, rp=torch.tensor(rows, dtype=torch.int32, device=dev),
                      row_len=torch.full((B,), lp.L, dtype=torch.int32, device=dev),
                      row_cand=torch.full((B,), -1, dtype=torch.int32, device=dev), in_place=rows == list(range(B)))
            if r.paged:                                            # the group's block tables on the device (rows in member order)
                self._push_tables(lp, st, range(B))
        return st

    def _push_tables(self, lp, st, ks) -> None:
        """Rows ``ks`` of the group's device block tables from the requests' lists (at the start, after a compaction, and for a
        request in the iteration it crosses into a new block)."""
        fill = self.r._fill(lp.L)
        m = lp.members
        bt = np.full((len(ks), fill.max_cols), -1, dtype=np.int32)
        for i, k in enumerate(ks):
            t = lp.seqs[int(m[k])].block_table
            bt[i, :len(t)] = t
        ks = list(ks)
        if ks == list(range(ks[0], ks[0] + len(ks))):
            fill.block_tables[ks[0]:ks[0] + len(ks)].copy_(torch.from_numpy(bt), non_blocking=True)
        else:
            fill.block_tables[torch.tensor(ks, dtype=torch.int64, device=fill.device)] = torch.from_numpy(bt).to(fill.device)

    @torch.inference_mode()
    def __call__(self, lp) -> torch.Tensor:
        r = self.r
        B, L, bs = lp.B, lp.L, r.block_size
        st = self._state(lp)
        prof = r.profiler
        prof.start("jacobi.block_alloc")
        m = lp.members
        S = lp.seq_len_h[m]
        s_max = int(S.max())
        if s_max + L - 1 > r.config.max_model_len:
            raise RuntimeError(f"Sequence needs {s_max + L - 1} positions but max_model_len={r.config.max_model_len}")
        need = (S + L - 1 + bs - 1) // bs                                                      # MR:1166-1199
        committed = (S + bs - 1) // bs
        cur = st["bt_len"][m]
        if (need != cur).any():
            bm = r.block_manager
            changed = np.flatnonzero(need != cur).tolist()
            for k in changed:
                seq = lp.seqs[int(m[k])]
                r._fit_block_table(seq, int(need[k]), bm)
                st["bt_len"][m[k]] = len(seq.block_table)
            if r.paged:
                self._push_tables(lp, st, changed)
        st["perm"][m] = np.maximum(st["perm"][m], need - committed)
        prof.stop("jacobi.block_alloc")
        prof.start("jacobi.forward")
        if r.paged:
            # slot mapping of the whole batch from the loop's device lengths and the device block tables: jf_engine_fill, no host copy
            fill = r._fill(L)
            _pos, slots, _cached, bt = fill.fill_device(lp.draft, lp.kv_start + 1, B)
            logits = r.model.forward(lp.draft, lp.positions, r.kv_cache, row_prompt=st["row_cand"], row_cand=st["row_cand"], row_len=st["row_len"],
                                     kv_len_rows=lp.kv_start, any_candidates=False, s_cur=s_max - 1 + L, logit_index=r._logit_index(B, L),
                                     paged_slots=slots, block_tables=bt)
        else:
            logits = r.model.forward(lp.draft, lp.positions, r.kv_cache, row_prompt=st["rp"], row_cand=st["row_cand"], row_len=st["row_len"],
                                     kv_len_rows=lp.kv_start, any_candidates=False, s_cur=s_max - 1 + L, logit_index=r._logit_index(B, L),
                                     rows_in_place=st["in_place"])                              # seed re-forwarded at S-1
        prof.stop("jacobi.forward")
        return logits.view(B, L - 1, logits.shape[-1])

    def finish(self, lp) -> None:
        """Once per chunk, after the decoder has appended the committed tokens: the counters MR / BM keep per request."""
        st = getattr(lp, "_mr", None)
        if st is None:
            return
        if self.r.paged:                                          # a draft position without a block (MR:1190-1191), reported by the fill launches
            err = int(self.r._fill(lp.L).err.item())
            if err:
                self.r._fill(lp.L).err.zero_()
                raise RuntimeError(f"Sequence {err - 1}: a draft position has no KV block (Cannot allocate blocks for draft tokens)")
        bm, bs = self.r.block_manager, self.r.block_size
        for slot, seq in enumerate(lp.seqs):
            seq.num_permanent_spec_blocks = max(seq.num_permanent_spec_blocks, int(st["perm"][slot]))
            if bm is not None:                                    # BM:534-564 with the cache already at len(seq): surplus blocks go back
                keep = -(-len(seq) // bs) + seq.num_permanent_spec_blocks
                while len(seq.block_table) > keep:
                    bm._give_back(seq.block_table.pop())


class ModelRunner:
    # bench.py: weights of the architecture `config` names that are already on the device (the engine sections of the bench
    # line run on the same random-init model as the headline instead of drawing 15 GB again)
    shared_weights: Optional[Qwen2Weights] = None

    def __init__(self, config: Config, rank: int = 0, event=None, device: Optional[str] = None):
        self.config = config
        self.rank, self.world_size = 0, 1
        dev = device or os.environ.get("JF_DEVICE") or ("cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(dev)
        if self.device.type == "cuda":
            _native.lib()                                     # fail loudly when the HIP extension is missing
        hf = config.hf_config
        dtype = torch.bfloat16 if self.device.type == "cuda" else torch.float32
        if os.environ.get("JF_DTYPE"):
            dtype = getattr(torch, os.environ["JF_DTYPE"])
        sw = ModelRunner.shared_weights
        if sw is not None and sw.embed.device == self.device and tuple(sw.embed.shape) == (hf.vocab_size, hf.hidden_size) \
                and len(sw.layers) == hf.num_hidden_layers:
            self.weights = sw
            dtype = sw.embed.dtype
        else:
            self.weights = Qwen2Weights(hf, self.device, dtype=dtype, seed=int(os.environ.get("JF_WEIGHT_SEED", "0")),
                                        init_std=float(os.environ.get("JF_INIT_STD", "0.02")))
            if list(Path(config.model_path).glob("*.safetensors")):
                self.weights.load_safetensors(config.model_path, hf)
            else:
                import sys
                print(f"[ModelRunner] no *.safetensors under {config.model_path}: using random-init weights", file=sys.stderr, flush=True)
        self.model = Qwen2Model(hf, self.weights)
        self.block_size = config.kvcache_block_size
        self.max_rows = int(min(config.max_num_seqs, int(os.environ.get("JF_MAX_ROWS", "64"))))
        if config.num_kvcache_blocks <= 0:
            config.num_kvcache_blocks = self.max_rows * ((config.max_model_len + self.block_size - 1) // self.block_size + 4)
        # "paged" (Config.kv_cache_layout, or JF_KV_LAYOUT for tools): the reference's memory model — a pool of blocks, K/V of a request
        # wherever its block table says (MR:1204-1265).  The pool is a StaticKVCache whose rows are BLOCKS ([num_blocks, H_kv, block_size, D]
        # per layer): the append launch takes the reference's slot mapping as it is, attention gathers a row's blocks in table order.
        self.paged = (os.environ.get("JF_KV_LAYOUT") or config.kv_cache_layout) == "paged"
        if self.paged:
            self.kv_cache = StaticKVCache(hf, config.num_kvcache_blocks, self.block_size, 0, 1, self.device, dtype=dtype)
            self.max_cols = (config.max_model_len + self.block_size - 1) // self.block_size + 4
            self._fills = {}                                      # block length -> ops.PagedFill (the reference's jacobi_buffers, MR:650-686)
        else:
            self.kv_cache = StaticKVCache(hf, self.max_rows, config.max_model_len, 0, 1, self.device, dtype=dtype)
        self.free_rows = list(range(self.max_rows))
        self.block_manager = None
        self.jacobi_decoder = None
        self._mb_decoders = {}
        self.profiler = ProfileTimer(self.device)
        self._kv_host = [0] * self.max_rows
        self._logit_idx = {}

    # ------------------------------------------------------------------------------------------
    def call(self, method_name, *args):
        return getattr(self, method_name)(*args)

    def exit(self):
        if self.profiler.enabled:
            print(self.profiler.report(), flush=True)

    def _row(self, seq: Sequence) -> int:
        if seq.cache_row < 0:
            if not self.free_rows:
                raise RuntimeError("no free KV cache row (raise max_num_seqs / JF_MAX_ROWS)")
            seq.cache_row = self.free_rows.pop(0)
        return seq.cache_row

    def release(self, seq: Sequence) -> None:
        if seq.cache_row >= 0:
            self.free_rows.append(seq.cache_row)
            seq.cache_row = -1

    # ---- paged layout ---------------------------------------------------------------------------
    def _grow_block_table(self, seq: Sequence, need: int) -> None:
        """Blocks for ``need`` * block_size positions (never fewer than the request holds): prefill-with-draft forwards len(seq) +
        block_len tokens (MR:777-963), which the scheduler's allocation of len(seq) tokens does not cover."""
        bm = self.block_manager
        while len(seq.block_table) < need:
            if bm is None or not bm.free_block_ids:
                raise RuntimeError("Cannot allocate blocks for draft tokens")
            seq.block_table.append(bm._allocate_block_no_clear(bm.free_block_ids[0]))
            seq.block_table_version += 1

    def _paged_slots(self, seqs: List[Sequence], starts: List[int], lens: List[int], T: int):
        """Slot of every token of rows that continue at ``starts`` (block_table[pos // bs] * bs + pos % bs, -1 = padding) and the rows'
        block tables — what the reference's prepare_prefill / prepare_decode build on the host (MR:460-560); the Jacobi step's own
        buffers come from jf_engine_fill instead (``_fill``)."""
        bs, B = self.block_size, len(seqs)
        for seq, s0, n in zip(seqs, starts, lens):
            self._grow_block_table(seq, -(-(s0 + n) // bs))
        C = max(len(s.block_table) for s in seqs)
        bt = np.full((B, C), -1, dtype=np.int32)
        slots = np.full((B, T), -1, dtype=np.int64)
        for b, (seq, s0, n) in enumerate(zip(seqs, starts, lens)):
            t = np.asarray(seq.block_table, dtype=np.int64)
            bt[b, :t.size] = t
            pos = s0 + np.arange(n, dtype=np.int64)
            slots[b, :n] = t[pos // bs] * bs + pos % bs
        return torch.from_numpy(slots).to(self.device), torch.from_numpy(bt).to(self.device)

    def _fill(self, L: int) -> "ops.PagedFill":
        f = self._fills.get(L)
        if f is None:
            f = self._fills[L] = ops.PagedFill(self.max_rows, L, self.max_cols, self.block_size, self.device)
        return f

    def _forward_rows(self, seqs: List[Sequence], ids: torch.Tensor, starts: List[int], lens: List[int],
                      logits_rows=None, logit_index=None) -> torch.Tensor:
        """Forward ``ids`` [B, T] where row b continues cache row seqs[b].cache_row from position starts[b]."""
        dev = self.device
        B, T = ids.shape
        st = torch.tensor(starts, dtype=torch.int32, device=dev)
        pos = st.view(B, 1) + torch.arange(T, dtype=torch.int32, device=dev).view(1, T)
        if self.paged:
            slots, bt = self._paged_slots(seqs, starts, lens, T)
            neg = torch.full((B,), -1, dtype=torch.int32, device=dev)
            return self.model.forward(ids.to(dev), pos, self.kv_cache, row_prompt=neg, row_cand=neg,
                                      row_len=torch.tensor(lens, dtype=torch.int32, device=dev), kv_len_rows=st, any_candidates=False,
                                      logits_rows=logits_rows, s_cur=max(starts) + T, logit_index=logit_index, paged_slots=slots, block_tables=bt)
        rows = [self._row(s) for s in seqs]
        rp = torch.tensor(rows, dtype=torch.int32, device=dev)
        return self.model.forward(ids.to(dev), pos, self.kv_cache, row_prompt=rp, row_cand=torch.full((B,), -1, dtype=torch.int32, device=dev),
                                  row_len=torch.tensor(lens, dtype=torch.int32, device=dev), kv_len_rows=st,
                                  any_candidates=False, logits_rows=logits_rows, s_cur=max(starts) + T, logit_index=logit_index,
                                  rows_in_place=rows == list(range(B)))     # freed rows are reused in any order

    def _prompt_forward(self, seqs: List[Sequence], rows: List[List[int]], want: List[range]) -> List[torch.Tensor]:
        """Forward whole token rows from position 0 (prefill), several ragged rows per launch: rows are padded to the longest
        of their group (a group holds at most JF_PREFILL_TOKENS padded tokens) and lm_head runs only on the positions in
        ``want[i]``.  Returns the logits of those positions per row."""
        budget = int(os.environ.get("JF_PREFILL_TOKENS", "16384"))
        out: List[torch.Tensor] = [None] * len(rows)
        i = 0
        while i < len(rows):
            j, tmax = i, 0
            while j < len(rows) and (j == i or (j - i + 1) * max(tmax, len(rows[j])) <= budget):
                tmax = max(tmax, len(rows[j]))
                j += 1
            if tmax > self.config.max_model_len:
                raise RuntimeError(f"prompt + draft ({tmax}) exceeds max_model_len={self.config.max_model_len}")
            ids = torch.zeros((j - i, tmax), dtype=torch.int64)
            for r in range(i, j):
                ids[r - i, :len(rows[r])] = torch.tensor(rows[r], dtype=torch.int64)
            idx = torch.tensor([(r - i) * tmax + t for r in range(i, j) for t in want[r]], dtype=torch.int32, device=self.device)
            logits = self._forward_rows(seqs[i:j], ids, [0] * (j - i), [len(rows[r]) for r in range(i, j)], logit_index=idx)
            o = 0
            for r in range(i, j):
                out[r] = logits[o:o + len(want[r])]
                o += len(want[r])
            i = j
        return out

    # ------------------------------------------------------------------------------------------ MR:777-963
    @torch.inference_mode()
    def _jacobi_prefill_with_drafting(self, seqs: List[Sequence]):
        rows, want = [], []
        for seq in seqs:
            sp = getattr(seq, "sampling_params", None)
            block_len = getattr(sp, "jacobi_block_len", 64) if sp else 64
            draft = [random.choice(seq.token_ids) for _ in range(block_len)]                  # MR:797
            rows.append(seq.token_ids + draft)
            want.append(range(len(seq) - 1, len(seq) + block_len - 1))
        for seq, logits in zip(seqs, self._prompt_forward(seqs, rows, want)):
            seq._prefill_draft = ops.argmax_rows(logits).cpu().tolist()                        # MR:914-918
            seq.num_cached_tokens = len(seq)                                                   # MR:929-947 (roll back)
            seq.draft_tokens = None
        return [[] for _ in seqs]

    # ------------------------------------------------------------------------------------------ MR:1134-1418
    @torch.inference_mode()
    def _jacobi_forward_step_batch(self, seqs: List[Sequence], draft_tokens_batch: torch.Tensor) -> torch.Tensor:
        B, L = len(seqs), draft_tokens_batch.size(1)
        if L < 2:
            raise ValueError("Draft must have at least 2 tokens (seed + 1 speculative)")
        seeds = draft_tokens_batch[:, 0].cpu().tolist()
        for i, seq in enumerate(seqs):
            seq.draft_tokens_gpu = draft_tokens_batch[i]
            if seq.token_ids[-1] != seeds[i]:                                                  # MR:1157-1162
                raise ValueError(f"Seed mismatch: seq[-1]={seq.token_ids[-1]}, draft[0]={seeds[i]}")
        prof = self.profiler
        prof.start("jacobi.block_alloc")
        bm = self.block_manager
        for seq in seqs:                                                                        # MR:1166-1199
            S = len(seq)
            if S + L - 1 > self.config.max_model_len:
                raise RuntimeError(f"Sequence needs {S + L - 1} positions but max_model_len={self.config.max_model_len}")
            need = (S + L - 1 + self.block_size - 1) // self.block_size
            committed = (S + self.block_size - 1) // self.block_size
            self._fit_block_table(seq, need, bm)
            seq.num_permanent_spec_blocks = max(seq.num_permanent_spec_blocks, need - committed)
        prof.stop("jacobi.block_alloc")
        if self.paged:
            # MR:1204-1265 ("jacobi.buffer_fill" + _get_slot_mapping_pattern) as ONE launch: ids, positions, slot mapping and cached
            # lengths of the whole batch from the lengths and block tables (the reference: a Python loop, ~8 launches per sequence)
            prof.start("jacobi.buffer_fill")
            dev = self.device
            _ids, positions, slots, _cq, _ck, cache_seqlens, bt, max_k = self._fill(L).fill(draft_tokens_batch, [len(s) for s in seqs],
                                                                                             [s.block_table for s in seqs])
            prof.stop("jacobi.buffer_fill")
            prof.start("jacobi.forward")
            neg = torch.full((B,), -1, dtype=torch.int32, device=dev)
            logits = self.model.forward(draft_tokens_batch.to(dev), positions.view(B, L).to(torch.int32), self.kv_cache, row_prompt=neg, row_cand=neg,
                                        row_len=torch.full((B,), L, dtype=torch.int32, device=dev), kv_len_rows=cache_seqlens,
                                        any_candidates=False, s_cur=max_k, logit_index=self._logit_index(B, L), paged_slots=slots, block_tables=bt)
            prof.stop("jacobi.forward")
        else:
            prof.start("jacobi.forward")
            logits = self._forward_rows(seqs, draft_tokens_batch, [len(s) - 1 for s in seqs], [L] * B, logit_index=self._logit_index(B, L))   # seed re-forwarded at S-1
            prof.stop("jacobi.forward")
        for seq in seqs:
            seq.num_cached_tokens = (len(seq) - 1) + L                                         # MR:1407-1408
        return logits.view(B, L - 1, logits.shape[-1])

    def _fit_block_table(self, seq: Sequence, need: int, bm) -> None:
        """MR:1166-1199 for one request: its block table holds exactly the blocks S + L - 1 positions need."""
        cur = len(seq.block_table)
        if cur > need:
            seq.block_table = seq.block_table[:need]
            seq.block_table_version += 1
        elif bm is not None:
            for _ in range(need - cur):
                if not bm.can_append(seq) or not bm.free_block_ids:
                    raise RuntimeError("Cannot allocate blocks for draft tokens")
                bid = bm.free_block_ids[0]
                bm._allocate_block_no_clear(bid)
                seq.block_table.append(bid)
                seq.block_table_version += 1

    def _logit_index(self, B: int, L: int) -> torch.Tensor:
        """lm_head on the L-1 positions per row that verify a speculative token (the reference computes all L and slices,
        MR:1413-1416; slicing afterwards would make the verify kernels' [B*(L-1), V] view a 600 MB copy at batch 64)."""
        key = (B, L)
        idx = self._logit_idx.get(key)
        if idx is None:
            idx = self._logit_idx[key] = (torch.arange(B, dtype=torch.int32, device=self.device).view(B, 1) * L +
                                          torch.arange(L - 1, dtype=torch.int32, device=self.device).view(1, L - 1)).reshape(-1)
        return idx

    def _jacobi_forward_step(self, seq: Sequence, draft_tokens: torch.Tensor) -> torch.Tensor:
        return self._jacobi_forward_step_batch([seq], draft_tokens)

    def _ensure_jacobi_decoder_initialized(self, seqs):                                         # MR:293-373
        jac = [s for s in seqs if s.decode_strategy == "jacobi"]
        if not jac:
            return
        if any(getattr(s, "jacobi_on_policy", False) for s in jac):                              # MR:317, 334-343
            if not isinstance(self.jacobi_decoder, JacobiDecoderNonGreedyOnPolicy):
                self.jacobi_decoder = JacobiDecoderNonGreedyOnPolicy(
                    block_manager=self.block_manager, forward_step=self._jacobi_forward_step,
                    forward_step_batch=self._jacobi_forward_step_batch, eos_token_id=self.config.eos,
                    pad_token_id=self.config.pad, vocab_size=self.config.hf_config.vocab_size, device=self.device)
            return
        temps = [float(getattr(s, "temperature", 0.0)) for s in jac]
        all_greedy, all_ng = all(t == 0.0 for t in temps), all(t > 0.0 for t in temps)
        if not (all_greedy or all_ng):
            raise NotImplementedError("Mixed temperature modes in Jacobi batch not supported. Got some sequences with "
                                      f"temperature=0 (greedy) and some with temperature>0 (non-greedy). Temperatures: {temps}")
        want = JacobiDecoder if all_greedy else JacobiDecoderNonGreedy
        if not isinstance(self.jacobi_decoder, want):
            self.jacobi_decoder = want(block_manager=self.block_manager, forward_step=self._jacobi_forward_step,
                                       forward_step_batch=self._jacobi_forward_step_batch, eos_token_id=self.config.eos,
                                       pad_token_id=self.config.pad, vocab_size=self.config.hf_config.vocab_size,
                                       device=self.device,
                                       # JF_ENGINE_LOOP=0: the callbacks above per iteration (the reference's contract) instead
                                       forward_step_loop=LoopForward(self) if os.environ.get("JF_ENGINE_LOOP", "1") != "0" else None)
        self.jacobi_decoder.profiler = self.profiler

    # ------------------------------------------------------------------------------------------ multiblock (new)
    @torch.inference_mode()
    def _run_multiblock(self, seqs: List[Sequence]):
        sp0 = seqs[0].sampling_params
        key = (len(seqs), sp0.jacobi_block_len, sp0.jacobi_max_blocks, sp0.jacobi_spawn_ratio, sp0.jacobi_lookahead_start_ratio,
               sp0.jacobi_n_gram_pool_size, sp0.jacobi_max_iterations, sp0.ignore_eos)
        for s in seqs[1:]:
            sp = s.sampling_params
            if (sp.jacobi_block_len, sp.jacobi_max_blocks, sp.jacobi_spawn_ratio, sp.jacobi_lookahead_start_ratio,
                    sp.jacobi_n_gram_pool_size, sp.jacobi_max_iterations, sp.ignore_eos) != key[1:]:
                raise NotImplementedError("multiblock requests in one batch must share their jacobi_* parameters")
        dec = self._mb_decoders.get(key)
        if dec is None:
            prm = ops.MultiblockParams(n=sp0.jacobi_block_len, K=sp0.jacobi_max_blocks, r=sp0.jacobi_spawn_ratio,
                                       lookahead_start_ratio=sp0.jacobi_lookahead_start_ratio,
                                       n_gram_pool_size=sp0.jacobi_n_gram_pool_size,
                                       eos_token_id=None if sp0.ignore_eos else self.config.eos, pad_token_id=self.config.pad,
                                       max_iteration_count=sp0.jacobi_max_iterations)
            dec = MultiblockJacobiDecoder(self.model, len(seqs), prm, max_seq_len=self.config.max_model_len)
            self._mb_decoders = {key: dec}                      # one live decoder (its KV cache is the big allocation)
        # every request decodes to ITS OWN budget (whole blocks are appended, so the last call may overshoot it by less than
        # a block, like the reference's accepts, JD E3); a request that outgrew its cache row is finished, not re-queued
        budgets = [max(s.max_tokens - s.num_completion_tokens, 0) for s in seqs]
        stats, gen_s, iters = dec.generate([s.token_ids for s in seqs], max_new_tokens=budgets, max_calls=1 << 30,
                                           seed=int(os.environ.get("JF_DRAFT_SEED", "1234")))
        self.last_multiblock = dict(stats=stats, gen_seconds=gen_s, iterations=iters)
        out = []
        for s, st in zip(seqs, stats):
            toks = st.token_ids
            s.extend_tokens(toks)
            s.num_cached_tokens = len(s)
            if st.stop_reason == "max_seq_len":
                s.max_tokens = min(s.max_tokens, s.num_completion_tokens)      # postprocess_jacobi then finishes it
            out.append(toks)
        return out

    # ------------------------------------------------------------------------------------------ autoregressive
    @torch.inference_mode()
    def _run_autoregressive(self, seqs: List[Sequence], is_prefill: bool):
        if is_prefill:                                          # ragged prompts, padded per group; last position's logits
            logits = torch.cat(self._prompt_forward(seqs, [s.token_ids for s in seqs],
                                                    [range(len(s) - 1, len(s)) for s in seqs]), 0)
        else:                                                   # decode: the whole batch in one forward (one token per row)
            ids = torch.tensor([[seq.last_token] for seq in seqs], dtype=torch.int64)
            logits = self._forward_rows(seqs, ids, [len(seq) - 1 for seq in seqs], [1] * len(seqs))
        for seq in seqs:
            seq.num_cached_tokens = len(seq)
        temps = [float(seq.temperature) for seq in seqs]
        if all(t == 0.0 for t in temps):
            return [int(t) for t in ops.argmax_rows(logits).cpu().tolist()]
        toks = []
        for i, t in enumerate(temps):
            if t == 0.0:
                toks.append(int(ops.argmax_rows(logits[i:i + 1])[0]))
            else:                                               # Gumbel-max sampling like layers/sampler.py:10-24
                p = torch.softmax(logits[i:i + 1].float() / t, dim=-1)
                # top_k / top_p planted on the request (the Jacobi decoders read them the same way, JDN:117-118): the same target
                # distribution for the autoregressive baseline, so that the two decoders can be compared sample for sample
                k, tp = ops.active_filters(getattr(seqs[i], "sampling_params", None), p.shape[-1])
                if k:
                    v, idx = torch.topk(p, k, dim=-1)
                    p = torch.zeros_like(p).scatter_(-1, idx, v)
                    p = p / p.sum(dim=-1, keepdim=True).clamp_min(1e-12)
                if tp:
                    sp, si = torch.sort(p, dim=-1, descending=True)
                    keep = torch.cumsum(sp, dim=-1) <= tp
                    keep[..., 0] = True                         # at least the most likely token
                    sp = sp * keep
                    p = torch.zeros_like(p).scatter_(-1, si, sp / sp.sum(dim=-1, keepdim=True).clamp_min(1e-12))
                toks.append(int(torch.argmax(p / torch.empty_like(p).exponential_(1).clamp_min_(1e-10), dim=-1)[0]))
        return toks

    # ------------------------------------------------------------------------------------------ MR:1464-1550
    def run(self, seqs: List[Sequence], is_prefill: bool):
        mb = [s for s in seqs if s.decode_strategy == "jacobi_multiblock_rejection_recycling"]
        jac = [s for s in seqs if s.decode_strategy == "jacobi"]
        if mb and len(mb) != len(seqs) or jac and len(jac) != len(seqs):
            raise NotImplementedError("Mixed decode strategies in same batch not supported. "
                                      f"Got {len(jac)} Jacobi, {len(mb)} multiblock and {len(seqs) - len(jac) - len(mb)} autoregressive sequences.")
        if mb:
            if is_prefill:
                for s in seqs:
                    s.num_cached_tokens = len(s)
                return [[] for _ in seqs]
            return self._run_multiblock(seqs)
        if jac:
            self._ensure_jacobi_decoder_initialized(seqs)
            if is_prefill:
                return self._jacobi_prefill_with_drafting(seqs)
            if any(getattr(s, "jacobi_on_policy", False) for s in seqs):                         # MR:1483-1510
                if not isinstance(self.jacobi_decoder, JacobiDecoderNonGreedyOnPolicy):
                    raise RuntimeError("On-policy decoder not initialized correctly. Expected JacobiDecoderNonGreedyOnPolicy "
                                       f"but got {type(self.jacobi_decoder)}")
                return self.jacobi_decoder.generate_rollout_records_batch(
                    seqs, n_token_seq_len=getattr(seqs[0], "jacobi_block_len", 64), return_metrics=False)
            return self.jacobi_decoder.generate_chunk_batch(seqs)
        return self._run_autoregressive(seqs, is_prefill)
