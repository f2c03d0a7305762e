"""HF-style seam: drop-in replacements for the functions the reference monkey-patches onto ``Qwen2ForCausalLM``

    Qwen2ForCausalLM.jacobi_forward_greedy_multiblock = jacobi_forward_greedy_multiblock      (DRV-MR:24-25)
    Qwen2ForCausalLM.jacobi_forward_greedy            = jacobi_forward_greedy                 (SB driver)

with the reference's signatures and return tuples (MB:141-167, 219-225, 547/614/740; SB:35-52, 138, 227/247/273/276).
``self`` must expose ``self.jf_backend``: an object with

    new_cache()                         -> cache  (has .get_seq_length())
    forward(rows[B,T] int64, cache)     -> logits [B*T, V]; row b continues the committed prefix, rows b>0 are candidates
    commit(cache, src_row, dst, length) -> keep candidate row ``src_row``'s K/V for [dst, dst+length)   (MB:500-502)
    set_length(cache, n)                -> committed length := n                                          (MB:36-59)

``Qwen2Backend`` implements it over the PyTorch-ROCm forward + static KV cache; tests use a scripted backend.  The loop
body is the HIP path (jf_mb_verify for the multiblock function, jf_argmax_partial + jf_sb_step for the single-block one):
one descriptor read-back per iteration, no ``.item()`` per span, per pool entry or per accepted prefix.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _native as N
from . import ops
from .modeling.qwen2 import Qwen2Model, StaticKVCache


class Qwen2Backend:
    """One prompt (batch 1, like the reference's HF functions) over a static cache with candidate scratch rows."""

    def __init__(self, model: Qwen2Model, max_seq_len: int = 8192, max_rows: int = 8, max_tokens: int = 512, t_align: int = 1):
        self.model, self.max_seq_len, self.max_rows, self.max_tokens = model, max_seq_len, max_rows, max_tokens
        self.device = model.device
        # rows are padded to a multiple of t_align tokens inside the forward (masked, no logits for them): keeps the GEMM row
        # count on the tuned grid (tuning.grid_alignment: 64 for one prompt) — off it the library's kernels are 1.1-1.3x slower
        self.t_align = max(int(t_align), 1)
        self._idx = {}

    def new_cache(self):
        tpad = -(-self.max_tokens // self.t_align) * self.t_align
        c = StaticKVCache(self.model.cfg, 1, self.max_seq_len, self.max_rows - 1, tpad, self.device, dtype=self.model.dtype)
        c.length = 0
        c.get_seq_length = lambda: c.length
        return c

    def forward(self, rows: torch.Tensor, cache) -> torch.Tensor:
        B, T = rows.shape
        dev = self.device
        kv = cache.length
        # max_tokens sizes the candidate scratch rows only: a single row (prefill of prompt + draft, any B == 1 forward)
        # writes the main cache and may be as long as the cache itself
        if kv + T > self.max_seq_len or B > self.max_rows:
            raise RuntimeError(f"forward of {B}x{T} tokens at position {kv} exceeds the static cache "
                               f"(max_seq_len={self.max_seq_len}, max_rows={self.max_rows})")
        Tp = -(-T // self.t_align) * self.t_align
        if Tp != T and kv + Tp > self.max_seq_len:
            Tp = T
        if B > 1 and Tp > cache.T_max:                  # runaway block lists (K >= 3, small r) with candidate rows: grow the scratch
            cache.grow_candidates(max(2 * cache.T_max, -(-Tp // 64) * 64))
        rows = rows.to(dev)
        idx = None
        if Tp != T:
            rows = torch.nn.functional.pad(rows, (0, Tp - T))
            if B > 1:                                   # one row: lm_head runs on all Tp rows (on the grid) and the prefix is returned
                idx = self._idx.get((B, T, Tp))
                if idx is None:
                    idx = self._idx[(B, T, Tp)] = (torch.arange(B, dtype=torch.int32, device=dev).view(B, 1) * Tp +
                                                   torch.arange(T, dtype=torch.int32, device=dev).view(1, T)).reshape(-1)
        pos = (kv + torch.arange(Tp, dtype=torch.int32, device=dev)).view(1, Tp).expand(B, Tp).contiguous()
        z = torch.zeros(B, dtype=torch.int32, device=dev)
        cand = torch.arange(-1, B - 1, dtype=torch.int32, device=dev)
        logits = self.model.forward(rows, pos, cache, row_prompt=z, row_cand=cand, row_len=z + T, kv_len_rows=z + kv,
                                    any_candidates=B > 1, s_cur=kv + Tp, logit_index=idx)
        return logits[:T] if (Tp != T and B == 1) else logits

    def commit(self, cache, src_row: int, dst: int, length: int) -> None:
        d = torch.zeros((1, N.DESC_INTS), dtype=torch.int32)
        f = N.DESC_FIELDS.index
        d[0, f("kv_src_row")], d[0, f("kv_copy_dst")], d[0, f("kv_copy_len")] = src_row, dst, length
        cache.committer.commit(d.to(self.device))

    def set_length(self, cache, n: int) -> None:
        cache.length = int(n)


def _backend(self):
    b = getattr(self, "jf_backend", None)
    if b is None:
        raise AttributeError("jacobi_forward_*: `self.jf_backend` is not set (see jacobiforcing_amd.hf_seam.Qwen2Backend)")
    return b


def _prefill(self, input_ids, past_key_values, n):
    """MB:175-225 / SB:68-138: forward prompt ⧺ draft, n-gram = argmax(logits[:, -n-1:-1]), cache cut back by n."""
    be = _backend(self)
    cache = past_key_values if past_key_values is not None else be.new_cache()
    logits = be.forward(input_ids, cache)                                   # [T, V]
    T = input_ids.shape[1]
    ngram = ops.argmax_rows(logits[T - n - 1:T - 1]).view(1, n)
    be.set_length(cache, cache.get_seq_length() + T - n)
    return cache, ngram[0], ngram, 0                                        # Q1: "first_correct_token" is the whole row


@torch.inference_mode()
def jacobi_forward_greedy_multiblock(self, input_ids, attention_mask=None, position_ids=None, past_key_values=None,
                                     use_cache=None, prefill_phase=False, n_token_seq_len: int = 32, K: int = 2,
                                     r: float = 0.85, lookahead_start_ratio=0.0, n_gram_pool_size=4, temperature: float = 1.0,
                                     top_p: float = 0.2, top_k=None, repetition_penalty=None, lenience: float = 1.0,
                                     accept_threshold: float = 0.99, tokenizer=None, eos_token_id: Optional[int] = None,
                                     pad_token_id: Optional[int] = None, max_iteration_count: int = 128):
    n = int(n_token_seq_len)
    if prefill_phase:
        return _prefill(self, input_ids, past_key_values, n)
    assert past_key_values is not None, "past_key_values must be provided during generation."          # MB:230
    be = _backend(self)
    cache = past_key_values
    dev = input_ids.device
    prm = ops.MultiblockParams(n=n, K=K, r=r, lookahead_start_ratio=lookahead_start_ratio, n_gram_pool_size=n_gram_pool_size,
                               eos_token_id=eos_token_id, pad_token_id=pad_token_id, max_iteration_count=max_iteration_count)
    key = (n, K, r, lookahead_start_ratio, n_gram_pool_size, eos_token_id, pad_token_id, max_iteration_count)
    st = getattr(cache, "_jf_batch", None)
    if st is None or st[0] != key:
        st = (key, ops.MultiblockBatch(1, prm, getattr(be, "device", dev)))
        cache._jf_batch = st
    batch = st[1]
    f = {k: N.DESC_FIELDS.index(k) for k in N.DESC_FIELDS}
    d = batch.begin(input_ids.view(1, n), torch.tensor([cache.get_seq_length()], dtype=torch.int32))
    while True:
        pk = batch.pack(d)
        if pk is None:
            break
        rows = pk[0]                                                        # [B, T] (one prompt: no padding)
        logits = be.forward(rows, cache)
        kv_before = cache.get_seq_length()
        d = batch.verify(logits)
        if d[0, f["kv_copy_len"]] > 0:
            be.commit(cache, int(d[0, f["kv_src_row"]]), int(d[0, f["kv_copy_dst"]]), int(d[0, f["kv_copy_len"]]))
        be.set_length(cache, int(d[0, f["kv_len"]]))
        if d[0, f["done"]]:
            break
    res = batch.results(d)[0]
    ret = torch.tensor([res["ret"]], dtype=input_ids.dtype, device=dev)
    nxt = torch.tensor([[res["next_token"]]], dtype=input_ids.dtype, device=dev)
    return cache, nxt, ret, res["iters"]


@torch.inference_mode()
def jacobi_forward_greedy(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, use_cache=None,
                          prefill_phase=False, n_token_seq_len=64, temperature=1.0, top_p=0.9, top_k=None,
                          repetition_penalty=None, lenience=1.0, accept_threshold=0.99, tokenizer=None,
                          eos_token_id: Optional[int] = None):
    """Single-block Jacobi call (SB:34-276): accept the longest matching prefix, re-draft the tail from the greedy
    predictions, bonus token on a full accept, EOS inside the accepted prefix / as next token ends the call."""
    if input_ids is None:
        raise ValueError("You must specify exactly input_ids")                                          # SB:54-55
    n = int(n_token_seq_len)
    eos_enabled = eos_token_id is not None
    if not eos_enabled:
        print("!!! WARNING: EOS handling disabled since eos_token_id is None !!!")                      # SB:61-62
    if prefill_phase:
        return _prefill(self, input_ids, past_key_values, n)
    assert past_key_values is not None                                                                  # SB:142
    be = _backend(self)
    cache = past_key_values
    dev = input_ids.device
    st = ops.SingleBlockStepper(input_ids, getattr(be, "device", dev))
    itr = 0
    d = None
    while st.total < n:                                                                                  # SB:150
        itr += 1
        logits = be.forward(st.draft(), cache)                                                           # [L, V]
        d = st.step(logits, eos_token_id if eos_enabled else None, cache.get_seq_length())               # SB:197-273 on the device
        be.set_length(cache, d["kv_len"])
        if d["done"]:
            break
    acc = st.acc[:min(st.total, st.cap)].view(1, -1).to(device=dev, dtype=input_ids.dtype)
    nxt = None if d is None else torch.full((1, 1), d["next_token"], device=dev, dtype=input_ids.dtype)
    return cache, nxt, acc, itr
