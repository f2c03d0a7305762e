"""CPU restatement of the reference's HF decode path *including the forward and its DynamicCache handling*,
used ONLY as the timed ``cpu_baseline`` of bench.py and by tests (TEST INFRASTRUCTURE; the product never
imports it).

It is the oracle state machine (oracle/jacobi_oracle.py, pinned to the reference's golden vectors) driving a plain
torch-CPU Qwen2 forward whose KV cache is handled the way the reference handles HF's DynamicCache:
concatenate on update, ``expand(...).contiguous()`` to B candidate rows (MB:93-127), narrow to the best row with
``.contiguous()`` (MB:500-502), trim by ``narrow`` (MB:36-59); logits are ``.float()``-ed and argmax'd with torch
(MB:463-476).
"""
from __future__ import annotations

import time
from typing import List, Optional

import torch
import torch.nn.functional as F

from . import jacobi_oracle as O


class CpuCache:
    def __init__(self, layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * layers
        self.v: List[Optional[torch.Tensor]] = [None] * layers

    def seq_len(self) -> int:
        return 0 if self.k[0] is None else self.k[0].size(-2)

    def rows(self) -> int:
        return 1 if self.k[0] is None else self.k[0].size(0)

    def resize(self, new_B: int) -> None:                       # MB:93-127
        if self.k[0] is None or self.k[0].size(0) == new_B:
            return
        cur = self.k[0].size(0)
        for i in range(len(self.k)):
            k, v = self.k[i], self.v[i]
            if new_B > cur:
                if cur == 1:
                    self.k[i] = k.expand(new_B, -1, -1, -1).contiguous()
                    self.v[i] = v.expand(new_B, -1, -1, -1).contiguous()
                else:
                    reps = (new_B + cur - 1) // cur
                    self.k[i] = k.repeat(reps, 1, 1, 1)[:new_B].contiguous()
                    self.v[i] = v.repeat(reps, 1, 1, 1)[:new_B].contiguous()
            else:
                self.k[i] = k[:new_B].contiguous()
                self.v[i] = v[:new_B].contiguous()

    def narrow_row(self, b: int) -> None:                        # MB:500-502
        for i in range(len(self.k)):
            self.k[i] = self.k[i][b:b + 1].contiguous()
            self.v[i] = self.v[i][b:b + 1].contiguous()

    def trim(self, num_false: int) -> None:                      # MB:36-59
        if num_false <= 0 or self.k[0] is None:
            return
        new_len = max(0, self.seq_len() - num_false)
        for i in range(len(self.k)):
            self.k[i] = self.k[i].narrow(-2, 0, new_len)
            self.v[i] = self.v[i].narrow(-2, 0, new_len)


class CpuQwen2:
    """Weights are the same tensors the GPU model uses (moved to the host)."""

    def __init__(self, cfg, weights, dtype=torch.bfloat16):
        self.cfg = cfg
        cv = lambda t: t.detach().to("cpu", dtype)
        self.embed = cv(weights.embed)
        self.layers = [{k: cv(v) for k, v in L.items()} for L in weights.layers]
        self.norm = cv(weights.norm)
        self.lm_head = self.embed if cfg.tie_word_embeddings else cv(weights.lm_head)
        hd = cfg.head_dim
        self.inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        self.dtype = dtype

    def _norm(self, x, w):
        xf = x.float()
        xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.cfg.rms_norm_eps)
        return xf.to(x.dtype) * w

    def _rope(self, x, pos):
        fr = pos.float().unsqueeze(-1) * self.inv_freq                  # [B,T,hd/2]
        cos, sin = fr.cos().unsqueeze(2), fr.sin().unsqueeze(2)
        h = x.shape[-1] // 2
        x1, x2 = x[..., :h].float(), x[..., h:].float()
        return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], -1).to(x.dtype)

    @torch.inference_mode()
    def forward(self, ids: torch.Tensor, cache: CpuCache) -> torch.Tensor:
        cfg = self.cfg
        B, T = ids.shape
        nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        past = cache.seq_len()
        pos = torch.arange(past, past + T).view(1, T).expand(B, T)
        x = self.embed[ids]
        causal = torch.ones(T, past + T, dtype=torch.bool).tril(diagonal=past)
        for i, L in enumerate(self.layers):
            h = self._norm(x, L["ln1"])
            qkv = F.linear(h, L["wqkv"], L["bqkv"]).view(B, T, nq + 2 * nkv, hd)
            q = self._rope(qkv[:, :, :nq], pos).transpose(1, 2)
            k = self._rope(qkv[:, :, nq:nq + nkv], pos).transpose(1, 2)
            v = qkv[:, :, nq + nkv:].transpose(1, 2)
            if cache.k[i] is None:
                cache.k[i], cache.v[i] = k.contiguous(), v.contiguous()
            else:                                                       # DynamicCache.update: concatenate
                cache.k[i] = torch.cat([cache.k[i], k], dim=-2)
                cache.v[i] = torch.cat([cache.v[i], v], dim=-2)
            o = F.scaled_dot_product_attention(q, cache.k[i], cache.v[i], attn_mask=causal, enable_gqa=True)
            x = x + F.linear(o.transpose(1, 2).reshape(B, T, nq * hd), L["wo"])
            hh = self._norm(x, L["ln2"])
            g, u = F.linear(hh, L["wgu"]).chunk(2, dim=-1)
            x = x + F.linear(F.silu(g) * u, L["wd"])
        x = self._norm(x, self.norm)
        return F.linear(x, self.lm_head).float()                         # MB:463


class _Hooked(O.MultiblockOracle):
    """The oracle state machine with the tensor cache mirrored operation by operation."""

    def __init__(self, *a, cache: CpuCache, **kw):
        super().__init__(*a, **kw)
        self.cache = cache

    def _kv_trim(self, num_false):
        super()._kv_trim(num_false)
        self.cache.trim(num_false)

    def _kv_resize(self, new_B):
        super()._kv_resize(new_B)
        self.cache.resize(new_B)

    def _kv_narrow(self, best_idx):
        super()._kv_narrow(best_idx)
        self.cache.narrow_row(best_idx)


def cpu_multiblock_call(model: CpuQwen2, cache: CpuCache, input_ids, kv_tokens, deadline: Optional[float] = None, **kw):
    """One generation call (MB:227-740) on the CPU.  Returns the oracle state (ret, next_token, iters).  With a
    ``deadline`` (perf_counter seconds, or a callable taking the tokens accepted so far in this call) the call is abandoned
    between iterations once it is passed / returns True; ``st.partial`` then holds the tokens the real-active block had
    accepted so far (bounded CPU sample for bench.py)."""
    st = _Hooked(input_ids, kv_tokens, cache=cache, **kw)
    st.partial = None
    while True:
        if deadline is not None:
            acc = sum(len(a) for b, a in enumerate(st.out_acc) if not st.need_reverify[b])
            if deadline(acc) if callable(deadline) else time.perf_counter() > deadline:
                st.partial = acc
                return st
        step = st.begin_iteration()
        if step is None:
            break
        out, spans = step
        logits = model.forward(torch.tensor(out, dtype=torch.int64), cache)
        greedy = torch.argmax(logits, dim=-1).tolist()                  # MB:476 over every row (superset of the spans)
        st.end_iteration(greedy)
        if st.done:
            break
    st.finalize()
    assert cache.seq_len() == len(st.kv_tokens), (cache.seq_len(), len(st.kv_tokens))
    return st


def cpu_prefill(model: CpuQwen2, prompt: List[int], draft: List[int]):
    cache = CpuCache(model.cfg.num_hidden_layers)
    logits = model.forward(torch.tensor([list(prompt) + list(draft)], dtype=torch.int64), cache)   # all S+n rows (MB:216)
    n = len(draft)
    ngram = torch.argmax(logits[:, -n - 1:-1, :], dim=-1)[0].tolist()
    cache.trim(n)
    return ngram, cache


def timed_tokens_per_second(model: CpuQwen2, prompt: List[int], rng, *, n, K, r, pool, eos, pad, budget_s=20.0,
                            max_calls=64, min_tokens=0, hard_cap_s=None):
    """Bounded CPU sample of the same workload: prefill (untimed) then generation calls until ``budget_s`` has passed AND at
    least ``min_tokens`` tokens were accepted (never beyond ``hard_cap_s``)."""
    text = list(prompt)
    ngram, cache = cpu_prefill(model, prompt, [rng.choice(text) for _ in range(n)])
    kv = list(prompt)
    inp = ngram
    t0 = time.perf_counter()
    tokens = iters = calls = 0
    cap = budget_s if hard_cap_s is None else max(hard_cap_s, budget_s)
    while calls < max_calls:
        el = time.perf_counter() - t0
        if el >= cap or (el >= budget_s and tokens >= min_tokens):
            break
        # a call is abandoned between iterations once the budget has passed and the token floor is met, or at the hard cap
        def stop(acc_in_call, _done=tokens):
            e = time.perf_counter() - t0
            return e >= cap or (e >= budget_s and _done + acc_in_call >= min_tokens)
        st = cpu_multiblock_call(model, cache, inp, kv, deadline=stop, n=n, K=K, r=r,
                                 n_gram_pool_size=pool, eos_token_id=eos, pad_token_id=pad)
        if st.partial is not None:                 # budget hit inside a call: count what was accepted so far
            tokens += st.partial
            iters += st.iters
            calls += 1
            break
        kv = st.kv_tokens
        text += st.ret
        tokens += len(st.ret)
        iters += st.iters
        calls += 1
        if eos is not None and eos in st.ret:
            break
        inp = [st.next_token] + [rng.choice(text) for _ in range(n - 1)]
    dt = time.perf_counter() - t0
    return dict(tokens=tokens, iterations=iters, calls=calls, seconds=dt, tokens_per_sec=tokens / dt if dt > 0 else 0.0)
