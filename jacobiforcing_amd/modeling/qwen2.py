"""Minimal Qwen2 forward in stock PyTorch-ROCm over a static, preallocated KV cache.

The north star keeps the transformer forward in PyTorch; this module is that forward, written
so the Jacobi loop body never has to copy the cache:

* committed K/V of prompt ``p`` live once in ``k_main[layer][p, :, :kv_len[p]]`` — "trim" is a
  length decrement, "append" of an accepted row-0 token is free (its K/V is already in place);
* recycled candidate rows (MB:575-588) write their speculative K/V into a small scratch
  ``k_cand`` and are attended over the shared prefix; the winner's accepted rows are copied to
  the main cache by ``jf_kv_commit`` (replacing MB:93-127 / MB:500-502's full-cache copies).

It mirrors what the reference's forward computes (HF Qwen2: embed -> N x [RMSNorm, QKV+bias,
RoPE, GQA attention, o_proj, RMSNorm, SwiGLU] -> RMSNorm -> lm_head, cf. MB:428-463 and
inference_engine/models/qwen3.py:185-215) but shares no code with it.
"""
from __future__ import annotations

import os

import json
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import torch
import torch.nn.functional as F

from .. import ops


@dataclass
class Qwen2Config:
    vocab_size: int = 152064
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    max_position_embeddings: int = 32768
    tie_word_embeddings: bool = False
    eos_token_id: int = 151645
    pad_token_id: int = 151643

    @classmethod
    def qwen2_5_coder_7b(cls) -> "Qwen2Config":
        """Qwen2.5-Coder-7B-Instruct (public HF config; the checkpoint family README.md:114 names)."""
        return cls()

    @classmethod
    def tiny(cls, vocab_size=512, hidden_size=128, layers=2, heads=4, kv_heads=2, head_dim=32, inter=256) -> "Qwen2Config":
        return cls(vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=inter, num_hidden_layers=layers,
                   num_attention_heads=heads, num_key_value_heads=kv_heads, head_dim=head_dim,
                   max_position_embeddings=4096, eos_token_id=vocab_size - 1, pad_token_id=vocab_size - 2)

    @classmethod
    def from_json(cls, path) -> "Qwen2Config":
        d = json.loads(Path(path).read_text())
        hd = d.get("head_dim") or d["hidden_size"] // d["num_attention_heads"]
        eos = d.get("eos_token_id", 151645)
        if isinstance(eos, list):
            eos = eos[0]
        return cls(vocab_size=d["vocab_size"], hidden_size=d["hidden_size"], intermediate_size=d["intermediate_size"],
                   num_hidden_layers=d["num_hidden_layers"], num_attention_heads=d["num_attention_heads"],
                   num_key_value_heads=d.get("num_key_value_heads", d["num_attention_heads"]), head_dim=hd,
                   rms_norm_eps=d.get("rms_norm_eps", 1e-6),
                   rope_theta=d.get("rope_theta") or (d.get("rope_parameters") or {}).get("rope_theta") or 1e6,
                   max_position_embeddings=d.get("max_position_embeddings", 32768),
                   tie_word_embeddings=d.get("tie_word_embeddings", False), eos_token_id=eos,
                   pad_token_id=d.get("pad_token_id") or 151643)


class Qwen2Weights:
    """Flat weight container (fused QKV and gate/up so each layer is 4 GEMMs)."""

    def __init__(self, cfg: Qwen2Config, device, dtype=torch.bfloat16, seed: int = 0, init_std: float = 0.02):
        g = torch.Generator(device=device).manual_seed(seed)
        H, I = cfg.hidden_size, cfg.intermediate_size
        nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim

        def rnd(*shape, std=init_std):
            return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

        self.embed = rnd(cfg.vocab_size, H)
        self.layers = []
        for _ in range(cfg.num_hidden_layers):
            self.layers.append(dict(
                ln1=torch.ones(H, device=device, dtype=dtype),
                wqkv=rnd((nq + 2 * nkv) * hd, H), bqkv=rnd((nq + 2 * nkv) * hd, std=0.01),
                wo=rnd(H, nq * hd),
                ln2=torch.ones(H, device=device, dtype=dtype),
                wgu=rnd(2 * I, H), wd=rnd(H, I)))
        self.norm = torch.ones(H, device=device, dtype=dtype)
        self.lm_head = self.embed if cfg.tie_word_embeddings else rnd(cfg.vocab_size, H)

    def load_state_dict(self, sd, cfg: Qwen2Config) -> None:
        """Adopt the tensors of a HF ``Qwen2ForCausalLM.state_dict()`` (fuses q/k/v and gate/up; casts to this dtype)."""
        dev, dt = self.embed.device, self.embed.dtype
        get = lambda k: sd[k].detach().to(device=dev, dtype=dt)
        self.embed = get("model.embed_tokens.weight")
        for i, L in enumerate(self.layers):
            pre = f"model.layers.{i}."
            L["ln1"] = get(pre + "input_layernorm.weight")
            L["ln2"] = get(pre + "post_attention_layernorm.weight")
            L["wqkv"] = torch.cat([get(pre + f"self_attn.{n}_proj.weight") for n in "qkv"], 0)
            L["bqkv"] = torch.cat([get(pre + f"self_attn.{n}_proj.bias") for n in "qkv"], 0)
            L["wo"] = get(pre + "self_attn.o_proj.weight")
            L["wgu"] = torch.cat([get(pre + "mlp.gate_proj.weight"), get(pre + "mlp.up_proj.weight")], 0)
            L["wd"] = get(pre + "mlp.down_proj.weight")
        self.norm = get("model.norm.weight")
        self.lm_head = self.embed if cfg.tie_word_embeddings else get("lm_head.weight")

    def load_safetensors(self, model_dir, cfg: Qwen2Config) -> None:
        """Load a HF Qwen2 checkpoint directory (*.safetensors) when one is available."""
        from safetensors import safe_open
        nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        tensors = {}
        if not list(Path(model_dir).glob("*.safetensors")):
            raise FileNotFoundError(f"no *.safetensors under {model_dir}")
        for f in sorted(Path(model_dir).glob("*.safetensors")):
            with safe_open(str(f), "pt", "cpu") as sf:
                for k in sf.keys():
                    tensors[k] = sf.get_tensor(k)
        dev, dt = self.embed.device, self.embed.dtype
        get = lambda k: tensors[k].to(device=dev, dtype=dt)
        self.embed = get("model.embed_tokens.weight")
        for i, L in enumerate(self.layers):
            pre = f"model.layers.{i}."
            L["ln1"] = get(pre + "input_layernorm.weight")
            L["ln2"] = get(pre + "post_attention_layernorm.weight")
            L["wqkv"] = torch.cat([get(pre + f"self_attn.{n}_proj.weight") for n in "qkv"], 0)
            L["bqkv"] = torch.cat([get(pre + f"self_attn.{n}_proj.bias") for n in "qkv"], 0)
            L["wo"] = get(pre + "self_attn.o_proj.weight")
            L["wgu"] = torch.cat([get(pre + "mlp.gate_proj.weight"), get(pre + "mlp.up_proj.weight")], 0)
            L["wd"] = get(pre + "mlp.down_proj.weight")
        self.norm = get("model.norm.weight")
        self.lm_head = self.embed if cfg.tie_word_embeddings else get("lm_head.weight")


def load_model_directory(model_dir, device, dtype=None, allow_random_init: bool = False):
    """(config, weights) of a HF Qwen2 directory: ``config.json`` + its ``*.safetensors``.  A directory without safetensors is an
    error (a checkpoint shipped as ``pytorch_model.bin``, or a path one level off, must not turn into numbers from random
    weights) unless ``allow_random_init`` is given; the notice then goes to stderr (the drivers' stdout is their data)."""
    import sys
    device = torch.device(device)
    cfg = Qwen2Config.from_json(Path(model_dir) / "config.json")
    has = bool(list(Path(model_dir).glob("*.safetensors")))
    if not has:
        other = [f.name for pat in ("*.bin", "*.pt", "*.pth") for f in Path(model_dir).glob(pat)]
        if other:
            raise FileNotFoundError(f"{model_dir} holds {other[:3]} but no *.safetensors: convert the checkpoint "
                                    "(save_pretrained(..., safe_serialization=True)); refusing to decode with random weights")
        if not allow_random_init:
            raise FileNotFoundError(f"no *.safetensors under {model_dir} (pass allow_random_init=True / --allow-random-init to "
                                    "run this architecture with random-init weights)")
    w = Qwen2Weights(cfg, device, dtype=dtype or (torch.bfloat16 if device.type == "cuda" else torch.float32))
    if has:
        w.load_safetensors(model_dir, cfg)
    else:
        print(f"[qwen2] no *.safetensors under {model_dir}: random-init weights (allow_random_init)", file=sys.stderr, flush=True)
    return cfg, w


class StaticKVCache:
    """Preallocated K/V: main [P, H_kv, S_max, D] per layer (+ candidate scratch [P*cand_rows, H_kv, T_max, D])."""

    def __init__(self, cfg: Qwen2Config, P: int, S_max: int, cand_rows: int, T_max: int, device, dtype=torch.bfloat16):
        nkv, hd, NL = cfg.num_key_value_heads, cfg.head_dim, cfg.num_hidden_layers
        self.P, self.S_max, self.cand_rows, self.T_max = P, S_max, max(cand_rows, 0), T_max
        z = lambda *s: torch.zeros(*s, device=device, dtype=dtype)
        self.k = [z(P, nkv, S_max, hd) for _ in range(NL)]
        self.v = [z(P, nkv, S_max, hd) for _ in range(NL)]
        cr = max(self.cand_rows, 1)
        self.ck = [z(P * cr, nkv, T_max, hd) for _ in range(NL)]
        self.cv = [z(P * cr, nkv, T_max, hd) for _ in range(NL)]
        self.kv_len = torch.zeros(P, dtype=torch.int32, device=device)     # committed length per prompt
        self.committer = ops.KVCommitter(self.k, self.v, self.ck, self.cv, cr) if self.cand_rows > 0 else None

    def grow_candidates(self, T_new: int) -> None:
        """Re-allocate the candidate scratch for rows of up to ``T_new`` tokens.  The scratch only lives from a forward to the
        KV commit behind its convergence launch, so between two iterations there is nothing to carry over.  Needed when the
        reference's block lists run away (K >= 3 with a small spawn ratio: rows of up to ~77 n tokens,
        tests/golden/mb_cases_v3.json) and such a row carries candidates; the usual rows fit the initial (K + 2) n."""
        if T_new <= self.T_max:
            return
        like = self.ck[0]
        shape = (like.shape[0], like.shape[1], int(T_new), like.shape[3])
        n_layers, dev, dt = len(self.ck), like.device, like.dtype
        # a clear capacity error instead of an out-of-memory failure in the middle of generation.  JF_CAND_SCRATCH_MAX_GB
        # is a hard cap when set; otherwise the allocation is simply tried (the caching allocator gives back what it holds reserved
        # but unused before it fails: ADVICE r05) and an out-of-memory error becomes the message below
        need = 2 * n_layers * like.element_size() * shape[0] * shape[1] * shape[2] * shape[3]
        msg = (f"candidate K/V scratch for rows of {T_new} tokens needs {need / 2**30:.1f} GiB ({shape[0]} candidate rows x {n_layers} layers) — a "
               f"runaway block list (K >= 3 with a small spawn ratio) outgrew the cache: lower max_prompts / K")
        cap = os.environ.get("JF_CAND_SCRATCH_MAX_GB")
        if cap is not None and need > float(cap) * 2**30:
            raise RuntimeError(f"{msg} or raise JF_CAND_SCRATCH_MAX_GB (= {cap})")
        del like
        self.ck = self.cv = self.committer = None                          # free before allocating the larger scratch
        z = lambda: torch.zeros(shape, device=dev, dtype=dt)
        try:
            self.ck = [z() for _ in range(n_layers)]
            self.cv = [z() for _ in range(n_layers)]
        except torch.OutOfMemoryError as e:
            self.ck = self.cv = None
            raise RuntimeError(f"{msg} ({e})") from None
        self.T_max = int(T_new)
        if self.cand_rows > 0:
            self.committer = ops.KVCommitter(self.k, self.v, self.ck, self.cv, max(self.cand_rows, 1))

    def get_seq_length(self, p: int = 0) -> int:
        return int(self.kv_len[p])


class Qwen2Model:
    @classmethod
    def from_hf(cls, hf_model, device=None, dtype=None) -> "Qwen2Model":
        """Build the forward from a loaded ``transformers`` Qwen2ForCausalLM (route 2)."""
        c = hf_model.config
        cfg = Qwen2Config(vocab_size=c.vocab_size, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                          num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                          num_key_value_heads=c.num_key_value_heads,
                          head_dim=getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads,
                          rms_norm_eps=c.rms_norm_eps,
                          rope_theta=(getattr(c, "rope_theta", None) or (getattr(c, "rope_parameters", None) or {}).get("rope_theta") or 1e6),
                          max_position_embeddings=c.max_position_embeddings, tie_word_embeddings=c.tie_word_embeddings,
                          eos_token_id=c.eos_token_id if isinstance(c.eos_token_id, int) else 151645,
                          pad_token_id=c.pad_token_id or 151643)
        p0 = next(hf_model.parameters())
        w = Qwen2Weights.__new__(Qwen2Weights)
        w.embed = torch.empty(0, device=device or p0.device, dtype=dtype or p0.dtype)
        w.layers = [dict() for _ in range(cfg.num_hidden_layers)]
        w.load_state_dict(hf_model.state_dict(), cfg)
        return cls(cfg, w)

    def __init__(self, cfg: Qwen2Config, weights: Qwen2Weights):
        self.cfg = cfg
        self.w = weights
        self.device = weights.embed.device
        self.dtype = weights.embed.dtype
        self._rope_tables(cfg.max_position_embeddings)

    def _rope_tables(self, n_pos: int) -> None:
        """cos / sin [n_pos, hd/2] fp32.  The fused RoPE + KV-append launch indexes them with raw positions, so they always
        cover every position a cache row can hold: forward() grows them to cache.S_max before the first launch."""
        hd = self.cfg.head_dim
        inv = 1.0 / (self.cfg.rope_theta ** (torch.arange(0, hd, 2, device=self.device, dtype=torch.float32) / hd))
        fr = torch.outer(torch.arange(int(n_pos), device=self.device, dtype=torch.float32), inv)
        self.cos = fr.cos()
        self.sin = fr.sin()

    # -- pieces -----------------------------------------------------------------------------
    def _norm(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        return F.rms_norm(x, (x.shape[-1],), w, self.cfg.rms_norm_eps)

    @staticmethod
    def _rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
        """x [R, T, heads, hd]; cos/sin [R, T, 1, hd/2] in x's dtype -> rotate-half RoPE (HF convention)."""
        h = x.shape[-1] // 2
        x1, x2 = x[..., :h], x[..., h:]
        return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1)

    def _mlp_residual(self, L, x2d, h2d):
        """x += down(silu(gate(h)) * up(h)) with the gate in one HIP launch and the residual in the GEMM epilogue (in place:
        an out-of-place addmm first copies x into its result, one more launch per GEMM)."""
        return x2d.addmm_(ops.swiglu(F.linear(h2d, L["wgu"])), L["wd"].t())

    # -- forward over a static cache -------------------------------------------------------------
    @torch.inference_mode()
    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, cache: StaticKVCache,
                row_prompt: torch.Tensor, row_cand: torch.Tensor, row_len: torch.Tensor,
                kv_len_rows: torch.Tensor, any_candidates: bool, logits_rows: Optional[slice] = None,
                s_cur: Optional[int] = None, logit_index: Optional[torch.Tensor] = None,
                rows_in_place: Optional[bool] = None, n_main: Optional[int] = None,
                paged_slots: Optional[torch.Tensor] = None, block_tables: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One forward over R rows of (padded) length T.

        input_ids [R,T] int64, positions [R,T] int32 (= kv_len + t), row_prompt [R] (cache row of the prefix),
        row_cand [R] (-1: the row writes into the main cache, else index into the candidate scratch),
        row_len [R] valid tokens per row, kv_len_rows [R] committed prefix length per row, s_cur = max(kv_len)+T when the
        caller knows it (saves a device read), rows_in_place = False when row r is NOT cache row r although R == P (a
        caller whose rows are a permutation of the cache rows; None: rows of one batch in cache order), n_main = the rows come
        in the loop API's order (jf_mb_loop, order 1): the first n_main rows write the main cache (row 0 of every running
        prompt, in prompt order), the rest are candidate rows — only THOSE rows' K/V prefixes are gathered, the main rows
        attend in place when they are the cache rows in order (what is left of MB:93-127).  Returns logits [R*T, V] in the weight dtype (or ``logits_rows`` of it, or
        the rows listed in ``logit_index`` — flat positions, negative entries are list padding and yield a junk row).

        PAGED layout (``Config.kv_cache_layout = "paged"``, the reference's cache: layers/attention.py:10-40, MR:1204-1265): ``cache`` is a
        pool ``[num_blocks, H_kv, block_size, D]`` per layer (a StaticKVCache whose "rows" are blocks), ``paged_slots`` [R*T] the slot of
        every new token (block * block_size + offset, -1 = padding: jf_engine_fill's slot_mapping) and ``block_tables`` [R, C] the blocks
        of each row in order (-1 = none).  The append is the same launch; a row's keys are its blocks gathered in table order (stock
        SDPA has no block-table argument; flash_attn_with_kvcache is not in this image), so logical key index = position."""
        cfg, w = self.cfg, self.w
        R, T = input_ids.shape
        nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        G = nq // nkv
        dev = self.device
        pos = positions.long()
        if s_cur is None:
            s_cur = (int(kv_len_rows.max().item()) + T) if R else T
        paged = block_tables is not None
        S_cap = cache.S_max * int(block_tables.shape[1]) if paged else cache.S_max
        S_cur = min(int(s_cur), S_cap)
        if self.cos.shape[0] < S_cap:                        # a cache longer than max_position_embeddings: never read past the table
            self._rope_tables(S_cap)
        ar_t = torch.arange(T, device=dev)
        ar_s = torch.arange(S_cur, device=dev)
        kvl = kv_len_rows.long()
        rlen = row_len.long()
        rel = ar_s.view(1, 1, S_cur) - kvl.view(R, 1, 1)                            # key index relative to the new block
        mask = (rel < 0) | ((rel <= ar_t.view(1, T, 1)) & (rel < rlen.view(R, 1, 1)))
        mask = mask.view(R, 1, 1, T, S_cur).expand(R, 1, G, T, S_cur).reshape(R, 1, G * T, S_cur)
        # additive mask built once per forward (a bool mask is converted inside every SDPA call otherwise), as a view of a
        # buffer whose row stride is a multiple of 16 elements: SDPA otherwise pads it itself — a fill and a copy in front of
        # EVERY layer's attention kernel (tools/sdpa_probe.py: 6.6 + 7.8 us per layer at 64 prompts)
        S_pad = (S_cur + 15) // 16 * 16
        bias = torch.zeros((R, 1, G * T, S_pad), dtype=self.dtype, device=dev)[..., :S_cur]
        bias.masked_fill_(~mask, float("-inf"))
        # slots of the freshly computed K/V rows
        valid = ar_t.view(1, T) < rlen.view(R, 1)
        main_rows = row_cand < 0
        neg = torch.full_like(pos, -1)
        if paged:
            if any_candidates or n_main is not None:
                raise NotImplementedError("candidate rows over the paged layout (the engine's decoders are single-block: MR:1468-1473)")
            slot_main = paged_slots.long().reshape(-1)
            nblk = -(-S_cur // cache.S_max)
            bt = block_tables[:, :nblk].clamp(min=0).long()                           # (blocks a row does not have: masked keys)
        else:
            slot_main = torch.where(valid & main_rows.view(R, 1), row_prompt.long().view(R, 1) * cache.S_max + pos, neg).reshape(-1)
        rp = row_prompt.long()
        split = n_main is not None
        if split:
            RA = int(n_main)
            RB = R - RA
            any_candidates = RB > 0
            direct_a = RA == cache.P and rows_in_place is not False
            if RB:
                slot_cand = torch.where(valid[RA:], row_cand[RA:].long().view(RB, 1) * cache.T_max + ar_t.view(1, T), neg[RA:])
                slot_cand = torch.cat([neg[:RA].reshape(-1), slot_cand.reshape(-1)])
                tail_idx = (kvl[RA:].view(-1, 1) + ar_t.view(1, T)).clamp_(max=S_cur - 1).view(-1, 1, T, 1).expand(-1, nkv, T, hd)
                crc, rp_b = row_cand[RA:].long(), rp[RA:]
            if not direct_a:
                rp_a = rp[:RA]
        elif any_candidates:
            slot_cand = torch.where(valid & (~main_rows).view(R, 1), row_cand.long().view(R, 1) * cache.T_max + ar_t.view(1, T),
                                    neg).reshape(-1)
            cr = (~main_rows).nonzero(as_tuple=True)[0]
            if cr.numel():
                tail_idx = (kvl[cr].view(-1, 1) + ar_t.view(1, T)).clamp_(max=S_cur - 1)          # [C,T]
                tail_idx = tail_idx.view(-1, 1, T, 1).expand(-1, nkv, T, hd)
                crc = row_cand[cr].long()
        pos32 = positions.to(torch.int32).reshape(-1).contiguous()
        direct = (not paged) and (not any_candidates) and R == cache.P and rows_in_place is not False   # row r is cache row r: attend in place

        x = w.embed[input_ids].view(R * T, cfg.hidden_size)                           # [R*T, H]
        for li, L in enumerate(w.layers):
            qkv = F.linear(self._norm(x, L["ln1"]), L["wqkv"], L["bqkv"])                 # [R*T, (nq+2nkv)*hd]
            # one HIP launch: RoPE on q/k, queries re-laid out per KV head, K/V rows appended to the cache(s) (a18)
            qh = ops.rope_kv_append(qkv, T, nq, nkv, hd, pos32, self.cos, self.sin, cache.k[li], cache.v[li], slot_main,
                                    cache.ck[li] if any_candidates else None, cache.cv[li] if any_candidates else None,
                                    slot_cand if any_candidates else None)
            if split:
                if direct_a:
                    Ka, Va = cache.k[li][:, :, :S_cur], cache.v[li][:, :, :S_cur]
                else:
                    Ka, Va = cache.k[li][rp_a, :, :S_cur], cache.v[li][rp_a, :, :S_cur]
                oa = F.scaled_dot_product_attention(qh[:RA], Ka, Va, attn_mask=bias[:RA])
                o = torch.empty((R * T, nq * hd), dtype=oa.dtype, device=dev)
                o[:RA * T].view(RA, T, nkv, G, hd).copy_(oa.view(RA, nkv, G, T, hd).permute(0, 3, 1, 2, 4))
                if RB:
                    # candidate rows: the prompt's prefix (gathered for these rows only) with their own speculative tail
                    Kb = cache.k[li][rp_b, :, :S_cur].scatter_(2, tail_idx, cache.ck[li][crc, :, :T])
                    Vb = cache.v[li][rp_b, :, :S_cur].scatter_(2, tail_idx, cache.cv[li][crc, :, :T])
                    ob = F.scaled_dot_product_attention(qh[RA:], Kb, Vb, attn_mask=bias[RA:])
                    o[RA * T:].view(RB, T, nkv, G, hd).copy_(ob.view(RB, nkv, G, T, hd).permute(0, 3, 1, 2, 4))
                x.addmm_(o, L["wo"].t())
                x = self._mlp_residual(L, x, self._norm(x, L["ln2"]))
                continue
            if any_candidates:
                Kf = cache.k[li][rp, :, :S_cur]                                       # [R,nkv,S,hd] gathered prefix (+ row-0 tail)
                Vf = cache.v[li][rp, :, :S_cur]
                if cr.numel():
                    # candidate rows see their own speculative tail instead of row 0's
                    Kf[cr] = Kf[cr].scatter(2, tail_idx, cache.ck[li][crc, :, :T])
                    Vf[cr] = Vf[cr].scatter(2, tail_idx, cache.cv[li][crc, :, :T])
            elif paged:                                                                # [R, nblk, H, bs, D] -> [R, H, nblk * bs, D]
                Kf = cache.k[li][bt].permute(0, 2, 1, 3, 4).reshape(R, nkv, nblk * cache.S_max, hd)[:, :, :S_cur]
                Vf = cache.v[li][bt].permute(0, 2, 1, 3, 4).reshape(R, nkv, nblk * cache.S_max, hd)[:, :, :S_cur]
            elif direct:
                Kf, Vf = cache.k[li][:, :, :S_cur], cache.v[li][:, :, :S_cur]
            else:
                Kf, Vf = cache.k[li][rp, :, :S_cur], cache.v[li][rp, :, :S_cur]
            o = F.scaled_dot_product_attention(qh, Kf, Vf, attn_mask=bias)
            o = o.view(R, nkv, G, T, hd).permute(0, 3, 1, 2, 4).reshape(R * T, nq * hd)
            x.addmm_(o, L["wo"].t())                                                  # residual in the GEMM epilogue, in place
            x = self._mlp_residual(L, x, self._norm(x, L["ln2"]))
        flat = self._norm(x, w.norm)
        if logits_rows is not None:
            flat = flat[logits_rows]
        if logit_index is not None:
            flat = flat.index_select(0, logit_index.clamp(min=0).long())
        return F.linear(flat, w.lm_head)
