#!/usr/bin/env python3
"""Generate golden vectors by running the UNMODIFIED reference in this container.

Run (build container only — /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py            # rewrites all 22 files, byte for byte
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py --check    # regenerates into a temporary directory and compares
    (--full-vocabulary / --skip-full-vocabulary: only / everything but fullvocab_cases.json; --out DIR: write there)

What it does
------------
* imports the reference's decode functions from /root/reference
  (modeling/cllm2_qwen2_modeling_kv_terminate_on_eos_improved_multiblock_lookahead_unified.py,
   modeling/cllm2_qwen2_modeling_kv_terminate_on_eos_improved.py,
   inference_engine/engine/jacobi_decoding.py, .../jacobi_decoding_nongreedy.py)
  with two in-process shims (a transformers-4.53-style ``DynamicCache`` stand-in and an
  empty ``flash_attn`` module) and NO edits to the reference;
* drives them with the deterministic scripted model of ``oracle/scripted_model.py``
  through a duck-typed ``self`` (HF path) or ``forward_step[_batch]`` callbacks (engine
  path), injecting every random draw (draft initialisation, random pads, uniforms,
  residual samples) from counter-based streams so the run is reproducible anywhere;
* records inputs and outputs as plain data under tests/golden/*.json.

Only DATA is written (token ids, lengths, counters).  No reference source text is copied.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import random
import sys
import types
import zlib
from pathlib import Path

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")
if not REF.is_dir():
    sys.exit("reference tree not present; golden vectors can only be regenerated in the build container")
sys.path.insert(0, str(REF))

from oracle.scripted_model import ScriptedModel, mix32  # noqa: E402

GOLDEN_DIR = Path(__file__).resolve().parent
OUT_DIR = GOLDEN_DIR

# --------------------------------------------------------------------------------------
# shims + imports of the reference
# --------------------------------------------------------------------------------------
import modeling.cllm2_qwen2_modeling_kv_terminate_on_eos_improved_multiblock_lookahead_unified as mb  # noqa: E402
import modeling.cllm2_qwen2_modeling_kv_terminate_on_eos_improved as sb  # noqa: E402


class FakeCache:
    """transformers-4.53 DynamicCache surface the reference touches."""

    def __init__(self):
        self.key_cache = []
        self.value_cache = []

    def get_seq_length(self):
        return self.key_cache[0].size(-2) if self.key_cache else 0


FakeCache.delete_false_key_value = mb._delete_false_key_value
mb.DynamicCache = FakeCache
mb.create_causal_mask = lambda **kw: None


class FakeCacheSB(FakeCache):
    pass


FakeCacheSB.delete_false_key_value = sb.delete_false_key_value
sb.DynamicCache = FakeCacheSB
sb.create_causal_mask = lambda **kw: None

_fa = types.ModuleType("flash_attn")
_fa.flash_attn_varlen_func = None
_fa.flash_attn_with_kvcache = None
sys.modules["flash_attn"] = _fa
from inference_engine.engine.jacobi_decoding import JacobiDecoder  # noqa: E402
from inference_engine.engine.jacobi_decoding_nongreedy import JacobiDecoderNonGreedy  # noqa: E402
import inference_engine.engine.jacobi_decoding_nongreedy as jdn_mod  # noqa: E402
import inference_engine.engine.jacobi_decoding_nongreedy_on_policy as jdo_mod  # noqa: E402
from inference_engine.engine.block_manager import BlockManager  # noqa: E402
from inference_engine.engine.sequence import Sequence  # noqa: E402
from inference_engine.sampling_params import SamplingParams  # noqa: E402


# --------------------------------------------------------------------------------------
# duck-typed HF "self" around the scripted model
# --------------------------------------------------------------------------------------
class _Trace:
    def __init__(self):
        self.forwards = []  # dicts: kv_len, out [B][T], greedy [B][T]


class FakeLayer:
    attention_type = "full_attention"

    def __init__(self, owner):
        self.owner = owner

    def __call__(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                 use_cache=None, cache_position=None, position_embeddings=None):
        cache = past_key_value
        tok = hidden_states[..., 0].long()
        B, T = tok.shape
        kv_len = cache.get_seq_length()
        if cache.key_cache:
            assert cache.key_cache[0].size(0) == B, (cache.key_cache[0].shape, B)
            prefix = cache.key_cache[0][:, 0, :, 0]
        else:
            prefix = torch.empty((B, 0), dtype=torch.long)
        model = self.owner.scripted
        greedy = []
        for b in range(B):
            g = model.greedy_rows(prefix[b].tolist(), [tok[b].tolist()])[0]
            greedy.append(g)
        k_new = tok.view(B, 1, T, 1).clone()
        v_new = position_ids.expand(B, T).reshape(B, 1, T, 1).clone()
        if cache.key_cache:
            cache.key_cache[0] = torch.cat([cache.key_cache[0], k_new], dim=-2)
            cache.value_cache[0] = torch.cat([cache.value_cache[0], v_new], dim=-2)
        else:
            cache.key_cache.append(k_new)
            cache.value_cache.append(v_new)
        self.owner.trace.forwards.append(dict(kv_len=kv_len, out=tok.tolist(), greedy=greedy))
        g = torch.tensor(greedy, dtype=torch.double)
        return (torch.stack([tok.double(), g], dim=-1),)


class FakeInner:
    def __init__(self, owner):
        self.layers = [FakeLayer(owner)]
        self.has_sliding_layers = False
        self.config = types.SimpleNamespace(num_hidden_layers=1)

    def embed_tokens(self, ids):
        return ids.double().unsqueeze(-1)

    def rotary_emb(self, h, pos):
        return pos

    def norm(self, h):
        return h


class FakeSelf:
    def __init__(self, scripted: ScriptedModel):
        self.scripted = scripted
        self.trace = _Trace()
        self.model = FakeInner(self)
        self.config = types.SimpleNamespace()

    def lm_head(self, h):
        g = h[..., 1].long()
        B, T = g.shape
        V = self.scripted.vocab
        # cheap deterministic noise in (-1,1) + planted max (the reference then calls .float())
        idx = torch.arange(V).view(1, 1, V)
        noise = (((idx * 2654435761 + (g.unsqueeze(-1) + 1) * 40503) % 65536).float() / 32768.0) - 1.0
        noise.scatter_(-1, g.unsqueeze(-1), 8.0)
        return noise


def kv_tokens(cache):
    if not cache.key_cache:
        return []
    assert cache.key_cache[0].size(0) >= 1
    return cache.key_cache[0][0, 0, :, 0].tolist()


BANNERS = {
    "======New block added": "spawn",
    "============= SWITCHING REAL ACTIVE BLOCK": "switch",
    "EARLY STOPPING": "early_stop",
}


def banners_of(text: str):
    out = []
    for line in text.splitlines():
        for k, v in BANNERS.items():
            if line.startswith(k):
                out.append(v)
    return out


# --------------------------------------------------------------------------------------
# HF multiblock (MB) cases — driver mirrors DRV-MR:152-240
# --------------------------------------------------------------------------------------
def rows_crc(rows) -> int:
    """Digest of a list of token rows (or one token list): crc32 over the little-endian int64 image, shape included."""
    a = np.asarray(rows, dtype="<i8")
    return zlib.crc32(a.tobytes(), zlib.crc32(np.asarray(a.shape, dtype="<i8").tobytes()))


def digest_mb_case(case):
    """Runaway cases (K >= 3 with a small spawn ratio: rows of 1000+ tokens over hundreds of forwards) are stored as digests:
    per forward (kv_len, B, T, crc of the rows, crc of the greedy tokens), per call the crc of the cache content; ret /
    next_token / iters / kv_len / banners stay in full."""
    for cl in case["calls"]:
        cl["forwards"] = [dict(kv_len=f["kv_len"], B=len(f["out"]), T=len(f["out"][0]), out_crc=rows_crc(f["out"]),
                               greedy_crc=rows_crc(f["greedy"])) for f in cl["forwards"]]
        if "kv_tokens" in cl:
            cl["kv_tokens_crc"] = rows_crc(cl.pop("kv_tokens"))
    case["digest"] = True
    return case


def run_mb_case(name, *, vocab, seed, robust, prompt_len, n, K, r, pool, lookahead=0.0,
                eos_pos=None, period=0, max_iter=128, max_calls=6, max_new_tokens=10 ** 9, allow_raise=False):
    eos_id, pad_id = vocab - 1, vocab - 2
    model = ScriptedModel(vocab, seed, robust, prompt_len, eos_id=eos_id, eos_pos=eos_pos,
                          reserved=(pad_id,), period=period)
    fs = FakeSelf(model)
    rng = random.Random(seed * 7919 + 13)
    prompt = model.prompt()
    input_ids = torch.tensor([prompt], dtype=torch.long)
    generated = list(prompt)
    calls = []

    # prefill (DRV-MR:174-198)
    draft0 = [rng.choice(generated) for _ in range(n)]
    fs.trace = _Trace()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        cache, first_tok, ngram, it0 = mb.jacobi_forward_greedy_multiblock(
            fs, input_ids=torch.tensor([prompt + draft0], dtype=torch.long), attention_mask=None,
            past_key_values=None, use_cache=True, prefill_phase=True, n_token_seq_len=n,
            K=K, r=r, lookahead_start_ratio=lookahead, n_gram_pool_size=pool,
            eos_token_id=eos_id, pad_token_id=pad_id, max_iteration_count=max_iter)
    prefill = dict(draft=draft0, ngram=ngram[0].tolist(), first_correct_token=first_tok.tolist(),
                   iters=int(it0), kv_len=cache.get_seq_length(), kv_tokens=kv_tokens(cache),
                   forward=fs.trace.forwards[0] if False else dict(
                       kv_len=fs.trace.forwards[0]["kv_len"], greedy_tail=fs.trace.forwards[0]["greedy"][0][-n - 1:-1]))
    first_correct = None
    ncall = 0
    stop_reason = None
    total_new = 0
    while True:
        gen_part = generated[prompt_len:]
        if eos_id in gen_part:
            stop_reason = "eos"
            break
        if total_new >= max_new_tokens:
            stop_reason = "max_new_tokens"
            break
        if ncall >= max_calls:
            stop_reason = "max_calls"
            break
        if ncall == 0:
            inp = ngram[0].tolist()
        else:
            inp = [int(first_correct.view(-1)[0])] + [rng.choice(generated) for _ in range(n - 1)]
        fs.trace = _Trace()
        buf = io.StringIO()
        kv_before = cache.get_seq_length()
        try:
            with contextlib.redirect_stdout(buf):
                cache, first_correct, acc, iters = mb.jacobi_forward_greedy_multiblock(
                    fs, input_ids=torch.tensor([inp], dtype=torch.long), attention_mask=None,
                    past_key_values=cache, use_cache=True, prefill_phase=False, n_token_seq_len=n,
                    K=K, r=r, lookahead_start_ratio=lookahead, n_gram_pool_size=pool,
                    eos_token_id=eos_id, pad_token_id=pad_id, max_iteration_count=max_iter)
        except RuntimeError as e:
            if not allow_raise:
                raise
            # the reference itself dies here (torch cannot broadcast the rows at MB:482): record where and how
            calls.append(dict(input=inp, kv_len_before=kv_before, error=str(e), banners=banners_of(buf.getvalue()),
                              forwards=fs.trace.forwards))
            stop_reason = "reference_raised"
            break
        ret = acc[0].tolist()
        generated += ret
        total_new += len(ret)
        calls.append(dict(input=inp, kv_len_before=kv_before, ret=ret,
                          next_token=first_correct.view(-1).tolist(), next_token_shape=list(first_correct.shape),
                          iters=int(iters), kv_len=cache.get_seq_length(), kv_tokens=kv_tokens(cache),
                          kv_batch=int(cache.key_cache[0].size(0)),
                          banners=banners_of(buf.getvalue()), forwards=fs.trace.forwards))
        ncall += 1
    new_tokens = total_new - 1  # DRV-MR:243 "subtract prefill"
    total_iters = sum(c.get("iters", 0) for c in calls)
    return dict(name=name, kind="mb",
                params=dict(n=n, K=K, r=r, pool=pool, lookahead=lookahead, eos_id=eos_id, pad_id=pad_id,
                            max_iter=max_iter, max_calls=max_calls),
                model=model.describe(), prompt=prompt, prefill=prefill, calls=calls,
                summary=dict(new_tokens=new_tokens, calls=ncall + 1, total_iterations=total_iters,
                             stop_reason=stop_reason, generated=generated[prompt_len:]))


# --------------------------------------------------------------------------------------
# HF single block (SB) cases
# --------------------------------------------------------------------------------------
def run_sb_case(name, *, vocab, seed, robust, prompt_len, n, eos_pos=None, max_calls=5):
    eos_id = vocab - 1
    model = ScriptedModel(vocab, seed, robust, prompt_len, eos_id=eos_id, eos_pos=eos_pos)
    fs = FakeSelf(model)
    rng = random.Random(seed * 104729 + 7)
    prompt = model.prompt()
    generated = list(prompt)
    draft0 = [rng.choice(generated) for _ in range(n)]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        cache, first_tok, ngram, _ = sb.jacobi_forward_greedy(
            fs, input_ids=torch.tensor([prompt + draft0], dtype=torch.long), past_key_values=None,
            use_cache=True, prefill_phase=True, n_token_seq_len=n, eos_token_id=eos_id)
    prefill = dict(draft=draft0, ngram=ngram[0].tolist(), kv_len=cache.get_seq_length())
    calls = []
    ncall = 0
    first_correct = None
    while True:
        if eos_id in generated[prompt_len:] or ncall >= max_calls:
            break
        if ncall == 0:
            inp = ngram[0].tolist()
        else:
            inp = [int(first_correct.view(-1)[0])] + [rng.choice(generated) for _ in range(n - 1)]
        fs.trace = _Trace()
        kv_before = cache.get_seq_length()
        with contextlib.redirect_stdout(buf):
            cache, first_correct, acc, itr = sb.jacobi_forward_greedy(
                fs, input_ids=torch.tensor([inp], dtype=torch.long), past_key_values=cache,
                use_cache=True, prefill_phase=False, n_token_seq_len=n, eos_token_id=eos_id)
        ret = acc[0].tolist()
        generated += ret
        calls.append(dict(input=inp, kv_len_before=kv_before, ret=ret, next_token=first_correct.view(-1).tolist(),
                          iters=int(itr), kv_len=cache.get_seq_length(), kv_tokens=kv_tokens(cache),
                          forwards=fs.trace.forwards))
        ncall += 1
    return dict(name=name, kind="sb", params=dict(n=n, eos_id=eos_id), model=model.describe(), prompt=prompt,
                prefill=prefill, calls=calls, summary=dict(generated=generated[prompt_len:]))


# --------------------------------------------------------------------------------------
# engine (JD / JDN) cases
# --------------------------------------------------------------------------------------
class CounterStream:
    """Counter-based injected randomness: element k = mix32(seed, k)."""

    def __init__(self, seed):
        self.seed = seed
        self.k = 0

    def next_u32(self):
        v = mix32(self.seed, self.k)
        self.k += 1
        return v

    def randint(self, low, high, size, **kw):
        nel = int(np.prod(size))
        return torch.tensor([low + self.next_u32() % (high - low) for _ in range(nel)], dtype=torch.long).view(*size)

    def uniform(self):
        # 24-bit uniform in [0,1): exactly representable in fp32
        return (self.next_u32() >> 8) / float(1 << 24)


@contextlib.contextmanager
def patched(obj, attr, value):
    old = getattr(obj, attr)
    setattr(obj, attr, value)
    try:
        yield
    finally:
        setattr(obj, attr, old)


class EngineHarness:
    """Caller side of the decoder seam, following MR:1134-1199 and MR:1407-1416
    (seed check, block-table growth without clearing, num_cached_tokens update)."""

    def __init__(self, vocab, num_blocks=64, block_size=256, logits_dtype=torch.float32):
        self.vocab = vocab
        self.block_size = block_size
        self.bm = BlockManager(num_blocks, block_size)
        self.models = {}
        self.trace = []
        self.logits_dtype = logits_dtype

    def add_seq(self, model: ScriptedModel, sp: SamplingParams, prefill_draft=None):
        seq = Sequence(model.prompt(), sp)
        self.bm.allocate(seq)
        seq.num_cached_tokens = len(seq)  # after prefill (MR:941)
        seq._prefill_draft = prefill_draft
        self.models[seq.seq_id] = model
        return seq

    def forward_step_batch(self, seqs, draft):
        B, L = draft.shape
        if L < 2:
            raise ValueError("Draft must have at least 2 tokens (seed + 1 speculative)")
        rows = []
        for i, seq in enumerate(seqs):
            seq.draft_tokens_gpu = draft[i]
            if seq.token_ids[-1] != int(draft[i, 0]):
                raise ValueError("Seed mismatch")
            S = len(seq)
            need = (S + L - 1 + self.block_size - 1) // self.block_size
            committed_blocks = (S + self.block_size - 1) // self.block_size
            cur = len(seq.block_table)
            if cur >= need:
                if cur > need:
                    seq.block_table = seq.block_table[:need]
                    seq.block_table_version += 1
            else:
                for _ in range(need - cur):
                    if not self.bm.can_append(seq):
                        raise RuntimeError("Cannot allocate blocks for draft tokens")
                    bid = self.bm.free_block_ids[0]
                    self.bm._allocate_block_no_clear(bid)
                    seq.block_table.append(bid)
                    seq.block_table_version += 1
            seq.num_permanent_spec_blocks = max(seq.num_permanent_spec_blocks, need - committed_blocks)
            model = self.models[seq.seq_id]
            lg = model.logits_rows(seq.token_ids[:-1], [draft[i].tolist()])[0]  # [L, V]
            rows.append(torch.from_numpy(lg))
        logits = torch.stack(rows, 0).to(self.logits_dtype)
        for seq in seqs:
            seq.num_cached_tokens = (len(seq) - 1) + L
        self.trace.append(dict(seq_ids=[s.seq_id for s in seqs], draft=draft.tolist(),
                               seq_lens=[len(s) for s in seqs]))
        return logits[:, :-1, :]

    def forward_step(self, seq, draft):
        return self.forward_step_batch([seq], draft)


def run_jd_case(name, *, vocab, seeds, robust, prompt_lens, block_lens, max_tokens, eos_pos=None,
                use_prefill_draft=True, pad_seed=99, batch=True, max_iters=128):
    eos_id, pad_id = vocab - 1, vocab - 2
    H = EngineHarness(vocab)
    dec = JacobiDecoder(H.bm, forward_step=H.forward_step, forward_step_batch=H.forward_step_batch,
                        eos_token_id=eos_id, pad_token_id=pad_id, vocab_size=vocab, device=torch.device("cpu"))
    seqs, descr = [], []
    base_id = None
    for i, (sd, pl, bl) in enumerate(zip(seeds, prompt_lens, block_lens)):
        ep = None if eos_pos is None else eos_pos[i]
        m = ScriptedModel(vocab, sd, robust, pl, eos_id=eos_id, eos_pos=ep, reserved=(pad_id,))
        sp = SamplingParams(temperature=0.0, max_tokens=max_tokens[i], decode_strategy="jacobi",
                            jacobi_block_len=bl, jacobi_max_iterations=max_iters)
        pd = None
        if use_prefill_draft:
            # what MR:914-918 stores: greedy over the prompt's last position + draft window (random draft)
            rr = random.Random(sd)
            d0 = [rr.choice(m.prompt()) for _ in range(bl)]
            pd = m.greedy_rows(m.prompt()[:-1], [[m.prompt()[-1]] + d0])[0][:bl]
        s = H.add_seq(m, sp, pd)
        if base_id is None:
            base_id = s.seq_id
        seqs.append(s)
        descr.append(dict(model=m.describe(), prompt=m.prompt(), block_len=bl, max_tokens=max_tokens[i],
                          prefill_draft=pd))
    stream = CounterStream(pad_seed)
    with patched(torch, "randint", stream.randint):
        if batch:
            out = dec.generate_chunk_batch(seqs)
        else:
            out = [dec.generate_chunk(s) for s in seqs]
    for t in H.trace:
        t["seq_idx"] = [sid - base_id for sid in t.pop("seq_ids")]
    return dict(name=name, kind="jd", params=dict(vocab=vocab, eos_id=eos_id, pad_id=pad_id, pad_seed=pad_seed,
                                                  batch=batch, max_iters=max_iters),
                seqs=descr, outputs=out, stats=dec.stats,
                final=[dict(token_ids=s.token_ids, num_cached_tokens=s.num_cached_tokens,
                            num_blocks=len(s.block_table)) for s in seqs],
                pads_consumed=stream.k, forwards=H.trace)


def scripted_multinomial(stream):
    def _mn(probs, num_samples=1, **kw):
        assert probs.dim() == 1 and num_samples == 1
        u = stream.uniform()
        c = torch.cumsum(probs.double(), 0)
        thr = u * float(c[-1])
        idx = int(torch.searchsorted(c, torch.tensor(thr, dtype=torch.double), right=True))
        idx = min(idx, probs.numel() - 1)
        return torch.tensor([idx], dtype=torch.long)
    return _mn


def run_jdn_case(name, *, vocab, seeds, robust, prompt_lens, block_len, max_tokens, temperature,
                 eos_pos=None, rng_seed=5, batch=True, logits_dtype="f32", peak=8.0, top_k=None, top_p=None):
    eos_id, pad_id = vocab - 1, vocab - 2
    H = EngineHarness(vocab, logits_dtype=TORCH_DTYPES[logits_dtype])
    dec = JacobiDecoderNonGreedy(H.bm, forward_step=H.forward_step, forward_step_batch=H.forward_step_batch,
                                 eos_token_id=eos_id, pad_token_id=pad_id, vocab_size=vocab,
                                 device=torch.device("cpu"))
    seqs, descr = [], []
    for i, (sd, pl) in enumerate(zip(seeds, prompt_lens)):
        ep = None if eos_pos is None else eos_pos[i]
        m = ScriptedModel(vocab, sd, robust, pl, eos_id=eos_id, eos_pos=ep, reserved=(pad_id,), peak=peak)
        sp = SamplingParams(temperature=temperature, max_tokens=max_tokens, decode_strategy="jacobi",
                            jacobi_block_len=block_len)
        # top_k / top_p are not SamplingParams fields: _build_target_probs reads them with getattr (JDN:117-118), so a caller
        # has to plant them on the instance — which is what this does (round 5)
        if top_k is not None:
            sp.top_k = top_k
        if top_p is not None:
            sp.top_p = top_p
        seqs.append(H.add_seq(m, sp, None))
        descr.append(dict(model=m.describe(), prompt=m.prompt()))
    pads = CounterStream(rng_seed * 3 + 1)
    unis = CounterStream(rng_seed * 3 + 2)
    bonus = CounterStream(rng_seed * 3 + 3)
    events = []

    def _rand(size=(), **kw):
        u = unis.uniform()
        events.append(("u", u))
        return torch.tensor(u, dtype=torch.float32)

    mn = scripted_multinomial(bonus)

    def _mn(probs, num_samples=1, **kw):
        r = mn(probs, num_samples)
        events.append(("bonus", int(r)))
        return r

    with patched(torch, "randint", pads.randint), patched(torch, "rand", _rand), patched(torch, "multinomial", _mn):
        if batch:
            out = dec.generate_chunk_batch(seqs)
        else:
            out = [dec.generate_chunk(s) for s in seqs]
    params = dict(vocab=vocab, eos_id=eos_id, pad_id=pad_id, rng_seed=rng_seed, batch=batch,
                  block_len=block_len, max_tokens=max_tokens, temperature=temperature)
    if logits_dtype != "f32":                 # recorded only when it differs: round-1 files regenerate byte-identical
        params["logits_dtype"] = logits_dtype
    if top_k is not None:
        params["top_k"] = top_k
    if top_p is not None:
        params["top_p"] = top_p
    return dict(name=name, kind="jdn", params=params,
                seqs=descr, outputs=out, stats=dec.stats,
                final=[dict(token_ids=s.token_ids, num_cached_tokens=s.num_cached_tokens) for s in seqs],
                draws=dict(pads=pads.k, uniforms=unis.k, bonus=bonus.k), forwards=H.trace)


TORCH_DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16}


class ScriptedRandom:
    """Stand-in for the ``random`` module inside the on-policy decoder (draft initialisation, JDO:254-266, 474)."""

    def __init__(self, stream):
        self.stream = stream

    def choice(self, seq):
        return seq[self.stream.next_u32() % len(seq)]

    def randrange(self, n):
        return self.stream.next_u32() % n


def run_jdo_case(name, *, vocab, seeds, robust, prompt_lens, block_len, max_tokens, temperature, max_blocks=128,
                 eos_pos=None, extra_stop=None, rng_seed=9, left_pad=0, logits_dtype="f32", peak=8.0, top_k=None, top_p=None):
    """JacobiDecoderNonGreedyOnPolicy.generate_rollout_records_batch (JDO:494-614) with every random draw injected."""
    eos_id, pad_id = vocab - 1, vocab - 2
    stop_ids = [eos_id] + ([extra_stop] if extra_stop is not None else [])
    H = EngineHarness(vocab, logits_dtype=TORCH_DTYPES[logits_dtype])
    dec = jdo_mod.JacobiDecoderNonGreedyOnPolicy(H.bm, forward_step=H.forward_step, forward_step_batch=H.forward_step_batch,
                                                 eos_token_id=stop_ids if len(stop_ids) > 1 else eos_id,
                                                 pad_token_id=pad_id, vocab_size=vocab, device=torch.device("cpu"))
    seqs, descr = [], []
    for i, (sd, pl) in enumerate(zip(seeds, prompt_lens)):
        ep = None if eos_pos is None else eos_pos[i]
        reserved = (pad_id,) + ((extra_stop,) if extra_stop is not None else ())
        m = ScriptedModel(vocab, sd, robust, pl, eos_id=eos_id, eos_pos=ep, reserved=reserved, peak=peak)
        sp = SamplingParams(temperature=temperature, max_tokens=max_tokens, decode_strategy="jacobi",
                            jacobi_block_len=block_len, jacobi_max_iterations=max_blocks, jacobi_on_policy=True)
        if top_k is not None:                   # planted: JDO:132-133 reads them with getattr, like JDN
            sp.top_k = top_k
        if top_p is not None:
            sp.top_p = top_p
        seq = H.add_seq(m, sp, None)
        seqs.append(seq)
        descr.append(dict(model=m.describe(), prompt=m.prompt()))
    inits = CounterStream(rng_seed * 5 + 1)
    unis = CounterStream(rng_seed * 5 + 2)
    multi = CounterStream(rng_seed * 5 + 3)

    def _rand(size=(), **kw):
        return torch.tensor(unis.uniform(), dtype=torch.float32)

    with patched(jdo_mod, "random", ScriptedRandom(inits)), patched(torch, "rand", _rand), \
            patched(torch, "multinomial", scripted_multinomial(multi)):
        records, metrics = dec.generate_rollout_records_batch(seqs, n_token_seq_len=None, return_metrics=True)
    params = dict(vocab=vocab, eos_id=eos_id, pad_id=pad_id, stop_ids=stop_ids, rng_seed=rng_seed,
                  block_len=block_len, max_tokens=max_tokens, temperature=temperature, max_blocks=max_blocks)
    if logits_dtype != "f32":
        params["logits_dtype"] = logits_dtype
    if top_k is not None:
        params["top_k"] = top_k
    if top_p is not None:
        params["top_p"] = top_p
    return dict(name=name, kind="jdo", params=params,
                seqs=descr, records=[{str(k): v for k, v in r.items()} for r in records], metrics=metrics,
                final=[dict(token_ids=s.token_ids, num_cached_tokens=s.num_cached_tokens) for s in seqs],
                draws=dict(inits=inits.k, uniforms=unis.k, multinomial=multi.k), forwards=H.trace)


# --------------------------------------------------------------------------------------
# kernel-level vectors: torch.argmax / accept-length semantics (ties, NaN, -inf, bf16)
# --------------------------------------------------------------------------------------
def run_argmax_vectors():
    g = torch.Generator().manual_seed(1234)
    cases = []
    V = 97
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(12, V, generator=g).to(dtype)
        x[1, 5] = x[1, 40] = 9.0                       # exact tie -> first index
        x[2, 96] = 11.0                                 # max at last index
        x[3, 0] = 11.0                                  # max at first index
        x[4, 17] = float("nan")                         # NaN counts as max
        x[5, 30] = float("nan"); x[5, 9] = float("nan")  # two NaNs -> first NaN
        x[6, :] = -float("inf")                         # all -inf -> index 0
        x[7, 20] = float("inf"); x[7, 3] = float("inf")  # +inf tie -> first
        x[8, :] = 0.0                                   # all equal -> 0
        x[9, 50] = float("inf"); x[9, 60] = float("nan")  # NaN beats +inf
        x[10, 10] = -0.0; x[10, :10] = -1.0; x[10, 11:] = -1.0; x[10, 44] = 0.0  # -0.0 == 0.0 tie -> first
        am = torch.argmax(x, dim=-1).tolist()
        cases.append(dict(dtype=str(dtype).split(".")[-1],
                          bits=(x.view(torch.int32) if dtype == torch.float32 else x.view(torch.int16).int()).tolist(),
                          argmax=am))
    # accept-length semantics (MB:482-486 / JD:253-293)
    acc = []
    rr = random.Random(7)
    for _ in range(40):
        L = rr.randint(1, 9)
        B = rr.randint(1, 4)
        draft = [[rr.randint(0, 3) for _ in range(L)] for _ in range(B)]
        greedy = [[rr.randint(0, 3) for _ in range(L)] for _ in range(B)]
        d, gt = torch.tensor(draft), torch.tensor(greedy)
        mismatch = d[:, 1:] != gt[:, :-1]
        a = ((mismatch.cumsum(dim=-1) == 0).sum(dim=-1) + 1).tolist()
        best = int(torch.argmax(torch.tensor(a)))
        acc.append(dict(draft=draft, greedy=greedy, accepted=a, best_idx=best))
    return dict(kind="kernel_vectors", argmax=cases, accept=acc)


# --------------------------------------------------------------------------------------
# block bookkeeping the Jacobi decoders drive (BM:114-121, 195-276, 534-564): scripted op sequences on the reference's
# BlockManager + Sequence, state recorded after every op
# --------------------------------------------------------------------------------------
def run_softmax_vectors():
    """_build_target_probs of the two non-greedy decoders (JDN:110-123, JDO:128-136) on bf16 and fp32 logits: the
    temperature-scaled logits and the probabilities as bit patterns.  The bf16 scaling (one rounding of a float32
    quotient) is reproducible bit for bit; the softmax differs between float32 implementations in the last place."""
    from inference_engine.engine.jacobi_decoding_nongreedy import _build_target_probs, _softmax_with_temperature
    out = []
    g = torch.Generator().manual_seed(77)
    for V, scale in ((64, 1.0), (257, 3.0), (1000, 2.0)):
        for T in (1.0, 0.7, 1.3, 0.25):
            x = (torch.randn(2, V, generator=g) * scale).to(torch.bfloat16)
            x[0, 5] = 9.0
            sp = SamplingParams(temperature=T, max_tokens=4, decode_strategy="jacobi")
            scaled = x if T == 1.0 else x / float(T)                              # JDN:66-69
            probs = _build_target_probs(x, sp)
            assert probs.dtype == torch.bfloat16 and torch.equal(probs, _softmax_with_temperature(x, T))
            xf = x.float()
            pf = _build_target_probs(xf, sp)
            out.append(dict(V=V, temperature=T,
                            logits_bf16=x.view(torch.int16).numpy().astype(np.uint16).reshape(-1).tolist(),
                            scaled_bf16=scaled.view(torch.int16).numpy().astype(np.uint16).reshape(-1).tolist(),
                            probs_bf16=probs.view(torch.int16).numpy().astype(np.uint16).reshape(-1).tolist(),
                            probs_f32_of_f32_logits=pf.view(torch.int32).numpy().astype(np.uint32).reshape(-1).tolist()))
    return out


def run_filter_vectors():
    """_build_target_probs (JDN:110-123) with top_k / top_p planted on the request object, on bf16 and on float32 logits: the
    filtered, renormalised distribution as bit patterns.  Peaked rows (a trained model's shape: the kept set ends between
    clearly different probabilities) and flat rows (bf16: the kept set usually ends inside a group of EQUAL probabilities,
    where torch.topk / torch.sort choose by kernel, not by rule)."""
    from inference_engine.engine.jacobi_decoding_nongreedy import _build_target_probs
    out = []
    g = torch.Generator().manual_seed(91)
    for V, scale in ((64, 2.0), (300, 3.0), (2000, 2.5), (2000, 6.0)):
        for T in (1.0, 0.7):
            for top_k, top_p in ((None, 0.9), (5, None), (20, 0.8), (3, 0.5), (50, 0.95), (None, 0.3), (1, None), (None, 0.999), (V - 1, 0.05)):
                x = (torch.randn(3, V, generator=g) * scale).to(torch.bfloat16)
                x[0, 7] = 9.0
                sp = SamplingParams(temperature=T, max_tokens=4, decode_strategy="jacobi")
                if top_k is not None:
                    sp.top_k = top_k
                if top_p is not None:
                    sp.top_p = top_p
                pb = _build_target_probs(x, sp)
                pf = _build_target_probs(x.float(), sp)
                assert pb.dtype == torch.bfloat16 and pf.dtype == torch.float32
                out.append(dict(V=V, temperature=T, top_k=top_k, top_p=top_p,
                                logits_bf16=x.view(torch.int16).numpy().astype(np.uint16).reshape(-1).tolist(),
                                probs_bf16=pb.view(torch.int16).numpy().astype(np.uint16).reshape(-1).tolist(),
                                probs_f32_of_f32_logits=pf.view(torch.int32).numpy().astype(np.uint32).reshape(-1).tolist()))
    return out


def run_bm_case(seed, num_blocks=48, block_size=256):
    rr = random.Random(seed)
    bm = BlockManager(num_blocks, block_size)
    ops_log = []
    seqs = []
    for _ in range(rr.randint(1, 3)):
        plen = rr.choice([1, 5, 200, 255, 256, 257, 511, 600])
        seq = Sequence([rr.randrange(50) for _ in range(plen)], SamplingParams(temperature=0.0, max_tokens=4096))
        bm.allocate(seq)
        seq.num_cached_tokens = len(seq)
        seqs.append(seq)
        ops_log.append(dict(op="allocate", seq=len(seqs) - 1, prompt_len=plen))

    def snap():
        return dict(tables=[len(s.block_table) for s in seqs], cached=[s.num_cached_tokens for s in seqs],
                    lens=[len(s) for s in seqs], spec=[s.num_permanent_spec_blocks for s in seqs], free=len(bm.free_block_ids))
    ops_log[-1]["after"] = snap()
    for _ in range(40):
        i = rr.randrange(len(seqs))
        seq = seqs[i]
        kind = rr.choice(["jacobi", "jacobi", "jacobi", "ar"])
        if kind == "ar":
            seq.append_token(rr.randrange(50))
            bm.may_append(seq)
            seq.num_cached_tokens = len(seq)
            ops_log.append(dict(op="ar", seq=i, after=snap()))
            continue
        L = rr.choice([2, 4, 16, 33, 64, 300])
        S = len(seq)
        need = (S + L - 1 + block_size - 1) // block_size                      # MR:1166-1198 (the forward's table growth)
        committed = (S + block_size - 1) // block_size
        cur = len(seq.block_table)
        if cur > need:
            seq.block_table = seq.block_table[:need]
        for _k in range(max(0, need - cur)):
            bid = bm.free_block_ids[0]
            bm._allocate_block_no_clear(bid)
            seq.block_table.append(bid)
        seq.num_permanent_spec_blocks = max(seq.num_permanent_spec_blocks, need - committed)
        seq.num_cached_tokens = S - 1 + L
        acc = rr.randint(1, L)                                                    # acc_len incl. the seed (JD:609-646)
        if acc > 1:
            seq.extend_tokens([rr.randrange(50) for _ in range(acc - 1)])
            bm.may_append_batch(seq, acc - 1)
            spec = acc - 1
        else:
            seq.append_token(rr.randrange(50))
            bm.may_append(seq)
            spec = 1
        trim = L - 1 - spec
        if trim > 0:
            bm.trim_kv_only_fast(seq, trim)
        ops_log.append(dict(op="jacobi", seq=i, L=L, acc=acc, after=snap()))
    return dict(seed=seed, num_blocks=num_blocks, block_size=block_size, ops=ops_log)


# --------------------------------------------------------------------------------------
# paged-KV slot mapping pattern of the batched Jacobi forward (MR:965-986), run from the reference's ModelRunner method
# --------------------------------------------------------------------------------------
def run_slot_pattern_vectors():
    from inference_engine.engine.model_runner import ModelRunner
    real_arange = torch.arange

    def cpu_arange(*a, **k):
        k.pop("device", None)                      # the method asks for device='cuda'
        return real_arange(*a, **k)

    out = []
    for block_size in (256, 512):
        me = types.SimpleNamespace(block_size=block_size, slot_mapping_lut={}, slot_mapping_cache={})
        for S, L in [(1, 2), (1, 64), (5, 8), (255, 2), (256, 2), (256, 16), (257, 33), (511, 64), (512, 64), (513, 3), (1000, 32),
                     (2047, 64), (4096, 17)]:
            with patched(torch, "arange", cpu_arange):
                blk, off = ModelRunner._get_slot_mapping_pattern(me, S, L)
            out.append(dict(block_size=block_size, seq_len=S, draft_len=L, block_indices=blk.tolist(), offsets=off.tolist()))
    return out


# --------------------------------------------------------------------------------------
FULL_V = 152064          # the width of Qwen2.5-Coder-7B's logits (BASELINE.json's model): every entry point once more at it


def run_full_vocabulary_cases():
    """Round 5: the reference's five entry points at V = 152 064 — its own argmax / softmax / multinomial shim over rows of the
    real width (earlier files stop at V = 2 000; at this width an id needs 18 bits, a row is many chunks of the argmax stream,
    the softmax sum runs over 152 064 terms and the bf16 image of the noise collides everywhere).  The fixtures stay small: the
    scripted model regenerates the logits from its descriptor."""
    import itertools
    # the non-greedy records keep the reference's request ids (``seq_ids``), which come from a process-global counter of its
    # Sequence class: restart it here, so that this file does not depend on what ran before it in the process
    Sequence.counter = itertools.count()
    V = FULL_V
    mbs = [
        run_mb_case("fv_mb_baseline_knobs", vocab=V, seed=9001, robust=70, prompt_len=24, n=32, K=2, r=0.85, pool=4, max_calls=3),
        run_mb_case("fv_mb_candidates_period7", vocab=V, seed=9002, robust=45, prompt_len=14, n=32, K=2, r=0.6, pool=8, period=7,
                    max_calls=3),
        run_mb_case("fv_mb_eos_n16", vocab=V, seed=9003, robust=80, prompt_len=9, n=16, K=2, r=0.5, pool=4, eos_pos=9 + 37, max_calls=4),
    ]
    sbs = [
        run_sb_case("fv_sb_n32", vocab=V, seed=9010, robust=70, prompt_len=21, n=32, max_calls=3),
        run_sb_case("fv_sb_n16_eos", vocab=V, seed=9011, robust=60, prompt_len=12, n=16, eos_pos=12 + 20, max_calls=4),
    ]
    jds = [
        run_jd_case("fv_jd_batch4_L32", vocab=V, seeds=[9020, 9021, 9022, 9023], robust=75, prompt_lens=[9, 14, 250, 31],
                    block_lens=[32, 32, 32, 32], max_tokens=[70, 40, 64, 50], eos_pos=[None, 14 + 22, None, None], pad_seed=911),
        run_jd_case("fv_jd_batch2_mixedL", vocab=V, seeds=[9024, 9025], robust=60, prompt_lens=[7, 12], block_lens=[16, 8],
                    max_tokens=[40, 30], pad_seed=912),
    ]
    # sampling: ln(V) = 11.9 — the planted id needs a peak of ~13 to carry real mass against 152 062 noise ids
    jdns = [
        run_jdn_case("fv_jdn_f32_T1", vocab=V, seeds=[9030, 9031], robust=80, prompt_lens=[9, 14], block_len=16, max_tokens=40,
                     temperature=1.0, rng_seed=71, peak=13.0),
        run_jdn_case("fv_jdn_bf16_T08_L32", vocab=V, seeds=[9032, 9033], robust=80, prompt_lens=[8, 11], block_len=32,
                     max_tokens=40, temperature=0.8, rng_seed=72, logits_dtype="bf16", peak=16.0),
        run_jdn_case("fv_jdn_bf16_T1_eos", vocab=V, seeds=[9035, 9036], robust=85, prompt_lens=[8, 6], block_len=16, max_tokens=48,
                     temperature=1.0, eos_pos=[8 + 19, None], rng_seed=73, logits_dtype="bf16", peak=14.0),
        run_jdn_case("fv_jdn_f32_k50_p09", vocab=V, seeds=[9037, 9038], robust=75, prompt_lens=[9, 7], block_len=16, max_tokens=40,
                     temperature=0.8, rng_seed=74, peak=12.0, top_k=50, top_p=0.9),
    ]
    jdos = [
        run_jdo_case("fv_jdo_f32_T1", vocab=V, seeds=[9040, 9041], robust=85, prompt_lens=[8, 12], block_len=16, max_tokens=40,
                     temperature=1.0, rng_seed=81, peak=13.0),
        run_jdo_case("fv_jdo_bf16_T08_stop", vocab=V, seeds=[9042, 9043], robust=90, prompt_lens=[8, 5], block_len=16, max_tokens=40,
                     temperature=0.8, eos_pos=[8 + 13, None], rng_seed=82, logits_dtype="bf16", peak=13.0),
        # (the one above is NOT reproducible by construction: its multinomial draws land among the noise ids, where torch's bf16
        # softmax — one bf16 ulp off the exactly rounded value on < 1 % of 152 064 entries — shifts the running sum by more than
        # an id's mass; kept as evidence, tests/test_oracle_golden.py SOFTMAX_ULP_OBSERVABLE.  With the mass on the planted ids:)
        run_jdo_case("fv_jdo_bf16_T08_peak16", vocab=V, seeds=[9044, 9045], robust=75, prompt_lens=[8, 5], block_len=16, max_tokens=40,
                     temperature=0.8, eos_pos=[8 + 13, None], rng_seed=83, logits_dtype="bf16", peak=16.0),
    ]
    return dict(mb=mbs, sb=sbs, jd=jds, jdn=jdns, jdo=jdos)


def check() -> int:
    """Regenerate every fixture into a temporary directory and compare it with the committed file, byte for byte."""
    import tempfile
    global OUT_DIR
    with tempfile.TemporaryDirectory() as d:
        OUT_DIR = Path(d)
        with contextlib.redirect_stdout(io.StringIO()):
            generate([a for a in sys.argv[1:] if a in ("--full-vocabulary", "--skip-full-vocabulary")])
        made = sorted(p.name for p in OUT_DIR.glob("*.json"))
        want = sorted(p.name for p in GOLDEN_DIR.glob("*.json"))
        if "--full-vocabulary" in sys.argv:
            want = ["fullvocab_cases.json"]
        elif "--skip-full-vocabulary" in sys.argv:
            want = [w for w in want if w != "fullvocab_cases.json"]
        bad = [n for n in want if n not in made or (OUT_DIR / n).read_bytes() != (GOLDEN_DIR / n).read_bytes()]
        extra = [n for n in made if n not in want]
    print(f"gen_golden --check: {len(want) - len(bad)} of {len(want)} fixture files regenerate byte-identical from {REF}"
          + (f"; DIFFERENT: {bad}" if bad else "") + (f"; not committed: {extra}" if extra else ""))
    return 1 if bad or extra else 0


def main():
    if "--out" in sys.argv:
        global OUT_DIR
        OUT_DIR = Path(sys.argv[sys.argv.index("--out") + 1])
        OUT_DIR.mkdir(parents=True, exist_ok=True)
    if "--check" in sys.argv:
        sys.exit(check())
    generate(sys.argv[1:])


def generate(argv):
    torch.manual_seed(0)
    if "--full-vocabulary" in argv:              # only that file (the others take ~50 s)
        with open(OUT_DIR / "fullvocab_cases.json", "w") as f:
            json.dump(run_full_vocabulary_cases(), f, separators=(",", ":"))
        print("wrote", OUT_DIR / "fullvocab_cases.json", (OUT_DIR / "fullvocab_cases.json").stat().st_size // 1024, "KiB")
        return
    mbs = []
    add = lambda *a, **k: mbs.append(run_mb_case(*a, **k))
    # basic sweeps
    add("n8_K1", vocab=64, seed=1, robust=60, prompt_len=6, n=8, K=1, r=0.85, pool=4)
    add("n8_K2_r50_trace", vocab=1000, seed=2, robust=70, prompt_len=6, n=8, K=2, r=0.5, pool=4, max_calls=4)
    add("n16_K2_r85", vocab=1000, seed=3, robust=70, prompt_len=11, n=16, K=2, r=0.85, pool=4)
    add("n16_K3_r50", vocab=64, seed=4, robust=65, prompt_len=9, n=16, K=3, r=0.5, pool=4)
    add("n32_K1", vocab=1000, seed=5, robust=70, prompt_len=20, n=32, K=1, r=0.85, pool=4)
    add("n32_K2_r85_default", vocab=1000, seed=6, robust=70, prompt_len=24, n=32, K=2, r=0.85, pool=4)
    add("n32_K2_r50_pool8", vocab=64, seed=7, robust=70, prompt_len=17, n=32, K=2, r=0.5, pool=8)
    add("n32_K3_r50_pool4_small_vocab", vocab=32, seed=8, robust=55, prompt_len=13, n=32, K=3, r=0.5, pool=4)
    add("n16_pool1", vocab=64, seed=9, robust=60, prompt_len=7, n=16, K=2, r=0.5, pool=1)
    add("n16_period_candidates", vocab=64, seed=10, robust=45, prompt_len=10, n=16, K=2, r=0.6, pool=4, period=5)
    add("n32_period_candidates_pool8", vocab=48, seed=11, robust=40, prompt_len=12, n=32, K=2, r=0.6, pool=8, period=7)
    add("n16_period3_K3", vocab=40, seed=12, robust=35, prompt_len=8, n=16, K=3, r=0.4, pool=4, period=3)
    add("n16_lookahead_half", vocab=48, seed=13, robust=45, prompt_len=9, n=16, K=2, r=0.6, pool=4, period=5, lookahead=0.5)
    add("n8_low_robust_acc1", vocab=64, seed=14, robust=0, prompt_len=5, n=8, K=2, r=0.85, pool=4)
    add("n8_full_robust_all_accept", vocab=64, seed=15, robust=100, prompt_len=5, n=8, K=2, r=0.5, pool=4)
    add("n16_max_iter3", vocab=64, seed=16, robust=30, prompt_len=6, n=16, K=2, r=0.85, pool=4, max_iter=3)
    add("n64_K2", vocab=1000, seed=17, robust=75, prompt_len=30, n=64, K=2, r=0.85, pool=4, max_calls=3)
    add("n16_K4_r25", vocab=64, seed=18, robust=60, prompt_len=9, n=16, K=4, r=0.25, pool=4, period=6)
    # EOS placements: inside accepted prefix, as first token, as "next token", at block boundary (Q13)
    add("n8_eos_mid", vocab=64, seed=20, robust=70, prompt_len=6, n=8, K=2, r=0.5, pool=4, eos_pos=6 + 11)
    add("n8_eos_first_token", vocab=64, seed=21, robust=70, prompt_len=6, n=8, K=2, r=0.5, pool=4, eos_pos=6)
    add("n8_eos_at_block_end_Q13", vocab=64, seed=22, robust=100, prompt_len=6, n=8, K=1, r=0.5, pool=4, eos_pos=6 + 8)
    add("n8_eos_second_block_start", vocab=64, seed=23, robust=80, prompt_len=6, n=8, K=2, r=0.5, pool=4, eos_pos=6 + 8 + 1)
    add("n16_eos_in_pseudo", vocab=64, seed=24, robust=85, prompt_len=7, n=16, K=2, r=0.3, pool=4, eos_pos=7 + 20)
    add("n16_eos_next_token", vocab=64, seed=25, robust=50, prompt_len=7, n=16, K=2, r=0.85, pool=4, eos_pos=7 + 5)
    for sd in range(30, 42):
        rr = random.Random(sd)
        n = rr.choice([8, 16, 32])
        add(f"rand_{sd}", vocab=rr.choice([24, 48, 200]), seed=sd, robust=rr.choice([30, 50, 70, 90]),
            prompt_len=rr.randint(4, 20), n=n, K=rr.choice([1, 2, 2, 3]), r=rr.choice([0.3, 0.5, 0.85]),
            pool=rr.choice([2, 4, 8]), period=rr.choice([0, 0, 4, 9]),
            eos_pos=rr.choice([None, None, rr.randint(4, 20) + rr.randint(0, 3 * n)]), max_calls=4)
    # edge knobs (appended after the sweeps so earlier cases keep their indices)
    add("n16_pool0_no_recycling", vocab=64, seed=43, robust=45, prompt_len=9, n=16, K=2, r=0.6, pool=0, period=5)
    add("n64_K2_pool4_candidates", vocab=48, seed=44, robust=45, prompt_len=14, n=64, K=2, r=0.85, pool=4, period=7, max_calls=3)
    add("n4_K2_tiny_block", vocab=64, seed=45, robust=60, prompt_len=5, n=4, K=2, r=0.5, pool=4, period=3)
    add("n16_r100_spawn_when_full", vocab=64, seed=46, robust=70, prompt_len=8, n=16, K=2, r=1.0, pool=4)
    add("n16_r005_spawn_at_once", vocab=64, seed=47, robust=70, prompt_len=8, n=16, K=2, r=0.05, pool=4, period=4)
    add("n32_lookahead_one", vocab=48, seed=48, robust=40, prompt_len=10, n=32, K=2, r=0.6, pool=8, period=6, lookahead=1.0)
    # configurations on which the reference itself raises (K >= 3: a pseudo block comes back with k candidate rows while the
    # real-active draft has B not in {1, k} rows, MB:482)
    raises = [run_mb_case("raise_n16_K3_r005", vocab=64, seed=47, robust=70, prompt_len=8, n=16, K=3, r=0.05, pool=4,
                          period=4, allow_raise=True)]
    for sd in range(60, 90):
        if len(raises) >= 4:
            break
        c = run_mb_case(f"raise_scan_{sd}", vocab=48, seed=sd, robust=45, prompt_len=9, n=16, K=3, r=0.3, pool=8, period=5,
                        allow_raise=True)
        if c["summary"]["stop_reason"] == "reference_raised":
            raises.append(c)
    assert all(c["summary"]["stop_reason"] == "reference_raised" for c in raises)
    sbs = [
        run_sb_case("sb_n16", vocab=1000, seed=50, robust=70, prompt_len=12, n=16),
        run_sb_case("sb_n8_all_accept", vocab=64, seed=51, robust=100, prompt_len=5, n=8),
        run_sb_case("sb_n8_eos_mid", vocab=64, seed=52, robust=70, prompt_len=5, n=8, eos_pos=5 + 10),
        run_sb_case("sb_n8_eos_next", vocab=64, seed=53, robust=40, prompt_len=5, n=8, eos_pos=5 + 3),
        run_sb_case("sb_n8_eos_bonus", vocab=64, seed=54, robust=100, prompt_len=5, n=8, eos_pos=5 + 8),
        run_sb_case("sb_n32", vocab=200, seed=55, robust=60, prompt_len=21, n=32),
    ]
    jds = [
        run_jd_case("jd_single_L8", vocab=64, seeds=[60], robust=70, prompt_lens=[9], block_lens=[8],
                    max_tokens=[40], batch=False),
        run_jd_case("jd_single_L16_noprefill", vocab=200, seeds=[61], robust=60, prompt_lens=[14], block_lens=[16],
                    max_tokens=[48], batch=False, use_prefill_draft=False),
        run_jd_case("jd_single_eos", vocab=64, seeds=[62], robust=80, prompt_lens=[7], block_lens=[8],
                    max_tokens=[64], eos_pos=[7 + 13], batch=False),
        run_jd_case("jd_batch3_mixedL", vocab=64, seeds=[63, 64, 65], robust=70, prompt_lens=[9, 12, 6],
                    block_lens=[8, 16, 8], max_tokens=[24, 30, 40]),
        run_jd_case("jd_batch4_eos_overshoot", vocab=100, seeds=[66, 67, 68, 69], robust=75,
                    prompt_lens=[5, 8, 250, 255], block_lens=[16, 16, 16, 16], max_tokens=[24, 24, 40, 40],
                    eos_pos=[5 + 9, None, None, 255 + 30]),
        run_jd_case("jd_batch2_robust0_ar_fallback", vocab=64, seeds=[70, 71], robust=0, prompt_lens=[6, 6],
                    block_lens=[8, 8], max_tokens=[12, 12]),
        run_jd_case("jd_batch2_robust100", vocab=64, seeds=[72, 73], robust=100, prompt_lens=[6, 9],
                    block_lens=[8, 8], max_tokens=[30, 30]),
        run_jd_case("jd_batch2_maxiters", vocab=64, seeds=[74, 75], robust=20, prompt_lens=[6, 9],
                    block_lens=[8, 8], max_tokens=[100, 100], max_iters=5),
    ]
    jdns = [
        run_jdn_case("jdn_single_T1", vocab=64, seeds=[80], robust=70, prompt_lens=[8], block_len=8,
                     max_tokens=32, temperature=1.0, batch=False),
        run_jdn_case("jdn_batch3_T07", vocab=64, seeds=[81, 82, 83], robust=70, prompt_lens=[8, 5, 11], block_len=8,
                     max_tokens=24, temperature=0.7),
        run_jdn_case("jdn_batch2_eos", vocab=64, seeds=[84, 85], robust=90, prompt_lens=[8, 5], block_len=16,
                     max_tokens=48, temperature=0.5, eos_pos=[8 + 6, 5 + 20]),
    ]
    jdos = [
        run_jdo_case("jdo_single_T1", vocab=64, seeds=[90], robust=70, prompt_lens=[8], block_len=8, max_tokens=24,
                     temperature=1.0),
        run_jdo_case("jdo_batch2_stop_T07", vocab=64, seeds=[91, 92], robust=90, prompt_lens=[8, 5], block_len=8,
                     max_tokens=40, temperature=0.7, eos_pos=[8 + 11, 5 + 3]),
        run_jdo_case("jdo_budget_padfill", vocab=64, seeds=[93, 94], robust=80, prompt_lens=[6, 9], block_len=16,
                     max_tokens=21, temperature=0.5),
        run_jdo_case("jdo_maxblocks2", vocab=64, seeds=[95], robust=60, prompt_lens=[7], block_len=8, max_tokens=64,
                     temperature=1.0, max_blocks=2),
        run_jdo_case("jdo_two_stop_ids", vocab=100, seeds=[96, 97, 98], robust=85, prompt_lens=[5, 12, 7], block_len=8,
                     max_tokens=48, temperature=0.8, eos_pos=[None, 12 + 9, 7 + 30], extra_stop=41),
        run_jdo_case("jdo_hot_T2", vocab=48, seeds=[99], robust=100, prompt_lens=[6], block_len=8, max_tokens=16,
                     temperature=2.0),
    ]
    # round 2: the block length BASELINE config 5 is quoted at (n = 32), flatter distributions (peak < 8: real accept /
    # reject / bonus traffic), larger vocabularies, and bfloat16 logits — what the engine hands the verifier (MR:1382)
    jdns2 = [
        run_jdn_case("jdn2_f32_L32_batch4", vocab=300, seeds=[100, 101, 102, 103], robust=80, prompt_lens=[8, 5, 11, 20],
                     block_len=32, max_tokens=96, temperature=1.0, peak=5.0, rng_seed=21),
        run_jdn_case("jdn2_f32_L32_V2000_T08", vocab=2000, seeds=[104, 105], robust=85, prompt_lens=[9, 14], block_len=32,
                     max_tokens=80, temperature=0.8, peak=9.0, rng_seed=22),
        run_jdn_case("jdn2_bf16_single_T1", vocab=64, seeds=[110], robust=70, prompt_lens=[8], block_len=8,
                     max_tokens=32, temperature=1.0, batch=False, logits_dtype="bf16", peak=4.0, rng_seed=23),
        run_jdn_case("jdn2_bf16_batch3_T07", vocab=64, seeds=[111, 112, 113], robust=70, prompt_lens=[8, 5, 11], block_len=8,
                     max_tokens=24, temperature=0.7, logits_dtype="bf16", peak=4.0, rng_seed=24),
        run_jdn_case("jdn2_bf16_batch2_eos_T05", vocab=64, seeds=[114, 115], robust=90, prompt_lens=[8, 5], block_len=16,
                     max_tokens=48, temperature=0.5, eos_pos=[8 + 6, 5 + 20], logits_dtype="bf16", rng_seed=25),
        run_jdn_case("jdn2_bf16_L32_batch4_T1", vocab=300, seeds=[116, 117, 118, 119], robust=80, prompt_lens=[8, 5, 11, 20],
                     block_len=32, max_tokens=96, temperature=1.0, logits_dtype="bf16", peak=5.0, rng_seed=26),
        run_jdn_case("jdn2_bf16_L32_V2000_T13", vocab=2000, seeds=[120, 121, 122], robust=85, prompt_lens=[9, 14, 6],
                     block_len=32, max_tokens=80, temperature=1.3, logits_dtype="bf16", peak=10.0, rng_seed=27),
        run_jdn_case("jdn2_bf16_L32_batch8_T09", vocab=500, seeds=list(range(123, 131)), robust=75,
                     prompt_lens=[6, 7, 8, 9, 10, 11, 12, 13], block_len=32, max_tokens=64, temperature=0.9,
                     logits_dtype="bf16", peak=6.0, rng_seed=28),
        run_jdn_case("jdn2_bf16_onehot_collisions", vocab=64, seeds=[131, 132], robust=60, prompt_lens=[7, 9], block_len=8,
                     max_tokens=40, temperature=0.25, logits_dtype="bf16", peak=8.0, rng_seed=29),
    ]
    jdos2 = [
        run_jdo_case("jdo2_f32_L32", vocab=300, seeds=[140, 141], robust=80, prompt_lens=[8, 12], block_len=32,
                     max_tokens=70, temperature=1.0, peak=5.0, rng_seed=31),
        run_jdo_case("jdo2_bf16_single_T1", vocab=64, seeds=[142], robust=70, prompt_lens=[8], block_len=8, max_tokens=24,
                     temperature=1.0, logits_dtype="bf16", peak=4.0, rng_seed=32),
        run_jdo_case("jdo2_bf16_batch2_stop_T07", vocab=64, seeds=[143, 144], robust=90, prompt_lens=[8, 5], block_len=8,
                     max_tokens=40, temperature=0.7, eos_pos=[8 + 11, 5 + 3], logits_dtype="bf16", rng_seed=33),
        run_jdo_case("jdo2_bf16_L32_V1000_T12", vocab=1000, seeds=[145, 146], robust=85, prompt_lens=[10, 7], block_len=32,
                     max_tokens=70, temperature=1.2, logits_dtype="bf16", peak=7.0, rng_seed=34),
        run_jdo_case("jdo2_bf16_two_stop_ids", vocab=100, seeds=[147, 148, 149], robust=85, prompt_lens=[5, 12, 7],
                     block_len=8, max_tokens=48, temperature=0.8, eos_pos=[None, 12 + 9, 7 + 30], extra_stop=41,
                     logits_dtype="bf16", peak=5.0, rng_seed=35),
        run_jdo_case("jdo2_bf16_hot_T2", vocab=48, seeds=[150], robust=100, prompt_lens=[6], block_len=8, max_tokens=16,
                     temperature=2.0, logits_dtype="bf16", rng_seed=36),
    ]
    # round 2: the multiblock function at BASELINE's knobs (n = 32, K = 2, r = 0.85, pool = 4) over seeded acceptance
    # patterns, plus a seeded sweep of the other knobs
    mbs2 = []
    for sd in range(400, 408):
        rr = random.Random(sd)
        pl = rr.randint(6, 28)
        mbs2.append(run_mb_case(f"mb2_baseline_{sd}", vocab=rr.choice([48, 200, 1000]), seed=sd, robust=rr.choice([35, 55, 75, 90]),
                                prompt_len=pl, n=32, K=2, r=0.85, pool=4, period=rr.choice([0, 5, 9, 13]),
                                eos_pos=rr.choice([None, None, pl + rr.randint(20, 90)]), max_calls=3))
    for sd in range(408, 416):
        rr = random.Random(sd)
        n = rr.choice([8, 16, 24, 48])
        pl = rr.randint(4, 20)
        mbs2.append(run_mb_case(f"mb2_rand_{sd}", vocab=rr.choice([24, 64, 300]), seed=sd, robust=rr.choice([25, 50, 70, 95]),
                                prompt_len=pl, n=n, K=rr.choice([1, 2, 2]), r=rr.choice([0.25, 0.6, 0.85, 1.0]),
                                pool=rr.choice([0, 1, 2, 4, 8]), period=rr.choice([0, 0, 3, 7]),
                                lookahead=rr.choice([0.0, 0.0, 0.4]),
                                eos_pos=rr.choice([None, None, pl + rr.randint(0, 3 * n)]), max_calls=3))
    # round 4: K >= 3 with a small spawn ratio — the reference's block lists run away (promotion decrements num_blocks without
    # removing list entries, active_blocks goes negative, MB:629-716 / SURVEY Q3, Q4): a spawn per iteration, rows of up to
    # ~77 n tokens, a ret of several hundred tokens per call.  These pin the LARGEST rows the reference really produces
    # (every configuration below runs to the end without the MB:482 crash; all have pool = 2, i.e. never a stacked candidate).
    mbs3 = [run_mb_case("mb3_runaway_n4_K3_r005_full", vocab=48, seed=2172, robust=30, prompt_len=14, n=4, K=3, r=0.05, pool=2,
                        period=3, max_calls=2)]
    for nm, kw in [
        ("n8_K3_r005", dict(vocab=48, seed=2311, robust=30, prompt_len=20, n=8, K=3, r=0.05, pool=2, period=5)),
        ("n16_K3_r005", dict(vocab=64, seed=2195, robust=60, prompt_len=5, n=16, K=3, r=0.05, pool=2, period=3)),
        ("n32_K3_r005", dict(vocab=24, seed=2196, robust=75, prompt_len=15, n=32, K=3, r=0.05, pool=2, period=0)),
        ("n32_K3_r025", dict(vocab=64, seed=2276, robust=75, prompt_len=14, n=32, K=3, r=0.25, pool=2, period=0)),
        ("n8_K4_r025", dict(vocab=24, seed=2233, robust=30, prompt_len=6, n=8, K=4, r=0.25, pool=2, period=0)),
        ("n16_K4_r005", dict(vocab=200, seed=2179, robust=60, prompt_len=20, n=16, K=4, r=0.05, pool=2, period=7)),
        ("n32_K4_r005", dict(vocab=64, seed=2117, robust=75, prompt_len=5, n=32, K=4, r=0.05, pool=2, period=3)),
        ("n16_K5_r005", dict(vocab=24, seed=1160, robust=45, prompt_len=19, n=16, K=5, r=0.05, pool=2, period=7)),
    ]:
        mbs3.append(digest_mb_case(run_mb_case(f"mb3_runaway_{nm}", **kw)))
    # round 2: seeded sweeps of the single-block function and the engine decoder (the first sets are hand-picked and small)
    sbs2 = []
    for sd in range(200, 216):
        rr = random.Random(sd)
        n = rr.choice([4, 8, 16, 16, 32, 64])
        pl = rr.randint(3, 30)
        sbs2.append(run_sb_case(f"sb2_rand_{sd}", vocab=rr.choice([32, 64, 500]), seed=sd, robust=rr.choice([20, 50, 80, 100]),
                                prompt_len=pl, n=n, eos_pos=rr.choice([None, None, pl + rr.randint(0, 3 * n)]), max_calls=4))
    jds2 = []
    for sd in range(300, 312):
        rr = random.Random(sd)
        B = rr.choice([1, 2, 3, 5, 8])
        same = rr.random() < 0.5
        L0 = rr.choice([4, 8, 16, 32, 64])
        bls = [L0 if same else rr.choice([4, 8, 16, 32]) for _ in range(B)]
        pls = [rr.choice([rr.randint(3, 40), rr.randint(250, 262)]) if rr.random() < 0.25 else rr.randint(3, 40) for _ in range(B)]
        mts = [rr.randint(6, 90) for _ in range(B)]
        eps = [(pl + rr.randint(0, 40)) if rr.random() < 0.35 else None for pl in pls]
        jds2.append(run_jd_case(f"jd2_rand_{sd}", vocab=rr.choice([48, 64, 300]), seeds=[1000 + 10 * sd + i for i in range(B)],
                                robust=rr.choice([0, 30, 60, 85, 100]), prompt_lens=pls, block_lens=bls, max_tokens=mts,
                                eos_pos=eps if any(e is not None for e in eps) else None,
                                use_prefill_draft=rr.random() < 0.7, pad_seed=500 + sd, batch=(B > 1) or rr.random() < 0.5,
                                max_iters=rr.choice([128, 128, 128, 3])))
    # round 2: seeded sweeps of the two sampling decoders, alternating float32 / bfloat16 logits
    jdns3, jdos3 = [], []
    for sd in range(500, 510):
        rr = random.Random(sd)
        B = rr.choice([1, 2, 4, 6])
        pls = [rr.randint(4, 24) for _ in range(B)]
        L = rr.choice([8, 16, 32])
        jdns3.append(run_jdn_case(f"jdn3_rand_{sd}", vocab=rr.choice([64, 300, 1000]), seeds=[2000 + 10 * sd + i for i in range(B)],
                                  robust=rr.choice([50, 70, 90]), prompt_lens=pls, block_len=L, max_tokens=rr.randint(12, 3 * L),
                                  temperature=rr.choice([0.5, 0.8, 1.0, 1.0, 1.4]),
                                  eos_pos=[(pl + rr.randint(2, 40)) if rr.random() < 0.3 else None for pl in pls] if rr.random() < 0.5 else None,
                                  rng_seed=100 + sd, batch=B > 1, logits_dtype="bf16" if sd % 2 else "f32",
                                  peak=rr.choice([3.0, 5.0, 8.0])))
    for sd in range(520, 526):
        rr = random.Random(sd)
        B = rr.choice([1, 2, 3])
        pls = [rr.randint(4, 16) for _ in range(B)]
        L = rr.choice([8, 16, 32])
        jdos3.append(run_jdo_case(f"jdo3_rand_{sd}", vocab=rr.choice([64, 200, 1000]), seeds=[3000 + 10 * sd + i for i in range(B)],
                                  robust=rr.choice([60, 80, 95]), prompt_lens=pls, block_len=L, max_tokens=rr.randint(10, 2 * L + 8),
                                  temperature=rr.choice([0.6, 1.0, 1.3]), max_blocks=rr.choice([128, 128, 2]),
                                  eos_pos=[(pl + rr.randint(2, 30)) if rr.random() < 0.4 else None for pl in pls],
                                  rng_seed=200 + sd, logits_dtype="bf16" if sd % 2 else "f32", peak=rr.choice([4.0, 6.0, 8.0])))
    # round 5: the same decoder with top_k / top_p on the request objects (JDN:72-123)
    jdns4 = [
        run_jdn_case("jdn4_f32_topk5_T1", vocab=300, seeds=[700, 701, 702], robust=75, prompt_lens=[8, 5, 11], block_len=16,
                     max_tokens=40, temperature=1.0, rng_seed=41, peak=4.0, top_k=5),
        run_jdn_case("jdn4_f32_topp09_T08", vocab=1000, seeds=[703, 704], robust=80, prompt_lens=[9, 14], block_len=32,
                     max_tokens=64, temperature=0.8, rng_seed=42, peak=5.0, top_p=0.9),
        run_jdn_case("jdn4_f32_k20_p08_batch4", vocab=2000, seeds=[705, 706, 707, 708], robust=70, prompt_lens=[8, 5, 11, 20],
                     block_len=32, max_tokens=48, temperature=1.0, rng_seed=43, peak=3.0, top_k=20, top_p=0.8),
        run_jdn_case("jdn4_f32_single_p05", vocab=64, seeds=[709], robust=60, prompt_lens=[8], block_len=8,
                     max_tokens=24, temperature=1.3, rng_seed=44, batch=False, peak=3.0, top_p=0.5),
        run_jdn_case("jdn4_bf16_topk3_T07", vocab=300, seeds=[710, 711, 712], robust=75, prompt_lens=[8, 5, 11], block_len=16,
                     max_tokens=40, temperature=0.7, rng_seed=45, logits_dtype="bf16", peak=6.0, top_k=3),
        run_jdn_case("jdn4_bf16_topp09_L32", vocab=2000, seeds=[713, 714, 715], robust=80, prompt_lens=[9, 14, 6], block_len=32,
                     max_tokens=64, temperature=1.0, rng_seed=46, logits_dtype="bf16", peak=8.0, top_p=0.9),
        run_jdn_case("jdn4_bf16_k50_p095_eos", vocab=500, seeds=[716, 717], robust=85, prompt_lens=[8, 5], block_len=16,
                     max_tokens=40, temperature=0.9, eos_pos=[20, None], rng_seed=47, logits_dtype="bf16", peak=8.0, top_k=50, top_p=0.95),
        run_jdn_case("jdn4_bf16_flat_p08", vocab=1000, seeds=[718, 719], robust=60, prompt_lens=[7, 9], block_len=16,
                     max_tokens=32, temperature=1.0, rng_seed=48, logits_dtype="bf16", peak=2.0, top_p=0.8),
        # small vocabularies: the scripted logits' bf16 images hardly collide, the kept sets end between DIFFERENT probabilities
        run_jdn_case("jdn4_bf16_V64_p09", vocab=64, seeds=[720, 721, 722], robust=70, prompt_lens=[8, 5, 11], block_len=16,
                     max_tokens=40, temperature=1.0, rng_seed=51, logits_dtype="bf16", peak=3.0, top_p=0.9),
        run_jdn_case("jdn4_bf16_V64_p07_T12", vocab=64, seeds=[720, 721, 722], robust=70, prompt_lens=[8, 5, 11], block_len=16,
                     max_tokens=40, temperature=1.2, rng_seed=51, logits_dtype="bf16", peak=2.0, top_p=0.7),
        run_jdn_case("jdn4_bf16_V128_p09", vocab=128, seeds=[720, 721, 722], robust=70, prompt_lens=[8, 5, 11], block_len=16,
                     max_tokens=40, temperature=1.0, rng_seed=51, logits_dtype="bf16", peak=4.0, top_p=0.9),
        run_jdn_case("jdn4_bf16_V64_k7", vocab=64, seeds=[720, 721, 722], robust=70, prompt_lens=[8, 5, 11], block_len=16,
                     max_tokens=40, temperature=1.0, rng_seed=51, logits_dtype="bf16", peak=2.0, top_k=7),
        run_jdn_case("jdn4_bf16_V64_k10_p08_T08", vocab=64, seeds=[720, 721, 722], robust=70, prompt_lens=[8, 5, 11], block_len=16,
                     max_tokens=40, temperature=0.8, rng_seed=51, logits_dtype="bf16", peak=3.0, top_k=10, top_p=0.8),
    ]
    jdos4 = [
        run_jdo_case("jdo4_f32_k5", vocab=300, seeds=[800, 801], robust=80, prompt_lens=[8, 12], block_len=16, max_tokens=40,
                     temperature=1.0, rng_seed=61, peak=4.0, top_k=5),
        run_jdo_case("jdo4_f32_p09_T08", vocab=1000, seeds=[802, 803, 804], robust=85, prompt_lens=[9, 6, 14], block_len=32, max_tokens=48,
                     temperature=0.8, rng_seed=62, peak=5.0, top_p=0.9),
        run_jdo_case("jdo4_f32_k20_p08_stop", vocab=200, seeds=[805, 806], robust=90, prompt_lens=[8, 5], block_len=8, max_tokens=30,
                     temperature=1.0, eos_pos=[18, None], rng_seed=63, peak=3.0, top_k=20, top_p=0.8),
        run_jdo_case("jdo4_bf16_V64_p09", vocab=64, seeds=[807, 808], robust=75, prompt_lens=[8, 11], block_len=16, max_tokens=40,
                     temperature=1.0, rng_seed=64, logits_dtype="bf16", peak=3.0, top_p=0.9),
        run_jdo_case("jdo4_bf16_V64_k7_T12", vocab=64, seeds=[809, 810], robust=75, prompt_lens=[7, 9], block_len=8, max_tokens=24,
                     temperature=1.2, rng_seed=65, logits_dtype="bf16", peak=2.0, top_k=7),
        run_jdo_case("jdo4_bf16_V64_k10_p08", vocab=64, seeds=[811], robust=80, prompt_lens=[10], block_len=16, max_tokens=40,
                     temperature=0.8, rng_seed=66, logits_dtype="bf16", peak=3.0, top_k=10, top_p=0.8),
    ]
    flt = run_filter_vectors()
    smx = run_softmax_vectors()
    kv = run_argmax_vectors()
    slots = run_slot_pattern_vectors()
    bms = [run_bm_case(sd) for sd in range(12)]

    def dump(fname, obj):
        p = OUT_DIR / fname
        with open(p, "w") as f:
            json.dump(obj, f, separators=(",", ":"))
        print(f"wrote {p} ({p.stat().st_size / 1024:.1f} KiB)")

    dump("mb_cases.json", mbs)
    dump("mb_raises.json", raises)
    dump("sb_cases.json", sbs)
    dump("jd_cases.json", jds)
    dump("jdn_cases.json", jdns)
    dump("jdo_cases.json", jdos)
    dump("mb_cases_v2.json", mbs2)
    dump("mb_cases_v3.json", mbs3)
    dump("sb_cases_v2.json", sbs2)
    dump("jd_cases_v2.json", jds2)
    dump("jdn_cases_v3.json", jdns3)
    dump("jdo_cases_v3.json", jdos3)
    dump("jdn_cases_v2.json", jdns2)
    dump("jdo_cases_v2.json", jdos2)
    dump("jdn_cases_v4.json", jdns4)
    dump("jdo_cases_v4.json", jdos4)
    dump("filter_vectors.json", flt)
    dump("softmax_vectors.json", smx)
    dump("kernel_vectors.json", kv)
    dump("slot_cases.json", slots)
    dump("bm_cases.json", bms)
    if "--skip-full-vocabulary" not in argv:
        dump("fullvocab_cases.json", run_full_vocabulary_cases())
    # quick human summary
    for c in mbs + mbs3:
        fw = [f for cl in c["calls"] for f in cl["forwards"]]
        maxB = max((f["B"] if "B" in f else len(f["out"]) for f in fw), default=0)
        maxT = max((f["T"] if "T" in f else len(f["out"][0]) for f in fw), default=0)
        s = c["summary"]
        tpf = (s["new_tokens"] / s["total_iterations"]) if s["total_iterations"] else 0
        ban = sum((cl["banners"] for cl in c["calls"]), [])
        print(f"{c['name']:34s} calls={s['calls']} new={s['new_tokens']} iters={s['total_iterations']} tpf={tpf:.2f} "
              f"maxB={maxB} maxT={maxT} stop={s['stop_reason']} spawn={ban.count('spawn')} switch={ban.count('switch')}")


if __name__ == "__main__":
    main()
