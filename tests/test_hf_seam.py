"""HF-style seam functions (jacobiforcing_amd.hf_seam) against the golden call records of the reference's
jacobi_forward_greedy_multiblock (mb_cases.json) and jacobi_forward_greedy (sb_cases.json), and end to end on the
PyTorch Qwen2 backend."""
import types

import numpy as np
import pytest
import torch

from jacobiforcing_amd import hf_seam
from oracle.scripted_model import ScriptedModel

from .backends import device_for, use_backend
from .conftest import load_golden
from .test_decoder_e2e import scratch_forward, tiny_model

MB = load_golden("mb_cases.json")
MB2 = load_golden("mb_cases_v2.json")
FV = load_golden("fullvocab_cases.json")                 # round 5: the reference at V = 152 064
SB = load_golden("sb_cases.json") + load_golden("sb_cases_v2.json") + FV["sb"]
BACKENDS = [pytest.param("hostsim", id="hostsim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


class ScriptedBackend:
    def __init__(self, model: ScriptedModel, dev):
        self.model, self.device = model, torch.device(dev)

    def new_cache(self):
        c = types.SimpleNamespace(tokens=[], spec=None, best=0)
        c.get_seq_length = lambda: len(c.tokens)
        return c

    def forward(self, rows, cache):
        r = rows.cpu().tolist()
        lg = self.model.logits_rows(cache.tokens, r)
        cache.spec = [cache.tokens + row for row in r]
        cache.best = 0
        return torch.from_numpy(lg.reshape(-1, lg.shape[-1])).to(self.device)

    def commit(self, cache, src_row, dst, length):
        assert dst == len(cache.tokens)
        cache.best = src_row

    def set_length(self, cache, n):
        src = cache.spec[cache.best] if cache.spec is not None else cache.tokens
        cache.tokens = src[:n]
        cache.spec = None


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", MB[:12] + MB[18:24] + MB2[:8] + FV["mb"], ids=[c["name"] for c in MB[:12] + MB[18:24] + MB2[:8] + FV["mb"]])
def test_multiblock_seam_golden(case, backend):
    with use_backend(backend):
        dev = device_for(backend)
        p = case["params"]
        me = types.SimpleNamespace(jf_backend=ScriptedBackend(ScriptedModel.from_dict(case["model"]), dev))
        kw = dict(n_token_seq_len=p["n"], K=p["K"], r=p["r"], lookahead_start_ratio=p["lookahead"], n_gram_pool_size=p["pool"],
                  eos_token_id=p["eos_id"], pad_token_id=p["pad_id"], max_iteration_count=p["max_iter"], use_cache=True)
        ids = torch.tensor([case["prompt"] + case["prefill"]["draft"]], dtype=torch.int64, device=dev)
        cache, first, ngram, it = hf_seam.jacobi_forward_greedy_multiblock(me, ids, past_key_values=None, prefill_phase=True, **kw)
        assert ngram.cpu().tolist() == [case["prefill"]["ngram"]] and first.cpu().tolist() == case["prefill"]["first_correct_token"]
        assert it == 0 and cache.get_seq_length() == case["prefill"]["kv_len"]
        for call in case["calls"]:
            ids = torch.tensor([call["input"]], dtype=torch.int64, device=dev)
            cache, nxt, ret, iters = hf_seam.jacobi_forward_greedy_multiblock(me, ids, past_key_values=cache, prefill_phase=False, **kw)
            assert ret.cpu().tolist() == [call["ret"]]
            assert list(nxt.shape) == call["next_token_shape"] and nxt.view(-1).cpu().tolist() == call["next_token"]
            assert iters == call["iters"]
            assert cache.tokens == call["kv_tokens"]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", SB, ids=[c["name"] for c in SB])
def test_singleblock_seam_golden(case, backend, capsys):
    with use_backend(backend):
        dev = device_for(backend)
        n, eos = case["params"]["n"], case["params"]["eos_id"]
        me = types.SimpleNamespace(jf_backend=ScriptedBackend(ScriptedModel.from_dict(case["model"]), dev))
        ids = torch.tensor([case["prompt"] + case["prefill"]["draft"]], dtype=torch.int64, device=dev)
        cache, _, ngram, _ = hf_seam.jacobi_forward_greedy(me, ids, past_key_values=None, prefill_phase=True, n_token_seq_len=n,
                                                           eos_token_id=eos, use_cache=True)
        assert ngram.cpu().tolist() == [case["prefill"]["ngram"]]
        for call in case["calls"]:
            ids = torch.tensor([call["input"]], dtype=torch.int64, device=dev)
            cache, nxt, ret, itr = hf_seam.jacobi_forward_greedy(me, ids, past_key_values=cache, prefill_phase=False,
                                                                 n_token_seq_len=n, eos_token_id=eos)
            assert ret.cpu().tolist() == [call["ret"]]
            assert nxt.view(-1).cpu().tolist() == call["next_token"]
            assert itr == call["iters"]
            assert cache.tokens == call["kv_tokens"]


@pytest.mark.parametrize("backend", BACKENDS)
def test_seam_on_qwen2_backend_equals_autoregressive(backend):
    """Driver loop of JacobiForcing/jacobi_forcing_inference_MR_humaneval.py:152-240 over the real (tiny) Qwen2 backend:
    the concatenated accepted n-grams equal greedy AR decoding."""
    import random
    with use_backend(backend):
        dev = device_for(backend)
        model = tiny_model(dev, seed=17)
        V = model.cfg.vocab_size
        me = types.SimpleNamespace(jf_backend=hf_seam.Qwen2Backend(model, max_seq_len=256, max_rows=4, max_tokens=128))
        me.jacobi_forward_greedy_multiblock = types.MethodType(hf_seam.jacobi_forward_greedy_multiblock, me)
        n, rng = 16, random.Random(3)
        prompt = [int(t) for t in np.random.default_rng(2).integers(0, V - 2, size=11)]
        text = list(prompt)
        draft = [rng.choice(text) for _ in range(n)]
        kw = dict(n_token_seq_len=n, K=2, r=0.5, n_gram_pool_size=4, eos_token_id=None, pad_token_id=V - 2, use_cache=True)
        cache, _, ngram, _ = me.jacobi_forward_greedy_multiblock(torch.tensor([prompt + draft], device=dev), past_key_values=None,
                                                                 prefill_phase=True, **kw)
        inp, gen = ngram, []
        for _ in range(3):
            cache, first, acc, iters = me.jacobi_forward_greedy_multiblock(inp, past_key_values=cache, prefill_phase=False, **kw)
            gen += acc[0].cpu().tolist()
            text += acc[0].cpu().tolist()
            assert cache.get_seq_length() == len(prompt) + len(gen)
            inp = torch.cat([first.view(1, 1), torch.tensor([[rng.choice(text) for _ in range(n - 1)]], device=dev)], dim=-1)
        fwd = scratch_forward(model)
        toks, ar = list(prompt), []
        for _ in range(len(gen)):
            nxt = fwd([toks[:-1]], [[toks[-1]]])[0][0]
            ar.append(nxt); toks.append(nxt)
        assert gen == ar


@pytest.mark.parametrize("backend", BACKENDS)
def test_seam_row_padding_changes_nothing(backend):
    """Qwen2Backend(t_align=...) pads every forward's rows to the tuned GEMM grid (masked tokens, no logits for them): the
    multiblock and the single-block function return exactly what they return without padding."""
    import random
    with use_backend(backend):
        dev = device_for(backend)
        model = tiny_model(dev, seed=23)
        V = model.cfg.vocab_size
        out = {}
        for ta in (1, 24):
            me = types.SimpleNamespace(jf_backend=hf_seam.Qwen2Backend(model, max_seq_len=320, max_rows=4, max_tokens=64, t_align=ta))
            n, rng = 16, random.Random(9)
            prompt = [int(t) for t in np.random.default_rng(4).integers(0, V - 2, size=13)]
            text = list(prompt)
            kw = dict(n_token_seq_len=n, K=2, r=0.5, n_gram_pool_size=4, eos_token_id=V - 1, pad_token_id=V - 2, use_cache=True)
            draft = [rng.choice(text) for _ in range(n)]
            cache, _, ngram, _ = hf_seam.jacobi_forward_greedy_multiblock(me, torch.tensor([prompt + draft], device=dev),
                                                                          past_key_values=None, prefill_phase=True, **kw)
            inp, got = ngram, []
            for _ in range(3):
                cache, first, acc, iters = hf_seam.jacobi_forward_greedy_multiblock(me, inp, past_key_values=cache, prefill_phase=False, **kw)
                got.append((acc[0].cpu().tolist(), int(iters), cache.get_seq_length()))
                text += acc[0].cpu().tolist()
                inp = torch.cat([first.view(1, 1), torch.tensor([[rng.choice(text) for _ in range(n - 1)]], device=dev)], dim=-1)
            cache2, _, ng2, _ = hf_seam.jacobi_forward_greedy(me, input_ids=torch.tensor([prompt + draft], device=dev), past_key_values=None,
                                                              use_cache=True, prefill_phase=True, n_token_seq_len=n, eos_token_id=V - 1)
            c2, f2, a2, it2 = hf_seam.jacobi_forward_greedy(me, input_ids=ng2, past_key_values=cache2, use_cache=True, prefill_phase=False,
                                                            n_token_seq_len=n, eos_token_id=V - 1)
            got.append((a2[0].cpu().tolist(), int(it2), c2.get_seq_length()))
            out[ta] = got
        assert out[1] == out[24]


@pytest.mark.parametrize("backend", BACKENDS)
def test_seam_prefill_longer_than_the_candidate_scratch(backend):
    """max_tokens sizes the candidate scratch rows only: a one-row forward (the prefill of prompt + draft, every single-block
    forward) may be as long as the cache row.  A 70-token prompt with max_tokens = 32 must prefill and decode."""
    with use_backend(backend):
        dev = device_for(backend)
        model = tiny_model(dev, seed=19)
        V = model.cfg.vocab_size
        me = types.SimpleNamespace(jf_backend=hf_seam.Qwen2Backend(model, max_seq_len=256, max_rows=4, max_tokens=32))
        n = 8
        prompt = [int(t) for t in np.random.default_rng(5).integers(0, V - 2, size=70)]
        kw = dict(n_token_seq_len=n, K=2, r=0.5, n_gram_pool_size=4, eos_token_id=None, pad_token_id=V - 2, use_cache=True)
        cache, _, ngram, _ = hf_seam.jacobi_forward_greedy_multiblock(me, torch.tensor([prompt + prompt[:n]], device=dev),
                                                                       past_key_values=None, prefill_phase=True, **kw)
        assert cache.get_seq_length() == len(prompt)
        cache, first, acc, iters = hf_seam.jacobi_forward_greedy_multiblock(me, ngram, past_key_values=cache, prefill_phase=False, **kw)
        fwd = scratch_forward(model)
        toks, ar = list(prompt), []
        for _ in range(acc.shape[1]):
            nxt = fwd([toks[:-1]], [[toks[-1]]])[0][0]
            ar.append(nxt); toks.append(nxt)
        assert acc[0].cpu().tolist() == ar
        # candidates (B > 1) longer than the scratch: the scratch grows (runaway block lists, K >= 3 with a small spawn ratio)
        t0 = cache.T_max
        out = me.jf_backend.forward(torch.zeros((2, 40), dtype=torch.int64, device=dev), cache)
        assert cache.T_max >= 40 > t0 and out.shape[0] == 2 * 40
        with pytest.raises(RuntimeError):                       # more rows than the backend was built for are still refused
            me.jf_backend.forward(torch.zeros((5, 8), dtype=torch.int64, device=dev), cache)


# ------------------------------------------------------------------------------------- streaming driver (applications/)
class _StubTokenizer:
    """Token ids <-> "<id>" pieces; enough of the HF tokenizer surface for jacobi_stream_chat."""

    def __init__(self, prompt_ids, eos, pad):
        self.prompt_ids, self.eos_token_id, self.pad_token_id = prompt_ids, eos, pad

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True):
        return "PROMPT"

    def __call__(self, text, return_tensors="pt", add_special_tokens=True):
        if isinstance(text, list):
            return {"input_ids": torch.tensor([self.prompt_ids], dtype=torch.int64)}
        ids = [int(x) for x in text.replace(">", " ").replace("<", " ").split()]
        return types.SimpleNamespace(input_ids=torch.tensor([ids], dtype=torch.int64))

    def decode(self, ids, skip_special_tokens=True, clean_up_tokenization_spaces=False):
        return "".join(f"<{t}>" for t in ids)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("per_token", [True, False], ids=["per_token", "per_chunk"])
def test_stream_chat_driver(per_token, backend):
    """jacobi_stream_chat (applications/jacobi_streaming_driver.py:7-193 surface): the streamed text is the model's greedy
    continuation up to EOS, on_text sees a growing prefix of it, per-token and per-chunk streaming agree."""
    from jacobiforcing_amd.drivers.stream_chat import jacobi_stream_chat
    with use_backend(backend):
        dev = device_for(backend)
        V, eos, pad = 200, 199, 198
        m = ScriptedModel(V, 77, 80, 12, eos_id=eos, eos_pos=12 + 41, reserved=(pad,))
        me = types.SimpleNamespace(jf_backend=ScriptedBackend(m, dev))
        tok = _StubTokenizer(m.prompt(), eos, pad)
        seen = []
        torch.manual_seed(3)
        text, final_ids, n_new, gen_time = jacobi_stream_chat(me, tok, [{"role": "user", "content": "hi"}], n_token_seq_len=16,
                                                              max_new_tokens=200, K=2, r=0.8, n_gram_pool_size=4,
                                                              on_text=seen.append, stream_per_token=per_token)
        want = m.ar_continuation(12, 41)                                  # tokens before the EOS at position 53
        assert final_ids[0].tolist() == want and n_new == len(want) and gen_time > 0
        assert text == "".join(f"<{t}>" for t in want)
        assert seen and seen[-1] == text and all(b.startswith(a) for a, b in zip(seen, seen[1:]))
        assert len(seen) == len(want) if per_token else len(seen) < len(want)


# ------------------------------------------------------------------------------------- DRV-SB / AR baseline counterparts
@pytest.mark.parametrize("backend", BACKENDS)
def test_single_block_driver_matches_ar_baseline(backend):
    """drivers/sb_math500.decode_one (the reference MATH500 driver's loop over jacobi_forward_greedy) produces exactly the
    greedy AR continuation that drivers/ar_baseline.generate_greedy decodes token by token from the same weights — the
    reference's own correctness criterion — and its row follows the driver's conventions."""
    from jacobiforcing_amd.drivers import ar_baseline, sb_math500
    import random
    with use_backend(backend):
        dev = device_for(backend)
        model = tiny_model(dev, seed=9)
        prompt = [5, 17, 33, 2, 9, 41, 7]
        me = types.SimpleNamespace(jf_backend=hf_seam.Qwen2Backend(model, max_seq_len=256, max_rows=1, max_tokens=64))
        row, toks = sb_math500.decode_one(me, prompt, n=8, eos_id=None, alt_eos_id=None, max_new_tokens=40, max_calls=64,
                                          rng=random.Random(0))
        ar, _ = ar_baseline.generate_greedy(model, prompt, max_new_tokens=len(toks))
        assert toks == ar
        assert row["stop_reason"] == "max_new_tokens" and row["new_tokens"] == len(toks) - 1 and row["calls"] >= 2
        assert row["total_iterations"] >= row["calls"] - 1 and row["prompt_tokens"] == len(prompt)
        assert abs(row["avg_iter_per_token"] - row["total_iterations"] / row["new_tokens"]) < 1e-12
