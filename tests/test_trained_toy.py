"""Jacobi decoding on a TRAINED checkpoint: tests/golden/toy_periodic/ (tests/golden/train_toy_checkpoint.py: a tiny Qwen2 that has
learnt token[i] = PERM[token[i - 6]]).  Random-init weights accept ~1 token per forward and the bench's 3.9 are planted logits, so
this is where multi-token acceptance runs through the REAL forward and KV cache: a wrong K/V row after a multi-token commit, a wrong
position, a wrong cached length would change the next logits and the tokens would leave the autoregressive ones.  The reference's own
criterion (inference_engine/tests/test_jacobi_decoding_greedy.py:180-206): greedy Jacobi == greedy AR, token for token — here at 3-6
tokens per forward, for every decoder behind ``LLM.generate`` (engine single block through the chunk loop and through the callback
contract, multiblock + rejection recycling at n = 16 / 32), over the contiguous and the paged KV layout."""
import random
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from jacobiforcing_amd import LLM, SamplingParams

from .backends import device_for, use_backend

TOY = Path(__file__).resolve().parent / "golden" / "toy_periodic"
sys.path.insert(0, str(TOY.parent))
from train_toy_checkpoint import PERIOD, corpus  # noqa: E402

BACKENDS = [pytest.param("hostsim", id="hostsim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
N_NEW = 96


def _prompts():
    return [row[:n].tolist() for row, n in zip(corpus(np.random.default_rng(5), 6, 300), (13, 7, 25, 18, 120, 161))]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("layout", ["contiguous", "paged"])
def test_jacobi_equals_autoregressive_at_several_tokens_per_forward(backend, layout, monkeypatch):
    monkeypatch.setenv("JF_DTYPE", "float32")
    with use_backend(backend):
        torch.manual_seed(0)
        random.seed(0)
        llm = LLM(str(TOY), tokenizer_path="none", device=device_for(backend), max_model_len=512, max_num_batched_tokens=4096, max_num_seqs=8,
                  kv_cache_layout=layout)
        prompts = _prompts()
        ar = [o["token_ids"] for o in llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=N_NEW, ignore_eos=True), use_tqdm=False)]
        # the language itself: the continuation is the permuted repetition (the checkpoint has learnt it; AR is the reference here)
        want = [row[n:n + N_NEW].tolist() for row, n in zip(corpus(np.random.default_rng(5), 6, 300 + N_NEW), (13, 7, 25, 18, 120, 161))]
        agree = [float(np.mean(np.asarray(a) == np.asarray(w))) for a, w in zip(ar, want)]
        assert min(agree[:4]) > 0.9, agree                       # (the two prompts beyond the training length are not asked to extrapolate)
        for loop_on in ("1", "0"):                               # the chunk loop on device arrays / the reference's callback contract
            monkeypatch.setenv("JF_ENGINE_LOOP", loop_on)
            llm.model_runner.jacobi_decoder = None
            out = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=N_NEW, ignore_eos=True, decode_strategy="jacobi", jacobi_block_len=16),
                               use_tqdm=False)
            st = llm.model_runner.jacobi_decoder.stats
            assert [o["token_ids"][:N_NEW] for o in out] == ar, loop_on
            per_forward = st["tokens_accepted"] / st["num_jacobi_iterations"] / len(prompts)
            assert per_forward > 3.5, per_forward               # (up to PERIOD = 6: a block's first PERIOD positions follow from committed tokens)
        for n in (16, 32):
            out = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=N_NEW, ignore_eos=True,
                                                       decode_strategy="jacobi_multiblock_rejection_recycling", jacobi_block_len=n), use_tqdm=False)
            assert [o["token_ids"][:N_NEW] for o in out] == ar, n
            lm = llm.model_runner.last_multiblock
            tokens, forwards = sum(len(s.token_ids) for s in lm["stats"]), sum(s.total_iterations for s in lm["stats"])
            assert tokens / forwards > 2.5, (n, tokens / forwards)


@pytest.mark.parametrize("backend", BACKENDS)
def test_sampling_at_a_low_temperature_follows_the_trained_language(backend, monkeypatch):
    """The non-greedy decoder on the same checkpoint: at T = 0.3 the learnt continuation holds nearly all the mass, so nearly every
    drafted token is accepted and the samples are the language's continuation — several tokens per forward through jf_rs_probs /
    jf_rs_step on real logits, with and without top_k / top_p planted."""
    monkeypatch.setenv("JF_DTYPE", "float32")
    with use_backend(backend):
        torch.manual_seed(1)
        random.seed(1)
        llm = LLM(str(TOY), tokenizer_path="none", device=device_for(backend), max_model_len=512, max_num_batched_tokens=4096, max_num_seqs=8)
        prompts = _prompts()[:4]
        want = [row[n:n + 48].tolist() for row, n in zip(corpus(np.random.default_rng(5), 4, 400), (13, 7, 25, 18))]
        for filters in (None, (20, 0.95)):
            sp = SamplingParams(temperature=0.3, max_tokens=48, ignore_eos=True, decode_strategy="jacobi", jacobi_block_len=16)
            if filters:
                sp.top_k, sp.top_p = filters
            llm.model_runner.jacobi_decoder = None
            out = [o["token_ids"][:48] for o in llm.generate(prompts, sp, use_tqdm=False)]
            st = llm.model_runner.jacobi_decoder.stats
            agree = np.mean([np.mean(np.asarray(o) == np.asarray(w)) for o, w in zip(out, want)])
            assert agree > 0.9, (filters, agree)
            assert st["tokens_accepted"] / st["num_jacobi_iterations"] / len(prompts) > 2.0


@pytest.mark.parametrize("backend", BACKENDS)
def test_hf_seam_drivers_on_the_trained_checkpoint(backend):
    """The HF-style entry points on the same checkpoint: the single-block driver (drivers/sb_math500.decode_one over
    hf_seam.jacobi_forward_greedy = SB:140-276) and the batch-1 multiblock decoder behind jacobi_forward_greedy_multiblock decode
    exactly what drivers/ar_baseline.generate_greedy decodes token by token, in fewer than half as many forwards."""
    import types
    from jacobiforcing_amd import hf_seam, ops
    from jacobiforcing_amd.drivers import ar_baseline, sb_math500
    from jacobiforcing_amd.engine.multiblock_decoder import MultiblockJacobiDecoder
    from jacobiforcing_amd.modeling.qwen2 import Qwen2Model, load_model_directory
    with use_backend(backend):
        dev = torch.device(device_for(backend))
        cfg, w = load_model_directory(str(TOY), dev, dtype=torch.float32)
        model = Qwen2Model(cfg, w)
        prompt = _prompts()[2]
        me = types.SimpleNamespace(jf_backend=hf_seam.Qwen2Backend(model, max_seq_len=512, max_rows=1, max_tokens=64))
        row, toks = sb_math500.decode_one(me, prompt, n=16, eos_id=None, alt_eos_id=None, max_new_tokens=80, max_calls=64, rng=random.Random(0))
        ar, _ = ar_baseline.generate_greedy(model, prompt, max_new_tokens=len(toks))
        assert toks == ar and row["avg_iter_per_token"] < 0.5, row            # more than two tokens per forward
        prm = ops.MultiblockParams(n=16, K=2, r=0.85, n_gram_pool_size=4, eos_token_id=None, pad_token_id=cfg.pad_token_id)
        dec = MultiblockJacobiDecoder(model, 1, prm, max_seq_len=512)
        stats, _, iters = dec.generate([prompt], max_new_tokens=80, max_calls=1 << 20, seed=3)
        assert stats[0].token_ids[:80] == ar[:80] and len(stats[0].token_ids) / stats[0].total_iterations > 2.0
