"""Randomised sweep: the device state-machine source (hostsim on CPU, real kernels with -m gpu) vs the CPU oracle over
many (n, K, r, pool, lookahead, max_iter, vocab, robustness, periodicity, EOS) combinations, several calls each."""
import os

import numpy as np
import pytest
import torch

from jacobiforcing_amd import ops
from oracle import jacobi_oracle as O
from oracle.scripted_model import ScriptedModel

from .backends import device_for, use_backend
from .test_multiblock import run_calls

BACKENDS = [pytest.param("hostsim", id="hostsim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


# seeds 0..119 run on both backends; 120..359 only through the real kernels (the CPU suite stays short).  JF_FUZZ_SCALE=k
# multiplies the kernel-only range for a soak run on a GPU box (profiles/soak_r02.txt).
FUZZ_SCALE = max(int(os.environ.get("JF_FUZZ_SCALE", "1")), 1)
CASES = [pytest.param(seed, b, id=f"{b}-{seed}", marks=[pytest.mark.gpu] if b == "hip" else [])
         for seed in range(360 * FUZZ_SCALE) for b in (("hostsim", "hip") if seed < 120 else ("hip",))]


@pytest.mark.parametrize("seed,backend", CASES)
def test_fuzz_vs_oracle(seed, backend):
    rng = np.random.default_rng(10_000 + seed)
    n = int(rng.choice([4, 8, 12, 16, 24, 32, 48, 64]))
    K = int(rng.choice([1, 2, 2, 2, 3, 4]))
    r = float(rng.choice([0.2, 0.5, 0.75, 0.85, 1.0]))
    pool = int(rng.choice([0, 1, 2, 4, 4, 8]))
    look = float(rng.choice([0.0, 0.0, 0.3, 0.9]))
    max_iter = int(rng.choice([128, 128, 128, 5, 2]))
    V = int(rng.choice([12, 24, 64, 300]))
    robust = int(rng.choice([0, 20, 50, 70, 90, 100]))
    period = int(rng.choice([0, 0, 3, 5, 9]))
    P = int(rng.integers(1, 5))
    eos_id, pad_id = V - 1, V - 2
    with use_backend(backend) as lib:
        old_fast = lib.jf_mb_set_fast_path(0 if seed % 5 == 4 else 1)     # every fifth seed: the general code only
        try:
            _fuzz_body(seed, backend, rng, n, K, r, pool, look, max_iter, V, robust, period, P, eos_id, pad_id)
        finally:
            lib.jf_mb_set_fast_path(old_fast)


def _fuzz_body(seed, backend, rng, n, K, r, pool, look, max_iter, V, robust, period, P, eos_id, pad_id):
    if True:
        dev = device_for(backend)
        models, kvs = [], []
        for p in range(P):
            pl = int(rng.integers(2, 30))
            eos_pos = None if rng.random() < 0.5 else pl + int(rng.integers(0, 3 * n))
            m = ScriptedModel(V, 500 + 31 * seed + p, robust, pl, eos_id=eos_id, eos_pos=eos_pos, reserved=(pad_id,), period=period)
            models.append(m)
            kvs.append(m.prompt())
        prm = ops.MultiblockParams(n=n, K=K, r=r, lookahead_start_ratio=look, n_gram_pool_size=pool, eos_token_id=eos_id,
                                   pad_token_id=pad_id, max_iteration_count=max_iter)
        batch = ops.MultiblockBatch(P, prm, dev)
        batch.fused = seed % 4 != 3        # every fourth seed: jf_argmax_* + jf_mb_step as two launches instead of jf_mb_verify
        fwd = [(lambda m: (lambda kv_rows, rows: [m.greedy_rows(kv_rows[b], [rows[b]])[0] for b in range(len(rows))]))(m)
               for m in models]
        inputs = [O.mb_prefill(fwd[p], kvs[p], [int(x) for x in rng.choice(kvs[p], size=n)])[0] for p in range(P)]
        okvs = [list(k) for k in kvs]
        kw = dict(n=n, K=K, r=r, lookahead_start_ratio=look, n_gram_pool_size=pool, eos_token_id=eos_id, pad_token_id=pad_id,
                  max_iteration_count=max_iter)
        for call in range(3):
            want, err = [], None
            for p in range(P):
                try:
                    want.append(O.mb_generation_call(fwd[p], inputs[p], okvs[p], **kw))
                except (RuntimeError, AssertionError) as e:      # the reference itself dies here (K>=3 broadcast, MB:667 assert)
                    err = e
                    break
            if err is not None:
                with pytest.raises((RuntimeError, ValueError)):
                    run_calls(batch, models, kvs, inputs, dev)
                return
            res = run_calls(batch, models, kvs, inputs, dev)
            nxt = []
            for p in range(P):
                st = want[p]
                ctx = f"seed {seed} call {call} prompt {p} (n={n} K={K} r={r} pool={pool})"
                assert res[p]["ret"] == st.ret, ctx
                assert res[p]["next_token"] == (st.next_token if st.next_token is not None else -1), ctx
                assert res[p]["iters"] == st.iters, ctx
                assert res[p]["kv_tokens"] == st.kv_tokens, ctx
                assert res[p]["banners"] == st.banners, ctx
                okvs[p] = st.kv_tokens
                kvs[p] = res[p]["kv_tokens"]
                nt = st.next_token if st.next_token is not None else 0
                nxt.append([nt] + [int(x) for x in rng.choice(kvs[p], size=n - 1)])
            inputs = nxt
