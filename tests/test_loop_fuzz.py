"""Randomised sweep of the LOOP API end to end: MultiblockJacobiDecoder (jf_mb_loop_*: calls restarted on the device or on the
host, the pack step's list order, the convergence launch's slot hand-off) against the CPU oracle's driver, at acceptance
rates a trained checkpoint has (several tokens per forward: spawns, promotions, call ends, candidate rows in most
iterations).  The logits come from the planted-acceptance hook (jacobiforcing_amd/synthetic.py: pure integer arithmetic on
ids / positions), restated below in Python for the oracle's forward — so both sides see the same greedy tokens whatever the
forward's numerics are, and every difference is a difference of the loop."""
import os

import numpy as np
import pytest
import torch

from jacobiforcing_amd import ops
from jacobiforcing_amd.engine.multiblock_decoder import MultiblockJacobiDecoder
from jacobiforcing_amd.synthetic import ScriptedAcceptance
from oracle import jacobi_oracle as O

from .backends import device_for, use_backend
from .test_decoder_e2e import oracle_generate, tiny_model

_M = (1 << 31) - 1
FUZZ_SCALE = max(int(os.environ.get("JF_FUZZ_SCALE", "1")), 1)
# seeds 0..7 on both backends, 8..47 only through the real kernels (JF_FUZZ_SCALE multiplies that range for a soak)
CASES = [pytest.param(seed, b, id=f"{b}-{seed}", marks=[pytest.mark.gpu] if b == "hip" else [])
         for seed in range(8 + 40 * FUZZ_SCALE) for b in (("hostsim", "hip") if seed < 8 else ("hip",))]


def _h(a: int, b: int, salt: int) -> int:
    x = (a * 1103515245 + b * 12345 + salt) & _M
    x ^= x >> 15
    x = (x * 48271) & _M
    x ^= x >> 13
    x = (x * 69621) & _M
    return x ^ (x >> 16)


class PlantedForward:
    """The hook's rule as the oracle's forward for prompt p (ScriptedAcceptance.__call__: decode rows and prefill rows)."""

    def __init__(self, hook: ScriptedAcceptance, p: int, plen: int):
        self.h, self.p, self.plen = hook, int(p), int(plen)

    def target(self, pos: int) -> int:
        return _h(pos, self.p + self.h.seed, 0x9E37) % self.h.vocab_hi

    def decode(self, kv_rows, out_rows):
        res = []
        for kv, row in zip(kv_rows, out_rows):
            ok, g = True, []
            for t, tok in enumerate(row):
                pos = len(kv) + t
                ok = ok and tok == self.target(pos)
                rob = _h(tok, pos, 0x51ED) % 100 < self.h.robust
                g.append(self.target(pos + 1) if (ok or rob) else _h(tok, pos + self.p, 0x7777) % self.h.vocab_hi)
            res.append(g)
        return res

    def prefill(self, kv_rows, out_rows):
        row = out_rows[0]                                           # prompt ⧺ draft; only the last n+1 positions are looked at
        g = []
        for pos in range(len(row)):
            rob = _h(pos, self.p, 0x51ED) % 100 < self.h.robust
            g.append(self.target(pos + 1) if (pos < self.plen or rob) else _h(pos, self.p, 0x7777) % self.h.vocab_hi)
        return [g]


@pytest.mark.parametrize("seed,backend", CASES)
def test_loop_fuzz_vs_oracle(seed, backend):
    rng = np.random.default_rng(50_000 + seed)
    n = int(rng.choice([8, 16, 16, 32]))
    K = int(rng.choice([1, 2, 2, 2, 3]))
    r = float(rng.choice([0.5, 0.7, 0.85, 0.85]))
    pool = int(rng.choice([0, 2, 4, 4]))
    look = float(rng.choice([0.0, 0.0, 0.5]))
    P = int(rng.integers(1, 7))
    robust = int(rng.choice([55, 70, 82, 95]))
    resident = bool(rng.integers(0, 4) != 0)
    t_align = int(rng.choice([1, 4, 8]))
    max_new = int(rng.choice([n, 2 * n + 3, 4 * n]))
    max_calls = int(rng.choice([3, 6, 12]))
    compact = bool(rng.integers(0, 5) != 0)                       # one in five over the padded [R, Tpad] rectangle of logits
    logit_align = int(rng.choice([1, 8, 64]))
    max_iter = int(rng.choice([128, 128, 128, 6]))
    use_eos = bool(rng.integers(0, 3) == 0)                       # one in three with an EOS id the planted sequence can hit
    if K >= 3 and rng.integers(0, 2) == 0:
        # K >= 3 with a small spawn ratio: the reference's block lists run away (tests/golden/mb_cases_v3.json pins
        # rows of up to ~77 n tokens) — the decoder follows (its candidate scratch grows on demand), nothing is cut off
        r = float(rng.choice([0.05, 0.25]))
        # (hundreds of forwards with rows of hundreds of tokens, restated token by token in Python on the oracle's side: two
        #  prompts, three calls and a small block keep such a seed at a few seconds)
        P, max_calls, n = min(P, 2), min(max_calls, 3), min(n, 16)
        max_new = min(max_new, 4 * n)
    with use_backend(backend):
        dev = device_for(backend)
        # a vocabulary whose rows are not 16-byte multiples takes the convergence check as its two launches (and the pack
        # launch behind them copies the descriptor tables into the mailbox itself): one seed in four
        model = tiny_model(dev, seed=seed, vocab=381 if seed % 4 == 1 else 384)
        V = model.cfg.vocab_size
        eos = int(rng.integers(0, 40)) if use_eos else None        # one id of the 382 the planted sequence draws from: a prompt in five meets it
        prm = ops.MultiblockParams(n=n, K=K, r=r, lookahead_start_ratio=look, n_gram_pool_size=pool, eos_token_id=eos,
                                   pad_token_id=V - 2, max_iteration_count=max_iter)
        prompts = [[int(t) for t in rng.integers(0, V - 2, size=int(L))] for L in rng.integers(3, 40, size=P)]
        hook = ScriptedAcceptance(V, robust_pct=robust, seed=int(rng.integers(1, 1000)), vocab_hi=V - 2)
        dec = MultiblockJacobiDecoder(model, P, prm, max_seq_len=512 if K < 3 else 6144, resident=resident, t_align=t_align,
                                      logits_hook=hook, compact_logits=compact, logit_align=logit_align)
        draw_seed = int(rng.integers(1, 1 << 20))
        failed = None
        stream = seed % 3 == 0              # every third seed through the streaming generator (chunks per finished call)
        chunks = {p: [] for p in range(P)}
        try:
            if stream:
                it = dec.generate_stream(prompts, max_new_tokens=max_new, max_calls=max_calls, seed=draw_seed)
                while True:
                    try:
                        p_, toks = next(it)
                        chunks[p_] += toks
                    except StopIteration as stop:
                        stats, _, iters = stop.value
                        break
            else:
                stats, _, iters = dec.generate(prompts, max_new_tokens=max_new, max_calls=max_calls, seed=draw_seed)
        except RuntimeError as e:
            # K >= 3 only: the reference's own crash at MB:482 (the oracle must fail on the same decode, below)
            assert K >= 3 and "size of tensor" in str(e), e
            failed = e
        refs, ref_failed = [], 0
        for p, prompt in enumerate(prompts):
            pf = PlantedForward(hook, p, len(prompt))
            draws = ops.DrawStreams(P, seed=draw_seed).rng(p)

            class _Fwd:                                             # first call of the oracle's driver is the prefill
                calls = 0

                def __call__(self, kv_rows, out_rows):
                    self.calls += 1
                    return pf.prefill(kv_rows, out_rows) if self.calls == 1 else pf.decode(kv_rows, out_rows)
            try:
                refs.append(oracle_generate(_Fwd(), prompt, prm, max_new, max_calls, draws))
            except RuntimeError as oe:
                assert "size of tensor" in str(oe)
                ref_failed += 1
                refs.append(None)
        if failed is not None:
            assert ref_failed >= 1
            return
        assert ref_failed == 0
        for p, ref in enumerate(refs):
            assert stats[p].token_ids == ref["tokens"], f"prompt {p}"
            assert (stats[p].calls, stats[p].total_iterations, stats[p].stop_reason) == (ref["calls"], ref["iters"], ref["stop"]), p
            assert int(dec.kv_len_host[p]) == ref["kv_len"]
            assert not stream or chunks[p] == ref["tokens"], f"streamed chunks of prompt {p}"
        tpf = sum(len(s.token_ids) for s in stats) / max(sum(s.total_iterations for s in stats), 1)
        assert tpf > 1.2 or robust < 70 or max_iter < 10 or eos is not None, tpf   # the sweep runs where several tokens are accepted per forward


@pytest.mark.gpu
@pytest.mark.parametrize("P", [300, 700])
@pytest.mark.parametrize("resident", [True, False], ids=["resident", "hostdriven"])
def test_hundreds_of_prompts_through_the_loop(resident, P):
    """300 prompts: one fused convergence launch with 300 steppers, the position list ordered over 300 descriptors.  700 prompts:
    more steppers than the fused launch may carry (half of the resident workgroups, 640 on an MI355X), so every iteration runs
    as argmax + step launches and the pack launch behind them copies all 700 descriptors into the mailbox and stamps it.
    Every prompt decodes the planted sequence; every seventh is also compared with the oracle's driver (tokens, calls,
    iterations, committed length)."""
    with use_backend("hip"):
        dev = device_for("hip")
        model = tiny_model(dev, seed=5)
        V = model.cfg.vocab_size
        n = 8
        rng = np.random.default_rng(77)
        prm = ops.MultiblockParams(n=n, K=2, r=0.85, n_gram_pool_size=4, eos_token_id=None, pad_token_id=V - 2)
        prompts = [[int(t) for t in rng.integers(0, V - 2, size=int(L))] for L in rng.integers(3, 20, size=P)]
        hook = ScriptedAcceptance(V, robust_pct=82, seed=11, vocab_hi=V - 2)
        dec = MultiblockJacobiDecoder(model, P, prm, max_seq_len=512, resident=resident, t_align=4, logits_hook=hook)
        stats, _, iters = dec.generate(prompts, max_new_tokens=2 * n, max_calls=6, seed=31)
        for p, st in enumerate(stats):
            pos = torch.arange(len(prompts[p]), len(prompts[p]) + len(st.token_ids))
            assert st.token_ids == hook.target(pos, torch.full_like(pos, p)).tolist(), p
        for p in range(0, P, 7):
            pf = PlantedForward(hook, p, len(prompts[p]))

            class _Fwd:
                calls = 0

                def __call__(self, kv_rows, out_rows):
                    self.calls += 1
                    return pf.prefill(kv_rows, out_rows) if self.calls == 1 else pf.decode(kv_rows, out_rows)
            ref = oracle_generate(_Fwd(), prompts[p], prm, 2 * n, 6, ops.DrawStreams(P, seed=31).rng(p))
            assert stats[p].token_ids == ref["tokens"], p
            assert (stats[p].calls, stats[p].total_iterations, stats[p].stop_reason) == (ref["calls"], ref["iters"], ref["stop"]), p
            assert int(dec.kv_len_host[p]) == ref["kv_len"], p


BACKENDS = [pytest.param("hostsim", id="hostsim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cfg", [dict(seed=5, n=8, K=3, r=0.05, pool=2, robust=30),       # one row per prompt, 551 tokens long
                                 dict(seed=16, n=8, K=3, r=0.05, pool=4, robust=45),      # candidate rows of 577 tokens: scratch grows x3
                                 dict(seed=11, n=8, K=3, r=0.25, pool=4, robust=30),
                                 dict(seed=0, n=8, K=4, r=0.05, pool=2, robust=30)],
                         ids=lambda c: f"n{c['n']}K{c['K']}r{c['r']}p{c['pool']}")
def test_runaway_block_lists_are_followed_not_cut(cfg, backend):
    """K >= 3 with a small spawn ratio: the reference's block lists run away (promotion decrements num_blocks without removing
    list entries, MB:711-716; tests/golden/mb_cases_v3.json pins rows of up to ~77 n tokens on the reference itself).  The
    decoder follows — rows far beyond (K + 2) n, the candidate scratch growing on demand — and decodes what the oracle's
    driver decodes: tokens, calls, iterations, committed lengths."""
    rng = np.random.default_rng(cfg["seed"])
    n, K, P, max_new, max_calls = cfg["n"], cfg["K"], 2, 64, 4
    with use_backend(backend):
        dev = device_for(backend)
        model = tiny_model(dev, seed=cfg["seed"], vocab=384)
        V = model.cfg.vocab_size
        prm = ops.MultiblockParams(n=n, K=K, r=cfg["r"], n_gram_pool_size=cfg["pool"], eos_token_id=None, pad_token_id=V - 2)
        prompts = [[int(t) for t in rng.integers(0, V - 2, size=int(L))] for L in rng.integers(3, 20, size=P)]
        hook = ScriptedAcceptance(V, robust_pct=cfg["robust"], seed=int(rng.integers(1, 1000)), vocab_hi=V - 2)
        dec = MultiblockJacobiDecoder(model, P, prm, max_seq_len=4096, logits_hook=hook)
        t_first = dec.cache.T_max
        longest = [0]
        fwd = dec._forward

        def counting(s):
            longest[0] = max(longest[0], s.Tpad)
            return fwd(s)
        dec._forward = counting
        stats, _, iters = dec.generate(prompts, max_new_tokens=max_new, max_calls=max_calls, seed=7)
        assert longest[0] > (K + 2) * n, longest                   # the lists did run away
        if cfg["pool"] > 2 and cfg["seed"] == 16:
            assert dec.cache.T_max > t_first                       # ... with candidate rows: the scratch grew
        for p, prompt in enumerate(prompts):
            pf = PlantedForward(hook, p, len(prompt))

            class _Fwd:
                calls = 0

                def __call__(self, kv_rows, out_rows):
                    self.calls += 1
                    return pf.prefill(kv_rows, out_rows) if self.calls == 1 else pf.decode(kv_rows, out_rows)
            ref = oracle_generate(_Fwd(), prompt, prm, max_new, max_calls, ops.DrawStreams(P, seed=7).rng(p))
            assert stats[p].token_ids == ref["tokens"], p
            assert (stats[p].calls, stats[p].total_iterations, stats[p].stop_reason) == (ref["calls"], ref["iters"], ref["stop"]), p
            assert int(dec.kv_len_host[p]) == ref["kv_len"], p
