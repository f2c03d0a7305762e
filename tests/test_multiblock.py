"""Multiblock Jacobi state machine (jf_mb_* through jacobiforcing_amd.ops.MultiblockBatch) against
the golden vectors recorded from the reference and against the CPU oracle.

The same test body runs on two backends:
  * hostsim (CPU): the device source compiled single-lane — checks the logic without a GPU;
  * hip (marked gpu): the real kernels through the C ABI on an MI355X.
"""
import numpy as np
import pytest
import torch

from jacobiforcing_amd import _native as N
from jacobiforcing_amd import ops
from oracle import jacobi_oracle as O
from oracle.scripted_model import ScriptedModel

from .backends import device_for, use_backend
from .conftest import forward_matches, kv_matches, load_golden

MB = load_golden("mb_cases.json") + load_golden("mb_cases_v2.json") + load_golden("mb_cases_v3.json") + \
    load_golden("fullvocab_cases.json")["mb"]            # round 5: the reference at V = 152 064
MBR = load_golden("mb_raises.json")

BACKENDS = [pytest.param("hostsim", id="hostsim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _params(p, max_blocks=None):
    return ops.MultiblockParams(n=p["n"], K=p["K"], r=p["r"], lookahead_start_ratio=p["lookahead"],
                                n_gram_pool_size=p["pool"], eos_token_id=p["eos_id"], pad_token_id=p["pad_id"],
                                max_iteration_count=p["max_iter"], max_blocks=max_blocks)


def run_calls(batch, models, kvs, inputs, dev, dtype=torch.float32, traces=None):
    """Drive one generation call for P prompts side by side with the scripted models as the 'forward'."""
    P = batch.P
    d = batch.begin(torch.tensor(inputs, dtype=torch.int64), torch.tensor([len(k) for k in kvs], dtype=torch.int32))
    kvs = [list(k) for k in kvs]
    events = [[] for _ in range(P)]
    fast, steps = [0] * P, [0] * P
    while True:
        packed_in = batch.pack(d)
        if packed_in is None:
            break
        ids, pos, row_prompt, row_len = packed_in
        ids_h, pos_h = ids.cpu().numpy(), pos.cpu().numpy()
        rp, rl = row_prompt.cpu().numpy(), row_len.cpu().numpy()
        B, T = batch.desc_field(d, "B").copy(), batch.desc_field(d, "T").copy()
        R, Tpad = ids_h.shape
        V = models[0].vocab
        logits = np.zeros((R, Tpad, V), dtype=np.float32)
        r0 = 0
        rows_of = []
        for p in range(P):
            rows = [ids_h[r0 + b, :T[p]].tolist() for b in range(B[p])]
            rows_of.append(rows)
            if B[p]:
                assert (rp[r0:r0 + B[p]] == p).all() and (rl[r0:r0 + B[p]] == T[p]).all()
                assert (pos_h[r0:r0 + B[p], :T[p]] == len(kvs[p]) + np.arange(T[p])).all()
                lg = models[p].logits_rows(kvs[p], rows)
                logits[r0:r0 + B[p], :T[p]] = lg
                if traces is not None:
                    traces[p].append(dict(kv_len=len(kvs[p]), out=rows, greedy=models[p].greedy_rows(kvs[p], rows)))
            r0 += B[p]
        d = batch.verify(torch.from_numpy(logits).to(dtype).to(dev))
        for p in range(P):
            if B[p] == 0:
                continue
            new_kv = int(batch.desc_field(d, "kv_len")[p])
            src = int(batch.desc_field(d, "kv_src_row")[p]) if batch.desc_field(d, "kv_copy_len")[p] > 0 else 0
            if batch.desc_field(d, "kv_copy_len")[p] > 0:
                assert batch.desc_field(d, "kv_copy_dst")[p] == len(kvs[p])
                assert batch.desc_field(d, "kv_copy_len")[p] == new_kv - len(kvs[p])
            keep = new_kv - len(kvs[p])
            assert 0 <= keep <= T[p]
            kvs[p] = kvs[p] + rows_of[p][src][:keep]
            ev = int(batch.desc_field(d, "events")[p])
            events[p] += [nm for bit, nm in ((1, "spawn"), (2, "switch"), (4, "early_stop")) if ev & bit]
            fast[p] += 1 if ev & N.EVT_FAST else 0
            steps[p] += 1
        if batch.desc_field(d, "done").all():
            break
    res = batch.results(d)
    for p in range(P):
        res[p]["kv_tokens"] = kvs[p]
        res[p]["banners"] = events[p]
        res[p]["fast_steps"], res[p]["steps"] = fast[p], steps[p]
    return res


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", MB, ids=[c["name"] for c in MB])
def test_golden_calls(case, backend):
    with use_backend(backend):
        dev = device_for(backend)
        p = case["params"]
        model = ScriptedModel.from_dict(case["model"])
        batch = ops.MultiblockBatch(1, _params(p), dev)
        kv = case["prefill"]["kv_tokens"]
        for ci, call in enumerate(case["calls"]):
            traces = [[]]
            r = run_calls(batch, [model], [kv], [call["input"]], dev, traces=traces)[0]
            ctx = f"{case['name']} call {ci}"
            assert r["ret"] == call["ret"], ctx
            assert [r["next_token"]] == call["next_token"], ctx
            assert r["iters"] == call["iters"], ctx
            assert r["kv_len"] == call["kv_len"], ctx
            assert kv_matches(r["kv_tokens"], call), ctx
            assert r["banners"] == call["banners"], ctx
            assert len(traces[0]) == len(call["forwards"]), ctx
            for it, (a, b) in enumerate(zip(traces[0], call["forwards"])):        # mb_cases_v3.json: digests of the rows
                assert forward_matches(a, b, greedy=False), f"{ctx} iter {it}"
            kv = r["kv_tokens"]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("cfg", [dict(n=16, K=2, r=0.5, pool=4, vocab=64, period=5, robust=50),
                                 dict(n=32, K=2, r=0.85, pool=4, vocab=300, period=0, robust=70),
                                 dict(n=32, K=3, r=0.4, pool=8, vocab=48, period=7, robust=40),
                                 dict(n=64, K=2, r=0.85, pool=4, vocab=1000, period=0, robust=75)],
                         ids=lambda c: f"n{c['n']}K{c['K']}p{c['pool']}")
def test_batch_vs_oracle(cfg, dtype, backend):
    """P prompts side by side (BASELINE config 4 shape: independent state machines, one forward) vs the
    oracle run prompt by prompt, several consecutive calls, EOS for some prompts."""
    with use_backend(backend):
        dev = device_for(backend)
        P, n = 6, cfg["n"]
        V = cfg["vocab"]
        eos_id, pad_id = V - 1, V - 2
        rng = np.random.default_rng(1234 + n)
        models, kvs = [], []
        for p in range(P):
            pl = int(rng.integers(4, 40))
            eos_pos = None if p % 3 else pl + int(rng.integers(n // 2, 3 * n))
            m = ScriptedModel(V, 1000 + 17 * p + n, cfg["robust"], pl, eos_id=eos_id, eos_pos=eos_pos,
                              reserved=(pad_id,), period=cfg["period"])
            models.append(m)
            kvs.append(m.prompt())
        prm = ops.MultiblockParams(n=n, K=cfg["K"], r=cfg["r"], n_gram_pool_size=cfg["pool"], eos_token_id=eos_id,
                                   pad_token_id=pad_id)
        batch = ops.MultiblockBatch(P, prm, dev)
        fwd = [(lambda m: (lambda kv_rows, rows: [m.greedy_rows(kv_rows[b], [rows[b]])[0] for b in range(len(rows))]))(m)
               for m in models]
        inputs = []
        for p in range(P):
            draft = [int(x) for x in rng.choice(kvs[p], size=n)]
            ngram, _ = O.mb_prefill(fwd[p], kvs[p], draft)
            inputs.append(ngram)
        okvs = [list(k) for k in kvs]
        for call in range(3):
            res = run_calls(batch, models, kvs, inputs, dev, dtype=dtype)
            nxt_inputs = []
            for p in range(P):
                st = O.mb_generation_call(fwd[p], inputs[p], okvs[p], n=n, K=cfg["K"], r=cfg["r"],
                                          n_gram_pool_size=cfg["pool"], eos_token_id=eos_id, pad_token_id=pad_id)
                ctx = f"call {call} prompt {p}"
                assert res[p]["ret"] == st.ret, ctx
                assert res[p]["next_token"] == (st.next_token if st.next_token is not None else -1), ctx
                assert res[p]["iters"] == st.iters, ctx
                assert res[p]["kv_tokens"] == st.kv_tokens, ctx
                assert res[p]["banners"] == st.banners, ctx
                okvs[p] = st.kv_tokens
                kvs[p] = res[p]["kv_tokens"]
                nt = st.next_token if st.next_token is not None else 0
                nxt_inputs.append([nt] + [int(x) for x in rng.choice(kvs[p], size=n - 1)])
            inputs = nxt_inputs


@pytest.mark.parametrize("backend", BACKENDS)
def test_spawn_without_pad_raises(backend):
    """MB:631-632: spawning without pad_token_id is a ValueError."""
    with use_backend(backend):
        dev = device_for(backend)
        m = ScriptedModel(64, 3, 100, 5)
        prm = ops.MultiblockParams(n=8, K=2, r=0.25, eos_token_id=None, pad_token_id=None)
        batch = ops.MultiblockBatch(1, prm, dev)
        with pytest.raises(ValueError):
            run_calls(batch, [m], [m.prompt()], [m.ar_continuation(5, 8)], dev)


@pytest.mark.parametrize("backend", BACKENDS)
def test_bad_shapes(backend):
    with use_backend(backend):
        dev = device_for(backend)
        batch = ops.MultiblockBatch(2, ops.MultiblockParams(n=8, pad_token_id=0), dev)
        with pytest.raises(ValueError):
            batch.begin(torch.zeros((2, 7), dtype=torch.int64), torch.zeros((2,), dtype=torch.int32))


@pytest.mark.parametrize("backend", BACKENDS)
def test_pack_valid_index(backend):
    """jf_mb_pack's compacted position list: flat index row*Tpad + t of every draft-carrying position, prompt by prompt
    and row by row, rounded up to the requested multiple with -1."""
    with use_backend(backend):
        dev = device_for(backend)
        P, n = 3, 8
        prm = ops.MultiblockParams(n=n, K=2, r=0.5, n_gram_pool_size=4, eos_token_id=None, pad_token_id=0)
        batch = ops.MultiblockBatch(P, prm, dev)
        models = [ScriptedModel(64, 40 + p, 60, 5 + p, reserved=(0,), period=3) for p in range(P)]
        kvs = [m.prompt() for m in models]
        rng = np.random.default_rng(0)
        inputs = [[int(x) for x in rng.choice(kvs[p], size=n)] for p in range(P)]
        d = batch.begin(torch.tensor(inputs, dtype=torch.int64), torch.tensor([len(k) for k in kvs], dtype=torch.int32))
        seen_ragged = False
        for _ in range(12):
            packed_in = batch.pack(d, t_align=4, compact=True, valid_align=8)
            if packed_in is None:
                break
            ids = packed_in[0].cpu().numpy()
            B, T = batch.desc_field(d, "B").copy(), batch.desc_field(d, "T").copy()
            Tpad = ids.shape[1]
            exp, r0 = [], 0
            for p in range(P):
                for b in range(B[p]):
                    exp += [(r0 + b) * Tpad + t for t in range(T[p])]
                r0 += B[p]
            nv = len(exp)
            exp += [-1] * ((nv + 7) // 8 * 8 - nv)
            assert batch.Nvalid == nv
            assert batch.valid_index.cpu().tolist() == exp
            seen_ragged |= nv < ids.size
            V = models[0].vocab
            logits = np.zeros((ids.shape[0], Tpad, V), dtype=np.float32)
            r0 = 0
            for p in range(P):
                rows = [ids[r0 + b, :T[p]].tolist() for b in range(B[p])]
                if B[p]:
                    logits[r0:r0 + B[p], :T[p]] = models[p].logits_rows(kvs[p], rows)
                r0 += B[p]
            flat = torch.from_numpy(logits).reshape(-1, V)
            vi = batch.valid_index.cpu().long().clamp(min=0)
            before = [list(k) for k in kvs]
            d = batch.verify(flat[vi].contiguous().to(dev))          # compacted logits -> jf_argmax_scatter
            r0 = 0
            for p in range(P):
                if B[p]:
                    keep = int(batch.desc_field(d, "kv_len")[p]) - len(before[p])
                    src = int(batch.desc_field(d, "kv_src_row")[p]) if batch.desc_field(d, "kv_copy_len")[p] > 0 else 0
                    kvs[p] = before[p] + ids[r0 + src, :keep].tolist()
                r0 += B[p]
            if batch.desc_field(d, "done").all():
                break
        assert seen_ragged
        res = batch.results(d)
        for p in range(P):
            st = O.mb_generation_call((lambda m: (lambda kv_rows, rows: [m.greedy_rows(kv_rows[b], [rows[b]])[0]
                                                                        for b in range(len(rows))]))(models[p]),
                                      inputs[p], models[p].prompt(), n=n, K=2, r=0.5, n_gram_pool_size=4, eos_token_id=None,
                                      pad_token_id=0)
            assert res[p]["ret"] == st.ret and res[p]["iters"] == st.iters


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", MBR, ids=[c["name"] for c in MBR])
def test_reference_crash_is_reproduced(case, backend):
    """Where the reference itself raises (MB:482 cannot broadcast k candidate rows against B not in {1, k} draft rows), the
    state machine reports JF_E_SHAPE -> RuntimeError with torch's message, in the same call and after the same forwards."""
    with use_backend(backend):
        dev = device_for(backend)
        p = case["params"]
        model = ScriptedModel.from_dict(case["model"])
        batch = ops.MultiblockBatch(1, _params(p), dev)
        kv = case["prefill"]["kv_tokens"]
        for call in case["calls"]:
            traces = [[]]
            if "error" not in call:
                r = run_calls(batch, [model], [kv], [call["input"]], dev, traces=traces)[0]
                assert r["ret"] == call["ret"]
                kv = r["kv_tokens"]
                continue
            with pytest.raises(RuntimeError) as ei:
                run_calls(batch, [model], [kv], [call["input"]], dev, traces=traces)
            assert call["error"] in str(ei.value)
            assert [t["out"] for t in traces[0]] == [f["out"] for f in call["forwards"]]


@pytest.mark.parametrize("backend", BACKENDS)
def test_block_list_capacity_is_an_error_not_a_corruption(backend):
    """max_blocks caps the block LISTS (the reference's lists only grow, Q3/Q4): hitting the cap is JF_E_CAPACITY ->
    RuntimeError reported through the descriptor; with the default capacity (1 + max_iteration_count entries) the same call
    completes and matches the oracle."""
    with use_backend(backend):
        dev = device_for(backend)
        V, n, K, r = 64, 8, 4, 0.25
        m = ScriptedModel(V, 23, 80, 6, reserved=(V - 2,))
        fwd = lambda kv_rows, rows: [m.greedy_rows(kv_rows[b], [rows[b]])[0] for b in range(len(rows))]
        # try first-block guesses until one makes the block lists outgrow K = 4 (most do at r = 0.25)
        found = None
        for s in range(40):
            g = np.random.default_rng(s)
            inp = O.mb_prefill(fwd, m.prompt(), [int(x) for x in g.choice(m.prompt(), size=n)])[0]
            kw = dict(n=n, K=K, r=r, n_gram_pool_size=4, eos_token_id=None, pad_token_id=V - 2)
            st = O.mb_generation_call(fwd, inp, m.prompt(), **kw)
            tight = ops.MultiblockParams(max_blocks=K, **kw)
            try:
                run_calls(ops.MultiblockBatch(1, tight, dev), [m], [m.prompt()], [inp], dev)
            except RuntimeError as e:
                assert "capacity" in str(e)
                found = (inp, st)
                break
        assert found is not None, "no guess outgrew max_blocks = K"
        inp, st = found
        res = run_calls(ops.MultiblockBatch(1, ops.MultiblockParams(**kw), dev), [m], [m.prompt()], [inp], dev)[0]
        assert res["ret"] == st.ret and res["iters"] == st.iters and res["kv_tokens"] == st.kv_tokens


@pytest.mark.gpu
def test_a_missing_row_is_reported_not_waited_for_forever():
    """The steppers of the convergence launch wait for their rows inside the launch.  A position whose item never runs (here: its
    entry of the position list is struck out, so no workgroup writes its slot) must end in JF_E_LAUNCH in that prompt's
    descriptor after the 2 s wall-clock bound — not in a hung GPU — and the batch must be usable again afterwards."""
    import time
    with use_backend("hip"):
        dev = device_for("hip")
        V, n = 64, 8
        prm = ops.MultiblockParams(n=n, K=2, r=0.5, n_gram_pool_size=4, eos_token_id=None, pad_token_id=0)
        batch = ops.MultiblockBatch(2, prm, dev)
        rng = np.random.default_rng(5)
        ids = torch.tensor(rng.integers(1, V, size=(2, n)), dtype=torch.int64)
        kv = torch.tensor([11, 17], dtype=torch.int32)

        def one_iteration(strike):
            d = batch.begin(ids, kv)
            batch.pack(d, compact=True, valid_align=8)
            logits = torch.randn(batch.valid_index.numel(), V, device=dev)
            if strike:
                batch.valid_index[batch.Nvalid - 1] = -1              # the last position of the second prompt gets no item
            return batch.verify(logits, compacted=True)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match="state machine error"):
            one_iteration(True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert 1.5 < dt < 10.0, dt
        d = one_iteration(False)                                      # packed[] was cleared on the error path: a clean step
        assert not batch.desc_field(d, "error").any() and (batch.desc_field(d, "iters") == 2).all()   # stepped once: the second iteration is next


@pytest.mark.parametrize("backend,P", [pytest.param("hostsim", 150, id="hostsim-150"),
                                       pytest.param("hip", 150, id="hip-150", marks=pytest.mark.gpu),
                                       pytest.param("hip", 700, id="hip-700", marks=pytest.mark.gpu)])
def test_many_prompts_side_by_side(backend, P):
    """150 prompts in one batch: 150 stepper workgroups wait for their own rows inside one jf_mb_verify launch (rows of a
    prompt are spread over many item workgroups), rolling over two calls each, against the oracle prompt by prompt.  700
    prompts are more steppers than the launch may carry (half of the resident workgroups: 640 on an MI355X): the same call
    runs as its two launches."""
    n, V = 8, 96
    eos_id, pad_id = V - 1, V - 2
    rng = np.random.default_rng(123)
    with use_backend(backend):
        dev = device_for(backend)
        models, kvs = [], []
        for p in range(P):
            m = ScriptedModel(V, 7000 + p, int(rng.choice([30, 60, 90])), int(rng.integers(3, 12)), eos_id=eos_id,
                              eos_pos=None, reserved=(pad_id,), period=int(rng.choice([0, 4])))
            models.append(m)
            kvs.append(m.prompt())
        prm = ops.MultiblockParams(n=n, K=2, r=0.5, n_gram_pool_size=4, eos_token_id=eos_id, pad_token_id=pad_id)
        batch = ops.MultiblockBatch(P, prm, dev)
        fwd = [(lambda m: (lambda kv_rows, rows: [m.greedy_rows(kv_rows[b], [rows[b]])[0] for b in range(len(rows))]))(m)
               for m in models]
        inputs = [O.mb_prefill(fwd[p], kvs[p], [int(x) for x in rng.choice(kvs[p], size=n)])[0] for p in range(P)]
        okvs = [list(k) for k in kvs]
        kw = dict(n=n, K=2, r=0.5, lookahead_start_ratio=0.0, n_gram_pool_size=4, eos_token_id=eos_id, pad_token_id=pad_id,
                  max_iteration_count=128)
        for call in range(2):
            want = [O.mb_generation_call(fwd[p], inputs[p], okvs[p], **kw) for p in range(P)]
            res = run_calls(batch, models, kvs, inputs, dev)
            nxt = []
            for p in range(P):
                st = want[p]
                assert res[p]["ret"] == st.ret and res[p]["iters"] == st.iters and res[p]["kv_tokens"] == st.kv_tokens, (call, p)
                okvs[p] = st.kv_tokens
                kvs[p] = res[p]["kv_tokens"]
                nt = st.next_token if st.next_token is not None else 0
                nxt.append([nt] + [int(x) for x in rng.choice(kvs[p], size=n - 1)])
            inputs = nxt


@pytest.mark.parametrize("backend", BACKENDS)
def test_fast_path_is_taken_and_changes_nothing(backend):
    """Machine::step_fast (the steady-state iteration as straight-line code) against the general state-machine code on
    the same calls: identical results, and the fast path really is the common case at BASELINE's knobs."""
    with use_backend(backend) as lib:
        dev = device_for(backend)
        n, V, P = 32, 300, 5
        eos_id, pad_id = V - 1, V - 2
        out = {}
        for fast_on in (1, 0):
            old = lib.jf_mb_set_fast_path(fast_on)
            try:
                rng = np.random.default_rng(77)
                models = [ScriptedModel(V, 4000 + p, [0, 40, 70, 82, 95][p], 20 + 7 * p, eos_id=eos_id, eos_pos=None,
                                        reserved=(pad_id,), period=0) for p in range(P)]
                kvs = [m.prompt() for m in models]
                prm = ops.MultiblockParams(n=n, K=2, r=0.85, n_gram_pool_size=4, eos_token_id=eos_id, pad_token_id=pad_id)
                batch = ops.MultiblockBatch(P, prm, dev)
                fwd = [(lambda m: (lambda kv_rows, rows: [m.greedy_rows(kv_rows[b], [rows[b]])[0] for b in range(len(rows))]))(m)
                       for m in models]
                inputs = [O.mb_prefill(fwd[p], kvs[p], [int(x) for x in rng.choice(kvs[p], size=n)])[0] for p in range(P)]
                got = []
                for call in range(3):
                    res = run_calls(batch, models, kvs, inputs, dev)
                    got.append([(r["ret"], r["next_token"], r["iters"], r["kv_tokens"], r["banners"]) for r in res])
                    kvs = [r["kv_tokens"] for r in res]
                    inputs = [[max(r["next_token"], 0)] + [int(x) for x in rng.choice(kvs[p], size=n - 1)] for p, r in enumerate(res)]
                    if fast_on:
                        frac = sum(r["fast_steps"] for r in res) / max(sum(r["steps"] for r in res), 1)
                        assert frac > 0.5, frac
                    else:
                        assert sum(r["fast_steps"] for r in res) == 0
                out[fast_on] = got
            finally:
                lib.jf_mb_set_fast_path(old)
        assert out[1] == out[0]
