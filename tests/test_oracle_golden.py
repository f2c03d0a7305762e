"""Pin the CPU oracle (oracle/jacobi_oracle.py) against golden vectors recorded from the
unmodified reference (tests/golden/gen_golden.py).  CPU only."""
import json

import numpy as np
import pytest

from oracle import jacobi_oracle as O
from oracle.scripted_model import ScriptedModel

from .conftest import forward_matches, kv_matches, load_golden

FV = load_golden("fullvocab_cases.json")               # round 5: every entry point at V = 152 064
MB = load_golden("mb_cases.json") + load_golden("mb_cases_v2.json") + load_golden("mb_cases_v3.json") + FV["mb"]
SB = load_golden("sb_cases.json") + load_golden("sb_cases_v2.json") + FV["sb"]
JD = load_golden("jd_cases.json") + load_golden("jd_cases_v2.json") + FV["jd"]
JDN = load_golden("jdn_cases.json") + load_golden("jdn_cases_v2.json") + load_golden("jdn_cases_v3.json") + load_golden("jdn_cases_v4.json") + FV["jdn"]
FLT = load_golden("filter_vectors.json")
MBR = load_golden("mb_raises.json")
SLOTS = load_golden("slot_cases.json")
JDO = load_golden("jdo_cases.json") + load_golden("jdo_cases_v2.json") + load_golden("jdo_cases_v3.json") + load_golden("jdo_cases_v4.json") + FV["jdo"]
SMX = load_golden("softmax_vectors.json")


def scripted_forward(model):
    def fwd(kv_rows, out_rows):
        return [model.greedy_rows(kv_rows[b], [out_rows[b]])[0] for b in range(len(out_rows))]
    return fwd


# ----------------------------------------------------------------------------- kernel-level
def test_argmax_semantics(kernel_vectors):
    for case in kernel_vectors["argmax"]:
        bits = np.array(case["bits"], dtype=np.int64)
        if case["dtype"] == "float32":
            x = (bits & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
        else:
            x = O.bf16_bits_to_f32((bits & 0xFFFF).astype(np.uint16))
        assert O.argmax_rows(x).tolist() == case["argmax"]


def test_accept_lengths(kernel_vectors):
    for c in kernel_vectors["accept"]:
        acc = O.accept_lengths(c["draft"], c["greedy"])
        assert acc == c["accepted"]
        assert O.first_max_index(acc) == c["best_idx"]


def test_bf16_roundtrip():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(4096).astype(np.float32)
    b = O.f32_to_bf16_bits(x)
    import torch
    ref = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().astype(np.uint16)
    assert (b == ref).all()
    assert (O.bf16_bits_to_f32(b) == torch.from_numpy(x).to(torch.bfloat16).float().numpy()).all()


# ----------------------------------------------------------------------------- multiblock
@pytest.mark.parametrize("case", MB, ids=[c["name"] for c in MB])
def test_multiblock_calls(case):
    p = case["params"]
    model = ScriptedModel.from_dict(case["model"])
    assert model.prompt() == case["prompt"]
    fwd = scripted_forward(model)
    # prefill (MB:175-225)
    ngram, kv = O.mb_prefill(fwd, case["prompt"], case["prefill"]["draft"])
    assert ngram == case["prefill"]["ngram"]
    assert kv == case["prefill"]["kv_tokens"]
    for ci, call in enumerate(case["calls"]):
        assert len(kv) == call["kv_len_before"]
        st = O.mb_generation_call(fwd, call["input"], kv, n=p["n"], K=p["K"], r=p["r"],
                                  lookahead_start_ratio=p["lookahead"], n_gram_pool_size=p["pool"],
                                  eos_token_id=p["eos_id"], pad_token_id=p["pad_id"],
                                  max_iteration_count=p["max_iter"])
        ctx = f"{case['name']} call {ci}"
        assert st.ret == call["ret"], ctx
        assert [st.next_token] == call["next_token"], ctx
        assert st.iters == call["iters"], ctx
        assert st.kv_len() == call["kv_len"], ctx
        assert kv_matches(st.kv_tokens, call), ctx
        assert len(st.kv_rows) == call["kv_batch"], ctx
        assert st.banners == call["banners"], ctx
        # per-iteration forward inputs/outputs (the runaway cases of mb_cases_v3.json hold them as digests)
        assert len(st.trace) == len(call["forwards"]), ctx
        for it, (a, b) in enumerate(zip(st.trace, call["forwards"])):
            assert forward_matches(a, b), f"{ctx} iter {it}"
        kv = st.kv_tokens


def test_multiblock_greedy_equals_ar(mb_cases):
    """The reference's own correctness criterion (test_jacobi_decoding_greedy.py:180-206): greedy Jacobi
    output == greedy AR output.  Holds up to the EOS token for every recorded case."""
    for case in mb_cases:
        model = ScriptedModel.from_dict(case["model"])
        gen = case["summary"]["generated"]
        ar = model.ar_continuation(len(case["prompt"]), len(gen))
        assert gen == ar, case["name"]


# ----------------------------------------------------------------------------- single block (HF)
@pytest.mark.parametrize("case", SB, ids=[c["name"] for c in SB])
def test_singleblock_calls(case):
    model = ScriptedModel.from_dict(case["model"])
    fwd = scripted_forward(model)
    n, eos = case["params"]["n"], case["params"]["eos_id"]
    ngram, kv = O.mb_prefill(fwd, case["prompt"], case["prefill"]["draft"])
    assert ngram == case["prefill"]["ngram"]
    for ci, call in enumerate(case["calls"]):
        r = O.sb_generation_call(fwd, call["input"], kv, n=n, eos_token_id=eos)
        assert r["ret"] == call["ret"], (case["name"], ci)
        assert [r["next_token"]] == call["next_token"]
        assert r["iters"] == call["iters"]
        assert r["kv_tokens"] == call["kv_tokens"]
        for a, b in zip(r["trace"], call["forwards"]):
            assert a["out"] == b["out"][0] and a["greedy"] == b["greedy"][0]
        kv = r["kv_tokens"]


@pytest.mark.parametrize("case", MBR, ids=[c["name"] for c in MBR])
def test_multiblock_reference_crash_is_restated(case):
    """Configurations on which the reference raises (torch cannot broadcast the draft rows against the candidate rows at
    MB:482): the oracle fails in the same call, after the same forwards, with torch's message."""
    p = case["params"]
    fwd = scripted_forward(ScriptedModel.from_dict(case["model"]))
    ngram, kv = O.mb_prefill(fwd, case["prompt"], case["prefill"]["draft"])
    kw = dict(n=p["n"], K=p["K"], r=p["r"], lookahead_start_ratio=p["lookahead"], n_gram_pool_size=p["pool"],
              eos_token_id=p["eos_id"], pad_token_id=p["pad_id"], max_iteration_count=p["max_iter"])
    for call in case["calls"]:
        if "error" not in call:
            st = O.mb_generation_call(fwd, call["input"], kv, **kw)
            assert st.ret == call["ret"] and st.kv_tokens == call["kv_tokens"]
            kv = st.kv_tokens
            continue
        with pytest.raises(RuntimeError) as ei:
            O.mb_generation_call(fwd, call["input"], kv, **kw)
        assert str(ei.value) == call["error"]
        assert [t["out"] for t in ei.value.trace] == [f["out"] for f in call["forwards"]]


# ----------------------------------------------------------------------------- engine greedy
def _mk_seqs(case, max_iters=128):
    seqs, models = [], []
    for d in case["seqs"]:
        m = ScriptedModel.from_dict(d["model"])
        assert m.prompt() == d["prompt"]
        models.append(m)
        seqs.append(O.OracleSeq(d["prompt"], d.get("block_len", case["params"].get("block_len")),
                                d.get("max_tokens", case["params"].get("max_tokens")), max_iters=max_iters,
                                prefill_draft=d.get("prefill_draft")))
    return seqs, models


@pytest.mark.parametrize("case", JD, ids=[c["name"] for c in JD])
def test_engine_greedy(case):
    p = case["params"]
    seqs, models = _mk_seqs(case, p["max_iters"])
    by_id = {id(s): m for s, m in zip(seqs, models)}
    trace = []

    def fwd(ss, drafts):
        trace.append(dict(seq_idx=[seqs.index(s) for s in ss], draft=[list(d) for d in drafts],
                          seq_lens=[len(s) for s in ss]))
        return [by_id[id(s)].greedy_rows(s.token_ids[:-1], [d])[0][:-1] for s, d in zip(ss, drafts)]

    stream = O.CounterStream(p["pad_seed"])
    stats = O.new_stats()
    pads = stream.pads(p["vocab"])
    if p["batch"]:
        out = O.engine_generate_batch(fwd, seqs, p["eos_id"], pads, stats)
    else:
        out = [O.engine_generate_single(fwd, s, p["eos_id"], pads, stats) for s in seqs]
    assert out == case["outputs"]
    assert stats == case["stats"]
    assert stream.k == case["pads_consumed"]
    for s, f in zip(seqs, case["final"]):
        assert s.token_ids == f["token_ids"]
        assert s.num_cached_tokens == f["num_cached_tokens"]
        assert s.num_table_blocks == f["num_blocks"]
    assert trace == case["forwards"]


# ----------------------------------------------------------------------------- engine non-greedy
def _as_dtype(logits, logits_dtype):
    """What ``.to(torch.bfloat16)`` does to the scripted model's float32 logits (values kept as float32)."""
    return O.bf16_round(logits) if logits_dtype == "bf16" else logits


@pytest.mark.parametrize("case", SMX, ids=[f"V{c['V']}_T{c['temperature']}" for c in SMX])
def test_target_probs_follow_torch_rounding_points(case):
    """_build_target_probs (JDN:110-123) recorded from torch on bf16 logits: the temperature-scaled logits are one rounding of
    a float32 quotient and must agree bit for bit; the probabilities are a float32 softmax rounded to bf16 — float32 softmax
    implementations differ in the last place, so entries may sit one bf16 ulp apart, in well under 1 % of the positions."""
    V, T = case["V"], case["temperature"]
    lb = np.array(case["logits_bf16"], dtype=np.uint16).reshape(-1, V)
    x = O.bf16_bits_to_f32(lb)
    t = np.float32(T)
    scaled = x if T == 1.0 else O.bf16_round((x / t).astype(np.float32))
    assert np.array_equal(O.f32_to_bf16_bits(scaled), np.array(case["scaled_bf16"], dtype=np.uint16).reshape(-1, V))
    p = O.target_probs(x, T, "bf16")
    got = O.f32_to_bf16_bits(p).astype(np.int64)
    want = np.array(case["probs_bf16"], dtype=np.uint16).reshape(-1, V).astype(np.int64)
    assert np.abs(got - want).max() <= 1                     # positive bf16 payloads are ordered like the values
    assert (got != want).mean() < 0.01
    # float32 logits keep a float32 softmax (no rounding point): 1e-5 relative against torch
    pf = O.target_probs(x, T, "f32")
    wf = np.array(case["probs_f32_of_f32_logits"], dtype=np.uint32).reshape(-1, V).view(np.float32)
    assert np.allclose(pf, wf, rtol=2e-5, atol=0)
    assert np.allclose(p.astype(np.float64).sum(-1), 1.0, atol=5e-3)


@pytest.mark.parametrize("case", FLT, ids=[f"V{c['V']}_T{c['temperature']}_k{c['top_k']}_p{c['top_p']}_{i}" for i, c in enumerate(FLT)])
def test_filtered_target_probs(case):
    """_build_target_probs with top_k / top_p (JDN:72-123) recorded from torch on bf16 logits and on their float32 images.
    bf16: bit for bit wherever the kept set does not end inside a group of equal probabilities (there torch's topk / sort pick
    by kernel: the SAME number of ids with the SAME values must survive).  float32: the same kept set, values within a few
    float32 ulps (torch's softmax and its float32 sums differ from the exact ones in the last place)."""
    V, T, k, tp = case["V"], case["temperature"], case["top_k"], case["top_p"]
    x = O.bf16_bits_to_f32(np.array(case["logits_bf16"], dtype=np.uint16).reshape(-1, V))
    want_b = O.bf16_bits_to_f32(np.array(case["probs_bf16"], dtype=np.uint16).reshape(-1, V))
    want_f = np.array(case["probs_f32_of_f32_logits"], dtype=np.uint32).reshape(-1, V).view(np.float32)
    got_b = O.target_probs(x, T, "bf16", k, tp)
    got_f = O.target_probs(x, T, "f32", k, tp)
    pb, pf = O.target_probs(x, T, "bf16"), O.target_probs(x, T, "f32")
    for r in range(x.shape[0]):
        if O.filter_boundary_is_tied(pb[r], k, tp, 7):
            assert (got_b[r] > 0).sum() == (want_b[r] > 0).sum()
            assert np.array_equal(np.sort(got_b[r]), np.sort(want_b[r]))
        else:
            assert np.array_equal(got_b[r], want_b[r]), r
        a, b = got_f[r], want_f[r]
        if O.filter_boundary_is_tied(pf[r], k, tp, 23):                 # (equal LOGITS are equal probabilities in float32 too)
            assert (a > 0).sum() == (b > 0).sum()
            a, b = np.sort(a), np.sort(b)
        else:
            assert np.array_equal(a > 0, b > 0), r
        ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
        assert ulp.max() <= 16, (r, int(ulp.max()))             # (the smallest kept probabilities: torch's float32 softmax is a few ulps off there)
    assert np.allclose(got_b.astype(np.float64).sum(-1), 1.0, atol=2e-2) and np.allclose(got_f.astype(np.float64).sum(-1), 1.0, atol=1e-5)


# Records of the reference in which a top-p kept set ends INSIDE a group of equal bf16 probabilities AND a draw lands on one of
# the tied ids: which of them survive is torch.sort's kernel choice (not stable, not the same at every size), so the decode
# cannot be reproduced by any rule — kept in the fixture as evidence of exactly that (tests below), not as parity targets.
TIE_CHOICE_OBSERVABLE = {"jdn4_bf16_flat_p08", "jdn4_bf16_topp09_L32"}


def _tied_rows(case) -> tuple:
    """(rows the oracle's decode of this record forms filtered probabilities for, rows whose kept set ends inside a tie)."""
    p = case["params"]
    log = [0, 0]
    orig = O.target_probs

    def counting(logits, temperature, dt="f32", top_k=None, top_p=None):
        base = orig(logits, temperature, dt)
        for row in base.reshape(-1, base.shape[-1]):
            log[0] += 1
            log[1] += bool(O.filter_boundary_is_tied(row, top_k, top_p, 7 if dt == "bf16" else 23))
        return orig(logits, temperature, dt, top_k, top_p)
    O.target_probs = counting
    try:
        try:
            test_engine_nongreedy.__wrapped__(case)
            ok = True
        except AssertionError:
            ok = False
    finally:
        O.target_probs = orig
    return log[0], log[1], ok


def test_filtered_records_and_tied_boundaries():
    """Every other jdn4 record is reproduced draw for draw (some of them meet tied cuts too — 62 of 225 rows in the k=50 record —
    but no draw of theirs lands on a tied id); the two that are not meet a tied cut in most rows."""
    for case in load_golden("jdn_cases_v4.json"):
        rows, tied, ok = _tied_rows(json.loads(json.dumps(case)))
        if case["name"] in TIE_CHOICE_OBSERVABLE:
            assert not ok and tied > rows // 2, (case["name"], rows, tied)
        else:
            assert ok, (case["name"], rows, tied)


def _engine_nongreedy(case):
    p = case["params"]
    seqs, models = _mk_seqs(case)
    by_id = {id(s): m for s, m in zip(seqs, models)}

    ldt = p.get("logits_dtype", "f32")

    def fwd(ss, drafts):
        return [_as_dtype(by_id[id(s)].logits_rows(s.token_ids[:-1], [d])[0][:-1], ldt) for s, d in zip(ss, drafts)]

    pads = O.CounterStream(p["rng_seed"] * 3 + 1)
    unis = O.CounterStream(p["rng_seed"] * 3 + 2)
    bonus = O.CounterStream(p["rng_seed"] * 3 + 3)
    stats = O.new_stats()
    args = (p["eos_id"], p["temperature"], pads.pads(p["vocab"]), unis.uniform, bonus.uniform, stats)
    flt = dict(top_k=p.get("top_k"), top_p=p.get("top_p"))      # (jdn_cases_v4.json: planted on the reference's SamplingParams instances)
    if p["batch"]:
        out = O.nongreedy_generate_batch(fwd, seqs, *args, logits_dtype=ldt, **flt)
    else:
        out = [O.nongreedy_generate_single(fwd, s, *args, logits_dtype=ldt, **flt) for s in seqs]
    assert out == case["outputs"]
    assert stats == case["stats"]
    assert dict(pads=pads.k, uniforms=unis.k, bonus=bonus.k) == case["draws"]
    for s, f in zip(seqs, case["final"]):
        assert s.token_ids == f["token_ids"]
        assert s.num_cached_tokens == f["num_cached_tokens"]


@pytest.mark.parametrize("case", [c for c in JDN if c["name"] not in TIE_CHOICE_OBSERVABLE],
                         ids=[c["name"] for c in JDN if c["name"] not in TIE_CHOICE_OBSERVABLE])
def test_engine_nongreedy(case):
    _engine_nongreedy(case)


test_engine_nongreedy.__wrapped__ = _engine_nongreedy


# ----------------------------------------------------------------------------- on-policy rollout records (JDO)
def run_oracle_jdo(case):
    p = case["params"]
    seqs, models = _mk_seqs(case, max_iters=p["max_blocks"])
    by_id = {id(s): m for s, m in zip(seqs, models)}
    trace = []
    ldt = p.get("logits_dtype", "f32")

    def fwd(ss, drafts):
        trace.append(dict(seq_idx=[seqs.index(s) for s in ss], draft=[list(d) for d in drafts], seq_lens=[len(s) for s in ss]))
        return [_as_dtype(by_id[id(s)].logits_rows(s.token_ids[:-1], [d])[0][:-1], ldt) for s, d in zip(ss, drafts)]

    inits = O.CounterStream(p["rng_seed"] * 5 + 1)
    unis = O.CounterStream(p["rng_seed"] * 5 + 2)
    multi = O.CounterStream(p["rng_seed"] * 5 + 3)
    records, metrics = O.onpolicy_rollout_records_batch(fwd, seqs, p["temperature"], p["stop_ids"], p["pad_id"], p["vocab"],
                                                        O.ScriptedRandom(inits), unis.uniform, multi.uniform,
                                                        logits_dtype=ldt, top_k=p.get("top_k"), top_p=p.get("top_p"))
    return seqs, records, metrics, dict(inits=inits.k, uniforms=unis.k, multinomial=multi.k), trace


def check_jdo(case, seqs, records, metrics, draws, trace=None):
    assert [{str(k): v for k, v in r.items()} for r in records] == case["records"]
    assert metrics == case["metrics"]
    assert draws == case["draws"]
    for s, f in zip(seqs, case["final"]):
        assert list(s.token_ids) == f["token_ids"]
        assert s.num_cached_tokens == f["num_cached_tokens"]
    if trace is not None:
        assert [dict(draft=t["draft"], seq_lens=t["seq_lens"]) for t in trace] == \
               [dict(draft=t["draft"], seq_lens=t["seq_lens"]) for t in case["forwards"]]


# A full-vocabulary bf16 rollout whose multinomial draws land among the 152 062 noise ids: torch's bf16 softmax is one bf16 ulp off
# the exactly rounded value on < 1 % of the entries, which moves the running sum of the inverse-CDF draw by more than a noise id's
# mass — the reference's own draw is then a NEIGHBOURING id.  Kept as evidence of where "bit for bit" ends.
SOFTMAX_ULP_OBSERVABLE = {"fv_jdo_bf16_T08_stop"}


@pytest.mark.parametrize("case", [c for c in JDO if c["name"] not in SOFTMAX_ULP_OBSERVABLE],
                         ids=[c["name"] for c in JDO if c["name"] not in SOFTMAX_ULP_OBSERVABLE])
def test_engine_onpolicy_records(case):
    seqs, records, metrics, draws, trace = run_oracle_jdo(case)
    check_jdo(case, seqs, records, metrics, draws, trace)


def test_full_vocabulary_bf16_draws_see_torchs_softmax_ulps():
    """Why the record in SOFTMAX_ULP_OBSERVABLE is not a parity target, shown on one row of its kind (V = 152 064, bf16, T = 0.8,
    a planted id of logit 13 over noise in (-1, 1)): torch's bf16 softmax and the definition (exact softmax rounded once) differ by
    exactly one bf16 ulp on ~1 % of the entries (1 364 of 152 064 here); the inverse-CDF draw the goldens inject then picks a
    different id for most of the uniforms that land among the noise ids — a few places away — while a draw that lands on the
    planted id is the same under both.  And the record itself: the oracle's first deviation from it is such a neighbouring id."""
    import torch
    V = 152064
    m = ScriptedModel(V, 9042, 90, 8, eos_id=V - 1, reserved=(V - 2,), peak=13.0)
    x = torch.from_numpy(m.logits_rows(m.prompt()[:-1], [[m.prompt()[-1], 5]])[0][:1]).to(torch.bfloat16)
    p_torch = torch.softmax(x / 0.8, dim=-1)[0].float().numpy()
    p_def = O.target_probs(x.float().numpy(), 0.8, "bf16")[0]
    diff = np.flatnonzero(p_torch != p_def)
    assert 0 < diff.size < V // 50
    ulp = np.abs(O.f32_to_bf16_bits(p_torch[diff]).astype(np.int64) - O.f32_to_bf16_bits(p_def[diff]).astype(np.int64))
    assert (ulp == 1).all()
    g = int(np.argmax(p_def))
    us = (np.arange(2000) + 0.5) / 2000.0
    a = np.array([O.inverse_cdf_sample(p_torch, float(u)) for u in us])
    b = np.array([O.inverse_cdf_sample(p_def, float(u)) for u in us])
    moved = a != b
    noise = (a != g) & (b != g)
    assert 0.005 < noise.mean() < 0.1                                  # the planted id holds 0.984 of the mass: ~1.6 % of the draws land elsewhere
    assert moved[noise].mean() > 0.5                                   # ... and MOST of those pick another id under torch's tensor (measured: 85 %)
    assert not moved[~noise].any()                                     # a draw on the planted id is the same under both
    assert np.abs(a[moved] - b[moved]).max() < 50                      # the other id is a few places away (measured: <= 9)
    case = [c for c in JDO if c["name"] in SOFTMAX_ULP_OBSERVABLE][0]
    _, records, _, _, _ = run_oracle_jdo(json.loads(json.dumps(case)))
    got, want = records[0][0]["answer_trajectory_ids"], case["records"][0]["0"]["answer_trajectory_ids"]
    first = next((i, j) for i, (r, w) in enumerate(zip(got, want)) for j, (x1, x2) in enumerate(zip(r, w)) if x1 != x2)
    assert abs(got[first[0]][first[1]] - want[first[0]][first[1]]) < 50


# ----------------------------------------------------------------------------- paged-KV slot mapping (MR:965-986, 1252-1254)
def test_engine_fill_matches_reference_slot_pattern():
    """The (block index, offset) pattern the reference's ModelRunner._get_slot_mapping_pattern returned for (seq_len, draft_len)
    pairs — first token in the last slot of a block, drafts crossing one and two block boundaries — fixes the oracle's
    slot_mapping = block_table[block] * block_size + offset, positions and cu_seqlens."""
    rng = np.random.default_rng(1)
    for c in SLOTS:
        bs, S, L = c["block_size"], c["seq_len"], c["draft_len"]
        table = [int(x) for x in rng.choice(10_000, size=max(c["block_indices"]) + 1, replace=False)]
        ref = O.engine_fill_ref([list(range(L))], [S], [table], bs, max_cols=len(table) + 2)
        want = [table[b] * bs + o for b, o in zip(c["block_indices"], c["offsets"])]
        assert ref["slot_mapping"] == want, c
        assert ref["positions"] == [S - 1 + j for j in range(L)]
        assert ref["cu_seqlens_q"] == [0, L] and ref["cu_seqlens_k"] == [0, S - 1 + L] and ref["cache_seqlens"] == [S - 1]


def test_exact_softmax_escalation_levels_agree(monkeypatch):
    """The definition of the probability tensor (exact softmax rounded once): float64 decides, elements near a rounding boundary
    are re-decided in 80-bit and then in 60-digit decimal arithmetic.  Forcing MANY elements through the two higher levels (absurdly
    wide near-tie bands) must not change a single probability — float64 was right for them — and the result is the correctly
    rounded quotient: checked against exact rational arithmetic on a small row."""
    from fractions import Fraction
    import decimal
    rng = np.random.default_rng(3)
    x = O.bf16_round((rng.standard_normal((3, 300)) * 3).astype(np.float32))
    base_bf16, base_f32 = O.target_probs(x, 0.7, "bf16"), O.target_probs(x, 0.7, "f32")
    monkeypatch.setattr(O, "NEAR_TIE_REL", 5e-3)                       # ~every element takes the extended-precision path
    before = list(O.NEAR_TIES_RESOLVED)
    assert np.array_equal(O.target_probs(x, 0.7, "bf16"), base_bf16) and np.array_equal(O.target_probs(x, 0.7, "f32"), base_f32)
    assert O.NEAR_TIES_RESOLVED[0] > before[0]
    monkeypatch.setattr(O, "NEAR_TIE_REL_LD", 5e-3)                    # ... and the decimal one
    assert np.array_equal(O.target_probs(x[:1], 0.7, "bf16"), base_bf16[:1]) and np.array_equal(O.target_probs(x[:1], 0.7, "f32"), base_f32[:1])
    assert O.NEAR_TIES_RESOLVED[1] > before[1]
    # bf16: the correctly rounded quotient, from 60-digit exponentials and exact rational rounding
    ctx = decimal.Context(prec=60)
    xs = O.bf16_round((x[0] / np.float32(0.7)).astype(np.float32))
    d = xs.astype(np.float64) - np.float64(xs.max())
    e = [Fraction(ctx.exp(decimal.Decimal(float(v)))) for v in d]
    S = sum(e)
    for i in range(0, 300, 7):
        q = e[i] / S
        got = Fraction(float(base_bf16[0, i]))
        # neighbours of `got` on the bf16 grid: the rounded value must be at least as close to q as either of them
        bits = int(O.f32_to_bf16_bits(np.array([base_bf16[0, i]], dtype=np.float32))[0])
        lo = Fraction(float(O.bf16_bits_to_f32(np.array([max(bits - 1, 0)], dtype=np.uint16))[0]))
        hi = Fraction(float(O.bf16_bits_to_f32(np.array([bits + 1], dtype=np.uint16))[0]))
        assert abs(q - got) <= abs(q - lo) and abs(q - got) <= abs(q - hi), i


# ------------------------------------------------------------------------------------- the fixtures ARE the reference's outputs
def test_fixtures_regenerate_from_the_reference_byte_for_byte():
    """tests/golden/gen_golden.py --check: import the unmodified reference (build container only: /root/reference), re-run every
    recorded call into a temporary directory and compare all 22 JSON files with the committed ones, byte for byte.  Skips where the
    reference is absent (the GPU box).  Two processes side by side (fullvocab_cases.json alone takes as long as the other 21)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    if not Path("/root/reference").is_dir():
        pytest.skip("the reference tree is not present here")
    if os.environ.get("JF_SKIP_GOLDEN_CHECK") == "1":
        pytest.skip("JF_SKIP_GOLDEN_CHECK=1")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, str(root / "tests" / "golden" / "gen_golden.py"), "--check"]
    procs = [subprocess.Popen(cmd + [flag], env=env, cwd=str(root), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for flag in ("--skip-full-vocabulary", "--full-vocabulary")]
    outs = [p.communicate(timeout=1500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "21 of 21 fixture files regenerate byte-identical" in outs[0] and "1 of 1 fixture files regenerate byte-identical" in outs[1], outs
    assert not list(Path("/root/reference").rglob("__pycache__")), "the import wrote bytecode into the reference tree"
