"""Engine single-block greedy decoder (jacobiforcing_amd.engine.jacobi_decoding.JacobiDecoder, HIP jf_engine_step)
against golden vectors recorded from the reference's JacobiDecoder (tests/golden/jd_cases.json)."""
import numpy as np
import pytest
import torch

from jacobiforcing_amd import ops
from jacobiforcing_amd.engine.block_manager import BlockManager
from jacobiforcing_amd.engine.jacobi_decoding import JacobiDecoder
from jacobiforcing_amd.engine.sequence import Sequence
from jacobiforcing_amd.sampling_params import SamplingParams
from oracle import jacobi_oracle as O
from oracle.jacobi_oracle import CounterStream
from oracle.scripted_model import ScriptedModel

from .backends import device_for, use_backend
from .conftest import load_golden

FV = load_golden("fullvocab_cases.json")                 # round 5: the reference at V = 152 064
JD = load_golden("jd_cases.json") + load_golden("jd_cases_v2.json") + FV["jd"]
BACKENDS = [pytest.param("hostsim", id="hostsim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


class Harness:
    """Caller side of the decoder seam (what MR:1134-1199 / 1407-1416 do around the model forward)."""

    def __init__(self, vocab, dev, dtype=torch.float32, block_size=256):
        self.bm = BlockManager(64, block_size)
        self.block_size = block_size
        self.models = {}
        self.trace = []
        self.dev, self.dtype = dev, dtype
        self.order = []

    def add(self, model, sp, prefill_draft):
        seq = Sequence(model.prompt(), sp)
        self.bm.allocate(seq)
        seq.num_cached_tokens = len(seq)
        seq._prefill_draft = prefill_draft
        self.models[seq.seq_id] = model
        self.order.append(seq.seq_id)
        return seq

    def forward_step_batch(self, seqs, draft):
        B, L = draft.shape
        if L < 2:
            raise ValueError("Draft must have at least 2 tokens (seed + 1 speculative)")
        d = draft.cpu()
        rows = []
        for i, seq in enumerate(seqs):
            seq.draft_tokens_gpu = draft[i]
            if seq.token_ids[-1] != int(d[i, 0]):
                raise ValueError("Seed mismatch")
            S = len(seq)
            need = (S + L - 1 + self.block_size - 1) // self.block_size
            committed = (S + self.block_size - 1) // self.block_size
            cur = len(seq.block_table)
            if cur > need:
                seq.block_table = seq.block_table[:need]
            for _ in range(max(0, need - cur)):
                bid = self.bm.free_block_ids[0]
                self.bm._allocate_block_no_clear(bid)
                seq.block_table.append(bid)
            seq.num_permanent_spec_blocks = max(seq.num_permanent_spec_blocks, need - committed)
            rows.append(torch.from_numpy(self.models[seq.seq_id].logits_rows(seq.token_ids[:-1], [d[i].tolist()])[0]))
        logits = torch.stack(rows, 0).to(self.dtype).to(self.dev)
        for seq in seqs:
            seq.num_cached_tokens = (len(seq) - 1) + L
        self.trace.append(dict(seq_idx=[self.order.index(s.seq_id) for s in seqs], draft=d.tolist(),
                               seq_lens=[len(s) for s in seqs]))
        return logits[:, :-1, :]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", JD, ids=[c["name"] for c in JD])
def test_engine_greedy_golden(case, dtype, backend):
    with use_backend(backend):
        dev = device_for(backend)
        p = case["params"]
        H = Harness(p["vocab"], dev, dtype)
        dec = JacobiDecoder(H.bm, forward_step=lambda s, d: H.forward_step_batch([s], d),
                            forward_step_batch=H.forward_step_batch, eos_token_id=p["eos_id"], pad_token_id=p["pad_id"],
                            vocab_size=p["vocab"], device=torch.device(dev))
        stream = CounterStream(p["pad_seed"])
        dec.set_pad_stream([stream.next_u32() % p["vocab"] for _ in range(max(case["pads_consumed"], 1) + 64)])
        seqs = []
        for d in case["seqs"]:
            m = ScriptedModel.from_dict(d["model"])
            sp = SamplingParams(temperature=0.0, max_tokens=d["max_tokens"], decode_strategy="jacobi",
                                jacobi_block_len=d["block_len"], jacobi_max_iterations=p["max_iters"])
            seqs.append(H.add(m, sp, d["prefill_draft"]))
        out = dec.generate_chunk_batch(seqs) if p["batch"] else [dec.generate_chunk(s) for s in seqs]
        assert out == case["outputs"]
        assert dec.stats == case["stats"]
        assert dec._pad_cursor == case["pads_consumed"]
        for s, f in zip(seqs, case["final"]):
            assert s.token_ids == f["token_ids"]
            assert s.num_cached_tokens == f["num_cached_tokens"]
            assert len(s.block_table) == f["num_blocks"]
        assert H.trace == case["forwards"]


def test_constructor_errors():
    with use_backend("hostsim"):
        with pytest.raises(ValueError):
            JacobiDecoder(None, vocab_size=10)
        with pytest.raises(ValueError):
            JacobiDecoder(None, forward_step=lambda s, d: None)


# ----------------------------------------------------------------------------- non-greedy (rejection sampling)
from jacobiforcing_amd.engine.jacobi_decoding_nongreedy import JacobiDecoderNonGreedy  # noqa: E402

# jdn_cases_v4.json: the reference with top_k / top_p planted on its SamplingParams instances (round 5); without the two records in
# which torch's choice among EQUAL probabilities at a cut is observable (tests/test_oracle_golden.py TIE_CHOICE_OBSERVABLE)
JDN = load_golden("jdn_cases.json") + load_golden("jdn_cases_v2.json") + load_golden("jdn_cases_v3.json") + \
    [c for c in load_golden("jdn_cases_v4.json") if c["name"] not in ("jdn4_bf16_flat_p08", "jdn4_bf16_topp09_L32")] + FV["jdn"]
# (fullvocab_cases.json holds one bf16 rollout whose draws see torch's softmax ulps: tests/test_oracle_golden.py SOFTMAX_ULP_OBSERVABLE)
JDO = load_golden("jdo_cases.json") + load_golden("jdo_cases_v2.json") + load_golden("jdo_cases_v3.json") + load_golden("jdo_cases_v4.json") + \
    [c for c in FV["jdo"] if c["name"] != "fv_jdo_bf16_T08_stop"]
BMC = load_golden("bm_cases.json")
TORCH_DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16}


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", JDN, ids=[c["name"] for c in JDN])
def test_engine_nongreedy_golden(case, backend):
    """Rejection-sampling verify with injected uniforms / residual draws / pads against the reference's
    JacobiDecoderNonGreedy, for float32 AND bfloat16 logits (``params.logits_dtype``; the reference keeps the logits dtype
    through softmax, JDN:64-70, so bf16 logits mean bf16-rounded probabilities).  Floating point enters through the
    float32 exp / sum inside softmax only; the committed token ids must still agree for these seeds."""
    with use_backend(backend):
        dev = device_for(backend)
        p = case["params"]
        H = Harness(p["vocab"], dev, TORCH_DTYPES[p.get("logits_dtype", "f32")])
        dec = JacobiDecoderNonGreedy(H.bm, forward_step=lambda s, d: H.forward_step_batch([s], d),
                                     forward_step_batch=H.forward_step_batch, eos_token_id=p["eos_id"],
                                     pad_token_id=p["pad_id"], vocab_size=p["vocab"], device=torch.device(dev))
        pads, unis, bonus = (CounterStream(p["rng_seed"] * 3 + k) for k in (1, 2, 3))
        dr = case["draws"]
        dec.set_streams([pads.next_u32() % p["vocab"] for _ in range(dr["pads"] + 64)],
                        [unis.uniform() for _ in range(dr["uniforms"] + 64)],
                        [bonus.uniform() for _ in range(dr["bonus"] + 64)])
        seqs = []
        for d in case["seqs"]:
            m = ScriptedModel.from_dict(d["model"])
            sp = SamplingParams(temperature=p["temperature"], max_tokens=p["max_tokens"], decode_strategy="jacobi",
                                jacobi_block_len=p["block_len"])
            for k in ("top_k", "top_p"):                     # planted on the instance, as the recording planted them (JDN:117-118)
                if k in p:
                    setattr(sp, k, p[k])
            seqs.append(H.add(m, sp, None))
        out = dec.generate_chunk_batch(seqs) if p["batch"] else [dec.generate_chunk(s) for s in seqs]
        assert out == case["outputs"]
        assert dec.stats == case["stats"]
        assert dict(pads=dec._cur[2], uniforms=dec._cur[0], bonus=dec._cur[1]) == case["draws"]
        for s, f in zip(seqs, case["final"]):
            assert s.token_ids == f["token_ids"] and s.num_cached_tokens == f["num_cached_tokens"]


@pytest.mark.parametrize("backend", BACKENDS)
def test_a_batch_that_mixes_settings_is_decoded_setting_by_setting(backend):
    """The reference builds the target distribution request by request (JDN:110-123), so one batch may mix (temperature, top_k,
    top_p).  Here a launch takes ONE setting: generate_chunk_batch decodes such a batch part by part, in order of first appearance
     — the same tokens, cached lengths and stream cursors as decoding the parts as separate batches."""
    case = next(c for c in JDN if c["name"] == "jdn4_f32_k20_p08_batch4")
    p = case["params"]
    settings = [dict(temperature=0.7, top_k=5), dict(temperature=1.3), dict(temperature=0.7, top_k=5), dict(temperature=1.3)]

    def build(order):
        dev = device_for(backend)
        H = Harness(p["vocab"], dev, torch.float32)
        dec = JacobiDecoderNonGreedy(H.bm, forward_step=lambda s, d: H.forward_step_batch([s], d), forward_step_batch=H.forward_step_batch,
                                     eos_token_id=p["eos_id"], pad_token_id=p["pad_id"], vocab_size=p["vocab"], device=torch.device(dev))
        pads, unis, bonus = (CounterStream(77 * 3 + k) for k in (1, 2, 3))
        dec.set_streams([pads.next_u32() % p["vocab"] for _ in range(4096)], [unis.uniform() for _ in range(4096)],
                        [bonus.uniform() for _ in range(4096)])
        seqs = {}
        for i in order:
            sp = SamplingParams(temperature=settings[i]["temperature"], max_tokens=p["max_tokens"], decode_strategy="jacobi",
                                jacobi_block_len=p["block_len"])
            if "top_k" in settings[i]:
                sp.top_k = settings[i]["top_k"]
            seqs[i] = H.add(ScriptedModel.from_dict(case["seqs"][i]["model"]), sp, None)
        return dec, seqs

    with use_backend(backend):
        dec_a, sa = build([0, 1, 2, 3])
        out_a = dec_a.generate_chunk_batch([sa[i] for i in range(4)])
        dec_b, sb = build([0, 2, 1, 3])
        out_02 = dec_b.generate_chunk_batch([sb[0], sb[2]])
        out_13 = dec_b.generate_chunk_batch([sb[1], sb[3]])
        assert out_a == [out_02[0], out_13[0], out_02[1], out_13[1]] and all(len(o) > 0 for o in out_a)
        assert dec_a._cur == dec_b._cur
        for i in range(4):
            assert sa[i].token_ids == sb[i].token_ids and sa[i].num_cached_tokens == sb[i].num_cached_tokens == len(sa[i])


# ------------------------------------------------------------------------------------- on-policy rollout records
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", JDO, ids=[c["name"] for c in JDO])
def test_engine_onpolicy_records_golden(case, backend):
    """generate_rollout_records_batch against the records the reference's JacobiDecoderNonGreedyOnPolicy produced with the
    same injected draws: block trajectories, prompts, teacher outputs, metrics, draw counts, forward inputs."""
    from jacobiforcing_amd.engine.jacobi_decoding_nongreedy_on_policy import JacobiDecoderNonGreedyOnPolicy
    with use_backend(backend):
        dev = device_for(backend)
        p = case["params"]
        H = Harness(p["vocab"], dev, TORCH_DTYPES[p.get("logits_dtype", "f32")])
        stop = p["stop_ids"] if len(p["stop_ids"]) > 1 else p["eos_id"]
        dec = JacobiDecoderNonGreedyOnPolicy(H.bm, forward_step=lambda s, d: H.forward_step_batch([s], d),
                                             forward_step_batch=H.forward_step_batch, eos_token_id=stop,
                                             pad_token_id=p["pad_id"], vocab_size=p["vocab"], device=torch.device(dev))
        inits, unis, multi = (CounterStream(p["rng_seed"] * 5 + k) for k in (1, 2, 3))
        dr = case["draws"]
        dec.set_streams(O.ScriptedRandom(inits), [unis.uniform() for _ in range(dr["uniforms"] + 64)],
                        [multi.uniform() for _ in range(dr["multinomial"] + 64)])
        seqs = []
        for d in case["seqs"]:
            m = ScriptedModel.from_dict(d["model"])
            sp = SamplingParams(temperature=p["temperature"], max_tokens=p["max_tokens"], decode_strategy="jacobi",
                                jacobi_block_len=p["block_len"], jacobi_max_iterations=p["max_blocks"], jacobi_on_policy=True)
            for k in ("top_k", "top_p"):                     # jdo_cases_v4.json: planted on the instance (JDO:132-133)
                if k in p:
                    setattr(sp, k, p[k])
            seqs.append(H.add(m, sp, None))
        records, metrics = dec.generate_rollout_records_batch(seqs, return_metrics=True)
        assert [{str(k): v for k, v in r.items()} for r in records] == case["records"]
        assert metrics == case["metrics"]
        assert dict(inits=inits.k, uniforms=dec._cur[0], multinomial=dec._cur[1]) == case["draws"]
        for s, f in zip(seqs, case["final"]):
            assert s.token_ids == f["token_ids"] and s.num_cached_tokens == f["num_cached_tokens"]
        assert [(t["draft"], t["seq_lens"]) for t in H.trace] == [(t["draft"], t["seq_lens"]) for t in case["forwards"]]
        if all(len(s.token_ids) - len(d["prompt"]) >= p["max_tokens"] for s, d in zip(seqs, case["seqs"])):
            assert dec.generate_rollout_records(seqs[0]) == {}            # budget is spent: no further blocks


def test_onpolicy_constructor_errors():
    from jacobiforcing_amd.engine.jacobi_decoding_nongreedy_on_policy import JacobiDecoderNonGreedyOnPolicy as D
    with pytest.raises(ValueError):
        D(None, eos_token_id=1, pad_token_id=0, vocab_size=8)
    f = lambda s, d: None
    with pytest.raises(ValueError):
        D(None, forward_step=f, pad_token_id=0, vocab_size=8)
    with pytest.raises(ValueError):
        D(None, forward_step=f, eos_token_id=1, vocab_size=8)
    with pytest.raises(ValueError):
        D(None, forward_step=f, eos_token_id=1, pad_token_id=0)
    assert D(None, forward_step=f, eos_token_id=[3, 4], pad_token_id=0, vocab_size=8, device="cpu").stop_token_ids == (3, 4)


# ------------------------------------------------------------------------------------- paged-KV index buffers (a16)
@pytest.mark.parametrize("backend", BACKENDS)
def test_paged_fill_matches_reference_arithmetic(backend):
    """jf_engine_fill vs the restatement of MR:1204-1265 / 965-986: positions, slot mapping through the block tables,
    cu_seqlens, cache_seqlens — including sequences that start a new block inside the draft and S = 1."""
    with use_backend(backend):
        dev = device_for(backend)
        bs, max_cols = 256, 12
        rng = np.random.default_rng(3)
        fill = ops.PagedFill(max_batch=16, max_block_len=64, max_blocks_per_seq=max_cols, block_size=bs, device=dev)
        for B, L in [(1, 2), (3, 16), (7, 33), (16, 64)]:
            seq_lens = [int(x) for x in rng.integers(1, 2000, size=B)]
            seq_lens[0] = 1
            if B > 2:
                seq_lens[1] = 256                               # seed is the last slot of block 0, draft starts block 1
                seq_lens[2] = 255 + 256
            tables = []
            for S in seq_lens:
                need = (S + L - 1 + bs - 1) // bs
                tables.append([int(x) for x in rng.choice(4096, size=need, replace=False)])
            draft = torch.from_numpy(rng.integers(0, 1000, size=(B, L))).to(torch.int64)
            out = fill.fill(draft.to(dev), seq_lens, tables)
            ref = O.engine_fill_ref(draft.tolist(), seq_lens, tables, bs, max_cols)
            names = ["input_ids", "positions", "slot_mapping", "cu_seqlens_q", "cu_seqlens_k", "cache_seqlens"]
            for name, got in zip(names, out[:6]):
                assert got.cpu().tolist() == list(ref[name]), (B, L, name)
            assert (out[6].cpu().numpy() == ref["block_tables"]).all()
            assert out[7] == ref["max_seqlen_k"]
        # a draft position without a block is an error, as in the reference (MR:1190-1191)
        with pytest.raises(RuntimeError):
            fill.fill(torch.zeros((1, 8), dtype=torch.int64, device=dev), [250], [[5]])
        with pytest.raises(ValueError):
            fill.fill(torch.zeros((1, 8), dtype=torch.int64, device=dev), [0], [[5]])
        with pytest.raises(ValueError):
            fill.fill(torch.zeros((1, 1), dtype=torch.int64, device=dev), [4], [[5]])


# ------------------------------------------------------------------------------------- block bookkeeping (a17)
@pytest.mark.parametrize("case", BMC, ids=[f"seed{c['seed']}" for c in BMC])
def test_block_manager_matches_reference_traces(case):
    """Scripted sequences of the block operations the decoders drive (forward-side table growth MR:1166-1198,
    may_append_batch BM:267-276, may_append BM:195-265, trim_kv_only_fast BM:534-564) were run on the reference's BlockManager +
    Sequence; this package's classes must go through the same states op by op: table lengths, num_cached_tokens, sequence
    lengths, permanent speculative blocks, free-block count."""
    import random
    rr = random.Random(case["seed"])
    bs = case["block_size"]
    bm = BlockManager(case["num_blocks"], bs)
    seqs = []

    def snap():
        return dict(tables=[len(s.block_table) for s in seqs], cached=[s.num_cached_tokens for s in seqs],
                    lens=[len(s) for s in seqs], spec=[s.num_permanent_spec_blocks for s in seqs], free=len(bm.free_block_ids))
    ops_iter = iter(case["ops"])
    for _ in range(rr.randint(1, 3)):                        # same draws as tests/golden/gen_golden.py::run_bm_case
        plen = rr.choice([1, 5, 200, 255, 256, 257, 511, 600])
        seq = Sequence([rr.randrange(50) for _ in range(plen)], SamplingParams(temperature=0.0, max_tokens=4096))
        bm.allocate(seq)
        seq.num_cached_tokens = len(seq)
        seqs.append(seq)
        op = next(ops_iter)
        assert op["op"] == "allocate" and op["prompt_len"] == plen
    assert snap() == op["after"]
    for step in range(40):
        i = rr.randrange(len(seqs))
        seq = seqs[i]
        kind = rr.choice(["jacobi", "jacobi", "jacobi", "ar"])
        op = next(ops_iter)
        assert op["op"] == kind and op["seq"] == i
        if kind == "ar":
            seq.append_token(rr.randrange(50))
            bm.may_append(seq)
            seq.num_cached_tokens = len(seq)
        else:
            L = rr.choice([2, 4, 16, 33, 64, 300])
            S = len(seq)
            need = (S + L - 1 + bs - 1) // bs
            committed = (S + bs - 1) // bs
            cur = len(seq.block_table)
            if cur > need:
                seq.block_table = seq.block_table[:need]
            for _k in range(max(0, need - cur)):
                bid = bm.free_block_ids[0]
                bm._allocate_block_no_clear(bid)
                seq.block_table.append(bid)
            seq.num_permanent_spec_blocks = max(seq.num_permanent_spec_blocks, need - committed)
            seq.num_cached_tokens = S - 1 + L
            acc = rr.randint(1, L)
            assert (L, acc) == (op["L"], op["acc"])
            if acc > 1:
                seq.extend_tokens([rr.randrange(50) for _ in range(acc - 1)])
                bm.may_append_batch(seq, acc - 1)
                spec = acc - 1
            else:
                seq.append_token(rr.randrange(50))
                bm.may_append(seq)
                spec = 1
            if L - 1 - spec > 0:
                bm.trim_kv_only_fast(seq, L - 1 - spec)
        assert snap() == op["after"], (step, kind)


# ------------------------------------------------------------------------------------- statistical criterion of the reference
@pytest.mark.parametrize("ldt", ["f32", "bf16"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_nongreedy_first_token_follows_the_target_distribution(backend, ldt):
    """The reference's own acceptance test for the non-greedy decoder is statistical: mean Jensen-Shannon divergence < 0.1
    between Jacobi and autoregressive sampling (inference_engine/tests/test_jacobi_decoding_nongreedy.py).  Rejection
    sampling with a delta proposal is distribution-preserving, so over many independent draws the FIRST committed token must
    follow softmax(logits / T) of its position, whatever the draft was."""
    from jacobiforcing_amd.engine.jacobi_decoding_nongreedy import JacobiDecoderNonGreedy
    with use_backend(backend):
        dev = device_for(backend)
        V, T = 12, 3.0                                   # scripted logits peak at 8.0: T = 3 spreads them (max p ~ 0.55)
        eos, pad = V - 1, V - 2
        model = ScriptedModel(V, 4242, 55, 6, eos_id=eos, eos_pos=None, reserved=(pad,))
        trials = 3000 if backend == "hostsim" else 1500
        rng = np.random.default_rng(0)
        counts = np.zeros(V)
        H = Harness(V, dev, TORCH_DTYPES[ldt])
        dec = JacobiDecoderNonGreedy(H.bm, forward_step=lambda s, d: H.forward_step_batch([s], d), forward_step_batch=H.forward_step_batch,
                                     eos_token_id=eos, pad_token_id=pad, vocab_size=V, device=torch.device(dev))
        for _ in range(trials):
            dec.set_streams(rng.integers(0, V, size=64), rng.random(64), rng.random(64))
            sp = SamplingParams(temperature=T, max_tokens=1, decode_strategy="jacobi", jacobi_block_len=4)
            seq = H.add(model, sp, None)
            out = dec.generate_chunk(seq)
            counts[out[0]] += 1
            H.bm.deallocate(seq)
        # the target: softmax of the logits that follow the prompt (the draft row's position 0 sees only the prompt)
        lg = model.logits_rows(model.prompt()[:-1], [[model.prompt()[-1], 0, 0, 0]])[0][0]
        if ldt == "bf16":      # the target is the ROUNDED distribution the reference samples from (bf16 probs need not sum to 1)
            lg = O.bf16_round(lg)
        p = O.target_probs(lg[None, :], T, ldt)[0].astype(np.float64)
        p = p / p.sum()
        q = counts / counts.sum()
        mid = 0.5 * (p + q)
        kl = lambda a, b: float(np.sum(np.where(a > 0, a * np.log(a / b), 0.0)))
        js = 0.5 * kl(p, mid) + 0.5 * kl(q, mid)
        assert js < 0.01, (js, p.round(3), q.round(3))
        assert q.max() < 0.95                                  # the distribution is not degenerate (a real test)


def test_top_k_and_top_p_are_read_like_the_reference_reads_them():
    """SamplingParams has no top_k / top_p (sampling_params.py:4-38); the reference honours such attributes when a caller
    attaches them and switches a stage off for None / out-of-range values (JDN:75, 92-96; JDO:100, 112-116).  Both sampling
    decoders apply them (jf_rs_filter)."""
    import types
    assert ops.active_filters(None, 100) == (0, 0.0)
    assert ops.active_filters(types.SimpleNamespace(temperature=1.0), 100) == (0, 0.0)
    assert ops.active_filters(types.SimpleNamespace(top_k=None, top_p=None), 100) == (0, 0.0)
    assert ops.active_filters(types.SimpleNamespace(top_k=0, top_p=1.0), 100) == (0, 0.0)          # inactive values (JDN:75, 96)
    assert ops.active_filters(types.SimpleNamespace(top_k=100, top_p=0.0), 100) == (0, 0.0)
    assert ops.active_filters(types.SimpleNamespace(top_k=5, top_p=0.9), 100) == (5, 0.9)
    sp = SamplingParams(temperature=0.8, max_tokens=4, decode_strategy="jacobi")
    sp.top_k, sp.top_p = 7, 0.5                                  # a SamplingParams instance takes the planted attributes
    assert ops.active_filters(sp, 100) == (7, 0.5)