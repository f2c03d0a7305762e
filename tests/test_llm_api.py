"""Drop-in surface: ``from jacobiforcing_amd import LLM, SamplingParams`` behaves like the reference's
``inference_engine`` on the Jacobi path.  The correctness criterion is the reference's own
(inference_engine/tests/test_jacobi_decoding_greedy.py:180-206): greedy Jacobi output == greedy AR output."""
import dataclasses
import json
import random

import pytest
import torch

from jacobiforcing_amd import LLM, SamplingParams
from jacobiforcing_amd.config import Config

from .backends import device_for, use_backend

BACKENDS = [pytest.param("hostsim", id="hostsim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture()
def model_dir(tmp_path):
    cfg = dict(vocab_size=320, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, max_position_embeddings=1024, rms_norm_eps=1e-6, rope_theta=10000.0,
               tie_word_embeddings=False, eos_token_id=319, pad_token_id=318, model_type="qwen2")
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    return str(tmp_path)


def test_sampling_params_surface():
    """Field names and defaults of inference_engine/sampling_params.py:4-38."""
    want = dict(temperature=1.0, max_tokens=64, ignore_eos=False, decode_strategy="autoregressive", jacobi_block_len=64,
                jacobi_max_iterations=128, jacobi_max_blocks=2, jacobi_spawn_ratio=0.85, jacobi_lookahead_start_ratio=0.0,
                jacobi_n_gram_pool_size=4, jacobi_on_policy=False)
    assert {f.name: f.default for f in dataclasses.fields(SamplingParams)} == want
    assert SamplingParams().use_jacobi
    with pytest.raises(AssertionError):
        SamplingParams(temperature=-1.0)
    with pytest.raises(ValueError):
        SamplingParams(temperature=0.0, jacobi_on_policy=True)


def test_config_checks(model_dir, tmp_path):
    c = Config(model_dir, max_model_len=512)
    assert c.hf_config.vocab_size == 320 and c.max_model_len == 512
    with pytest.raises(AssertionError):
        Config(model_dir, kvcache_block_size=100)
    with pytest.raises(AssertionError):
        Config(str(tmp_path / "missing"))
    with pytest.raises(NotImplementedError):
        Config(model_dir, tensor_parallel_size=2)


@pytest.mark.parametrize("backend", BACKENDS)
def test_jacobi_matches_autoregressive(model_dir, backend, monkeypatch):
    monkeypatch.setenv("JF_INIT_STD", "0.3")
    monkeypatch.setenv("JF_DTYPE", "float32")     # bf16 GEMV-vs-GEMM rounding flips near-tie argmaxes of a random model
    with use_backend(backend):
        dev = device_for(backend)
        llm = LLM(model_dir, tokenizer_path="none", device=dev, max_model_len=512, max_num_batched_tokens=512, max_num_seqs=4)
        prompts = [[5, 9, 200, 31, 7], [100, 101, 102, 103, 104, 105, 106, 107, 108], [250, 3]]
        N = 40
        ar = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=N, ignore_eos=True), use_tqdm=False)
        jac = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=N, ignore_eos=True, decode_strategy="jacobi",
                                                   jacobi_block_len=16), use_tqdm=False)
        mb = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=N, ignore_eos=True,
                                                  decode_strategy="jacobi_multiblock_rejection_recycling",
                                                  jacobi_block_len=16, jacobi_max_blocks=2, jacobi_spawn_ratio=0.5),
                          use_tqdm=False)
        assert [len(o["token_ids"]) for o in ar] == [N] * 3
        for a, j, m in zip(ar, jac, mb):
            assert len(j["token_ids"]) >= N and len(m["token_ids"]) >= N     # E3: the last iteration may overshoot
            assert j["token_ids"][:N] == a["token_ids"]
            assert m["token_ids"][:N] == a["token_ids"]
            assert j["text"] == ""
        # cross-mode (test_jacobi_decoding_greedy.py:314-491): one request at a time == the same request inside a batch
        for i, pr in enumerate(prompts):
            one_ar = llm.generate([pr], SamplingParams(temperature=0.0, max_tokens=N, ignore_eos=True), use_tqdm=False)
            one_j = llm.generate([pr], SamplingParams(temperature=0.0, max_tokens=N, ignore_eos=True, decode_strategy="jacobi",
                                                      jacobi_block_len=16), use_tqdm=False)
            assert one_ar[0]["token_ids"] == ar[i]["token_ids"]
            assert one_j[0]["token_ids"][:N] == ar[i]["token_ids"]
        tpf = llm.model_runner.jacobi_decoder.stats
        assert tpf["tokens_accepted"] >= 3 * N and tpf["num_jacobi_iterations"] >= 1
        # kwargs of LLM.generate (llm.py:22-37); the names the reference breaks on are mapped
        kw = llm.generate(prompts[:1], SamplingParams(temperature=0.0, max_tokens=20, ignore_eos=True), use_tqdm=False,
                          greedy=True, jacobi_block_len=8, jacobi_num_blocks=2, jacobi_ngram_pool_size=4, jacobi_spawn_ratio=0.5)
        assert kw[0]["token_ids"][:20] == ar[0]["token_ids"][:20]
        llm.exit()


@pytest.mark.parametrize("backend", BACKENDS)
def test_multiblock_requests_keep_their_own_budgets(model_dir, backend, monkeypatch):
    """Requests of one multiblock batch with different max_tokens: each stops at ITS budget (whole blocks are appended, so at
    most one block of overshoot), and a request that ran out of cache row is finished instead of being re-prefilled."""
    monkeypatch.setenv("JF_INIT_STD", "0.3")
    monkeypatch.setenv("JF_DTYPE", "float32")
    with use_backend(backend):
        dev = device_for(backend)
        llm = LLM(model_dir, tokenizer_path="none", device=dev, max_model_len=512, max_num_batched_tokens=512, max_num_seqs=4)
        prompts = [[5, 9, 200, 31, 7], [100, 101, 102, 103, 104, 105, 106, 107, 108]]
        mk = lambda n: SamplingParams(temperature=0.0, max_tokens=n, ignore_eos=True,
                                      decode_strategy="jacobi_multiblock_rejection_recycling", jacobi_block_len=8,
                                      jacobi_max_blocks=2, jacobi_spawn_ratio=0.5)
        out = llm.generate(prompts, [mk(10), mk(60)], use_tqdm=False)
        ar = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=60, ignore_eos=True), use_tqdm=False)
        n0, n1 = len(out[0]["token_ids"]), len(out[1]["token_ids"])
        assert 10 <= n0 < 10 + 2 * 8 + 2 and n1 >= 60
        assert out[0]["token_ids"] == ar[0]["token_ids"][:n0] and out[1]["token_ids"][:60] == ar[1]["token_ids"]
        llm.exit()


@pytest.mark.parametrize("backend", BACKENDS)
def test_nongreedy_and_errors(model_dir, backend, monkeypatch):
    monkeypatch.setenv("JF_INIT_STD", "0.3")
    with use_backend(backend):
        dev = device_for(backend)
        llm = LLM(model_dir, tokenizer_path="none", device=dev, max_model_len=512, max_num_batched_tokens=512, max_num_seqs=4)
        torch.manual_seed(0)
        out = llm.generate([[1, 2, 3, 4], [9, 8, 7]], SamplingParams(temperature=0.8, max_tokens=24, ignore_eos=True,
                                                                     decode_strategy="jacobi", jacobi_block_len=8), use_tqdm=False)
        assert all(len(o["token_ids"]) >= 24 for o in out)
        with pytest.raises(NotImplementedError):        # MR:1538-1542 mixed strategies
            llm.generate([[1, 2, 3], [4, 5, 6]], [SamplingParams(temperature=0.0, max_tokens=4, decode_strategy="jacobi"),
                                                 SamplingParams(temperature=0.0, max_tokens=4)], use_tqdm=False)
        llm2 = LLM(model_dir, tokenizer_path="none", device=dev, max_model_len=512, max_num_batched_tokens=512, max_num_seqs=4)
        with pytest.raises(NotImplementedError):        # MR:367-373 mixed temperatures
            llm2.generate([[1, 2, 3], [4, 5, 6]], [SamplingParams(temperature=0.0, max_tokens=4, decode_strategy="jacobi"),
                                                  SamplingParams(temperature=1.0, max_tokens=4, decode_strategy="jacobi")], use_tqdm=False)
        with pytest.raises(ValueError):
            llm2.generate(["text prompt"], SamplingParams(max_tokens=4), use_tqdm=False)


@pytest.mark.parametrize("backend", BACKENDS)
def test_on_policy_rollout_records(model_dir, backend, monkeypatch):
    """SamplingParams(jacobi_on_policy=True): generate() returns rollout records (JDO:7-28) instead of texts — one
    {block index -> record} dict per sequence, the batch's list repeated once per sequence exactly as the reference's
    LLMEngine does (ENG:99-116, 176-185)."""
    monkeypatch.setenv("JF_INIT_STD", "0.3")
    with use_backend(backend):
        dev = device_for(backend)
        llm = LLM(model_dir, tokenizer_path="none", device=dev, max_model_len=512, max_num_batched_tokens=512, max_num_seqs=4)
        torch.manual_seed(0)
        prompts = [[1, 2, 3, 4, 5], [9, 8, 7]]
        sp = SamplingParams(temperature=0.9, max_tokens=20, ignore_eos=True, decode_strategy="jacobi", jacobi_block_len=8,
                            jacobi_on_policy=True)
        recs = llm.generate(prompts, sp, use_tqdm=False)
        assert len(recs) == len(prompts) ** 2 and recs[:2] == recs[2:]
        for i, r in enumerate(recs[:2]):
            assert sorted(r) == list(range(len(r))) and len(r) >= 1
            total = 0
            for k in sorted(r):
                b = r[k]
                assert b["diffusion_itr_id"] == f"itr_{k}" and b["data_id"] == f"data_{i}"
                assert all(len(v) == 8 for v in b["answer_trajectory_ids"]) and len(b["answer_trajectory_ids"]) >= 2
                assert b["prompt_ids"][:len(prompts[i])] == prompts[i]
                assert b["teacher_output_ids"][:len(b["prompt_ids"])] == b["prompt_ids"]
                assert b["num_iters"] == k + 1 and b["num_forwards"] >= b["num_iters"]
                total = len(b["teacher_output_ids"]) - len(prompts[i])
            stop = {319}
            done = b["teacher_output_ids"]
            assert total >= 20 or done[-1] in stop            # budget reached, or a stop token ended the rollout
            # the last vector of block k is what was committed for that block
            first = r[0]
            assert first["answer_trajectory_ids"][-1][:min(8, total)] == done[len(prompts[i]):len(prompts[i]) + min(8, total)]


def test_string_prompts_go_through_the_directory_tokenizer(model_dir, tmp_path, monkeypatch):
    """`LLM(model, tokenizer_path)` with text prompts (ENG:59-63, 188-200): the tokenizer found in the directory encodes the
    prompt, supplies EOS / PAD ids and decodes the result; without one, a text prompt is refused."""
    pytest.importorskip("transformers")
    tk = pytest.importorskip("tokenizers")
    vocab = {f"w{i}": i for i in range(300)}
    vocab.update({"<pad>": 300, "<eos>": 301, "<unk>": 302})
    tok = tk.Tokenizer(tk.models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = tk.pre_tokenizers.Whitespace()
    tdir = tmp_path / "tok"                        # its own directory: config.json's model_type would pick Qwen2's tokenizer class
    tdir.mkdir()
    tok.save(str(tdir / "tokenizer.json"))
    (tdir / "tokenizer_config.json").write_text(json.dumps({"tokenizer_class": "PreTrainedTokenizerFast", "eos_token": "<eos>",
                                                                "pad_token": "<pad>", "unk_token": "<unk>"}))
    monkeypatch.setenv("JF_DTYPE", "float32")
    with use_backend("hostsim"):
        llm = LLM(model_dir, tokenizer_path=str(tdir), device="cpu", max_model_len=256, max_num_batched_tokens=256, max_num_seqs=2)
        assert (llm.config.eos, llm.config.pad) == (301, 300)
        sp = SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True)
        by_text = llm.generate(["w5 w9 w200 w31"], sp, use_tqdm=False)[0]
        by_ids = llm.generate([[5, 9, 200, 31]], sp, use_tqdm=False)[0]
        assert by_text["token_ids"] == by_ids["token_ids"] and len(by_text["token_ids"]) == 6
        assert by_text["text"] == llm.tokenizer.decode(by_text["token_ids"])
        bare = LLM(model_dir, tokenizer_path="none", device="cpu", max_model_len=256, max_num_batched_tokens=256, max_num_seqs=2)
        with pytest.raises(ValueError):
            bare.generate(["w5 w9"], sp, use_tqdm=False)


@pytest.mark.parametrize("backend", BACKENDS)
def test_more_requests_than_cache_rows_queue_up(model_dir, backend, monkeypatch):
    """Five requests on an engine with two cache rows (max_num_seqs=2): the rest of the queue waits for a row, as the
    reference's waits for KV blocks (SCH:27-47); every request decodes exactly as it does alone, in all three strategies."""
    monkeypatch.setenv("JF_INIT_STD", "0.3")
    monkeypatch.setenv("JF_DTYPE", "float32")
    with use_backend(backend):
        dev = device_for(backend)
        llm = LLM(model_dir, tokenizer_path="none", device=dev, max_model_len=256, max_num_batched_tokens=256, max_num_seqs=2)
        prompts = [[5, 9, 200, 31, 7], [100, 101, 102], [250, 3], [7, 7, 7, 8], [1, 2, 3, 4, 5, 6]]
        N = 10
        sp = SamplingParams(temperature=0.0, max_tokens=N, ignore_eos=True)
        alone = [llm.generate([p], sp, use_tqdm=False)[0]["token_ids"] for p in prompts]
        assert [o["token_ids"] for o in llm.generate(prompts, sp, use_tqdm=False)] == alone
        for strategy in ("jacobi", "jacobi_multiblock_rejection_recycling"):
            out = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=N, ignore_eos=True, decode_strategy=strategy,
                                                       jacobi_block_len=4), use_tqdm=False)
            assert [o["token_ids"][:N] for o in out] == alone, strategy


@pytest.mark.parametrize("backend", BACKENDS)
def test_engine_queue_stress_with_eos(tmp_path, backend, monkeypatch):
    """Fourteen requests of mixed prompt lengths and budgets over three cache rows, with an EOS id the random model really
    emits, so requests finish at different times and freed rows are reused in every order: each request's tokens equal
    the ones it produces alone (greedy AR), for the AR, Jacobi and multiblock strategies."""
    from collections import Counter
    import numpy as np
    monkeypatch.setenv("JF_INIT_STD", "0.3")
    monkeypatch.setenv("JF_DTYPE", "float32")
    base = dict(vocab_size=320, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, max_position_embeddings=1024, rms_norm_eps=1e-6, rope_theta=10000.0,
                tie_word_embeddings=False, eos_token_id=319, pad_token_id=318, model_type="qwen2")
    rng = np.random.default_rng(11)
    prompts = [[int(x) for x in rng.integers(0, 300, size=int(rng.integers(2, 40)))] for _ in range(14)]
    budgets = [int(rng.integers(3, 28)) for _ in prompts]
    kw = dict(tokenizer_path="none", max_model_len=256, max_num_batched_tokens=256, max_num_seqs=3)
    with use_backend(backend):
        dev = device_for(backend)
        d0 = tmp_path / "a"; d0.mkdir(); (d0 / "config.json").write_text(json.dumps(base))
        free = LLM(str(d0), device=dev, **kw).generate(prompts, [SamplingParams(temperature=0.0, max_tokens=b, ignore_eos=True)
                                                                for b in budgets], use_tqdm=False)
        eos = Counter(t for o in free for t in o["token_ids"][2:]).most_common(1)[0][0]      # an id that shows up mid-stream
        d1 = tmp_path / "b"; d1.mkdir(); (d1 / "config.json").write_text(json.dumps(dict(base, eos_token_id=int(eos))))
        llm = LLM(str(d1), device=dev, **kw)
        sps = lambda **extra: [SamplingParams(temperature=0.0, max_tokens=b, **extra) for b in budgets]
        alone = [llm.generate([p], sp, use_tqdm=False)[0]["token_ids"] for p, sp in zip(prompts, sps())]
        assert any(a and a[-1] == eos and len(a) < b for a, b in zip(alone, budgets))          # some requests do stop early
        assert [o["token_ids"] for o in llm.generate(prompts, sps(), use_tqdm=False)] == alone
        for strategy in ("jacobi", "jacobi_multiblock_rejection_recycling"):
            out = llm.generate(prompts, sps(decode_strategy=strategy, jacobi_block_len=4), use_tqdm=False)
            for o, a, b in zip(out, alone, budgets):
                got = o["token_ids"]
                if a[-1] == eos and len(a) < b:
                    assert got == a, strategy                           # stops on the same EOS, nothing after it
                else:
                    assert got[:b] == a, strategy


def test_preempted_request_gives_its_cache_row_back():
    """A tiny KV block pool forces the scheduler to preempt (SCH:63-73): the preempted request's static cache row returns to the
    free list, so the admission limit counts rows that are really free and a later admission never meets 'no free KV cache
    row'; everything still decodes to the greedy continuation."""
    import json, tempfile
    from pathlib import Path
    from jacobiforcing_amd import LLM, SamplingParams
    from tests.backends import use_backend
    cfgd = dict(vocab_size=97, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, max_position_embeddings=2048, rms_norm_eps=1e-6, rope_theta=10000.0,
                tie_word_embeddings=False, eos_token_id=96, pad_token_id=95, model_type="qwen2")
    d = tempfile.mkdtemp()
    (Path(d) / "config.json").write_text(json.dumps(cfgd))
    prompts = [[(7 * i + j) % 90 for j in range(200 + 40 * i)] for i in range(4)]
    sp = SamplingParams(temperature=0.0, max_tokens=150, ignore_eos=True)
    with use_backend("hostsim"):
        ref = LLM(d, tokenizer_path="none", device="cpu", max_model_len=1024, max_num_batched_tokens=4096, max_num_seqs=4)
        want = [r["token_ids"] for r in ref.generate(prompts, sp, use_tqdm=False)]
        # 256-token blocks: 6 blocks cannot hold four requests that each cross a block edge while decoding
        llm = LLM(d, tokenizer_path="none", device="cpu", max_model_len=1024, max_num_batched_tokens=4096, max_num_seqs=4,
                  num_kvcache_blocks=6)
        sched = llm.scheduler
        n_pre = [0]
        orig = sched.preempt

        def counting(seq):
            n_pre[0] += 1
            orig(seq)
            assert seq.cache_row < 0                       # the row went back with the blocks
        sched.preempt = counting
        got = [r["token_ids"] for r in llm.generate(prompts, sp, use_tqdm=False)]
        assert n_pre[0] >= 1, "the pool was meant to force a preemption"
        assert got == want
        assert sorted(llm.model_runner.free_rows) == list(range(llm.model_runner.max_rows))


# ----------------------------------------------------------------------------- the device-resident chunk loop
def _engine_state(llm, seqs_log):
    bm = llm.scheduler.block_manager
    return dict(free=list(bm.free_block_ids), used=sorted(bm.used_block_ids), seqs=seqs_log)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("temperature", [0.0, 0.8], ids=["greedy", "T08"])
def test_chunk_loop_on_device_arrays_equals_the_callback_contract(tmp_path, backend, temperature, monkeypatch):
    """The engine decoders with ModelRunner's ``forward_step_loop`` (draft / positions / cached lengths read from the loop's
    device arrays, request objects brought up to date once per chunk) against the same decoders driven through the reference's
    callback contract (``forward_step_batch(seqs, draft)``, request objects current before every forward: JF_ENGINE_LOOP=0):
    tokens, stats, every request's counters and block table, and the block pool, request by request — with an EOS id the
    random model emits (requests leave the batch at different iterations: the arrays are compacted), mixed budgets, two block
    lengths in one batch and a prompt that crosses a KV-block boundary mid-chunk."""
    from collections import Counter
    import numpy as np
    from jacobiforcing_amd.engine import llm_engine
    monkeypatch.setenv("JF_INIT_STD", "0.3")
    monkeypatch.setenv("JF_DTYPE", "float32")
    base = dict(vocab_size=320, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, max_position_embeddings=1024, rms_norm_eps=1e-6, rope_theta=10000.0,
                tie_word_embeddings=False, eos_token_id=319, pad_token_id=318, model_type="qwen2")
    rng = np.random.default_rng(5)
    prompts = [[int(x) for x in rng.integers(0, 300, size=int(rng.integers(2, 40)))] for _ in range(9)]
    prompts[3] = [int(x) for x in rng.integers(0, 300, size=254)]            # 254 + 4 - 1 positions: its first forward needs a second KV block
    budgets = [int(rng.integers(3, 40)) for _ in prompts]
    budgets[3] = 30
    blocks = [6 if i % 3 else 4 for i in range(len(prompts))]
    kw = dict(tokenizer_path="none", max_model_len=512, max_num_batched_tokens=2048, max_num_seqs=16)
    d0 = tmp_path / "a"; d0.mkdir(); (d0 / "config.json").write_text(json.dumps(base))

    def run(model_dir, loop_on):
        monkeypatch.setenv("JF_ENGINE_LOOP", "1" if loop_on else "0")
        torch.manual_seed(7)
        random.seed(7)                                         # (the prefill draft is random.choice of the prompt, MR:797)
        llm = LLM(str(model_dir), device=dev, **kw)
        log = []
        orig = llm.scheduler.postprocess_jacobi

        def spy(seqs, toks):                                   # the request objects as the engine sees them after every chunk
            for s, t in zip(seqs, toks):
                log.append((len(log), list(s.token_ids), s.num_cached_tokens, list(s.block_table), s.num_permanent_spec_blocks, list(t)))
            return orig(seqs, toks)
        llm.scheduler.postprocess_jacobi = spy
        sps = [SamplingParams(temperature=temperature, max_tokens=b, decode_strategy="jacobi", jacobi_block_len=L)
               for b, L in zip(budgets, blocks)]
        out = llm.generate(prompts, sps, use_tqdm=False)
        dec = llm.model_runner.jacobi_decoder
        assert (dec.forward_step_loop is not None) == loop_on
        return out, dict(dec.stats), _engine_state(llm, log)

    with use_backend(backend):
        dev = device_for(backend)
        free, _, _ = run(d0, False)
        eos = Counter(t for o in free for t in o["token_ids"][2:]).most_common(1)[0][0]
        d1 = tmp_path / "b"; d1.mkdir(); (d1 / "config.json").write_text(json.dumps(dict(base, eos_token_id=int(eos))))
        want = run(d1, False)
        got = run(d1, True)
    assert any(o["token_ids"] and o["token_ids"][-1] == eos and len(o["token_ids"]) < b for o, b in zip(want[0], budgets))
    assert any(len(e[3]) == 2 for e in want[2]["seqs"])             # a block table grew inside a chunk
    assert got[0] == want[0]
    assert got[1] == want[1]
    assert got[2] == want[2]


# ----------------------------------------------------------------------------- the reference's own acceptance test, in its own shape
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mode", ["autoregressive", "jacobi_loop", "jacobi_callbacks", "jacobi_T08"])
def test_paged_layout_decodes_like_the_contiguous_cache(tmp_path, backend, mode, monkeypatch):
    """``Config(kv_cache_layout="paged")``: the reference's memory model (a pool of 256-token blocks, K/V wherever the block
    table says, slot mapping per forward: layers/attention.py:10-40, MR:965-986, 1204-1265) behind the same decoders — the consumer
    of jf_engine_fill inside the package.  Same tokens and stats as the contiguous layout request by request:
    AR, greedy Jacobi through the chunk loop on device arrays (slot mapping from the loop's device lengths: PagedFill.fill_device) and
    through the callback contract (PagedFill.fill = MR:1204-1265), and T = 0.8 — with prompts that start next to and cross a
    256-token block edge, more requests than cache rows (blocks are handed back and re-used in another order) and two block lengths."""
    import numpy as np
    monkeypatch.setenv("JF_INIT_STD", "0.3")
    monkeypatch.setenv("JF_DTYPE", "float32")
    base = dict(vocab_size=320, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, max_position_embeddings=1024, rms_norm_eps=1e-6, rope_theta=10000.0,
                tie_word_embeddings=False, eos_token_id=-1, pad_token_id=318, model_type="qwen2")
    (tmp_path / "config.json").write_text(json.dumps(base))
    rng = np.random.default_rng(11)
    lens = [5, 254, 256, 257, 9, 250, 511 - 60]
    prompts = [[int(x) for x in rng.integers(0, 300, size=n)] for n in lens]
    budgets = [int(rng.integers(6, 22)) for _ in prompts]
    strategy = "autoregressive" if mode == "autoregressive" else "jacobi"
    T = 0.8 if mode == "jacobi_T08" else 0.0
    monkeypatch.setenv("JF_ENGINE_LOOP", "0" if mode == "jacobi_callbacks" else "1")
    sps = [SamplingParams(temperature=T, max_tokens=b, ignore_eos=True, decode_strategy=strategy, jacobi_block_len=8 if i % 2 else 5)
           for i, b in enumerate(budgets)]

    def run(layout):
        torch.manual_seed(3)
        random.seed(3)
        llm = LLM(str(tmp_path), tokenizer_path="none", device=dev, max_model_len=640, max_num_batched_tokens=4096, max_num_seqs=4,
                  kv_cache_layout=layout)
        assert llm.model_runner.paged == (layout == "paged")
        out = llm.generate(prompts, sps, use_tqdm=False)
        dec = llm.model_runner.jacobi_decoder
        bm = llm.scheduler.block_manager
        assert not bm.used_block_ids                                  # every block came back
        return [o["token_ids"] for o in out], (dict(dec.stats) if dec is not None else None)

    with use_backend(backend):
        dev = device_for(backend)
        want = run("contiguous")
        got = run("paged")
    assert all(len(t) >= b for t, b in zip(want[0], budgets)) if strategy == "autoregressive" else all(len(t) > 0 for t in want[0])
    assert got[0] == want[0]
    assert got[1] == want[1]
    with pytest.raises(ValueError):
        Config(str(tmp_path), kv_cache_layout="rows")


def _per_position_js(a, b, V):
    """Mean over positions of the Jensen-Shannon divergence (natural log) between the empirical token distributions of two sample
    sets (lists of token-id lists) at that position — compute_token_distributions + compare_distributions(metric="js") of
    inference_engine/tests/test_jacobi_decoding_nongreedy.py:204-320."""
    import numpy as np
    T = min(min(len(x) for x in a), min(len(x) for x in b))
    out = []
    for t in range(T):
        p = np.bincount([x[t] for x in a], minlength=V).astype(np.float64)
        q = np.bincount([x[t] for x in b], minlength=V).astype(np.float64)
        p, q = p / p.sum(), q / q.sum()
        m = 0.5 * (p + q)
        kl = lambda u, w: float(np.sum(np.where(u > 0, u * np.log(np.where(u > 0, u, 1.0) / np.where(w > 0, w, 1.0)), 0.0)))
        out.append(0.5 * kl(p, m) + 0.5 * kl(q, m))
    return float(np.mean(out)), out


@pytest.mark.gpu
@pytest.mark.parametrize("filters", [None, (50, 0.9)], ids=["plain", "top_k50_top_p09"])
def test_nongreedy_jacobi_matches_autoregressive_sampling_per_position(tmp_path, filters, monkeypatch):
    """The reference's acceptance test of the non-greedy decoder (test_jacobi_decoding_nongreedy.py:324-435): sample the same
    prompt N times with the autoregressive sampler and N times with decode_strategy="jacobi" at the same temperature, compare
    the empirical token distributions position by position (Jensen-Shannon), pass when the mean is below a threshold (0.1
    there; 0.05 here, and not above twice what two autoregressive sample sets differ by, i.e. the sampling noise of N draws).
    Through LLM.generate on the GPU, V = 1 024, 16 positions, N = 4 096, T = 0.8, with and without top_k = 50 / top_p = 0.9
    planted on the requests (both samplers read them, JDN:117-118).  A random model's per-position marginals are broad — the
    contexts diverge — so the sampling noise of N draws is what bounds the statistic: ~0.15 at N = 512 (two autoregressive
    sets against each other), ~0.02 at N = 4 096."""
    monkeypatch.setenv("JF_INIT_STD", "1.0")                # next-token distributions with a real choice (most of 512 samples are distinct)
    monkeypatch.setenv("JF_MAX_ROWS", "256")
    V, N, T, L, POS = 1024, 4096, 0.8, 8, 16
    cfg = dict(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, max_position_embeddings=1024, rms_norm_eps=1e-6, rope_theta=10000.0,
               tie_word_embeddings=False, eos_token_id=-1, pad_token_id=V - 2, model_type="qwen2")
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    with use_backend("hip"):
        torch.manual_seed(1234)
        llm = LLM(str(tmp_path), tokenizer_path="none", device="cuda", max_model_len=256, max_num_batched_tokens=65536, max_num_seqs=256)
        prompt = [5, 9, 200, 31, 7, 640, 77, 3]

        def sample(strategy, n):
            out = []
            for _ in range(n // 256):
                sp = SamplingParams(temperature=T, max_tokens=POS, ignore_eos=True, decode_strategy=strategy, jacobi_block_len=L)
                if filters:
                    sp.top_k, sp.top_p = filters
                out += [o["token_ids"][:POS] for o in llm.generate([prompt] * 256, sp, use_tqdm=False)]
            return out
        ar1, ar2, jac = sample("autoregressive", N), sample("autoregressive", N), sample("jacobi", N)
        stats = dict(llm.model_runner.jacobi_decoder.stats)
    noise, _ = _per_position_js(ar1, ar2, V)
    js, per_pos = _per_position_js(ar1, jac, V)
    js2, _ = _per_position_js(ar2, jac, V)
    distinct, distinct_ar = len({tuple(x) for x in jac}), len({tuple(x) for x in ar1})
    print(f"per-position JS: AR vs AR {noise:.4f}, AR vs Jacobi {js:.4f} / {js2:.4f}; distinct Jacobi samples {distinct} of {N} (autoregressive: {distinct_ar}); "
          f"tokens per iteration {stats['tokens_accepted'] / max(stats['num_jacobi_iterations'], 1) / 256:.2f}")
    assert all(len(x) == POS for x in jac)
    # a real distribution, not a degenerate one: as many different sequences as the autoregressive sampler draws (top_k = 50 / top_p = 0.9
    # leave this model's peaked rows a nucleus of a few ids: ~10^2 different sequences in 4 096 draws on either side)
    assert distinct > (32 if filters else N // 16) and distinct_ar // 2 < distinct < 2 * distinct_ar, (distinct, distinct_ar)
    assert js < 0.05 and js2 < 0.05, (js, js2, noise, per_pos)
    # (the autoregressive sampler filters float32 probabilities, the decoder the bf16 ones the reference's dtype rule gives it: with
    #  top-p the two nuclei can differ by a bf16 step of the running sum — a real, small difference on top of the sampling noise)
    assert max(js, js2) < max(2.0 * noise, 0.01) + 0.005, (js, js2, noise)
