"""The C restatement of the verify body (oracle/verify_ref.c, the kernel-level CPU baseline) and the CPU reference
forward used for bench.py's cpu_baseline, checked against the numpy oracle / golden vectors."""
import ctypes as C
import random

import numpy as np
import pytest
import torch

import __graft_entry__ as G
from jacobiforcing_amd import ops
from jacobiforcing_amd.modeling.qwen2 import Qwen2Config, Qwen2Model, Qwen2Weights
from oracle import cpu_reference as CR
from oracle import jacobi_oracle as O

from .backends import use_backend
from .test_decoder_e2e import oracle_generate, scratch_forward, tiny_model


@pytest.fixture(scope="module")
def clib():
    lib = C.CDLL(str(G.build_oracle_c()))
    lib.ref_argmax_rows.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    lib.ref_accept_lengths.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def test_c_argmax_golden(clib, kernel_vectors):
    for case in kernel_vectors["argmax"]:
        bits = np.array(case["bits"], dtype=np.int64)
        if case["dtype"] == "float32":
            x = np.ascontiguousarray((bits & 0xFFFFFFFF).astype(np.uint32))
            dt = 0
        else:
            x = np.ascontiguousarray((bits & 0xFFFF).astype(np.uint16))
            dt = 1
        out = np.zeros(x.shape[0], dtype=np.int64)
        clib.ref_argmax_rows(x.ctypes.data, dt, x.shape[0], x.shape[1], x.shape[1], out.ctypes.data)
        assert out.tolist() == case["argmax"]


def test_c_argmax_random_vs_numpy(clib):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((37, 5001)).astype(np.float32)
    x[3, 17] = x[3, 4000] = 9.0
    out = np.zeros(37, dtype=np.int64)
    clib.ref_argmax_rows(x.ctypes.data, 0, 37, 5001, 5001, out.ctypes.data)
    assert (out == O.argmax_rows(x)).all()
    b = O.f32_to_bf16_bits(x)
    clib.ref_argmax_rows(b.ctypes.data, 1, 37, 5001, 5001, out.ctypes.data)
    assert (out == O.argmax_rows(O.bf16_bits_to_f32(b))).all()


def test_c_accept_golden(clib, kernel_vectors):
    for c in kernel_vectors["accept"]:
        d = np.array(c["draft"], dtype=np.int64)
        g = np.array(c["greedy"], dtype=np.int64)
        acc = np.zeros(g.shape[0], dtype=np.int32)
        best = np.zeros(1, dtype=np.int32)
        clib.ref_accept_lengths(d.ctypes.data, d.shape[0], g.ctypes.data, g.shape[1], g.shape[0], d.shape[1],
                                acc.ctypes.data, best.ctypes.data)
        assert acc.tolist() == c["accepted"] and int(best[0]) == c["best_idx"]


def test_cpu_reference_forward_matches_oracle_tokens():
    """The CPU reference (DynamicCache-style cat/expand/narrow cache) yields the same tokens as the oracle over a
    from-scratch forward of the same weights."""
    with use_backend("hostsim"):
        model = tiny_model("cpu", seed=11)
        V = model.cfg.vocab_size
        prm = ops.MultiblockParams(n=8, K=2, r=0.5, n_gram_pool_size=4, eos_token_id=V - 1, pad_token_id=V - 2)
        prompt = [int(t) for t in np.random.default_rng(1).integers(0, V - 2, size=13)]
        ref = oracle_generate(scratch_forward(model), prompt, prm, 40, 6, random.Random(5))
        cpu = CR.CpuQwen2(model.cfg, model.w, dtype=torch.float32)
        rng = random.Random(5)
        text = list(prompt)
        ngram, cache = CR.cpu_prefill(cpu, prompt, [rng.choice(text) for _ in range(8)])
        kv, inp, gen, calls = list(prompt), ngram, [], 1
        while not (V - 1 in gen or len(gen) >= 40 or calls >= 6):
            st = CR.cpu_multiblock_call(cpu, cache, inp, kv, n=8, K=2, r=0.5, n_gram_pool_size=4, eos_token_id=V - 1,
                                        pad_token_id=V - 2)
            kv = st.kv_tokens
            gen += st.ret
            text += st.ret
            calls += 1
            inp = [st.next_token] + [rng.choice(text) for _ in range(7)]
        assert gen == ref["tokens"]
