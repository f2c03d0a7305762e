"""Kernel-level parity through the C ABI: argmax (a2), accepted-prefix scan (a3), engine step (a15),
KV append/commit (a9/a10/a18) vs the CPU oracle and the golden vectors.  GPU tests are marked."""
import ctypes
import os
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from jacobiforcing_amd import _native as N
from jacobiforcing_amd import ops
from oracle import jacobi_oracle as O

from .conftest import load_golden  # noqa: E402
from .backends import device_for, use_backend

ROOT = Path(__file__).resolve().parents[1]
BACKENDS = [pytest.param("hostsim", id="hostsim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
GPU = pytest.mark.gpu


# ------------------------------------------------------------------------------------- ABI surface
def test_library_exports_every_declared_symbol():
    """The shared library loads and exports every entry point include/jacobiforcing.h declares."""
    hdr = (ROOT / "include" / "jacobiforcing.h").read_text()
    declared = set(re.findall(r"\b(jf_[a-z_0-9]+)\s*\(", hdr)) - {"jf_mb_params", "jf_mb_desc", "jf_engine_row"}
    assert declared == set(N.EXPORTED_SYMBOLS), declared ^ set(N.EXPORTED_SYMBOLS)
    import __graft_entry__ as G
    lib_path = G.build_hip()
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(lib_path)], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert declared <= exported, declared - exported
    assert all(sym.startswith("jf_") for sym in exported), sorted(e for e in exported if not e.startswith("jf_"))[:5]
    raw = ctypes.CDLL(str(lib_path))   # loads without a GPU (no compute call here)
    for name in declared:
        assert hasattr(raw, name)
    raw.jf_version.restype = ctypes.c_int
    assert raw.jf_version() == int(re.search(r"#define JF_VERSION (\d+)", hdr).group(1)) == 600


def test_docs_state_the_headers_entry_point_count_and_version():
    """INTEGRATION.md and COVERAGE.md name the number of entry points and the ABI version: both must be the header's (they
    drifted once: "34 entry points, ABI 400" beside a header with 35 at version 410)."""
    hdr = (ROOT / "include" / "jacobiforcing.h").read_text()
    n = len(set(re.findall(r"^JF_API\s+[^;(]*?\b(jf_[a-z_0-9]+)\s*\(", hdr, flags=re.M)))
    ver = int(re.search(r"#define JF_VERSION (\d+)", hdr).group(1))
    assert n == len(N.EXPORTED_SYMBOLS)
    integ = (ROOT / "INTEGRATION.md").read_text()
    cov = (ROOT / "COVERAGE.md").read_text()
    counts = [int(x) for x in re.findall(r"(\d+) (?:`jf_\*` )?entry points", integ + cov)]
    assert counts and all(c == n for c in counts), (counts, n)
    assert f"ABI version {ver}" in integ and f"ABI {ver})" in cov


def test_plain_c_client(tmp_path):
    """The boundary is a C ABI: a C99 program including include/jacobiforcing.h compiles with gcc, links against the shared
    library and gets answers (sizes, argument validation with jf_last_error) without a GPU."""
    import __graft_entry__ as G
    lib = G.build_hip()
    exe = tmp_path / "abi_client"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "abi" / "abi_client.c"),
                           f"-L{lib.parent}", "-ljacobiforcing", f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    assert "version=600" in out and "desc=64" in out and "params=40" in out
    assert "rc=-1" in out and "null pointer" in out


def test_python_constants_are_the_headers(tmp_path):
    """The mailbox / driver-block / event constants of _native.py are the values a C compiler reads out of the header."""
    names = ["JF_MB_SEQ", "JF_MB_RTOT", "JF_MB_RMAIN", "JF_MB_TPAD", "JF_MB_TMAX", "JF_MB_NVALID", "JF_MB_NVALID_PAD", "JF_MB_NDONE",
             "JF_MB_MAXKV", "JF_MB_ERROR", "JF_MB_ACCEPTED", "JF_MB_NCALL_END", "JF_MB_MAILBOX_HDR", "JF_MB_FIN_INTS",
             "JF_DRV_HDR_INTS", "JF_DRV_ACTIVE", "JF_DRV_STOP", "JF_DRV_CALLS", "JF_DRV_ITERS", "JF_DRV_NEW", "JF_DRV_BUDGET",
             "JF_DRV_MAX_CALLS", "JF_DRV_TEXT_LEN", "JF_DRV_CURSOR", "JF_DRV_FIN_RET_LEN", "JF_DRV_FIN_NEXT", "JF_DRV_FIN_ITERS",
             "JF_DRV_FIN_OFF", "JF_STOP_NONE", "JF_STOP_EOS", "JF_STOP_MAX_NEW_TOKENS", "JF_STOP_MAX_CALLS", "JF_STOP_MAX_SEQ_LEN",
             "JF_STOP_TEXT_FULL", "JF_MB_INACTIVE", "JF_MB_KEEP", "JF_E_INVALID", "JF_E_CAPACITY", "JF_E_LAUNCH", "JF_E_SHAPE",
             "JF_EL_SEQ", "JF_EL_ERROR", "JF_EL_STEP_ERROR", "JF_EL_CURSORS", "JF_EL_HDR", "JF_EL_KIND_GREEDY", "JF_EL_KIND_SAMPLING",
             "JF_MB_LOOP_PUBLISH_FENCE"]
    src = tmp_path / "consts.c"
    src.write_text('#include <stdio.h>\n#include "jacobiforcing.h"\nint main(void) {\n' +
                   "".join(f'  printf("{n} %lld\\n", (long long)({n}));\n' for n in names) +
                   '  printf("MAILBOX_INTS_7 %lld\\n", (long long)JF_MB_MAILBOX_INTS(7));\n'
                   '  printf("PACKED_ENTRIES_100 %lld\\n", (long long)JF_MB_PACKED_ENTRIES(100));\n'
                   '  printf("LOOP_BYTES %lld\\n", (long long)sizeof(jf_mb_loop));\n'
                   '  printf("ENGINE_LOOP_BYTES %lld\\n", (long long)sizeof(jf_engine_loop));\n'
                   '  printf("EL_MAILBOX_INTS_7 %lld\\n", (long long)JF_EL_MAILBOX_INTS(7));\n  return 0;\n}\n')
    exe = tmp_path / "consts"
    subprocess.check_call(["gcc", "-std=c99", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)])
    c = {k: int(v) for k, v in (line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())}
    mb = ["SEQ", "RTOT", "RMAIN", "TPAD", "TMAX", "NVALID", "NVALID_PAD", "NDONE", "MAXKV", "ERROR", "ACCEPTED", "NCALL_END"]
    assert [c["JF_MB_" + k] for k in mb] == [getattr(N, "MB_" + k) for k in mb]
    assert (c["JF_MB_MAILBOX_HDR"], c["JF_MB_FIN_INTS"], c["JF_DRV_HDR_INTS"]) == (N.MB_MAILBOX_HDR, N.MB_FIN_INTS, N.DRV_HDR_INTS)
    assert c["MAILBOX_INTS_7"] == N.mailbox_ints(7) and c["PACKED_ENTRIES_100"] == 100 + N.MB_PACKED_EXTRA
    drv = ["ACTIVE", "STOP", "CALLS", "ITERS", "NEW", "BUDGET", "MAX_CALLS", "TEXT_LEN", "CURSOR", "FIN_RET_LEN", "FIN_NEXT", "FIN_ITERS",
           "FIN_OFF"]
    assert [c["JF_DRV_" + k] for k in drv] == list(range(len(N.DRV_FIELDS))) and len(drv) == len(N.DRV_FIELDS)
    assert sorted(c[k] for k in c if k.startswith("JF_STOP_")) == sorted(N.STOP_REASONS)
    assert (c["JF_MB_INACTIVE"], c["JF_MB_KEEP"]) == (N.JF_MB_INACTIVE, N.JF_MB_KEEP)
    assert c["LOOP_BYTES"] == ctypes.sizeof(N.MbLoop)
    assert c["ENGINE_LOOP_BYTES"] == ctypes.sizeof(N.EngineLoop) and c["EL_MAILBOX_INTS_7"] == N.EL_HDR + 7
    assert [c["JF_EL_" + k] for k in ("SEQ", "ERROR", "STEP_ERROR", "CURSORS", "HDR", "KIND_GREEDY", "KIND_SAMPLING")] == \
        [N.EL_SEQ, N.EL_ERROR, N.EL_STEP_ERROR, N.EL_CURSORS, N.EL_HDR, N.EL_KIND_GREEDY, N.EL_KIND_SAMPLING]
    assert c["JF_MB_LOOP_PUBLISH_FENCE"] == N.MB_LOOP_PUBLISH_FENCE


def _c_case_file(case, path):
    """One golden multiblock record (tests/golden/mb_cases*.json: calls of the unmodified reference) as the flat integer file the
    C client reads."""
    import math
    p = case["params"]
    V = case["model"]["vocab"] if "vocab" in case["model"] else case["model"]["V"]
    out = [p["n"], p["K"], int(math.ceil(p["r"] * p["n"])), p["pool"], -1 if p["eos_id"] is None else p["eos_id"],
           -1 if p["pad_id"] is None else p["pad_id"], p["max_iter"], 1 + p["max_iter"], V, len(case["calls"])]
    for call in case["calls"]:
        out += [call["kv_len_before"]] + list(call["input"]) + [len(call["forwards"])]
        for fw in call["forwards"]:
            B, T = len(fw["out"]), len(fw["out"][0])
            out += [B, T] + [t for row in fw["out"] for t in row] + [t for row in fw["greedy"] for t in row]
        nt = call["next_token"][0] if call["next_token"] else -1
        out += [len(call["ret"])] + list(call["ret"]) + [-1 if nt is None else nt, call["iters"], call["kv_len"]]
    path.write_text(" ".join(str(int(x)) for x in out))


@GPU
def test_plain_c_client_launches_kernels(tmp_path, mb_cases):
    """The same C99 client with -DJF_ABI_GPU: device buffers from the HIP runtime's C API, then jf_argmax_rows,
    jf_accept_lengths and jf_sb_step called from C and checked there — and the HOT PATH itself: golden records of the
    reference (BASELINE's knobs, a candidate-row case, a K = 3 case) driven through jf_mb_begin -> jf_mb_pack -> jf_mb_verify
    -> jf_mb_read_ret from C, every forward's rows and every call's ret / next_token / iters / kv_len compared in C.
    No Python, no torch between caller and kernels."""
    import __graft_entry__ as G
    lib = G.build_hip()
    exe = tmp_path / "abi_client_gpu"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-DJF_ABI_GPU", f"-I{ROOT / 'include'}", "-I/opt/rocm/include",
                           str(ROOT / "tests" / "abi" / "abi_client.c"), f"-L{lib.parent}", "-ljacobiforcing", "-L/opt/rocm/lib",
                           "-lamdhip64", f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    files = []
    for name in ("n32_K2_r85_default", "n32_period_candidates_pool8", "n16_K3_r50"):
        case = next(c for c in mb_cases if c["name"] == name)
        f = tmp_path / f"{name}.txt"
        _c_case_file(case, f)
        files.append(str(f))
    out = subprocess.run([str(exe)] + files, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu=ok" in out.stdout and "accepted=4" in out.stdout
    assert out.stdout.count("hot_path=ok") == 3, out.stdout


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(N.NativeLibraryError):
        N.load(tmp_path / "nope.so")


def test_struct_layouts_match_header():
    assert ctypes.sizeof(N.MbDesc) == 64 and ctypes.sizeof(N.EngineRow) == 32
    assert ctypes.sizeof(N.MbParams) == 40


# ------------------------------------------------------------------------------------- argmax
def _bits_to_tensor(case):
    bits = np.array(case["bits"], dtype=np.int64)
    if case["dtype"] == "float32":
        return torch.from_numpy((bits & 0xFFFFFFFF).astype(np.uint32).view(np.float32).copy())
    return torch.from_numpy((bits & 0xFFFF).astype(np.uint16).view(np.int16).copy()).view(torch.bfloat16)


@pytest.mark.parametrize("backend", BACKENDS)
def test_argmax_golden_vectors(kernel_vectors, backend):
    """torch.argmax corner cases recorded from torch in the build container: ties, NaN, +-inf, -0.0; V=97 exercises
    the unaligned (scalar) kernel."""
    with use_backend(backend):
        dev = device_for(backend)
        for case in kernel_vectors["argmax"]:
            x = _bits_to_tensor(case).to(dev)
            assert ops.argmax_rows(x).cpu().tolist() == case["argmax"], case["dtype"]


@GPU
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("R,V", [(1, 152064), (32, 152064), (7, 151936), (33, 4096), (5, 1000), (3, 1001), (2, 8),
                                 (300, 32000), (64, 152064)])
def test_argmax_vs_oracle(R, V, dtype):
    g = torch.Generator().manual_seed(R * 131 + V)
    x = torch.randn(R, V, generator=g)
    idx = torch.randint(0, V, (R,), generator=g)
    x[torch.arange(R), idx] = 7.0                       # planted max
    for r in range(0, R, 3):                            # planted exact ties (first index must win)
        j = int(torch.randint(0, V, (1,), generator=g))
        x[r, j] = 7.0
    if R > 2:
        x[2, V - 1] = float("nan")
    xd = x.to(dtype)
    got = ops.argmax_rows(xd.to("cuda")).cpu().numpy()
    ref = O.argmax_rows(xd.float().numpy())
    assert (got == ref).all()
    assert (got == torch.argmax(xd.float(), dim=-1).numpy()).all()


@GPU
@pytest.mark.parametrize("seed", range(24))
def test_argmax_randomized_shapes(seed):
    """Random row counts, vocabulary sizes (aligned and not), row strides, dtypes and special values (NaN, +-inf, -0.0,
    ties) against torch.argmax — covers the vector path, the scalar path, the NaN re-scan and both launch shapes."""
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    R = [1, 2, 3, 17, 64, 130, 257][ri(0, 6)]
    V = [8, 1001, 4096, 32000, 50257, 151936, 152064][ri(0, 6)]
    if R * V > 40_000_000:
        R = max(1, 40_000_000 // V)
    pad = [0, 0, 8, 3][ri(0, 3)]                              # row stride > V, possibly misaligned (scalar path)
    dtype = [torch.float32, torch.bfloat16][seed % 2]
    x = torch.randn(R, V + pad, generator=g)
    if seed % 3 == 0:
        x[:, :] = x[:, :].round()                              # many exact ties
    for _ in range(ri(0, 6)):
        r, c = ri(0, R - 1), ri(0, V - 1)
        x[r, c] = [float("nan"), float("inf"), float("-inf"), -0.0, 0.0, 1e30][ri(0, 5)]
    xd = x.to(dtype).cuda()
    view = xd[:, :V]
    got = ops.argmax_rows(view).cpu()
    ref = torch.argmax(view.float().cpu(), dim=-1)
    assert (got == ref).all(), (R, V, pad, dtype)


@GPU
def test_argmax_strided_rows_and_reuse_of_workspace():
    """logits[:, :-1] style views (row stride > V) and back-to-back launches on one workspace."""
    g = torch.Generator().manual_seed(5)
    big = torch.randn(6, 40, 2048, generator=g).to(torch.bfloat16).cuda()
    packed = ops.new_packed(6 * 40, "cuda")
    for _ in range(3):
        view = big[:, :-1, :]                             # MR:1416
        for b in range(6):
            got = ops.argmax_rows(view[b], packed)
            assert (got.cpu() == torch.argmax(view[b].float().cpu(), dim=-1)).all()
    assert int(packed.abs().sum()) == 0                   # consumers re-zero the workspace


@GPU
def test_argmax_full_size_property():
    """BASELINE full size (config 4 per-GPU shape: 512 rows x 152064): checksum-style properties instead of a
    slow CPU pass — planted maxima are found, and the result is invariant under a row permutation."""
    R, V = 512, 152064
    x = torch.randn(R, V, device="cuda", dtype=torch.bfloat16)
    idx = torch.randint(0, V, (R,), device="cuda")
    x[torch.arange(R, device="cuda"), idx] = 30.0
    got = ops.argmax_rows(x)
    assert (got == idx).all()
    perm = torch.randperm(R, device="cuda")
    assert (ops.argmax_rows(x[perm].contiguous()) == idx[perm]).all()
    assert (got == torch.argmax(x.float(), dim=-1)).all()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("R,V,slots", [(37, 4096, 64), (9, 1001, 9), (200, 32000, 512), (168, 152064, 192)])
def test_argmax_scatter_matches_torch(R, V, slots, dtype, backend):
    """jf_argmax_scatter: row i of the compacted logits lands in packed[out_index[i]]; rows with a negative index
    (list padding) are skipped, untouched slots stay zero."""
    if backend == "hostsim" and R * V > 4_000_000:
        pytest.skip("GPU-size case")
    with use_backend(backend):
        dev = device_for(backend)
        g = torch.Generator().manual_seed(R + V)
        x = torch.randn(R, V, generator=g).to(dtype)
        x[1, 5] = x[1, 900] = 9.0                              # tie: first index wins
        perm = torch.randperm(slots, generator=g)[:R].to(torch.int32)
        perm[R // 2] = -1
        perm[R - 1] = -1
        packed = ops.new_packed(slots, dev)
        ops.argmax_scatter(x.to(dev), perm.to(dev), packed)
        got = packed.cpu().numpy().astype(np.uint64)
        ref = torch.argmax(x.float(), dim=-1).numpy()
        used = np.zeros(slots, dtype=bool)
        for i in range(R):
            if perm[i] >= 0:
                assert int((~got[perm[i]]) & np.uint64(0xFFFFFFFF)) == ref[i], i
                used[int(perm[i])] = True
        assert (got[~used] == 0).all()


def test_argmax_rejects_bad_input():
    with use_backend("hostsim"):
        with pytest.raises(ValueError):
            ops.argmax_rows(torch.zeros(2, 8, dtype=torch.float16))
        with pytest.raises(ValueError):
            ops.argmax_partial(torch.zeros(2, 8).t(), ops.new_packed(8, "cpu"))


# ------------------------------------------------------------------------------------- accept scan
@pytest.mark.parametrize("backend", BACKENDS)
def test_accept_lengths_golden(kernel_vectors, backend):
    with use_backend(backend):
        dev = device_for(backend)
        for c in kernel_vectors["accept"]:
            d = torch.tensor(c["draft"], dtype=torch.int64, device=dev)
            g = torch.tensor(c["greedy"], dtype=torch.int64, device=dev)
            acc, best = ops.accept_lengths(d, g)
            assert acc.cpu().tolist() == c["accepted"]
            assert int(best.cpu()) == c["best_idx"]


@GPU
def test_accept_lengths_long_rows_and_broadcast():
    rng = np.random.default_rng(3)
    for L in (2, 64, 65, 129, 500):
        B = 5
        g = rng.integers(0, 3, size=(B, L))
        d = np.concatenate([rng.integers(0, 3, size=(1, 1)), g[:1, :-1]], axis=1)   # row 0 matches itself fully
        d = np.repeat(d, 1, axis=0)
        cut = rng.integers(0, L, size=B)
        acc, best = ops.accept_lengths(torch.tensor(d).cuda(), torch.tensor(g).cuda())
        ref = O.accept_lengths(d.tolist(), g.tolist())
        assert acc.cpu().tolist() == ref and int(best.cpu()) == O.first_max_index(ref)


# ------------------------------------------------------------------------------------- KV cache
@GPU
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_kv_append_matches_index_copy(dtype):
    rows, H, S, D, Ntok = 3, 4, 50, 128, 37
    g = torch.Generator().manual_seed(1)
    kc = torch.zeros(rows, H, S, D, dtype=dtype, device="cuda")
    vc = torch.zeros_like(kc)
    kn = torch.randn(Ntok, H, D, generator=g).to(dtype).cuda()
    vn = torch.randn(Ntok, H, D, generator=g).to(dtype).cuda()
    slot = torch.randperm(rows * S, generator=g)[:Ntok].to(torch.int64)
    slot[5] = -1
    ops.kv_append(kc, vc, kn, vn, slot.cuda())
    rk, rv = torch.zeros_like(kc), torch.zeros_like(vc)
    for i, s in enumerate(slot.tolist()):
        if s < 0:
            continue
        rk[s // S, :, s % S, :] = kn[i]
        rv[s // S, :, s % S, :] = vn[i]
    assert torch.equal(kc, rk) and torch.equal(vc, rv)


@GPU
def test_kv_commit_copies_candidate_rows():
    P, H, S, D, T, CR, layers = 3, 4, 64, 128, 16, 3, 2
    g = torch.Generator().manual_seed(2)
    mk = [torch.randn(P, H, S, D, generator=g).to(torch.bfloat16).cuda() for _ in range(layers)]
    mv = [torch.randn(P, H, S, D, generator=g).to(torch.bfloat16).cuda() for _ in range(layers)]
    ck = [torch.randn(P * CR, H, T, D, generator=g).to(torch.bfloat16).cuda() for _ in range(layers)]
    cv = [torch.randn(P * CR, H, T, D, generator=g).to(torch.bfloat16).cuda() for _ in range(layers)]
    ref_k, ref_v = [t.clone() for t in mk], [t.clone() for t in mv]
    desc = torch.zeros(P, N.DESC_INTS, dtype=torch.int32)
    f = N.DESC_FIELDS.index
    plan = {0: (2, 10, 5), 2: (1, 33, 16)}            # prompt -> (src_row, dst, len); prompt 1 copies nothing
    for p, (src, dst, ln) in plan.items():
        desc[p, f("kv_src_row")], desc[p, f("kv_copy_dst")], desc[p, f("kv_copy_len")] = src, dst, ln
        for l in range(layers):
            ref_k[l][p, :, dst:dst + ln] = ck[l][p * CR + src - 1, :, :ln]
            ref_v[l][p, :, dst:dst + ln] = cv[l][p * CR + src - 1, :, :ln]
    ops.KVCommitter(mk, mv, ck, cv, CR).commit(desc.cuda())
    for l in range(layers):
        assert torch.equal(mk[l], ref_k[l]) and torch.equal(mv[l], ref_v[l])


def _special_value_rows(dtype, V=40960):
    """Rows with -NaN / +NaN payloads, -0.0 vs +0.0 ties, all-negative and all -inf rows, ties across a chunk edge, +inf."""
    g = torch.Generator().manual_seed(9)

    def base(neg=False):
        x = torch.randn(V, generator=g)
        return -x.abs() - 0.5 if neg else x

    def bits(x):
        return x.view(torch.int32) if dtype == torch.float32 else x.view(torch.int16)

    def set_bits(x, pos, payload32):
        b = bits(x)
        b[pos] = payload32 if dtype == torch.float32 else (payload32 >> 16) - (0x10000 if (payload32 >> 16) >= 0x8000 else 0)
    cases = []
    x = base().to(dtype); set_bits(x, 20000, 0xFFC00000 - (1 << 32)); cases.append(x)                 # -NaN only
    x = base().to(dtype); set_bits(x, 30001, 0xFFC00000 - (1 << 32)); set_bits(x, 111, 0x7FC10000); cases.append(x)   # +NaN first
    x = base().to(dtype); set_bits(x, 5, 0xFFC00000 - (1 << 32)); set_bits(x, 4000, 0x7FFF0000); cases.append(x)      # -NaN first
    x = base(True).to(dtype); x[777] = -0.0; x[20000] = 0.0; cases.append(x)                          # -0 before +0
    x = base(True).to(dtype); x[20000] = -0.0; x[777] = 0.0; cases.append(x)                          # +0 before -0
    x = base(True).to(dtype); x[12345] = -0.0; cases.append(x)                                        # only -0
    x = base(True).to(dtype); cases.append(x)                                                         # all negative
    x = torch.full((V,), -float("inf")).to(dtype); cases.append(x)                                    # all -inf -> 0
    x = torch.full((V,), -float("inf")).to(dtype); x[V - 1] = -3.0e38; cases.append(x)                # last element
    x = base().to(dtype); x[8191] = 50.0; x[8192] = 50.0; cases.append(x)                             # tie across chunk edge
    x = base().to(dtype); x[3] = float("inf"); x[V - 3] = float("inf"); cases.append(x)
    return torch.stack(cases)


@GPU
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_argmax_special_values_vector_path(dtype):
    """-NaN / +NaN with payloads, -0.0 vs +0.0 ties, all-negative and all -inf rows on the 16-byte vector path
    (V multiple of 8, several chunks) against torch.argmax on the CPU."""
    V = 40960
    X = _special_value_rows(dtype, V)
    ref = torch.argmax(X.float(), dim=-1)
    got = ops.argmax_rows(X.cuda()).cpu()
    assert got.tolist() == ref.tolist()
    # a non-contiguous row stride keeps rows 16-byte aligned as well
    Y = torch.zeros(X.shape[0], V + 64, dtype=dtype)
    Y[:, :V] = X
    assert ops.argmax_rows(Y.cuda()[:, :V]).cpu().tolist() == ref.tolist()


# ------------------------------------------------------------------------------------- non-greedy softmax-gather
def _run_rs_probs(x, dn, temperature):
    R, V = x.shape
    st = ops.RsStepper(4, 16, "cuda", [0], [0.5], [0.5])
    ws = torch.zeros((int(N.lib().jf_rs_workspace_bytes(R, V)) // 4 + 4,), dtype=torch.float32, device="cuda")
    xd = x.cuda()
    p = torch.zeros(R, device="cuda"); m = torch.zeros(R, device="cuda"); s = torch.zeros(R, device="cuda")
    packed = ops.new_packed(R, "cuda")
    N.check(N.lib().jf_rs_probs(ops._ptr(xd), ops._dtype_code(xd), R, V, xd.stride(0), ops._ptr(dn.cuda()), temperature, ops._ptr(p),
                                ops._ptr(m), ops._ptr(s), ops._ptr(packed), ops._ptr(ws), ws.numel() * 4, ops._stream(xd.device)))
    del st
    # bf16 logits: the sign bit of p_draft marks "the float32 row sum cannot decide the rounding" (the value is then the lower
    # candidate): the probability itself is the magnitude
    return p.cpu().abs(), m.cpu(), s.cpu(), (~packed.cpu()) & 0xFFFFFFFF


@GPU
@pytest.mark.parametrize("temperature", [1.0, 0.7])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_rs_probs_argmax_special_values(dtype, temperature):
    """The greedy token jf_rs_probs records next to the probabilities follows torch.argmax on the same corner cases as the
    argmax kernel (its NaN detection is different: a negative NaN is seen by the exp sum, not by a running minimum), and the
    gathered probability is NaN exactly where torch's softmax is."""
    X = _special_value_rows(dtype)
    X[8, -1] = -3.0e4                 # the kernel forms exp2(fma(x, log2(e)/T, -max)): exact up to |x| ~ 5e8, not at 3e38
    dn = torch.full((X.shape[0],), 17, dtype=torch.int64)
    p, m, s, greedy = _run_rs_probs(X, dn, temperature)
    assert greedy.tolist() == torch.argmax(X.float(), dim=-1).tolist()
    ref = torch.softmax(X.float() / temperature, dim=-1)[:, 17]
    assert torch.isnan(p).tolist() == torch.isnan(ref).tolist()


@GPU
@pytest.mark.parametrize("temperature", [1.0, 0.7])
def test_rs_probs_vs_torch_softmax(temperature):
    """float32 logits: p_draft = softmax(logits / T)[draft] (JDN:65-70, 328) in fp32: tolerance 2e-5 relative (fp32 exp / sum
    order), argmax bit-exact."""
    R, V = 31, 152064
    g = torch.Generator().manual_seed(4)
    x = torch.randn(R, V, generator=g) * 3
    dn = torch.randint(0, V, (R,), generator=g)
    dn[0] = int(torch.argmax(x[0]))
    got, _, _, am = _run_rs_probs(x, dn, temperature)
    want = torch.softmax(x / temperature, dim=-1)[torch.arange(R), dn]
    assert torch.allclose(got, want, rtol=2e-5, atol=1e-12), float(((got - want).abs() / want).max())
    assert am.tolist() == torch.argmax(x, dim=-1).tolist()


def _bf16_bits(t):
    return t.to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF


@GPU
@pytest.mark.parametrize("temperature", [1.0, 0.7, 1.3, 0.25])
def test_rs_probs_bf16_follows_torch_rounding_points(temperature):
    """bfloat16 logits: the reference's probs tensor is torch.softmax(logits / T) IN bf16 (JDN:64-70 on MR:1382's logits).
    torch (CPU, in this test) is the reference: the scaled row maximum must agree bit for bit (one rounding of a float32
    quotient), p_draft must be a bf16 value within one bf16 ulp of torch's, and equal to it for all but a few rows (float32
    softmax internals differ in the last place between any two implementations, torch's own CPU kernels included)."""
    R, V = 248, 152064
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(R, V, generator=g) * 3).to(torch.bfloat16)
    dn = torch.randint(0, V, (R,), generator=g)
    am_ref = torch.argmax(x.float(), dim=-1)
    dn[: R // 2] = am_ref[: R // 2]                                     # half the rows ask for the heaviest token
    got, m, s, am = _run_rs_probs(x, dn, temperature)
    scaled = x if temperature == 1.0 else x / float(temperature)        # bf16 tensor, JDN:66-69
    want = torch.softmax(scaled, dim=-1)[torch.arange(R), dn]
    assert want.dtype == torch.bfloat16
    assert torch.equal(m, scaled.float().max(dim=-1).values)             # exact scaling, exact max
    assert torch.equal(got.to(torch.bfloat16).float(), got)              # a bf16 value held in a float
    d = (_bf16_bits(got) - _bf16_bits(want.float())).abs()
    assert int(d.max()) <= 1
    assert float((d != 0).float().mean()) < 0.03, d.nonzero().flatten().tolist()
    assert am.tolist() == am_ref.tolist()                                # next-draft argmax is over the RAW logits
    # the float32 row sum against an exact (float64) one
    sx = torch.exp(scaled.double() - scaled.double().max(dim=-1, keepdim=True).values).sum(-1)
    assert torch.allclose(s.double(), sx, rtol=2e-5)


# ------------------------------------------------------------------------------------- fused RoPE + Q layout + KV append
@GPU
@pytest.mark.parametrize("D", [32, 128, 24])          # 24: half a head is 12 elements — bf16 takes the element-wise kernel
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_rope_kv_append_matches_torch(dtype, D):
    R, T, nq, nkv, S_max, T_max = 3, 5, 8, 2, 40, 8
    g = torch.Generator().manual_seed(6)
    qkv = torch.randn(R * T, (nq + 2 * nkv) * D, generator=g).to(dtype).cuda()
    pos = torch.randint(0, 30, (R * T,), generator=g, dtype=torch.int32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.outer(torch.arange(64, dtype=torch.float32), inv)
    cos, sin = fr.cos().cuda(), fr.sin().cuda()
    kc = torch.zeros(2, nkv, S_max, D, dtype=dtype, device="cuda"); vc = torch.zeros_like(kc)
    ck = torch.zeros(3, nkv, T_max, D, dtype=dtype, device="cuda"); cv = torch.zeros_like(ck)
    slot_main = torch.full((R * T,), -1, dtype=torch.int64); slot_cand = torch.full((R * T,), -1, dtype=torch.int64)
    slot_main[:T] = 1 * S_max + 7 + torch.arange(T)                 # row 0 of "prompt 1" appends at 7..
    slot_cand[T:2 * T] = 2 * T_max + torch.arange(T)                # a candidate row writes the scratch
    q = ops.rope_kv_append(qkv, T, nq, nkv, D, pos.cuda(), cos, sin, kc, vc, slot_main.cuda(), ck, cv, slot_cand.cuda())
    x = qkv.float().cpu().view(R * T, nq + 2 * nkv, D)
    c, s_ = fr.cos()[pos.long()][:, None, :], fr.sin()[pos.long()][:, None, :]
    x1, x2 = x[..., :D // 2], x[..., D // 2:]
    rot = torch.cat([x1 * c - x2 * s_, x2 * c + x1 * s_], -1)
    G = nq // nkv
    q_ref = rot[:, :nq].view(R, T, nkv, G, D).permute(0, 2, 3, 1, 4).reshape(R, nkv, G * T, D).to(dtype)
    tol = dict(atol=2e-6, rtol=2e-6) if dtype == torch.float32 else dict(atol=1e-2, rtol=1e-2)   # fma contraction on the GPU
    assert torch.allclose(q.cpu().float(), q_ref.float(), **tol)
    k_ref, v_ref = rot[:, nq:nq + nkv].to(dtype), x[:, nq + nkv:].to(dtype)
    for t in range(T):
        assert torch.allclose(kc[1, :, 7 + t].cpu().float(), k_ref[t].float(), **tol)
        assert torch.equal(vc[1, :, 7 + t].cpu(), v_ref[t])
        assert torch.allclose(ck[2, :, t].cpu().float(), k_ref[T + t].float(), **tol)
        assert torch.equal(cv[2, :, t].cpu(), v_ref[T + t])
    assert float(kc[0].abs().sum()) == 0 and float(ck[:2].abs().sum()) == 0


@GPU
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_swiglu_matches_torch(dtype):
    g = torch.Generator().manual_seed(8)
    gu = (torch.randn(37, 2 * 1024, generator=g) * 2).to(dtype).cuda()
    out = ops.swiglu(gu)
    a, b = gu.float().chunk(2, dim=-1)
    ref = (torch.nn.functional.silu(a) * b)
    tol = dict(atol=1e-5, rtol=1e-5) if dtype == torch.float32 else dict(atol=2e-2, rtol=2e-2)
    assert torch.allclose(out.float(), ref, **tol)


# ------------------------------------------------------------------------------------- non-greedy step at batch scale
def _rs_case(B, L, V, seed, p_hit, u_value):
    """Logits whose softmax puts ~p_hit on the proposed token of every position; uniforms pinned to u_value (None: random
    24-bit uniforms)."""
    g = torch.Generator().manual_seed(seed)
    draft = torch.randint(0, V, (B, L), generator=g)
    logits = torch.randn(B, L - 1, V, generator=g) * 0.3
    boost = float(np.log(p_hit / (1 - p_hit) * (V - 1)))                 # logit gap that gives the proposed id mass ~p_hit
    logits.scatter_(2, draft[:, 1:].unsqueeze(-1), boost)
    n = 4 * B * L
    unis = torch.full((n,), u_value) if u_value is not None else torch.randint(0, 1 << 24, (n,), generator=g).float() / float(1 << 24)
    bonus = torch.randint(0, 1 << 24, (n,), generator=g).float() / float(1 << 24)
    pads = torch.randint(0, V, (n,), generator=g)
    return draft, logits, unis, bonus, pads


def _run_rs(backend, B, L, V, seed, p_hit, u_value, dtype, temperature=1.0, eos=None):
    with use_backend(backend):
        dev = device_for(backend)
        draft, logits, unis, bonus, pads = _rs_case(B, L, V, seed, p_hit, u_value)
        st = ops.RsStepper(B, L, dev, pads, unis, bonus)
        rows, toks, nd = st.step(draft.to(dev), logits.to(dtype).to(dev), temperature, eos, [L] * B, [3, 5, 7])
        return rows.copy(), toks.copy(), nd.cpu().numpy().copy(), st.cursors.cpu().tolist()


def _assert_rs_equal(a, b, B):
    f = N.RS_FIELDS.index
    assert (a[0] == b[0]).all(), (a[0].tolist(), b[0].tolist())
    for r in range(B):
        n = int(b[0][r, f("n_committed")])
        assert (a[1][r, :n] == b[1][r, :n]).all(), r
        if b[0][r, f("active_next")]:
            assert (a[2][r] == b[2][r]).all(), r
    assert a[3] == b[3]


@GPU
@pytest.mark.parametrize("p_hit,u_value", [(0.6, 0.95), (0.9, 0.95), (0.5, 0.3)], ids=["collisions", "many_collisions", "mixed"])
def test_rs_step_batch_matches_sequential_reference(p_hit, u_value):
    """jf_rs_step draws the bonus tokens of all rejected rows in parallel at ASSUMED stream positions and repairs the rows
    behind a row that needed several draws (its sample hit the proposed id).  Forced rejections with a heavy proposed id make
    such collisions frequent; results must equal the row-by-row restatement (JDN:581-639) incl. every stream cursor."""
    B, L, V = 48, 9, 2000
    a = _run_rs("hip", B, L, V, 11, p_hit, u_value, torch.float32)
    b = _run_rs("hostsim", B, L, V, 11, p_hit, u_value, torch.float32)
    f = N.RS_FIELDS.index
    assert (a[0][:, f("reject_pos")] == b[0][:, f("reject_pos")]).all()
    assert (a[0][:, f("n_bonus_draws")] == b[0][:, f("n_bonus_draws")]).all()
    if p_hit >= 0.6:
        assert (b[0][:, f("n_bonus_draws")] > 1).sum() >= 5          # the repair path really ran
    _assert_rs_equal(a, b, B)


@GPU
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("B,L,V", [(6, 5, 2048), (6, 5, 2049), (5, 9, 5000), (7, 5, 20000), (4, 9, 40000), (3, 5, 32768), (128, 9, 3000),
                                   (1, 2, 152064), (100, 5, 152064)],
                         ids=["one_segment", "two_segments", "three", "ten_bf16", "sixteen_f32", "sixteen_full", "128_rows", "one_row_one_test",
                              "more_workgroups_than_resident"])
def test_rs_step_segment_counts_and_batches(B, L, V, dtype):
    """The one-launch step gives a row one workgroup per NON-EMPTY vocabulary segment (1 ... 16 of them, by V and dtype), the
    last one standing in for the empty ones, and segment 0's workgroup draws the row's bonus token: every count of active
    segments, a batch of the launch's maximum of 128 rows, the smallest call (one row, one test), and 100 rows at the real
    vocabulary (3 + 1 500 workgroups for 1 024 resident slots: waits only point backwards in dispatch order), against the
    row-by-row restatement."""
    a = _run_rs("hip", B, L, V, 31 + V % 7, 0.5, None, dtype, 0.9, eos=3)
    b = _run_rs("hostsim", B, L, V, 31 + V % 7, 0.5, None, dtype, 0.9, eos=3)
    f = N.RS_FIELDS.index
    assert B < 3 or (b[0][:, f("reject_pos")] >= 0).sum() >= 1
    _assert_rs_equal(a, b, B)


@GPU
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("temperature,p_hit,u_value", [(1.0, 0.6, None), (0.8, 0.5, None), (1.3, 0.7, 0.9)],
                         ids=["T1", "T08", "T13_collisions"])
def test_rs_step_at_the_real_vocabulary(dtype, temperature, p_hit, u_value):
    """jf_rs_probs + jf_rs_step at V = 152064 (Qwen2.5's lm_head rows) against the row-by-row oracle restatement
    (JDN:299-354, 581-639; tests/backends.py) in both dtypes: accepted counts, rejected positions, bonus tokens of the
    two-level float64 inverse-CDF walk, draws, next drafts, every stream cursor.  bf16 runs torch's rounding points."""
    B, L, V = 10, 9, 152064
    a = _run_rs("hip", B, L, V, 21, p_hit, u_value, dtype, temperature, eos=7)
    b = _run_rs("hostsim", B, L, V, 21, p_hit, u_value, dtype, temperature, eos=7)
    f = N.RS_FIELDS.index
    assert (b[0][:, f("reject_pos")] >= 0).sum() >= 3                    # rejections (bonus draws) really happen
    _assert_rs_equal(a, b, B)


@GPU
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("temperature", [1.0, 0.7])
def test_rs_onpolicy_step_at_the_real_vocabulary(dtype, temperature):
    """jf_rs_probs + jf_rs_onpolicy_step at V = 152064 against the oracle restatement (JDO:270-327, 465-477): committed
    tokens, bonus draws, the re-drafted rows (one inverse-CDF draw each) and both cursors."""
    R, V = 12, 152064
    g = torch.Generator().manual_seed(31)
    proposed = torch.randint(0, V, (R,), generator=g)
    logits = torch.randn(R, V, generator=g) * 0.5
    boost = float(np.log(0.8 / 0.2 * (V - 1)))
    logits[torch.arange(R), proposed] = boost
    logits[4, proposed[4]] = 0.0                                         # position 4 is (almost surely) rejected
    unis = torch.randint(0, 1 << 24, (64,), generator=g).float() / float(1 << 24)
    unis[2:6] = 0.01                                                     # positions 0..3 are accepted (the cursor starts at 2)
    multi = torch.randint(0, 1 << 24, (64,), generator=g).float() / float(1 << 24)
    out = {}
    for backend in ("hip", "hostsim"):
        with use_backend(backend):
            dev = device_for(backend)
            st = ops.OnPolicyStepper(16, dev, unis, multi, [3])
            out[backend] = st.step(proposed.to(dev), logits.to(dtype).to(dev), temperature, [2, 5]) + (st.cursors.cpu().tolist(),)
    row_h, cm_h, rd_h, cur_h = out["hip"]
    row_o, cm_o, rd_o, cur_o = out["hostsim"]
    assert row_o["reject_pos"] == 4 and row_o["n_redraft"] == R - 5
    assert row_h == row_o and cm_h == cm_o and cur_h == cur_o
    n = row_o["n_committed"]
    assert rd_h[n:] == rd_o[n:]


@GPU
def test_rs_step_batch64_block32_planted_mass():
    """BASELINE config 5's shape: 64 rows x block 32 x V = 152064, bf16 logits.  Two ids per position carry ~all the mass
    (the proposed one ~0.5, a planted alternative ~0.5), the uniform is pinned above 0.5: every row rejects at position 0 and
    the residual draw must return the planted id (collisions with the proposed id are re-drawn, JDN:135-146; the rows behind
    a collision are repaired).  Size-independent properties: one committed token per row, cursors add up."""
    B, L, V = 64, 32, 152064
    g = torch.Generator(device="cuda").manual_seed(3)
    logits = torch.randn(B, L - 1, V, generator=g, device="cuda").to(torch.bfloat16)
    draft = torch.randint(0, V, (B, L), generator=g, device="cuda")
    alt = (draft[:, 1:] + 1 + torch.randint(0, V - 2, (B, L - 1), generator=g, device="cuda")) % V
    assert bool((alt != draft[:, 1:]).all())
    logits.scatter_(2, draft[:, 1:].unsqueeze(-1), 22.0)
    logits.scatter_(2, alt.unsqueeze(-1), 22.0)
    n = 4 * B
    unis = torch.full((n,), 0.75)
    bonus = torch.rand(n, generator=torch.Generator().manual_seed(9))
    pads = torch.randint(0, V, (4 * B * L,))
    st = ops.RsStepper(B, L, "cuda", pads, unis, bonus)
    rows, toks, nd = st.step(draft, logits, 1.0, None, [L] * B, [0, 0, 0])
    f = N.RS_FIELDS.index
    assert (rows[:, f("reject_pos")] == 0).all() and (rows[:, f("n_committed")] == 1).all()
    assert (rows[:, f("n_uniforms")] == 1).all()
    want = alt[:, 0].cpu().numpy()
    assert (toks[:, 0] == want).mean() >= 0.98                            # the rest of the vocabulary holds ~1e-3 of the mass
    assert (toks[:, 0] != draft[:, 1].cpu().numpy()).all()                # a bonus never equals the rejected proposal
    draws = rows[:, f("n_bonus_draws")]
    assert draws.min() >= 1 and draws.max() <= 16 and 1.3 < draws.mean() < 3.0   # geometric with p ~ 0.5
    assert st.cursors.cpu().tolist() == [B, int(draws.sum()), int(rows[:, f("n_pads")].sum())]
    # next draft: seed = the bonus, then the greedy tail (argmax of rows 1.. = proposed or alt: a tie -> lower id), then pads
    lo = torch.minimum(draft[:, 1:], alt).cpu().numpy()
    ndc = nd.cpu().numpy()
    assert (ndc[:, 0] == toks[:, 0]).all()
    assert (ndc[:, 1:L - 1] == lo[:, 1:]).all()


@GPU
@pytest.mark.parametrize("B,L,V", [(100, 65, 64), (1100, 4, 40), (300, 9, 50)], ids=["rows_x_block_over_lds", "batch_over_lds", "many_rejections"])
def test_rs_step_beyond_the_lds_tables(B, L, V):
    """Batches whose accept tests / row tables / bonus-stream window do not fit the LDS staging of rs_accept_kernel and
    rs_chain_kernel take the global-memory variants of the same serial walks: results must not change."""
    a = _run_rs("hip", B, L, V, 5, 0.5, None, torch.float32, 0.9, eos=3)
    b = _run_rs("hostsim", B, L, V, 5, 0.5, None, torch.float32, 0.9, eos=3)
    _assert_rs_equal(a, b, B)


# ------------------------------------------------------------------------------------- top-k / top-p (jf_rs_filter)
def _filter_rows(x: torch.Tensor, temperature: float, top_k: int, top_p: float) -> np.ndarray:
    """jf_rs_probs + jf_rs_filter on [R, V] logits: the per-row records, expanded (jf_rs_filter_expand) to the probability tensor the
    reference's _build_target_probs returns, as float32 values.  Checked on the way: p_draft is the final value of the drafted id,
    the rows' statistics are marked, and the records are self-consistent (the stages that are on, cuts inside [0, 1])."""
    R, V = x.shape
    dn = torch.arange(R, dtype=torch.int64, device=x.device) % V
    probs, p, rec = ops.filtered_probs(x, temperature, top_k, top_p, dn)
    q = probs.float().cpu().numpy()
    assert np.array_equal(p.cpu().numpy(), q[np.arange(R), dn.cpu().numpy()])                        # p_draft = the final value
    r = np.frombuffer(rec.cpu().numpy().tobytes(), dtype=np.dtype([("sum", "<f8"), ("row_max", "<f4"), ("x_keep", "<f4"), ("cut1", "<u4"),
                                                                   ("tie1", "<i4"), ("s1", "<f4"), ("cut2", "<u4"), ("tie2", "<i4"),
                                                                   ("s2", "<f4"), ("flags", "<u4"), ("rsv", "<u4")]))
    want_flags = (1 if 0 < int(top_k) < V else 0) | (2 if 0.0 < float(top_p) < 1.0 else 0)
    assert (r["flags"] == want_flags).all() and (r["cut1"] <= 0x3F800000).all() and (r["cut2"] <= 0x3F800000).all()
    assert ((r["tie1"] >= -1) & (r["tie1"] < V) & (r["tie2"] >= -1) & (r["tie2"] < V)).all()
    fin = np.isfinite(x.float().cpu().numpy()).all(-1) | True
    assert (r["sum"][fin] >= 0).all()
    return q


FLT = load_golden("filter_vectors.json")


@GPU
@pytest.mark.parametrize("case", FLT[::3], ids=[f"V{c['V']}_T{c['temperature']}_k{c['top_k']}_p{c['top_p']}_{i}" for i, c in enumerate(FLT)][::3])
def test_rs_filter_reproduces_the_references_tensors(case):
    """jf_rs_filter against _build_target_probs of the unmodified reference with top_k / top_p planted (tests/golden/
    filter_vectors.json): bf16 bit for bit in every row whose cuts fall between DIFFERENT probabilities (rows with a cut inside
    a group of equal values: the same number of survivors with the same values — which ids survive there is torch's kernel's
    choice, lowest id first here); float32: the same kept set, values within a few ulps; and exactly the oracle's definition."""
    V, T, k, tp = case["V"], case["temperature"], case["top_k"], case["top_p"]
    xb = torch.from_numpy(np.array(case["logits_bf16"], dtype=np.uint16).reshape(-1, V).view(np.int16)).view(torch.bfloat16)
    want_b = O.bf16_bits_to_f32(np.array(case["probs_bf16"], dtype=np.uint16).reshape(-1, V))
    want_f = np.array(case["probs_f32_of_f32_logits"], dtype=np.uint32).reshape(-1, V).view(np.float32)
    got_b = _filter_rows(xb.cuda(), T, k or 0, tp or 0.0)
    got_f = _filter_rows(xb.float().cuda(), T, k or 0, tp or 0.0)
    x = xb.float().numpy()
    assert np.array_equal(got_b, O.target_probs(x, T, "bf16", k, tp))
    assert np.array_equal(got_f, O.target_probs(x, T, "f32", k, tp))
    pb = O.target_probs(x, T, "bf16")
    for r in range(x.shape[0]):
        if O.filter_boundary_is_tied(pb[r], k, tp, 7):
            assert np.array_equal(np.sort(got_b[r]), np.sort(want_b[r]))
        else:
            assert np.array_equal(got_b[r], want_b[r]), r
        a, b = np.sort(got_f[r]), np.sort(want_f[r])
        assert (a > 0).sum() == (b > 0).sum()
        assert np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)).max() <= 16


@GPU
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("V,top_k,top_p,scale", [(152064, 50, 0.0, 3.0), (152064, 0, 0.9, 3.0), (152064, 40, 0.95, 3.0), (152064, 0, 0.9, 0.3),
                                                 (152064, 1000, 0.5, 0.3), (4099, 1, 0.0, 2.0), (4099, 4098, 0.999, 2.0), (5, 2, 0.6, 1.0)],
                         ids=["k50", "p09", "k40_p095", "flat_p09", "flat_k1000_p05", "k1", "all_but_one", "tiny"])
def test_rs_filter_at_the_real_vocabulary(dtype, V, top_k, top_p, scale):
    """jf_rs_filter at V = 152064 (and a ragged V, and 5 ids) against the oracle's filter_probs_row in both dtypes: peaked rows
    (N(0, 3^2) logits: cuts between distinct values) and flat rows (N(0, 0.3^2): in bf16 thousands of ids share the value at a
    cut — the first ones by id survive on both sides), T = 0.8.  Bit for bit."""
    g = torch.Generator().manual_seed(V + top_k)
    x = (torch.randn(3, V, generator=g) * scale).to(dtype)
    x[1, V // 2] = 12.0
    got = _filter_rows(x.cuda(), 0.8, top_k, top_p)
    want = O.target_probs(x.float().numpy(), 0.8, "bf16" if dtype == torch.bfloat16 else "f32", top_k or None, top_p or None)
    assert np.array_equal(got, want), [(int((got[r] != want[r]).sum()), int((got[r] > 0).sum()), int((want[r] > 0).sum())) for r in range(3)]
    if top_k:
        assert ((got > 0).sum(-1) <= top_k).all()
    assert np.allclose(got.astype(np.float64).sum(-1), 1.0, atol=3e-2 if dtype == torch.bfloat16 else 1e-5)


@GPU
@pytest.mark.parametrize("top_k,top_p", [(50, 0.0), (0, 0.9), (40, 0.95), (70000, 0.5)], ids=["k50", "p09", "k40_p095", "k70000_p05"])
@pytest.mark.parametrize("shape", ["equal_logits", "two_values", "many_patterns", "one_heavy_pattern"])
def test_rs_filter_bf16_rows_that_leave_the_fast_path(shape, top_k, top_p):
    """The bf16 filter keeps 16-bit counts of the row's scaled-logit patterns and solves on a list of at most 4 096 occupied patterns.
    Rows that do not fit: a pattern that occurs more than 65 535 times (all logits equal: one pattern 152 064 times — the plain adds
    wrap, the checksum notices, the row is counted again with 32-bit side counters; two patterns of 76 032 ids each; one heavy pattern
    among random ones) and rows with more than 4 096 occupied patterns (log-uniform magnitudes: the solver on the counters themselves).
    Bit for bit against the oracle, T = 0.8."""
    V = 152064
    g = torch.Generator().manual_seed(len(shape) * 100 + top_k)
    if shape == "equal_logits":
        x = torch.zeros(2, V)
        x[1] = 3.25
    elif shape == "two_values":
        x = torch.where(torch.arange(V) % 2 == 0, torch.tensor(1.0), torch.tensor(0.5)).repeat(2, 1)
        x[1] = x[1].flip(0)
    elif shape == "many_patterns":
        x = torch.sign(torch.randn(2, V, generator=g)) * torch.pow(2.0, torch.rand(2, V, generator=g) * 28.0 - 24.0)
    else:
        x = torch.randn(2, V, generator=g)
        x[:, torch.randperm(V, generator=g)[:70000]] = 0.75
    x = x.to(torch.bfloat16)
    if shape == "many_patterns":
        assert len(np.unique(x[0].view(torch.int16).numpy())) > 4096
    got = _filter_rows(x.cuda(), 0.8, top_k, top_p)
    want = O.target_probs(x.float().numpy(), 0.8, "bf16", top_k or None, top_p or None)
    assert np.array_equal(got, want), [(int((got[r] != want[r]).sum()), int((got[r] > 0).sum()), int((want[r] > 0).sum())) for r in range(2)]


@GPU
@pytest.mark.parametrize("top_k,top_p", [(50, 0.0), (0, 0.9), (40, 0.95)], ids=["k50", "p09", "k40_p095"])
@pytest.mark.parametrize("shape", ["below_the_exp_cut", "partly_below_the_exp_cut", "tiny_magnitudes", "zeros_of_both_signs", "huge_without_mass",
                                   "huge_with_mass", "ragged_unaligned"])
def test_rs_filter_bf16_zone_table_edges(shape, top_k, top_p):
    """The zone kernel's count pass indexes 8 192 counters by the packed bf16 pattern (2^-16 <= |x| < 2^16) and takes a per-element path
    only for vectors with an id outside them.  Its edges: patterns INSIDE the table that carry no mass (below max - 104: counted, then
    dropped when the list is built), magnitudes below 2^-16 (the 64-entry list: duplicates, both signs, +0 and -0), magnitudes of 2^16
    and more without mass (-1e6, -inf: ignored) and with mass (the maximum itself: the row goes to the other kernel), a ragged row on
    an odd stride (element loads, -inf padding).  Bit for bit against the oracle, T = 0.8 (T = 1 for the unaligned rows)."""
    V = 152064
    g = torch.Generator().manual_seed(len(shape) * 31 + top_k)
    x = torch.randn(3, V, generator=g) * 2.0
    T = 0.8
    if shape == "below_the_exp_cut":
        x[:, 777] = 120.0                                       # max / T - 104 = 46: every other id is counted and carries nothing
    elif shape == "partly_below_the_exp_cut":
        x[:, 5] = 100.0                                         # cut at 21 (scaled): the ids at 30 and 25 stay, the N(0, 4) bulk goes
        x[:, 1000:1040] = 30.0
        x[:, 2000:2100:3] = 25.0
        x[1, 3000:3010] = 16.75                                 # scaled 20.94 / 21.0 either side of the cut after bf16 rounding
    elif shape == "tiny_magnitudes":
        idx = torch.randperm(V, generator=g)[:48]
        x[:, idx] = torch.tensor([1e-6, -1e-6, 3e-7, 1e-6, -2.5e-9, 1.2e-5])[torch.arange(48) % 6]
    elif shape == "zeros_of_both_signs":
        x[:, 100:130] = 0.0
        x[:, 200:220] = -0.0
        x[2] = torch.where(torch.arange(V) % 3 == 0, torch.tensor(0.0), torch.tensor(-0.0))   # a row of zeros: > 64 small magnitudes
    elif shape == "huge_without_mass":
        x[:, 10:20] = -1.0e6
        x[:, 30:33] = float("-inf")
        x[1, 40] = -7.0e4
    elif shape == "huge_with_mass":
        x[0, 12345] = 70000.0                                   # the maximum itself is outside the table
        x[1, 5] = 65536.0
        x[1, 6] = 65536.0 - 256.0                               # 2^16 - 2^8: the largest magnitudes inside (scaled by 1 / 0.8 they are outside)
    if shape == "ragged_unaligned":
        V, T = 151999, 1.0
        buf = (torch.randn(3, V + 3, generator=g) * 2.0).to(torch.bfloat16).cuda()
        xd = buf[:, 1:V + 1]                                    # rows start 2 bytes off a 16-byte boundary, stride V + 3
        x = xd.cpu().float()
        probs, p, rec = ops.filtered_probs(xd, T, top_k, top_p, torch.arange(3, dtype=torch.int64, device="cuda"))
        got = probs.float().cpu().numpy()
    else:
        x = x.to(torch.bfloat16)
        got = _filter_rows(x.cuda(), T, top_k, top_p)
    want = O.target_probs(x.float().numpy(), T, "bf16", top_k or None, top_p or None)
    assert np.array_equal(got, want), [(int((got[r] != want[r]).sum()), int((got[r] > 0).sum()), int((want[r] > 0).sum())) for r in range(3)]


@GPU
@pytest.mark.parametrize("shape", ["one_bin", "quantised", "subnormal", "beyond_the_levels"])
@pytest.mark.parametrize("top_k,top_p", [(50, 0.0), (0, 0.9), (3000, 0.35), (0, 0.999)], ids=["k50", "p09", "k3000_p035", "p0999"])
def test_rs_filter_float32_levels(shape, top_k, top_p):
    """The float32 path's three levels of counters (15-bit bins with exact sums, bits 16:8 of one bin, the last 8 bits under one
    prefix) where they are stressed: nearly uniform rows (every probability in ONE or two bins: the whole search happens in the
    sub-tables), quantised logits (thousands of ids share each float32 value: a cut inside a group, kept by id), rows whose tail is
    subnormal or zero, and a row of more than 2^18 ids (the pass-per-step variant).  Bit for bit against the oracle."""
    V = {"one_bin": 152064, "quantised": 70001, "subnormal": 152064, "beyond_the_levels": (1 << 18) + 77}[shape]
    g = torch.Generator().manual_seed(len(shape) * 1000 + top_k)
    x = torch.randn(2, V, generator=g)
    if shape == "one_bin":
        x = x * 1e-3
    elif shape == "quantised":
        x = torch.round(x * 2) / 2
    elif shape == "subnormal":
        x = x * 25.0
    else:
        x = x * 2.0
    got = _filter_rows(x.cuda(), 0.8, top_k, top_p)
    want = O.target_probs(x.numpy(), 0.8, "f32", top_k or None, top_p or None)
    assert np.array_equal(got, want), [(int((got[r] != want[r]).sum()), int((got[r] > 0).sum()), int((want[r] > 0).sum())) for r in range(2)]


@GPU
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("top_k,top_p", [(8, 0.0), (0, 0.7), (20, 0.9)], ids=["k8", "p07", "k20_p09"])
def test_rs_step_samples_from_the_filtered_distribution(dtype, top_k, top_p):
    """The whole non-greedy step with filters: jf_rs_probs -> jf_rs_filter -> jf_rs_step on the probability rows, 24 rows x block
    9 at V = 5 000 and 6 rows at the real vocabulary, against the row-by-row restatement over the oracle's filtered distribution:
    accepted counts, rejected positions, bonus tokens (never a filtered-out id), draws, next drafts, cursors."""
    for B, L, V in ((24, 9, 5000), (6, 5, 152064)):
        g = torch.Generator().manual_seed(77 + top_k + V)
        logits = torch.randn(B, L - 1, V, generator=g) * 2.5                # a handful of heavy ids per row: the filters cut between real alternatives
        heavy = logits.argmax(-1)
        draft = torch.randint(0, V, (B, L), generator=g)
        pick = torch.rand(B, L - 1, generator=g) < 0.6                       # most proposals are the row's heaviest id (accepted about half the time) ...
        draft[:, 1:] = torch.where(pick, heavy, draft[:, 1:])               # ... the others are random ids: filtered out, probability 0, rejected
        n = 4 * B * L
        unis = torch.randint(0, 1 << 24, (n,), generator=g).float() / float(1 << 24)
        bonus = torch.randint(0, 1 << 24, (n,), generator=g).float() / float(1 << 24)
        pads = torch.randint(0, V, (n,), generator=g)
        res = {}
        for backend in ("hip", "hostsim"):
            with use_backend(backend):
                dev = device_for(backend)
                st = ops.RsStepper(B, L, dev, pads, unis, bonus)
                rows, toks, nd = st.step(draft.to(dev), logits.to(dtype).to(dev), 0.9, 3, [L] * B, [3, 5, 7], top_k, top_p)
                res[backend] = (rows.copy(), toks.copy(), nd.cpu().numpy().copy(), st.cursors.cpu().tolist())
        f = N.RS_FIELDS.index
        assert (res["hostsim"][0][:, f("reject_pos")] >= 0).sum() >= B // 3
        _assert_rs_equal(res["hip"], res["hostsim"], B)
        q = O.target_probs(logits.to(dtype).float().numpy().reshape(B * (L - 1), V), 0.9, "bf16" if dtype == torch.bfloat16 else "f32",
                           top_k or None, top_p or None).reshape(B, L - 1, V)
        rows, toks = res["hip"][0], res["hip"][1]
        for b in range(B):
            rej, n = int(rows[b, f("reject_pos")]), int(rows[b, f("n_committed")])
            if rej >= 0:
                assert q[b, rej, int(toks[b, n - 1])] > 0          # the bonus token has mass under the filtered distribution


def _run_rs_given(backend, draft, logits, unis, bonus, pads, temperature, eos=None):
    with use_backend(backend):
        dev = device_for(backend)
        B, L = draft.shape
        st = ops.RsStepper(B, L, dev, pads, unis, bonus)
        rows, toks, nd = st.step(draft.to(dev), logits.to(dev), temperature, eos, [L] * B, [3, 5, 7])
        return rows.copy(), toks.copy(), nd.cpu().numpy().copy(), st.cursors.cpu().tolist()


@GPU
@pytest.mark.parametrize("B,L,V", [(100, 65, 64), (8, 9, 2000)], ids=["outside_lds", "staged"])
def test_rs_step_every_accepted_test_needs_the_exact_probability(B, L, V):
    """Uniforms placed 2e-6 (relative) below the EXACT probability of every proposal: each test lies inside the error band of
    the streaming kernel's float32 row sum, so each one is resolved with the row's float64 sum — and then accepted, so a row of
    L - 1 tests carries L - 1 resolved words while the walk works through it.  Round 4's eight-entry patch table of the walk
    outside LDS overwrote its last entry when full and such a row never finished; the table is one word per
    position of the current row now.  Every proposal must come out accepted, no bonus draw, cursors = B (L - 1)."""
    g = torch.Generator().manual_seed(77)
    draft = torch.randint(0, V, (B, L), generator=g)
    logits = torch.randn(B, L - 1, V, generator=g) * 0.3
    logits.scatter_(2, draft[:, 1:].unsqueeze(-1), float(np.log(0.9 / 0.1 * (V - 1))))
    x = logits.double()
    pex = torch.softmax(x, dim=-1).gather(2, draft[:, 1:].unsqueeze(-1)).squeeze(-1)          # float64: the exact value to 1e-16
    unis = (pex * (1.0 - 2e-6)).float().reshape(-1)
    assert (unis.double() < pex.reshape(-1)).all()
    n = 4 * B * L
    unis = torch.cat([torch.full((3,), 0.5), unis, torch.full((n,), 0.5)])     # (_run_rs_given starts the stream at cursor 3)
    bonus = torch.randint(0, 1 << 24, (n,), generator=g).float() / float(1 << 24)
    pads = torch.randint(0, V, (n,), generator=g)
    a = _run_rs_given("hip", draft, logits, unis, bonus, pads, 1.0)
    b = _run_rs_given("hostsim", draft, logits, unis, bonus, pads, 1.0)
    f = N.RS_FIELDS.index
    assert (b[0][:, f("n_committed")] == L - 1).all() and (b[0][:, f("reject_pos")] == -1).all()
    _assert_rs_equal(a, b, B)


@GPU
@pytest.mark.parametrize("B,L,V", [(48, 9, 2000), (300, 33, 64)], ids=["staged", "outside_lds"])
def test_rs_step_float32_rows_with_a_large_scaled_maximum(B, L, V):
    """float32 logits around 30 at T = 0.1: the scaled maximum is ~300, the error bound of the streaming sum grows with it
    (rs_eps_row) and passes the 1e-4 the narrow accept band covers.  Such rows carry the WIDE band (2^-8, flagged in the word's
    lowest mantissa bit) — round 4 marked every one of their tests undecided and resolved them serially; the results are the
    oracle's either way."""
    T = 0.1
    g = torch.Generator().manual_seed(5)
    draft = torch.randint(0, V, (B, L), generator=g)
    logits = torch.randn(B, L - 1, V, generator=g) * 0.3
    logits.scatter_(2, draft[:, 1:].unsqueeze(-1), float(np.log(0.5 / 0.5 * (V - 1))))
    logits = logits * T + 30.0
    n = 4 * B * L
    unis = torch.randint(0, 1 << 24, (n,), generator=g).float() / float(1 << 24)
    bonus = torch.randint(0, 1 << 24, (n,), generator=g).float() / float(1 << 24)
    pads = torch.randint(0, V, (n,), generator=g)
    a = _run_rs_given("hip", draft, logits, unis, bonus, pads, T, eos=3)
    b = _run_rs_given("hostsim", draft, logits, unis, bonus, pads, T, eos=3)
    f = N.RS_FIELDS.index
    assert (b[0][:, f("reject_pos")] >= 0).sum() >= B // 4 and (b[0][:, f("n_committed")] > 1).sum() >= B // 8
    _assert_rs_equal(a, b, B)


@GPU
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("coarse", [False, True], ids=["distinct_logits", "tied_logits"])
def test_rs_step_masked_argmax_after_sixteen_collisions(dtype, coarse):
    """A proposal that holds ~0.97 of its position's mass and is rejected collides in (nearly) all 16 residual draws: the bonus
    token is the masked argmax (JDN:147-153) — first index of the largest ROUNDED probability among the other ids.  The kernels
    find it from the segments' largest logits instead of forming every probability; logits on a coarse grid make the maximum a
    many-way tie (first index wins, ids in several segments).  Real vocabulary, against the row-by-row restatement."""
    B, L, V = 6, 4, 152064
    g = torch.Generator().manual_seed(77)
    draft = torch.randint(0, V, (B, L), generator=g)
    logits = torch.randn(B, L - 1, V, generator=g) * 2
    if coarse:
        logits = torch.round(logits)                                   # hundreds of ids share the largest value
    boost = float(np.log(0.97 / 0.03) + np.log(V) + 2.0)
    logits.scatter_(2, draft[:, 1:].unsqueeze(-1), boost)
    n = 4 * B * L * 16
    unis = torch.full((n,), 0.995)                                      # every first test rejects its 0.97-mass proposal
    bonus = torch.randint(0, 1 << 24, (n,), generator=g).float() / float(1 << 24)
    pads = torch.randint(0, V, (n,), generator=g)
    out = {}
    for backend in ("hip", "hostsim"):
        with use_backend(backend):
            dev = device_for(backend)
            st = ops.RsStepper(B, L, dev, pads, unis, bonus)
            rows, toks, nd = st.step(draft.to(dev), logits.to(dtype).to(dev), 1.0, None, [L] * B, [1, 2, 3])
            out[backend] = (rows.copy(), toks.copy(), nd.cpu().numpy().copy(), st.cursors.cpu().tolist())
    f = N.RS_FIELDS.index
    assert (out["hostsim"][0][:, f("n_bonus_draws")] == 16).sum() >= 2      # the masked argmax really decided rows
    _assert_rs_equal(out["hip"], out["hostsim"], B)


@GPU
@pytest.mark.parametrize("B,L,V", [(8, 9, 20000), (300, 9, 50)], ids=["one_launch", "several_launches"])
def test_timing_events_ride_on_the_calls_dispatches(B, L, V):
    """jf_timing_arm: the events a caller arms are taken by the next jf_rs_probs / jf_rs_step call and carry the start of its
    first launch and the stop of its last one (hipExtLaunchKernel), or bracket the launches of a slow path; the call's results do
    not change, the events read a plausible duration, and nothing stays armed for the call after."""
    plain = _run_rs("hip", B, L, V, 9, 0.5, None, torch.bfloat16, 0.9, eos=3)
    pool, seen = [], {}

    def hook(name, phase, nbytes):
        assert phase == "arm", (name, phase)                 # armed calls get no "begin" / "end"
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()                               # their handles exist
        pool.append((a, b))
        seen.setdefault(name, []).append((a, b))
        return a, b
    ops.STAGE_HOOK = hook
    try:
        timed = _run_rs("hip", B, L, V, 9, 0.5, None, torch.bfloat16, 0.9, eos=3)
    finally:
        ops.STAGE_HOOK = None
    _assert_rs_equal(timed, plain, B)
    torch.cuda.synchronize()
    assert set(seen) == {"rs_probs", "rs_step"}
    for name, evs in seen.items():
        for a, b in evs:
            us = a.elapsed_time(b) * 1e3
            assert 0.5 < us < 5e4, (name, us)
    again = _run_rs("hip", B, L, V, 9, 0.5, None, torch.bfloat16, 0.9, eos=3)      # nothing armed any more
    _assert_rs_equal(again, plain, B)


@GPU
@pytest.mark.parametrize("env", [dict(JF_ARGMAX_REVERSE="1"), dict(JF_ARGMAX_REVERSE="2", JF_ARGMAX_ITEMS="4096"),
                                 dict(JF_ARGMAX_WAVE="1", JF_ARGMAX_REVERSE="2"), dict(JF_ARGMAX_NT="0", JF_ARGMAX_CHUNK="4096")],
                         ids=["rows-reversed", "chunk-major", "wave-chunk-major", "plain-small-chunks"])
def test_argmax_is_invariant_under_the_launch_shape_knobs(env):
    """The sweep overrides (item order, item count, per-wavefront items, load policy — read once per process) only change how
    the vocabulary is walked: torch.argmax semantics stay, ties and a NaN row included."""
    import os
    import subprocess
    import sys
    code = ("import torch; from jacobiforcing_amd import ops; g = torch.Generator().manual_seed(3); "
            "x = torch.randn(37, 152064, generator=g).to(torch.bfloat16); x[5, 777] = x[5, 151000] = 30.0; x[9, 4242] = float('nan'); "
            "assert ops.argmax_rows(x.cuda()).cpu().tolist() == torch.argmax(x.float(), -1).tolist(); "
            "y = torch.randn(700, 152064, generator=g).to(torch.bfloat16).cuda(); "
            "assert torch.equal(ops.argmax_rows(y), torch.argmax(y.float(), -1)); print('ok')")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300,
                       cwd=str(ROOT))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


@GPU
@pytest.mark.parametrize("env", [dict(JF_VERIFY_ITEM_WGS="7"), dict(JF_VERIFY_ITEM_WGS="0", JF_ARGMAX_ITEMS="4096"),
                                 dict(JF_ARGMAX_WAVE="1", JF_VERIFY_ITEM_WGS="33"), dict(JF_ARGMAX_CHUNK="4096")],
                         ids=["seven-item-workgroups", "one-per-item-many-chunks", "wave-items", "small-chunks"])
def test_convergence_launch_is_invariant_under_its_knobs(env):
    """How many item workgroups walk the list, how many chunk slots a position has and whether an item is a workgroup or a
    wavefront (overrides read once per process) change nothing about what the convergence launch computes: the golden records
    of the reference and the small full-vocabulary batches decode as before."""
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        str(ROOT / "tests" / "test_multiblock.py"), str(ROOT / "tests" / "test_bench_and_dist.py"),
                        "-k", "golden_calls or small_batches or many_prompts"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_convergence_launch_keeps_its_register_budget():
    """The fused convergence launch shares one kernel between the streaming argmax items and the per-prompt steppers: the
    steppers' code must not cost the items their occupancy (round 3 met both ways this breaks silently: a stepper that is no
    longer inlined — 248 VGPRs + scratch, 2 waves per SIMD, the stream 35 % slower — and a kernel-argument struct kept in
    scratch because its address was selected against null).  Cross-compiles jf_multiblock.hip with the resource remarks."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = ROOT / "jacobiforcing_amd" / "csrc"
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", f"-I{ROOT / 'include'}", f"-I{csrc}",
                          str(csrc / "jf_multiblock.hip"), "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", out.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "mb_verify_kernel" not in name:
            continue
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        assert vgprs <= 96 and scratch == 0, (name, vgprs, scratch)
        seen += 1
    assert seen == 8


# ------------------------------------------------------------------------------------- the loop around the engine steps (f3)
@pytest.mark.parametrize("backend", BACKENDS)
def test_engine_loop_mailboxes_are_pooled_and_their_sequence_numbers_continue(backend):
    """A loop lives for one chunk; its mailbox (mapped host memory the device mails and the host polls) does not: ops._MailboxPool
    hands the same block from loop to loop and the sequence numbers continue, so nothing is mapped, unmapped or re-zeroed per chunk
    (profiles/soak_r06.txt: a mailbox allocated per chunk lost its first record twice in 25 600 soak cases)."""
    with use_backend(backend):
        dev = device_for(backend)
        seen = []
        for it in range(3):
            B, L = 3 + it, 4                                        # (same size class: 128 words)
            lp = ops.EngineLoop(N.EL_KIND_GREEDY, L, dev, [10] * B, [8] * B)
            seen.append((lp._mb_ptr.value, lp.seq))
            rows = torch.zeros((B, N.ENGINE_ROW_INTS), dtype=torch.int32, device=dev)
            rows[:, 0], rows[:, 1], rows[:, 3] = 2, 1, 1             # acc_len 2, one new token, still active
            toks = torch.full((B, L), 7 + it, dtype=torch.int64, device=dev)
            lp.set_draft(torch.zeros((B, L), dtype=torch.int64))
            for _ in range(2):
                lp.next_buffer()
                lp.commit(rows, toks, torch.zeros((1,), dtype=torch.int64, device=dev))
                n, e, a, f = lp.wait()
                assert n.tolist() == [1] * B and a.tolist() == [1] * B
            ring, rl = lp.tokens_host()
            assert rl.tolist() == [2] * B and ring[0, :2].tolist() == [7 + it] * 2
            lp.close()
        assert seen[0][0] == seen[1][0] == seen[2][0]               # one block, three loops
        assert [s for _, s in seen] == [seen[0][1], seen[0][1] + 2, seen[0][1] + 4]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("kind", [N.EL_KIND_GREEDY, N.EL_KIND_SAMPLING], ids=["greedy_rows", "sampling_rows"])
@pytest.mark.parametrize("B,L", [(1, 2), (7, 5), (64, 32), (300, 9)])
def test_engine_loop_commit_maintains_the_device_arrays(B, L, kind, backend):
    """jf_engine_loop_commit against its definition in numpy, several iterations on the same loop (rings fill up, budgets and
    cached lengths move, positions follow), then once more after a compaction (rings stay in their slots): the arrays on the
    device, the record in the mailbox (n | eos | active | fallback per row, the stream cursors) and the sequence word."""
    with use_backend(backend):
        dev = device_for(backend)
        g = np.random.default_rng(B * 100 + L + kind)
        seq_lens = g.integers(1, 500, size=B)
        remaining = g.integers(1, 3 * L, size=B)
        lp = ops.EngineLoop(kind, L, dev, seq_lens.tolist(), remaining.tolist())
        ints = N.ENGINE_ROW_INTS if kind == N.EL_KIND_GREEDY else N.RS_ROW_INTS
        rows_dev = torch.zeros((B, ints), dtype=torch.int32, device=dev)
        toks_dev = torch.zeros((B, L), dtype=torch.int64, device=dev)
        cursors = torch.zeros((1 if kind == N.EL_KIND_GREEDY else 3,), dtype=torch.int64, device=dev)
        kv, rem = seq_lens - 1, remaining.copy()
        ring = [[] for _ in range(B)]
        members = np.arange(B)
        lp.set_draft(torch.zeros((B, L), dtype=torch.int64))
        for it in range(4):
            b = members.size
            if it == 3 and b > 1:                                   # a request left: the batch is compacted, ring rows stay put
                keep = np.flatnonzero(g.random(b) < 0.6)
                keep = keep if keep.size else np.array([0])
                lp.compact(keep)
                members, kv, rem = members[keep], kv[keep], rem[keep]
                b = members.size
                rows_dev, toks_dev = rows_dev[:b].clone(), toks_dev[:b].clone()
            room = np.array([lp.cap - len(ring[s]) for s in members])
            n = np.minimum(g.integers(1, L + 1, size=b), room)      # (a full ring is an error path of its own, below)
            eos, act = g.integers(0, 2, size=b), g.integers(0, 2, size=b)
            acc = np.where(g.random(b) < 0.3, 1, n + 1)
            rec = np.zeros((b, ints), dtype=np.int32)
            if kind == N.EL_KIND_GREEDY:
                n = np.where(acc == 1, np.minimum(1, room), n)
                rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = acc, n, eos, act
            else:
                rec[:, 0], rec[:, 1], rec[:, 6] = n, eos, act
            toks = g.integers(0, 152064, size=(b, L))
            rows_dev[:b].copy_(torch.from_numpy(rec))
            toks_dev[:b].copy_(torch.from_numpy(toks))
            cur = g.integers(0, 1 << 40, size=cursors.numel())
            cursors.copy_(torch.from_numpy(cur))
            lp.next_buffer()
            lp.commit(rows_dev, toks_dev, cursors)
            gn, ge, ga, gf = lp.wait()
            assert gn.tolist() == n.tolist() and ge.tolist() == eos.tolist() and ga.tolist() == act.tolist()
            assert gf.tolist() == ((acc == 1).astype(int).tolist() if kind == N.EL_KIND_GREEDY else [0] * b)
            assert lp.cursors_host[:cursors.numel()] == cur.tolist()
            kv, rem = kv + n, rem - n
            for r, s in enumerate(members):
                ring[s] += toks[r, :n[r]].tolist()
            assert lp.kv_start.cpu().tolist() == kv.tolist() and lp.remaining.cpu().tolist() == rem.tolist()
            assert lp.positions.cpu().tolist() == (kv[:, None] + np.arange(L)[None, :]).tolist()
            rg, rl = lp.tokens_host()
            assert rl.tolist() == [len(x) for x in ring]
            assert all(rg[s, :len(ring[s])].tolist() == ring[s] for s in range(B))
        # a ring without room for the row's tokens: reported, nothing written past the ring
        rec = np.zeros((members.size, ints), dtype=np.int32)
        rec[:, 1 if kind == N.EL_KIND_GREEDY else 0] = L
        rec[:, 0] = L + 1 if kind == N.EL_KIND_GREEDY else L
        rows_dev[:members.size].copy_(torch.from_numpy(rec))
        with pytest.raises(RuntimeError, match="ring"):
            for _ in range(lp.cap // L + 2):
                lp.next_buffer()
                lp.commit(rows_dev, toks_dev, cursors)
                lp.wait()
        lp.close()
