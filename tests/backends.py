"""Test-only backends for the product wrappers in jacobiforcing_amd.ops.

* ``hip``      — the real library (GPU box, ``-m gpu`` tests).
* ``hostsim``  — tests/hostsim/libjf_hostsim.so: the device state-machine source compiled
                 single-lane with g++, plus numpy stand-ins for the streaming kernels.  Used by the
                 CPU suite to check control logic and host plumbing without a GPU.  The product
                 package never loads it.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from jacobiforcing_amd import _native as N
from oracle import jacobi_oracle as O

ROOT = Path(__file__).resolve().parents[1]
HS_DIR = ROOT / "tests" / "hostsim"
HS_LIB = HS_DIR / "libjf_hostsim.so"


def build_hostsim(force: bool = False) -> Path:
    src = HS_DIR / "hostsim.cpp"
    core = ROOT / "jacobiforcing_amd" / "csrc" / "jf_mb_core.h"
    if force or not HS_LIB.exists() or HS_LIB.stat().st_mtime < max(src.stat().st_mtime, core.stat().st_mtime):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", f"-I{ROOT / 'include'}",
                               f"-I{ROOT / 'jacobiforcing_amd' / 'csrc'}", str(src), "-o", str(HS_LIB)])
    return HS_LIB


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, C.c_void_p):
        return p.value or 0
    return int(p)


def _view(ptr, count, dtype):
    a = _addr(ptr)
    nbytes = int(count) * np.dtype(dtype).itemsize
    buf = (C.c_char * nbytes).from_address(a)
    return np.frombuffer(buf, dtype=dtype, count=int(count))


class HostSimLib:
    """Duck-types the ctypes library object used by jacobiforcing_amd.ops (CPU tensors only)."""

    def __init__(self):
        self.hs = C.CDLL(str(build_hostsim()))
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        P = C.POINTER(N.MbParams)
        sig = {
            "hs_mb_state_ints": (i64, [P]), "hs_mb_max_rows": (i32, [P]), "hs_mb_max_tokens": (i32, [P]),
            "hs_mb_begin": (C.c_int, [vp, i64, C.c_int, P, vp, vp, vp]),
            "hs_mb_pack": (C.c_int, [vp, i64, C.c_int, i32, i64, vp, vp, vp, vp]),
            "hs_mb_step": (C.c_int, [vp, i64, C.c_int, vp, i64, vp]),
            "hs_mb_read_ret": (C.c_int, [vp, i64, C.c_int, vp, i32]),
            "hs_engine_step": (C.c_int, [vp, C.c_int, C.c_int, vp, i32, vp, vp, vp, vp, i64, vp, vp]),
        }
        for k, (r, a) in sig.items():
            f = getattr(self.hs, k)
            f.restype, f.argtypes = r, a
        self._err = b""

    # -- plumbing
    def jf_version(self):
        return 100

    def jf_last_error(self):
        return self._err

    # -- state machine (drop the trailing stream argument)
    def jf_mb_state_ints(self, p):
        return self.hs.hs_mb_state_ints(p)

    def jf_mb_max_rows(self, p):
        return self.hs.hs_mb_max_rows(p)

    def jf_mb_max_tokens(self, p):
        return self.hs.hs_mb_max_tokens(p)

    def jf_mb_begin(self, *a):
        return self.hs.hs_mb_begin(*a[:-1])

    def jf_mb_pack(self, *a):
        return self.hs.hs_mb_pack(*a[:-1])

    def jf_mb_step(self, *a):
        return self.hs.hs_mb_step(*a[:-1])

    def jf_mb_read_ret(self, *a):
        return self.hs.hs_mb_read_ret(*a[:-1])

    def jf_engine_step(self, *a):
        return self.hs.hs_engine_step(*a[:-1])

    # -- numpy stand-ins for the streaming kernels (oracle arithmetic)
    def jf_argmax_partial(self, logits, dtype, R, V, stride, packed, stream):
        if dtype == N.JF_F32:
            x = _view(logits, (R - 1) * stride + V, np.float32)
            rows = np.stack([x[r * stride:r * stride + V] for r in range(R)])
        else:
            b = _view(logits, (R - 1) * stride + V, np.uint16)
            rows = O.bf16_bits_to_f32(np.stack([b[r * stride:r * stride + V] for r in range(R)]))
        am = O.argmax_rows(rows).astype(np.uint64)
        pk = _view(packed, R, np.uint64)
        new = (np.uint64(1) << np.uint64(32)) | ((~am) & np.uint64(0xFFFFFFFF))
        pk[:] = np.maximum(pk, new)
        return 0

    def jf_argmax_decode(self, packed, R, greedy, stream):
        pk = _view(packed, R, np.uint64)
        g = _view(greedy, R, np.int64)
        g[:] = ((~pk) & np.uint64(0xFFFFFFFF)).astype(np.int64)
        pk[:] = 0
        return 0

    def jf_accept_lengths(self, draft, draft_rows, greedy, gstride, B, L, accepted, best_idx, stream):
        d = _view(draft, draft_rows * L, np.int64).reshape(draft_rows, L)
        g = _view(greedy, (B - 1) * gstride + max(L - 1, 0), np.int64)
        grows = [g[b * gstride:b * gstride + max(L - 1, 0)].tolist() + [0] for b in range(B)]
        acc = O.accept_lengths(d.tolist(), grows)
        _view(accepted, B, np.int32)[:] = acc
        if _addr(best_idx):
            _view(best_idx, 1, np.int32)[0] = O.first_max_index(acc)
        return 0


    # -- KV cache stand-ins (byte moves; same contracts as jf_kv_append / jf_kv_commit)
    def jf_kv_append(self, k_cache, v_cache, k_new, v_new, slot, Ntok, H, D, S_max, ks, vs, esz, stream):
        rowb = D * esz
        sl = _view(slot, Ntok, np.int64)
        for dst0, src0, tstride in ((_addr(k_cache), _addr(k_new), ks), (_addr(v_cache), _addr(v_new), vs)):
            for i in range(Ntok):
                s_ = int(sl[i])
                if s_ < 0:
                    continue
                brow, pos = divmod(s_, S_max)
                for h in range(H):
                    C.memmove(dst0 + ((brow * H + h) * S_max + pos) * rowb, src0 + (i * tstride + h * D) * esz, rowb)
        return 0

    def jf_kv_commit(self, main_k, main_v, cand_k, cand_v, layers, desc, P, cand_rows, H, D, S_max, T_max, esz, stream):
        rowb = D * esz
        tabs = [_view(t, layers, np.int64) for t in (main_k, main_v, cand_k, cand_v)]
        d = _view(desc, P * N.DESC_INTS, np.int32).reshape(P, N.DESC_INTS)
        f = N.DESC_FIELDS.index
        for p in range(P):
            ln, src, dst = int(d[p, f("kv_copy_len")]), int(d[p, f("kv_src_row")]), int(d[p, f("kv_copy_dst")])
            if ln <= 0 or src <= 0:
                continue
            crow = p * cand_rows + src - 1
            for l in range(layers):
                for mi, ci in ((0, 2), (1, 3)):
                    for h in range(H):
                        C.memmove(int(tabs[mi][l]) + ((p * H + h) * S_max + dst) * rowb,
                                  int(tabs[ci][l]) + ((crow * H + h) * T_max) * rowb, ln * rowb)
        return 0


@contextlib.contextmanager
def use_backend(name: str):
    """Temporarily install a backend as the library jacobiforcing_amd.ops talks to."""
    old = N._LIB
    if name == "hostsim":
        N._LIB = HostSimLib()
    elif name == "hip":
        N._LIB = N.load()
    else:
        raise ValueError(name)
    try:
        yield N._LIB
    finally:
        N._LIB = old


def device_for(name: str) -> str:
    return "cpu" if name == "hostsim" else "cuda:0"
