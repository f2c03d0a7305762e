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
            "hs_mb_pack": (C.c_int, [vp, i64, C.c_int, i32, i64, vp, vp, vp, vp, vp, i32]),
            "hs_mb_step": (C.c_int, [vp, i64, C.c_int, vp, i64, vp]),
            "hs_mb_read_ret": (C.c_int, [vp, i64, C.c_int, vp, i32]),
            "hs_engine_step": (C.c_int, [vp, C.c_int, C.c_int, vp, i32, vp, vp, vp, vp, i64, vp, vp]),
            "hs_sb_step": (C.c_int, [vp, C.c_int, vp, i32, i32, i32, vp, i32, vp]),
            "hs_engine_loop_commit": (C.c_int, [C.POINTER(N.EngineLoop), i32]),
            "hs_mb_loop_begin": (C.c_int, [C.POINTER(N.MbLoop), i32, P, vp, vp]),
            "hs_mb_loop_step": (C.c_int, [C.POINTER(N.MbLoop), i32, P, i32, i32, C.c_int]),
            "hs_mb_loop_pack": (C.c_int, [C.POINTER(N.MbLoop), i32, P]),
        }
        self._host_blocks = {}
        self._filtered = {}                # address of a jf_rs_filter record array -> the dense filtered tensor (oracle)
        for k, (r, a) in sig.items():
            f = getattr(self.hs, k)
            f.restype, f.argtypes = r, a
        self._err = b""

    # -- plumbing
    def jf_version(self):
        return 200

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

    def jf_mb_verify(self, logits, dtype, R, V, stride, out_index, states, state_ints, P, packed, packed_len, packed_cap, Tpad,
                     desc, params, stream):
        """the two stand-ins back to back (the fused launch computes the same thing)"""
        if _addr(out_index):
            rc = self.jf_argmax_scatter(logits, dtype, R, V, stride, out_index, packed, stream)
        else:
            rc = self.jf_argmax_partial(logits, dtype, R, V, stride, packed, stream)
        return rc or self.hs.hs_mb_step(states, state_ints, P, packed, packed_len, desc)

    def jf_mb_read_ret(self, *a):
        return self.hs.hs_mb_read_ret(*a[:-1])

    def jf_mb_set_fast_path(self, on):
        return self.hs.hs_set_fast_path(int(on))

    # -- the loop API: plain memory for the mailbox, the argmax stand-in, then the same bodies prompt after prompt
    def jf_host_alloc(self, nbytes, out):
        buf = (C.c_char * int(nbytes))()
        addr = C.addressof(buf)
        self._host_blocks[addr] = buf
        out._obj.value = addr
        return 0

    def jf_host_free(self, p):
        self._host_blocks.pop(_addr(p), None)
        return 0

    def jf_mailbox_wait(self, mailbox, seq, timeout_us, stream):
        got = int(_view(mailbox, 1, np.int32)[0])
        if got != seq:
            self._err = f"jf_mailbox_wait: sequence {seq} not published (mailbox holds {got})".encode()
            return N.JF_E_LAUNCH
        return 0

    def jf_mb_loop_begin(self, loop, seq, params, input_ids, kv_len, stream):
        return self.hs.hs_mb_loop_begin(loop, seq, params, input_ids, kv_len)

    def jf_mb_loop_pack(self, loop, seq, params, stream):
        return self.hs.hs_mb_loop_pack(loop, seq, params)

    def jf_mb_loop_iterate(self, loop, seq, logits, dtype, R, V, stride, compacted, Rtot, Tpad, params, queue_pack, ev_begin, ev_end, stream):
        if compacted:
            rc = self.jf_argmax_scatter(logits, dtype, R, V, stride, loop.valid_index, loop.packed, stream)
        else:
            rc = self.jf_argmax_partial(logits, dtype, R, V, stride, loop.packed, stream)
        return rc or self.hs.hs_mb_loop_step(loop, seq, params, Rtot, Tpad, queue_pack)

    def jf_engine_step(self, *a):
        return self.hs.hs_engine_step(*a[:-1])

    def jf_sb_step(self, *a):
        return self.hs.hs_sb_step(*a[:-1])

    def jf_engine_loop_commit(self, loop, seq, stream):
        return self.hs.hs_engine_loop_commit(loop, seq)

    def jf_engine_fill(self, draft, B, L, seq_len, block_tables, max_cols, block_size, input_ids, positions, slot_mapping,
                       cu_q, cu_k, cache_seqlens, err, stream):
        d = _view(draft, B * L, np.int64).reshape(B, L)
        S = _view(seq_len, B, np.int32)
        bt = _view(block_tables, B * max_cols, np.int32).reshape(B, max_cols)
        ii, pp, sm = _view(input_ids, B * L, np.int64), _view(positions, B * L, np.int64), _view(slot_mapping, B * L, np.int32)
        q, k, cs = _view(cu_q, B + 1, np.int32), _view(cu_k, B + 1, np.int32), _view(cache_seqlens, B, np.int32)
        q[0] = k[0] = 0
        for i in range(B):
            for j in range(L):
                pos = int(S[i]) - 1 + j
                ii[i * L + j], pp[i * L + j] = d[i, j], pos
                blk = pos // block_size
                ok = pos >= 0 and blk < max_cols and bt[i, blk] >= 0
                sm[i * L + j] = bt[i, blk] * block_size + pos % block_size if ok else -1
                if not ok and err:
                    e = _view(err, 1, np.int32)
                    if e[0] == 0:
                        e[0] = i + 1
            q[i + 1], k[i + 1], cs[i] = q[i] + L, k[i] + int(S[i]) - 1 + L, int(S[i]) - 1
        return 0

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

    def jf_argmax_scatter(self, logits, dtype, R, V, stride, out_index, packed, stream):
        oi = _view(out_index, R, np.int32)
        keep = np.nonzero(oi >= 0)[0]
        if dtype == N.JF_F32:
            x = _view(logits, (R - 1) * stride + V, np.float32)
            rows = np.stack([x[r * stride:r * stride + V] for r in keep]) if len(keep) else np.zeros((0, V), np.float32)
        else:
            b = _view(logits, (R - 1) * stride + V, np.uint16)
            rows = O.bf16_bits_to_f32(np.stack([b[r * stride:r * stride + V] for r in keep])) if len(keep) else np.zeros((0, V), np.float32)
        if len(keep):
            am = O.argmax_rows(rows).astype(np.uint64)
            pk = _view(packed, int(oi.max()) + 1, np.uint64)
            new = (np.uint64(1) << np.uint64(32)) | ((~am) & np.uint64(0xFFFFFFFF))
            pk[oi[keep]] = np.maximum(pk[oi[keep]], new)
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

    def jf_rope_kv_append(self, qkv, dtype, Ntok, T, nq, nkv, D, positions, cos_t, sin_t, q_out, k_cache, v_cache, slot_main,
                          S_max, k_cand, v_cand, slot_cand, T_max, stream):
        heads, half = nq + 2 * nkv, D // 2
        if dtype == N.JF_F32:
            x = _view(qkv, Ntok * heads * D, np.float32).reshape(Ntok, heads, D).astype(np.float32)
        else:
            x = O.bf16_bits_to_f32(_view(qkv, Ntok * heads * D, np.uint16)).reshape(Ntok, heads, D)
        pos = _view(positions, Ntok, np.int32).astype(np.int64)
        mp = int(pos.max()) + 1
        cos = _view(cos_t, mp * half, np.float32).reshape(mp, half)[pos][:, None, :]
        sin = _view(sin_t, mp * half, np.float32).reshape(mp, half)[pos][:, None, :]
        x1, x2 = x[..., :half], x[..., half:]
        rot = np.concatenate([x1 * cos - x2 * sin, x2 * cos + x1 * sin], axis=-1).astype(np.float32)
        out = np.where((np.arange(heads) < nq + nkv)[None, :, None], rot, x).astype(np.float32)
        esz = 4 if dtype == N.JF_F32 else 2

        def enc(a):
            return a.astype(np.float32) if dtype == N.JF_F32 else O.f32_to_bf16_bits(a.astype(np.float32))
        R, G = Ntok // T, nq // nkv
        q = out[:, :nq].reshape(R, T, nkv, G, D).transpose(0, 2, 3, 1, 4).reshape(-1)
        _view(q_out, q.size, np.float32 if dtype == N.JF_F32 else np.uint16)[:] = enc(q)
        rowb = D * esz

        def scatter(kc, vc, slots, cap):
            sl = _view(slots, Ntok, np.int64)
            for i in range(Ntok):
                s_ = int(sl[i])
                if s_ < 0:
                    continue
                brow, p_ = divmod(s_, cap)
                for h in range(nkv):
                    for base, src in ((_addr(kc), out[i, nq + h]), (_addr(vc), out[i, nq + nkv + h])):
                        buf = np.ascontiguousarray(enc(src))
                        C.memmove(base + ((brow * nkv + h) * cap + p_) * rowb, buf.ctypes.data, rowb)
        scatter(k_cache, v_cache, slot_main, S_max)
        if _addr(slot_cand):
            scatter(k_cand, v_cand, slot_cand, T_max)
        return 0

    def jf_swiglu(self, gu, dtype, M, I, out, stream):
        if dtype == N.JF_F32:
            x = _view(gu, M * 2 * I, np.float32).reshape(M, 2 * I).astype(np.float32)
        else:
            x = O.bf16_bits_to_f32(_view(gu, M * 2 * I, np.uint16)).reshape(M, 2 * I)
        g, u = x[:, :I], x[:, I:]
        r = ((g / (np.float32(1) + np.exp(-g, dtype=np.float32))) * u).astype(np.float32)
        if dtype == N.JF_F32:
            _view(out, M * I, np.float32)[:] = r.reshape(-1)
        else:
            _view(out, M * I, np.uint16)[:] = O.f32_to_bf16_bits(r.reshape(-1))
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


    # -- non-greedy stand-ins (oracle arithmetic with the kernel's stream/cursor contract)
    def _rows_f32(self, logits, dtype, R, V, stride):
        if dtype == N.JF_F32:
            x = _view(logits, (R - 1) * stride + V, np.float32)
            return np.stack([x[r * stride:r * stride + V] for r in range(R)])
        b = _view(logits, (R - 1) * stride + V, np.uint16)
        return O.bf16_bits_to_f32(np.stack([b[r * stride:r * stride + V] for r in range(R)]))

    @staticmethod
    def _ldt(dtype):
        return "bf16" if dtype == N.JF_BF16 else "f32"

    def jf_rs_workspace_bytes(self, R, V):
        return max(int(R), 0) * 8                       # the CPU stand-in keeps no per-chunk partials

    def jf_rs_step_workspace_bytes(self, rows):
        return max(int(rows), 0) * (16 * 8 + 16)

    def jf_rs_probs(self, logits, dtype, R, V, stride, draft_next, temperature, p_draft, row_max, row_sumexp, packed, ws, ws_bytes, stream):
        rows = self._rows_f32(logits, dtype, R, V, stride)
        t = np.float32(1.0 if temperature <= 0 else temperature)
        x = rows if t == np.float32(1.0) else (O.bf16_round(rows / t) if dtype == N.JF_BF16 else rows / t)
        m = x.max(axis=1)
        e = np.exp(x - m[:, None], dtype=np.float32)
        ssum = e.sum(axis=1, dtype=np.float32)
        dn = _view(draft_next, R, np.int64)
        _view(row_max, R, np.float32)[:] = m
        _view(row_sumexp, R, np.float32)[:] = ssum
        _view(p_draft, R, np.float32)[:] = O.target_probs(rows, float(temperature), self._ldt(dtype))[np.arange(R), dn]
        am = O.argmax_rows(rows).astype(np.uint64)
        pk = _view(packed, R, np.uint64)
        pk[:] = np.maximum(pk, (np.uint64(1) << np.uint64(32)) | ((~am) & np.uint64(0xFFFFFFFF)))
        return 0

    def jf_rs_filter_workspace_bytes(self, dtype, R, V):
        return 0

    def jf_rs_filter(self, logits, dtype, R, V, stride, draft_next, temperature, top_k, top_p, filt, p_draft, row_max, row_sumexp,
                     workspace, workspace_bytes, stream):
        """The stand-in keeps the oracle's dense filtered tensor beside the record array (keyed by its address): jf_rs_step /
        jf_rs_filter_expand called with these records sample from / return it."""
        rows = self._rows_f32(logits, dtype, R, V, stride)
        q = O.target_probs(rows, float(temperature), self._ldt(dtype), int(top_k) or None, float(top_p) or None).astype(np.float32)
        self._filtered[_addr(filt)] = q
        rec = _view(filt, R * C.sizeof(N.RsFilterRow) // 4, np.int32).reshape(R, -1)
        rec[:] = 0
        rec[:, N.RS_FILTER_FLAGS_WORD] = (1 if int(top_k) > 0 and int(top_k) < V else 0) | (2 if 0.0 < float(top_p) < 1.0 else 0)
        dn = _view(draft_next, R, np.int64)
        _view(p_draft, R, np.float32)[:] = q[np.arange(R), dn]
        _view(row_max, R, np.float32)[:] = np.inf
        _view(row_sumexp, R, np.float32)[:] = -1.0
        return 0

    def jf_rs_filter_expand(self, logits, dtype, R, V, stride, temperature, filt, probs, stream):
        q = self._filtered[_addr(filt)]
        if dtype == N.JF_BF16:
            _view(probs, R * V, np.uint16)[:] = O.f32_to_bf16_bits(q).reshape(-1)
        else:
            _view(probs, R * V, np.float32)[:] = q.reshape(-1)
        return 0

    def _probs_of(self, logits, dtype, R, V, stride, temperature, filt):
        """The target distribution the step samples from: with filter records, the tensor jf_rs_filter stored beside them."""
        if filt is not None and _addr(filt):
            return self._filtered[_addr(filt)]
        return O.target_probs(self._rows_f32(logits, dtype, R, V, stride), temperature, self._ldt(dtype))

    def jf_rs_step(self, logits, dtype, V, stride, draft, B, L, p_draft, row_max, row_sumexp, packed, temperature, eos_id,
                   remaining, u_stream, u_len, u_cursor, b_stream, b_len, b_cursor, pad_stream, pad_len, pad_cursor,
                   committed, next_draft, rows, ws, ws_bytes, filt, stream):
        R = B * (L - 1)
        probs = self._probs_of(logits, dtype, R, V, stride, temperature, filt)
        d = _view(draft, B * L, np.int64).reshape(B, L)
        us, bs, ps = _view(u_stream, u_len, np.float32), _view(b_stream, b_len, np.float32), _view(pad_stream, pad_len, np.int64)
        uc, bc, pc = _view(u_cursor, 1, np.int64), _view(b_cursor, 1, np.int64), _view(pad_cursor, 1, np.int64)
        rem = _view(remaining, B, np.int32)
        cm = _view(committed, B * L, np.int64).reshape(B, L)
        nd = _view(next_draft, B * L, np.int64).reshape(B, L)
        rw = _view(rows, B * N.RS_ROW_INTS, np.int32).reshape(B, N.RS_ROW_INTS)
        pk = _view(packed, R, np.uint64)
        greedy = ((~pk) & np.uint64(0xFFFFFFFF)).astype(np.int64).reshape(B, L - 1)
        eos = None if eos_id < 0 else int(eos_id)
        for b in range(B):
            used = {"u": 0, "b": 0}

            def nu():
                v = float(us[(uc[0] + used["u"]) % u_len]); used["u"] += 1; return v

            def nb():
                v = float(bs[(bc[0] + used["b"]) % b_len]); used["b"] += 1; return v
            toks, keep, e = O.rs_verify_row(d[b].tolist(), probs[b * (L - 1):(b + 1) * (L - 1)], eos, nu, nb)
            uc[0] += used["u"]; bc[0] += used["b"]
            cm[b, :len(toks)] = toks
            rej = used["u"] - 1 if (len(toks) == used["u"] and used["b"] > 0) or (used["b"] > 0) else -1
            active = (not e) and len(toks) < int(rem[b])
            npads = 0
            if active:
                acc_len = 1 + len(toks)
                row = [toks[-1]]
                if acc_len < L:
                    r_ = greedy[b, acc_len - 1:].tolist()
                    copy_len = min(len(r_), L - 1)
                    row += r_[:copy_len]
                else:
                    row += [int(greedy[b, -1])]
                    copy_len = 1
                npads = L - 1 - copy_len
                row += [int(ps[(pc[0] + i) % pad_len]) for i in range(npads)]
                nd[b] = row
                pc[0] += npads
            rw[b] = [len(toks), int(e), rej, used["b"], used["u"], npads, int(active), 0]
        pk[:] = 0
        return 0

    def jf_rs_onpolicy_step(self, logits, dtype, V, stride, proposed, R, p_draft, row_max, row_sumexp, packed, temperature,
                            stop_ids, n_stop, u_stream, u_len, u_cursor, m_stream, m_len, m_cursor, committed, redraft, row,
                            ws, ws_bytes, filt, stream):
        probs = self._probs_of(logits, dtype, R, V, stride, temperature, filt)
        prop = _view(proposed, R, np.int64).tolist()
        stops = _view(stop_ids, n_stop, np.int32).tolist() if n_stop else []
        us, ms = _view(u_stream, u_len, np.float32), _view(m_stream, m_len, np.float32)
        uc, mc = _view(u_cursor, 1, np.int64), _view(m_cursor, 1, np.int64)
        used = {"u": 0, "m": 0}

        def nu():
            v = float(us[(uc[0] + used["u"]) % u_len]); used["u"] += 1; return v

        def nm():
            v = float(ms[(mc[0] + used["m"]) % m_len]); used["m"] += 1; return v
        toks, stop_hit = O.onpolicy_verify(prop, probs, stops, nu, nm)
        draws = used["m"]
        rej = used["u"] - 1 if draws > 0 else -1
        cm, rd = _view(committed, R, np.int64), _view(redraft, R, np.int64)
        cm[:len(toks)] = toks
        n_re = (R - len(toks)) if (not stop_hit and len(toks) < R) else 0
        base = int(mc[0]) + draws
        for li in range(len(toks), len(toks) + n_re):
            rd[li] = O.inverse_cdf_sample(probs[li], nm())
        rw = _view(row, N.OP_ROW_INTS, np.int32)
        rw[:6] = [len(toks), int(stop_hit), rej, draws, used["u"], n_re]
        rw[6:8] = np.array([base], dtype=np.int64).view(np.int32)
        uc[0] += used["u"]; mc[0] += used["m"]
        _view(packed, R, np.uint64)[:] = 0
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
