"""Thin torch-facing wrappers over the C ABI (include/jacobiforcing.h).

torch is used for device memory and streams only; every operation below runs in the HIP
library (``jacobiforcing_amd/lib/libjacobiforcing.so``).  Reference citations follow the
header.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as N

EVT_SPAWN, EVT_SWITCH, EVT_EARLY = 1, 2, 4
VERIFY_HOOK = None      # bench.py: (before, after) callables around the verify launch (HIP events on the launch stream)
VERIFY_EVENTS = None    # bench.py: callable -> (begin, end) torch.cuda.Event pair (already recorded once, so their handles exist)
STAGE_HOOK = None       # bench.py: f(name, phase, nbytes) with phase "begin"/"end" around jf_rs_probs / jf_rs_step / jf_argmax+jf_sb_step;
                        # jf_rs_probs / jf_rs_step first ask f(name, "arm", nbytes): a (begin, end) pair of torch events (recorded once
                        # before, so their handles exist) is attached to the call's own dispatches (jf_timing_arm) and no "begin"/"end" follows
ENGINE_LOOP_HOOKS = None  # bench.py: {"body_begin", "body_end", "record_seen", "forward_begin"}: f(loop) around the engine decoders' iteration
                        # body (engine/chunk_loop.py) — events in front of the step and behind the commit launch, the host clock when the
                        # record has been seen, an event in front of the next forward's first kernel
LOOP_HOOKS = None       # bench.py: {"pack_end": f(batch), "forward_begin": f(batch)} — events behind the queued pack launch and in
                        # front of the next forward's first kernel (GPU idle time between the loop body and the forward);
                        # "mailbox_seen": f(batch) the moment the host's poll returns (host clock: control time to the next forward)


def _stage(name: str, nbytes: int):
    """STAGE_HOOK protocol around one timed library call: returns the callable to run behind the call."""
    hk = STAGE_HOOK
    if not hk:
        return lambda: None
    ev = hk(name, "arm", nbytes)
    if ev:
        N.check(N.lib().jf_timing_arm(C.c_void_p(ev[0].cuda_event), C.c_void_p(ev[1].cuda_event)), "jf_timing_arm")
        return lambda: None
    hk(name, "begin", nbytes)
    return lambda: hk(name, "end", 0)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device: torch.device):
    if device.type == "cuda":
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    return None


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return N.JF_F32
    if t.dtype == torch.bfloat16:
        return N.JF_BF16
    raise ValueError(f"logits dtype must be float32 or bfloat16, got {t.dtype}")


def _grown(buf: torch.Tensor, need_bytes: int) -> torch.Tensor:
    """``buf`` if it holds ``need_bytes``, else a zeroed buffer of the same dtype that does (workspaces sized by the library)."""
    if buf.numel() * buf.element_size() >= need_bytes:
        return buf
    return torch.zeros((-(-need_bytes // buf.element_size()),), dtype=buf.dtype, device=buf.device)


# --------------------------------------------------------------------------------------------
# (a2) argmax
# --------------------------------------------------------------------------------------------
def new_packed(rows: int, device) -> torch.Tensor:
    """Zeroed argmax workspace (uint64 payloads held in an int64 tensor)."""
    return torch.zeros((max(int(rows), 1),), dtype=torch.int64, device=device)


def argmax_partial(logits: torch.Tensor, packed: torch.Tensor) -> None:
    """logits [R, V] (last dim contiguous) -> packed[r] = (key << 32) | ~argmax.  packed must be zero."""
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError(f"logits must be [R, V] with a contiguous vocabulary axis, got {tuple(logits.shape)} strides {logits.stride()}")
    R, V = logits.shape
    if packed.numel() < R:
        raise ValueError("argmax workspace too small")
    N.check(N.lib().jf_argmax_partial(_ptr(logits), _dtype_code(logits), R, V, logits.stride(0) if R > 1 else V,
                                      _ptr(packed), _stream(logits.device)), "jf_argmax_partial")


def argmax_scatter(logits: torch.Tensor, out_index: torch.Tensor, packed: torch.Tensor) -> None:
    """As ``argmax_partial`` for logits of a compacted position list: row i -> packed[out_index[i]]; rows whose index is
    negative (list padding) are skipped without being read."""
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError(f"logits must be [R, V] with a contiguous vocabulary axis, got {tuple(logits.shape)} strides {logits.stride()}")
    R, V = logits.shape
    if out_index.dtype != torch.int32 or out_index.numel() < R or not out_index.is_contiguous():
        raise ValueError("out_index must be a contiguous int32 tensor with one entry per logits row")
    N.check(N.lib().jf_argmax_scatter(_ptr(logits), _dtype_code(logits), R, V, logits.stride(0) if R > 1 else V,
                                      _ptr(out_index), _ptr(packed), _stream(logits.device)), "jf_argmax_scatter")


def argmax_rows(logits: torch.Tensor, packed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch.argmax(logits, dim=-1) semantics (MB:476, SB:197, JD:357/567) for [..., V] logits."""
    shape = logits.shape[:-1]
    V = logits.shape[-1]
    flat = logits.reshape(-1, V)
    R = flat.shape[0]
    if packed is None:
        packed = new_packed(R, logits.device)
    greedy = torch.empty((R,), dtype=torch.int64, device=logits.device)
    if R:
        argmax_partial(flat, packed)
        N.check(N.lib().jf_argmax_decode(_ptr(packed), R, _ptr(greedy), _stream(logits.device)), "jf_argmax_decode")
    return greedy.view(shape)


# --------------------------------------------------------------------------------------------
# (a3) accepted-prefix scan
# --------------------------------------------------------------------------------------------
def accept_lengths(draft: torch.Tensor, greedy: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """accepted[b] = 1 + #leading matches of draft[b,1:] vs greedy[b,:L-1]; best = first argmax (MB:482-489)."""
    if draft.dim() != 2 or greedy.dim() != 2:
        raise ValueError("draft and greedy must be 2-D")
    draft = draft.contiguous()
    B, L = greedy.shape[0], draft.shape[1]
    if greedy.stride(1) != 1:
        greedy = greedy.contiguous()
    accepted = torch.empty((B,), dtype=torch.int32, device=draft.device)
    best = torch.zeros((1,), dtype=torch.int32, device=draft.device)
    N.check(N.lib().jf_accept_lengths(_ptr(draft), draft.shape[0], _ptr(greedy), greedy.stride(0), B, L, _ptr(accepted),
                                      _ptr(best), _stream(draft.device)), "jf_accept_lengths")
    return accepted, best


# --------------------------------------------------------------------------------------------
# multiblock state machine
# --------------------------------------------------------------------------------------------
@dataclass
class MultiblockParams:
    n: int = 32
    K: int = 2
    r: float = 0.85
    lookahead_start_ratio: float = 0.0
    n_gram_pool_size: int = 4
    eos_token_id: Optional[int] = None
    pad_token_id: Optional[int] = None
    max_iteration_count: int = 128
    # Capacity of the block lists.  The reference's counters can run away (active_blocks goes negative, Q4: K >= 3, or
    # K = 2 with a pseudo block that made no progress when the RA block filled) and it then spawns one block per
    # iteration, so exact parity needs 1 + max_iteration_count entries; None picks that (~240 KB of state per prompt
    # at n = 32).  A smaller value turns such runs into a capacity RuntimeError.
    max_blocks: Optional[int] = None

    def _blocks(self) -> int:
        if self.max_blocks is not None:
            return int(self.max_blocks)
        return 1 + int(self.max_iteration_count)

    def to_c(self) -> N.MbParams:
        return N.MbParams(n=int(self.n), K=int(self.K), spawn_threshold=int(math.ceil(self.r * self.n)),  # MB:262
                          pool_size=int(self.n_gram_pool_size),
                          eos_id=-1 if self.eos_token_id is None else int(self.eos_token_id),
                          pad_id=-1 if self.pad_token_id is None else int(self.pad_token_id),
                          max_iter=int(self.max_iteration_count), max_blocks=int(max(self._blocks(), self.K)),
                          lookahead_start_ratio=float(self.lookahead_start_ratio))


class MultiblockBatch:
    """P side-by-side multiblock Jacobi calls (one wavefront each) sharing one forward per iteration."""

    def __init__(self, P: int, params: MultiblockParams, device):
        self.P = int(P)
        self.params = params
        self.device = torch.device(device)
        self.c_params = params.to_c()
        lib = N.lib()
        self.state_ints = int(lib.jf_mb_state_ints(C.byref(self.c_params)))
        if self.state_ints <= 0:
            N.check(N.JF_E_INVALID, "jf_mb_state_ints")
        self.max_rows = int(lib.jf_mb_max_rows(C.byref(self.c_params)))
        self.max_tokens = int(lib.jf_mb_max_tokens(C.byref(self.c_params)))
        dev = self.device
        self.states = torch.zeros((self.P, self.state_ints), dtype=torch.int32, device=dev)
        self.desc_dev = torch.zeros((self.P, N.DESC_INTS), dtype=torch.int32, device=dev)
        pin = dev.type == "cuda"
        self.desc_host = torch.zeros((self.P, N.DESC_INTS), dtype=torch.int32, pin_memory=pin)
        rows = self.P * self.max_rows
        # one slot per position, plus room for the fused launch's per-chunk slots (JF_MB_PACKED_ENTRIES)
        self.packed = new_packed(rows * self.max_tokens + N.MB_PACKED_EXTRA, dev)
        self.input_ids = torch.zeros((rows * self.max_tokens,), dtype=torch.int64, device=dev)
        self.positions = torch.zeros((rows * self.max_tokens,), dtype=torch.int32, device=dev)
        self.row_prompt = torch.zeros((rows,), dtype=torch.int32, device=dev)
        self.row_len = torch.zeros((rows,), dtype=torch.int32, device=dev)
        self.valid_index_buf = torch.zeros((rows * self.max_tokens + 64,), dtype=torch.int32, device=dev)
        self.valid_index: Optional[torch.Tensor] = None     # set by pack(compact=True)
        self.Nvalid = 0
        self.ret_cap = self.max_tokens + 2
        self.ret_buf = torch.zeros((self.P, self.ret_cap), dtype=torch.int64, device=dev)
        self.Rtot = 0
        self.Tpad = 0
        # JF_FUSED_VERIFY=0: argmax and state machine as two launches (A/B measurements in tools/)
        self.fused = os.environ.get("JF_FUSED_VERIFY", "1") != "0"

    # -- descriptor readback: the one host sync of an iteration ---------------------------------
    def _read_desc(self) -> np.ndarray:
        self.desc_host.copy_(self.desc_dev, non_blocking=True)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        d = self.desc_host.numpy()
        err = d[:, N.DESC_FIELDS.index("error")]
        if err.any():
            self.packed.zero_()    # a launch that reported an error may have left stale keys behind
            p = int(np.nonzero(err)[0][0])
            N.raise_state_error(int(err[p]), f"multiblock prompt {p} (state-machine line {int(d[p, N.DESC_FIELDS.index('rsv0')])})",
                                aux=int(d[p, N.DESC_FIELDS.index("rsv1")]))
        return d

    def desc_field(self, d: np.ndarray, name: str) -> np.ndarray:
        return d[:, N.DESC_FIELDS.index(name)]

    def begin(self, input_ids: torch.Tensor, kv_len: torch.Tensor) -> np.ndarray:
        """MB:230-262 for every prompt: input_ids [P, n] int64, kv_len [P] int32."""
        n = self.params.n
        if tuple(input_ids.shape) != (self.P, n):
            raise ValueError(f"input_ids must be [{self.P}, {n}], got {tuple(input_ids.shape)}")
        input_ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        kv_len = kv_len.to(device=self.device, dtype=torch.int32).contiguous()
        N.check(N.lib().jf_mb_begin(_ptr(self.states), self.state_ints, self.P, C.byref(self.c_params), _ptr(input_ids),
                                    _ptr(kv_len), _ptr(self.desc_dev), _stream(self.device)), "jf_mb_begin")
        return self._read_desc()

    def pack(self, d: np.ndarray, t_align: int = 1, compact: bool = False, valid_align: int = 1):
        """Forward inputs of the current iteration (MB:417-436): returns (input_ids [R,Tpad], positions [R,Tpad],
        row_prompt [R], row_len [R]).  ``t_align`` rounds the padded row length up (keeps GEMM shapes on a small grid).
        ``compact=True`` also fills ``self.valid_index`` (flat positions that carry a draft token, rounded up to a
        multiple of ``valid_align`` with -1): lm_head and the argmax then run on those positions only."""
        B = self.desc_field(d, "B")
        T = self.desc_field(d, "T")
        self.Rtot = int(B.sum())
        self.Tpad = int(T.max()) if self.Rtot else 0
        if t_align > 1 and self.Tpad:
            self.Tpad = min(((self.Tpad + t_align - 1) // t_align) * t_align, self.max_tokens)
        if self.Rtot == 0:
            return None
        fill = self.params.pad_token_id if self.params.pad_token_id is not None else 0
        valid_align = max(int(valid_align), 1)
        self.Nvalid = int((B.astype(np.int64) * T).sum())
        nvp = (self.Nvalid + valid_align - 1) // valid_align * valid_align
        if compact and nvp > self.valid_index_buf.numel():
            raise RuntimeError("valid-position list exceeds its buffer")
        N.check(N.lib().jf_mb_pack(_ptr(self.states), self.state_ints, self.P, self.Tpad, int(fill), _ptr(self.input_ids),
                                   _ptr(self.positions), _ptr(self.row_prompt), _ptr(self.row_len),
                                   _ptr(self.valid_index_buf) if compact else None, valid_align, _stream(self.device)),
                "jf_mb_pack")
        self.valid_index = self.valid_index_buf[:nvp] if compact else None
        R, Tp = self.Rtot, self.Tpad
        return (self.input_ids[:R * Tp].view(R, Tp), self.positions[:R * Tp].view(R, Tp), self.row_prompt[:R],
                self.row_len[:R])

    def verify(self, logits: torch.Tensor, compacted: Optional[bool] = None) -> np.ndarray:
        """argmax over the vocabulary + the whole loop body (MB:467-721).  logits: [Rtot, Tpad, V] / [Rtot*Tpad, V], or —
        after ``pack(compact=True)`` — one row per entry of ``self.valid_index``.  ``compacted=None`` decides by the
        row count, preferring the compacted reading when ``pack`` was asked for the list (pass it explicitly if the
        rectangular tensor happens to have as many rows as the rounded-up list)."""
        V = logits.shape[-1]
        flat = logits.reshape(-1, V)
        if compacted is None:
            compacted = self.valid_index is not None and flat.shape[0] == self.valid_index.numel()
        if compacted:
            if self.valid_index is None or flat.shape[0] != self.valid_index.numel():
                raise ValueError(f"expected logits for the {0 if self.valid_index is None else self.valid_index.numel()} "
                                 f"listed positions, got {tuple(logits.shape)}")
        elif flat.shape[0] != self.Rtot * self.Tpad:
            raise ValueError(f"expected logits for {self.Rtot}x{self.Tpad} positions, got {tuple(logits.shape)}")
        if self.fused:
            return self.verify_fused(flat, self.valid_index if compacted else None)
        VERIFY_HOOK and VERIFY_HOOK[0](self, flat)
        if compacted:
            argmax_scatter(flat, self.valid_index, self.packed)
        else:
            argmax_partial(flat, self.packed)
        self._launch_step()
        VERIFY_HOOK and VERIFY_HOOK[1](self, flat)
        return self._read_desc()

    def verify_fused(self, flat: torch.Tensor, out_index: Optional[torch.Tensor]) -> np.ndarray:
        """jf_mb_verify: argmax items + one stepper workgroup per prompt in one launch (MB:473-721)."""
        if flat.dim() != 2 or flat.stride(1) != 1:
            raise ValueError(f"logits must be [R, V] with a contiguous vocabulary axis, got {tuple(flat.shape)} strides {flat.stride()}")
        R, V = flat.shape
        VERIFY_HOOK and VERIFY_HOOK[0](self, flat)
        N.check(N.lib().jf_mb_verify(_ptr(flat), _dtype_code(flat), R, V, flat.stride(0) if R > 1 else V, _ptr(out_index),
                                     _ptr(self.states), self.state_ints, self.P, _ptr(self.packed), self.Rtot * self.Tpad,
                                     self.packed.numel(), self.Tpad, _ptr(self.desc_dev),
                                     C.byref(self.c_params), _stream(self.device)), "jf_mb_verify")
        VERIFY_HOOK and VERIFY_HOOK[1](self, flat)
        return self._read_desc()

    def _launch_step(self) -> None:
        N.check(N.lib().jf_mb_step(_ptr(self.states), self.state_ints, self.P, _ptr(self.packed),
                                   self.Rtot * self.Tpad, _ptr(self.desc_dev), _stream(self.device)), "jf_mb_step")

    def step(self) -> np.ndarray:
        self._launch_step()
        return self._read_desc()

    def results(self, d: np.ndarray) -> List[dict]:
        N.check(N.lib().jf_mb_read_ret(_ptr(self.states), self.state_ints, self.P, _ptr(self.ret_buf), self.ret_cap,
                                       _stream(self.device)), "jf_mb_read_ret")
        ret = self.ret_buf.cpu().numpy()
        out = []
        for p in range(self.P):
            ln = int(self.desc_field(d, "ret_len")[p])
            out.append(dict(ret=ret[p, :ln].tolist(), next_token=int(self.desc_field(d, "next_token")[p]),
                            iters=int(self.desc_field(d, "iters")[p]), kv_len=int(self.desc_field(d, "kv_len")[p])))
        return out


# --------------------------------------------------------------------------------------------
# the loop around the step (jf_mb_loop_*): no host round trip on the critical path
# --------------------------------------------------------------------------------------------
class DrawStreams:
    """Pre-drawn 32-bit words, one stream per prompt, behind the reference driver's ``random.choice(generated_ids)``
    (DRV:176-180, 209-215): draw k of prompt p is ``text[(u[p, k % len] * len(text)) >> 32]``.  The same streams feed the
    host-side driver (``rng(p).choice``), the resident driver on the device and — in the tests — the oracle's driver."""

    def __init__(self, P: int, seed: int = 1234, length: int = 8192):
        self.P, self.length = int(P), int(length)
        self.words = np.stack([np.random.default_rng(int(seed) + p).integers(0, 1 << 32, size=self.length, dtype=np.uint64)
                               for p in range(self.P)]).astype(np.uint32) if self.P else np.zeros((0, self.length), np.uint32)
        self.cursor = np.zeros(self.P, dtype=np.int64)

    class _Rng:
        def __init__(self, owner, p):
            self.o, self.p = owner, p

        def choice(self, seq):
            o, p = self.o, self.p
            w = int(o.words[p, o.cursor[p] % o.length])
            o.cursor[p] += 1
            return seq[(w * len(seq)) >> 32]

    def rng(self, p: int) -> "DrawStreams._Rng":
        return DrawStreams._Rng(self, int(p))


class LoopSummary:
    """What the device published about the NEXT forward (mailbox header) + the descriptor table of the launch
    (``d`` [P, DESC_INTS], ``fin`` [P, MB_FIN_INTS] resident-driver records or None; both filled by ``snapshot``)."""
    __slots__ = ("seq", "Rtot", "Rmain", "Tpad", "Tmax", "Nvalid", "Nvalid_pad", "n_done", "max_kv", "error", "accepted",
                 "n_call_end", "d", "fin")

    def __init__(self, h):
        (self.seq, self.Rtot, self.Rmain, self.Tpad, self.Tmax, self.Nvalid, self.Nvalid_pad, self.n_done, self.max_kv,
         self.error, self.accepted, self.n_call_end) = h
        self.d = None
        self.fin = None


class MultiblockLoop:
    """jf_mb_loop_begin / jf_mb_loop_iterate over a ``MultiblockBatch``: per iteration ONE launch for the convergence check +
    loop body of every prompt (committed lengths written into the cache's ``kv_len``, finished calls restarted on the
    device when a resident driver is attached), the pack launch of the next forward queued right behind it, and a mailbox
    in mapped pinned host memory that the host polls for (Rtot, Tpad, ...) instead of copying descriptors and synchronising
    the stream.  Row order 1: row 0 of every prompt first (the cache rows in order, attended in place), candidate rows after."""

    # jf_mb_loop.flags per device: None = not decided yet.  Decided once per process and device by mailbox_selftest() — or by
    # JF_PUBLISH_FENCE=0/1 — when the first loop on that device is built.
    PUBLISH_FENCE: dict = {}
    SELFTEST_LOG: dict = {}            # device -> (rounds, stale) of the self-test that decided (tests / bench read it)

    @classmethod
    def publish_flags(cls, dev: torch.device) -> int:
        key = (dev.type, dev.index)
        if key not in cls.PUBLISH_FENCE:
            forced = os.environ.get("JF_PUBLISH_FENCE")
            if forced in ("0", "1"):
                cls.PUBLISH_FENCE[key] = int(forced)
            elif dev.type != "cuda" or os.environ.get("JF_MAILBOX_SELFTEST", "1") == "0":
                cls.PUBLISH_FENCE[key] = 0
            else:
                cls.PUBLISH_FENCE[key] = 0                         # (the self-test's own loop is built with the cheap order)
                rounds, stale = cls.mailbox_selftest(dev, int(os.environ.get("JF_MAILBOX_SELFTEST_ROUNDS", "1000")))
                cls.SELFTEST_LOG[key] = (rounds, stale)
                if stale:
                    cls.PUBLISH_FENCE[key] = N.MB_LOOP_PUBLISH_FENCE
                    print(f"jacobiforcing_amd: the mailbox self-test saw the sequence word before the tables in {stale} of {rounds} rounds on "
                          f"{dev}: publishing behind a release fence from now on (JF_MB_LOOP_PUBLISH_FENCE, ~7 us per launch)", file=sys.stderr)
        return cls.PUBLISH_FENCE[key]

    @staticmethod
    def mailbox_selftest(dev: torch.device, rounds: int = 1000, prompts: int = 48) -> tuple:
        """The check of tools/mailbox_stress.py through the shipped library: jf_mb_loop_begin restarts every prompt, its pack
        launch copies all descriptors into the mailbox and stamps it, the host waits for the stamp and looks at the table AT ONCE
        (the slots were overwritten with a marker before).  Returns (rounds, rounds in which the table was not there yet)."""
        P, n = int(prompts), 16
        prm = MultiblockParams(n=n, K=2, r=0.85, n_gram_pool_size=4, eos_token_id=None, pad_token_id=0)
        batch = MultiblockBatch(P, prm, dev)
        kvl = torch.zeros(P, dtype=torch.int32, device=dev)
        lp = MultiblockLoop(batch, kv_len=kvl, t_cap=64, t_align=1, valid_align=8, compact=True, cand_rows=3, order=1, max_seq_len=1 << 20)
        fB, fT, fkv = (N.DESC_FIELDS.index(k) for k in ("B", "T", "kv_len"))
        g = np.random.default_rng(1)
        ids = torch.from_numpy(g.integers(1, 1000, size=(P, n))).to(dev)
        kvs = [g.integers(5, 500, size=P).astype(np.int32) for _ in range(8)]
        kvd = [torch.from_numpy(k) for k in kvs]
        stale = 0
        for i in range(int(rounds)):
            lp.mailbox[N.MB_MAILBOX_HDR:N.MB_MAILBOX_HDR + P * N.DESC_INTS] = -7
            s = lp.begin(ids, kvd[i & 7])
            d = s.d
            ok = (d[:, fB] == 1).all() and (d[:, fT] == n).all() and (d[:, fkv] == kvs[i & 7]).all() and s.Rtot == P and s.Nvalid == P * n
            stale += 0 if ok else 1
        lp.close()
        return int(rounds), stale

    def __init__(self, batch: "MultiblockBatch", kv_len: Optional[torch.Tensor], t_cap: int, t_align: int = 1, valid_align: int = 1,
                 compact: bool = True, cand_rows: int = 1, order: int = 1, max_seq_len: int = 0,
                 drv: Optional[torch.Tensor] = None, draws: Optional[torch.Tensor] = None, wait_timeout_s: float = 30.0):
        b = self.batch = batch
        dev = b.device
        lib = N.lib()
        self.flags = self.publish_flags(torch.device(dev))
        self.compact = bool(compact)
        self.t_cap = int(min(t_cap, b.max_tokens))
        rows = b.P * b.max_rows
        self.row_cand = torch.full((rows,), -1, dtype=torch.int32, device=dev)
        self.row_kv = torch.zeros((rows,), dtype=torch.int32, device=dev)
        self.kv_len = kv_len
        self.drv, self.draws = drv, draws
        self.n_ints = N.mailbox_ints(b.P)
        # (a mailbox of an earlier loop, mapped once: its sequence numbers continue — _MailboxPool below has the reason)
        self._mb_key, ptr, self.seq = _MailboxPool.take(torch.device(dev), self.n_ints)
        self._mb_ptr = ptr
        self.mailbox = np.ctypeslib.as_array((C.c_int32 * self.n_ints).from_address(ptr.value))
        self._hdr = self.mailbox[:N.MB_MAILBOX_HDR]
        self._wait = lib.jf_mailbox_wait
        self._views, self._vi = {}, {}
        self.timeout_us = int(wait_timeout_s * 1e6)
        fill = b.params.pad_token_id if b.params.pad_token_id is not None else 0
        self.c_loop = N.MbLoop(
            states=b.states.data_ptr(), state_ints=b.state_ints, P=b.P, order=int(order),
            packed=b.packed.data_ptr(), packed_cap=b.packed.numel(), desc=b.desc_dev.data_ptr(),
            input_ids=b.input_ids.data_ptr(), positions=b.positions.data_ptr(), row_prompt=b.row_prompt.data_ptr(),
            row_len=b.row_len.data_ptr(), row_cand=self.row_cand.data_ptr(), row_kv_len=self.row_kv.data_ptr(),
            valid_index=b.valid_index_buf.data_ptr() if compact else None,
            rows_cap=rows, t_cap=self.t_cap, t_align=int(t_align), valid_align=int(valid_align), cand_rows=int(max(cand_rows, 1)),
            flags=int(self.flags), pad_fill=int(fill), kv_len=None if kv_len is None else kv_len.data_ptr(), mailbox=ptr.value,
            drv=None if drv is None else drv.data_ptr(),
            drv_ints=0 if drv is None else int(drv.shape[1]), draws=None if draws is None else draws.data_ptr(),
            draw_len=0 if draws is None else int(draws.shape[1]), max_seq_len=int(max_seq_len))
        self.last: Optional[LoopSummary] = None

    def close(self) -> None:
        if self._mb_ptr is not None:
            if self.batch.device.type == "cuda":             # a queued launch may still be mailing descriptors
                torch.cuda.synchronize(self.batch.device)
            self.mailbox = None
            self._hdr = None
            _MailboxPool.give(self._mb_key, self._mb_ptr, self.seq)
            self._mb_ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- launches -------------------------------------------------------------------------------------
    def begin(self, input_ids: torch.Tensor, kv_len: torch.Tensor) -> LoopSummary:
        """MB:230-262 for every prompt (kv_len: a length, JF_MB_KEEP or JF_MB_INACTIVE) + publish + pack."""
        b = self.batch
        n = b.params.n
        if tuple(input_ids.shape) != (b.P, n):
            raise ValueError(f"input_ids must be [{b.P}, {n}], got {tuple(input_ids.shape)}")
        self._in = input_ids.to(device=b.device, dtype=torch.int64).contiguous()
        self._kv = kv_len.to(device=b.device, dtype=torch.int32).contiguous()
        self.seq += 1
        N.check(N.lib().jf_mb_loop_begin(self.c_loop, self.seq, C.byref(b.c_params), _ptr(self._in), _ptr(self._kv),
                                         _stream(b.device)), "jf_mb_loop_begin")
        return self.wait()

    def iterate(self, logits: torch.Tensor) -> None:
        """Queue the convergence check + loop body + next pack behind the forward that produced ``logits`` (rows follow
        ``valid_index()`` when the loop was built with ``compact``, else the Rtot x Tpad rectangle).  Does not wait."""
        b, s = self.batch, self.last
        V = logits.shape[-1]
        flat = logits.reshape(-1, V)
        want = s.Nvalid_pad if self.compact else s.Rtot * s.Tpad
        if flat.shape[0] != want or flat.stride(1) != 1:
            raise ValueError(f"expected contiguous logits for {want} positions, got {tuple(logits.shape)}")
        self.seq += 1
        # bench.py: VERIFY_EVENTS() -> (begin, end) torch events recorded by the library around the convergence launch alone
        # (the pack launch is queued in the same call either way); VERIFY_HOOK (before, after) callables are told about it
        ev = VERIFY_EVENTS() if VERIFY_EVENTS else None
        hook = VERIFY_HOOK
        hook and hook[0](b, flat)
        N.check(N.lib().jf_mb_loop_iterate(self.c_loop, self.seq, _ptr(flat), _dtype_code(flat), flat.shape[0], V,
                                           flat.stride(0) if flat.shape[0] > 1 else V, 1 if self.compact else 0, s.Rtot, s.Tpad,
                                           C.byref(b.c_params), 1, C.c_void_p(ev[0].cuda_event) if ev else None,
                                           C.c_void_p(ev[1].cuda_event) if ev else None, _stream(b.device)), "jf_mb_loop_iterate")
        hook and hook[1](b, flat)
        if LOOP_HOOKS and "pack_end" in LOOP_HOOKS:
            LOOP_HOOKS["pack_end"](b)

    def wait(self, snapshot: bool = True) -> LoopSummary:
        """Poll the mailbox for the sequence number of the last launch.  ``snapshot=False`` returns the header only (all the
        next forward needs); call ``snapshot(s)`` for the descriptor table before the next ``iterate``/``begin`` — e.g. after
        the forward has been queued, so that the copy is off the critical path."""
        b = self.batch
        rc = self._wait(self._mb_ptr, self.seq, self.timeout_us, _stream(b.device))
        if rc:
            N.check(rc, "jf_mailbox_wait")
        if LOOP_HOOKS and "mailbox_seen" in LOOP_HOOKS:
            LOOP_HOOKS["mailbox_seen"](b)
        s = LoopSummary(self._hdr[:12].tolist())
        self.last = s
        b.Rtot, b.Tpad, b.Nvalid = s.Rtot, s.Tpad, s.Nvalid
        b.valid_index = None                   # valid_index() sets it for the forward that uses it
        if s.error:
            self.snapshot(s)
            p = s.error - 1
            b.packed.zero_()      # a failed launch may have left keys behind
            f = N.DESC_FIELDS.index
            N.raise_state_error(int(s.d[p, f("error")]), f"multiblock prompt {p} (state-machine line {int(s.d[p, f('rsv0')])})",
                                aux=int(s.d[p, f("rsv1")]))
        if snapshot:
            self.snapshot(s)
        return s

    def snapshot(self, s: LoopSummary) -> LoopSummary:
        """Copy the descriptor table (and the driver records) of summary ``s`` out of the mailbox."""
        if s.d is None:
            if s.seq != self.seq:
                raise RuntimeError("the mailbox has been overwritten by a later launch")
            m, P = self.mailbox, self.batch.P
            s.d = m[N.MB_MAILBOX_HDR:N.MB_MAILBOX_HDR + P * N.DESC_INTS].reshape(P, N.DESC_INTS).copy()
            if self.drv is not None:
                o = N.MB_MAILBOX_HDR + P * N.DESC_INTS
                s.fin = m[o:o + P * N.MB_FIN_INTS].reshape(P, N.MB_FIN_INTS).copy()
        return s

    # -- views of the next forward's inputs (written by the pack launch that is already queued) ---------
    def inputs(self):
        """(input_ids [R,Tpad], positions [R,Tpad], row_prompt, row_len, row_cand, row_kv [R]) of the next forward; the views
        are kept per shape (making six tensor views costs the host ~10 us, on the critical path behind the mailbox)."""
        b, s = self.batch, self.last
        key = (s.Rtot, s.Tpad)
        v = self._views.get(key)
        if v is None:
            R, Tp = key
            v = (b.input_ids[:R * Tp].view(R, Tp), b.positions[:R * Tp].view(R, Tp), b.row_prompt[:R], b.row_len[:R],
                 self.row_cand[:R], self.row_kv[:R])
            if len(self._views) > 4096:
                self._views.clear()
            self._views[key] = v
        return v

    def valid_index(self) -> Optional[torch.Tensor]:
        if not self.compact:
            return None
        n = self.last.Nvalid_pad
        v = self._vi.get(n)
        if v is None:
            v = self._vi[n] = self.batch.valid_index_buf[:n]
        self.batch.valid_index = v
        return v


# --------------------------------------------------------------------------------------------
# KV cache
# --------------------------------------------------------------------------------------------
def kv_append(k_cache: torch.Tensor, v_cache: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor,
              slot: torch.Tensor) -> None:
    """cache [rows, H_kv, S_max, D]; new [N, H_kv, D] (token stride free, heads/D contiguous); slot[i] = row * S_max +
    position (-1 skips) (ATT:10-40)."""
    rows, H, S_max, D = k_cache.shape
    Ntok = k_new.shape[0]
    if not (k_cache.is_contiguous() and v_cache.is_contiguous()):
        raise ValueError("kv_append expects contiguous caches")
    for t in (k_new, v_new):
        if tuple(t.shape) != (Ntok, H, D) or t.stride(2) != 1 or t.stride(1) != D:
            raise ValueError("kv_append: sources must be [N, H_kv, D] with contiguous heads")
    if slot.numel() != Ntok or slot.dtype != torch.int64:
        raise ValueError("kv_append: slot must be int64 [N]")
    ks = k_new.stride(0) if Ntok > 1 else H * D
    vs = v_new.stride(0) if Ntok > 1 else H * D
    N.check(N.lib().jf_kv_append(_ptr(k_cache), _ptr(v_cache), _ptr(k_new), _ptr(v_new), _ptr(slot), Ntok, H, D, S_max,
                                 ks, vs, k_cache.element_size(), _stream(k_cache.device)), "jf_kv_append")


def rope_kv_append(qkv: torch.Tensor, T: int, nq: int, nkv: int, D: int, positions: torch.Tensor, cos: torch.Tensor,
                   sin: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, slot_main: torch.Tensor,
                   k_cand: Optional[torch.Tensor] = None, v_cand: Optional[torch.Tensor] = None,
                   slot_cand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused RoPE + query re-layout + KV append of one layer.  qkv [N, (nq+2nkv)*D] contiguous; returns q [R, nkv, G*T, D]."""
    Ntok = qkv.shape[0]
    if not qkv.is_contiguous() or qkv.shape[1] != (nq + 2 * nkv) * D:
        raise ValueError("rope_kv_append: qkv must be contiguous [N, (nq+2nkv)*D]")
    if positions.dtype != torch.int32 or positions.numel() != Ntok or not positions.is_contiguous():
        raise ValueError("rope_kv_append: positions must be contiguous int32 [N]")
    R = Ntok // T
    G = nq // nkv
    q = torch.empty((R, nkv, G * T, D), dtype=qkv.dtype, device=qkv.device)
    S_max = k_cache.shape[2]
    T_max = k_cand.shape[2] if k_cand is not None else 0
    N.check(N.lib().jf_rope_kv_append(_ptr(qkv), _dtype_code(qkv), Ntok, T, nq, nkv, D, _ptr(positions), _ptr(cos), _ptr(sin), _ptr(q),
                                      _ptr(k_cache), _ptr(v_cache), _ptr(slot_main), S_max,
                                      _ptr(k_cand) if slot_cand is not None else None,
                                      _ptr(v_cand) if slot_cand is not None else None, _ptr(slot_cand), T_max,
                                      _stream(qkv.device)), "jf_rope_kv_append")
    return q


def swiglu(gu: torch.Tensor) -> torch.Tensor:
    """silu(gate) * up for the fused [M, 2I] projection output -> [M, I] (one HIP launch)."""
    if gu.dim() != 2 or not gu.is_contiguous() or gu.shape[1] % 2:
        raise ValueError("swiglu expects a contiguous [M, 2*I] tensor")
    M, I2 = gu.shape
    out = torch.empty((M, I2 // 2), dtype=gu.dtype, device=gu.device)
    N.check(N.lib().jf_swiglu(_ptr(gu), _dtype_code(gu), M, I2 // 2, _ptr(out), _stream(gu.device)), "jf_swiglu")
    return out


class KVCommitter:
    """Holds the per-layer pointer tables for jf_kv_commit (candidate row -> committed row, MB:500-502)."""

    def __init__(self, main_k: Sequence[torch.Tensor], main_v: Sequence[torch.Tensor], cand_k: Sequence[torch.Tensor],
                 cand_v: Sequence[torch.Tensor], cand_rows: int):
        dev = main_k[0].device
        self.layers = len(main_k)
        self.P, self.H, self.S_max, self.D = main_k[0].shape
        self.T_max = cand_k[0].shape[2]
        self.cand_rows = int(cand_rows)
        self.esz = main_k[0].element_size()
        self._keep = (list(main_k), list(main_v), list(cand_k), list(cand_v))
        mk = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev)
        self.tabs = [mk(main_k), mk(main_v), mk(cand_k), mk(cand_v)]
        self.device = dev

    def commit(self, desc_dev: torch.Tensor) -> None:
        N.check(N.lib().jf_kv_commit(_ptr(self.tabs[0]), _ptr(self.tabs[1]), _ptr(self.tabs[2]), _ptr(self.tabs[3]),
                                     self.layers, _ptr(desc_dev), self.P, self.cand_rows, self.H, self.D, self.S_max,
                                     self.T_max, self.esz, _stream(self.device)), "jf_kv_commit")


# --------------------------------------------------------------------------------------------
# engine single-block step (JD:567-710)
# --------------------------------------------------------------------------------------------
class EngineStepper:
    def __init__(self, max_rows: int, max_L: int, device, pad_stream: torch.Tensor):
        dev = torch.device(device)
        self.device = dev
        self.max_rows, self.max_L = int(max_rows), int(max_L)
        self.packed = new_packed(self.max_rows * self.max_L, dev)
        self.new_tokens = torch.zeros((self.max_rows, self.max_L), dtype=torch.int64, device=dev)
        self.next_draft = torch.zeros((self.max_rows, self.max_L), dtype=torch.int64, device=dev)
        self.rows_dev = torch.zeros((self.max_rows, N.ENGINE_ROW_INTS), dtype=torch.int32, device=dev)
        self.rows_host = torch.zeros((self.max_rows, N.ENGINE_ROW_INTS), dtype=torch.int32, pin_memory=dev.type == "cuda")
        self.tok_host = torch.zeros((self.max_rows, self.max_L), dtype=torch.int64, pin_memory=dev.type == "cuda")
        self.pad_stream = pad_stream.to(device=dev, dtype=torch.int64).contiguous()
        self.pad_cursor = torch.zeros((1,), dtype=torch.int64, device=dev)
        self.remaining = torch.zeros((self.max_rows,), dtype=torch.int32, device=dev)

    def step(self, draft: torch.Tensor, logits: torch.Tensor, eos_id: Optional[int], remaining: Sequence[int]):
        """draft [B, L] int64, logits [B, L-1, V] -> (rows ndarray [B, 8], new_tokens ndarray [B, L], next_draft tensor)."""
        B, L = draft.shape
        if L < 2:
            raise ValueError("Draft must have at least 2 tokens (seed + 1 speculative)")      # MR:1144-1145
        if logits.ndim != 3 or logits.size(0) != B or logits.size(1) != L - 1:                 # JD:244-248
            raise ValueError(f"forward must return logits [B, L-1, vocab] for verifying speculative tokens, "
                             f"expected [{B}, {L - 1}, *], got {tuple(logits.shape)}")
        if B > self.max_rows or L > self.max_L:
            raise RuntimeError("EngineStepper capacity exceeded")
        draft = draft.to(device=self.device, dtype=torch.int64).contiguous()
        flat = logits.reshape(B * (L - 1), logits.shape[-1])
        argmax_partial(flat, self.packed)
        rem = torch.tensor(list(remaining), dtype=torch.int32)
        self.remaining[:B].copy_(rem, non_blocking=True)
        nt = self.new_tokens.view(-1)[:B * L].view(B, L)
        nd = self.next_draft.view(-1)[:B * L].view(B, L)
        N.check(N.lib().jf_engine_step(_ptr(draft), B, L, _ptr(self.packed), -1 if eos_id is None else int(eos_id),
                                       _ptr(self.remaining), _ptr(nt), _ptr(nd), _ptr(self.pad_stream),
                                       self.pad_stream.numel(), _ptr(self.pad_cursor), _ptr(self.rows_dev),
                                       _stream(self.device)), "jf_engine_step")
        self.rows_host[:B].copy_(self.rows_dev[:B], non_blocking=True)
        th = self.tok_host.view(-1)[:B * L].view(B, L)
        th.copy_(nt, non_blocking=True)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        if (self.rows_host[:B, N.ENGINE_FIELDS.index("rsv0")] < 0).any():            # the pad workgroup gave up waiting for a row (2 s bound):
            self.rows_dev[:B, N.ENGINE_FIELDS.index("rsv0")] = 0                     # its marker sits in a word the row workgroups never write
            self.packed.zero_()
            N.check(N.JF_E_LAUNCH, "jf_engine_step (a row never published its hand-off word: launch incomplete)")
        return self.rows_host[:B].numpy(), th.numpy(), nd

    def step_loop(self, loop: "EngineLoop", logits: torch.Tensor, eos_id: Optional[int]) -> None:
        """The step on ``loop.draft`` with the loop's device arrays (budgets in place, the next draft into the loop's other
        buffer) and the commit launch behind it.  Nothing is read back: ``loop.wait()`` polls for the iteration's record."""
        B, L = loop.B, loop.L
        if logits.ndim != 3 or logits.size(0) != B or logits.size(1) != L - 1:                 # JD:244-248
            raise ValueError(f"forward must return logits [B, L-1, vocab] for verifying speculative tokens, "
                             f"expected [{B}, {L - 1}, *], got {tuple(logits.shape)}")
        if B > self.max_rows or L > self.max_L:
            raise RuntimeError("EngineStepper capacity exceeded")
        flat = logits.reshape(B * (L - 1), logits.shape[-1])
        done = _stage("engine_verify", flat.shape[0] * flat.shape[1] * flat.element_size())
        argmax_partial(flat, self.packed)
        nt = self.new_tokens.view(-1)[:B * L].view(B, L)
        N.check(N.lib().jf_engine_step(_ptr(loop.draft), B, L, _ptr(self.packed), -1 if eos_id is None else int(eos_id),
                                       _ptr(loop.remaining), _ptr(nt), _ptr(loop.next_buffer()), _ptr(self.pad_stream),
                                       self.pad_stream.numel(), _ptr(self.pad_cursor), _ptr(self.rows_dev),
                                       _stream(self.device)), "jf_engine_step")
        done()
        loop.commit(self.rows_dev, nt, self.pad_cursor)


# --------------------------------------------------------------------------------------------
# the loop around the engine steps (jf_engine_loop_commit)
# --------------------------------------------------------------------------------------------
class _MailboxPool:
    """Mapped host memory for the loops' records (EngineLoop, MultiblockLoop), allocated once and handed from loop to loop.  An engine
    loop lives for one chunk; allocating and freeing its mailbox per chunk (hipHostMalloc / hipHostFree: a map and an unmap of
    GPU-visible host pages each time) lost a record twice in 25 600 cases of the 100 x fuzz soak with twelve processes on one GPU —
    the commit launch had run, the stream had drained, and the freshly mapped word still read 0; the multiblock loop, which maps one
    mailbox per decoder, lost its first record once in 4 016 cases of the same soak (profiles/soak_r06.txt).  A mailbox that stays
    mapped has no such window: its sequence numbers simply continue from loop to loop (nothing to re-zero), and a chunk no longer
    pays two driver calls."""
    _free: dict = {}                                           # (library, device index, ints) -> [(pointer, last sequence number)]

    @classmethod
    def take(cls, dev: torch.device, n_ints: int):
        size = 128
        while size < n_ints:
            size *= 2
        # (the library object is part of the key — and so kept alive: the memory belongs to the library that allocated it; the
        #  tests' CPU stand-in owns its blocks per instance)
        key = (N.lib(), dev.index if dev.index is not None else -1, size)
        lst = cls._free.get(key)
        if lst:
            ptr, seq = lst.pop()
            return key, ptr, seq
        ptr = C.c_void_p()
        N.check(key[0].jf_host_alloc(size * 4, C.byref(ptr)), "jf_host_alloc")
        return key, ptr, 0

    @classmethod
    def give(cls, key, ptr, seq: int) -> None:
        if seq > 0x7F000000:                                   # (a mailbox retires long before its 32-bit sequence number wraps)
            key[0].jf_host_free(ptr)
            return
        cls._free.setdefault(key, []).append((ptr, int(seq)))


class EngineLoop:
    """One block-length group of an engine decoder's chunk on the device: the draft as ONE [B, L] tensor that the step's next
    draft replaces (two buffers, alternating), the rows' token budgets, cached lengths, next positions and committed-token
    rings as arrays the commit launch maintains, and one record per iteration in mapped host memory (n, eos, active per row +
    the stream cursors) that the host polls for — no per-row read-back, no stream synchronisation.  When a request leaves the
    group (``compact``) the arrays are gathered once; ring rows stay where they are (``slot``)."""

    def __init__(self, kind: int, L: int, device, seq_lens: Sequence[int], remaining: Sequence[int], wait_timeout_s: float = 30.0):
        dev = torch.device(device)
        self.kind, self.L, self.device = int(kind), int(L), dev
        n = self.n = len(seq_lens)
        self.members = np.arange(n, dtype=np.int64)            # ring slot (= row of the group as it started) of each current row
        self.cap = int(max(max(remaining), 0)) + self.L          # a row commits < remaining + L tokens (the last step may overshoot, JD E3)
        i32 = lambda a: torch.tensor(np.asarray(a, dtype=np.int32)).to(dev)
        self.kv_start = i32(np.asarray(seq_lens) - 1)
        self.remaining = i32(remaining)
        self.positions = (self.kv_start.view(n, 1) + torch.arange(self.L, dtype=torch.int32, device=dev).view(1, self.L)).contiguous()
        self.slot = torch.arange(n, dtype=torch.int32, device=dev)
        self.ring = torch.zeros((n, self.cap), dtype=torch.int64, device=dev)
        self.ring_len = torch.zeros((n,), dtype=torch.int32, device=dev)
        self._buf = [torch.empty((n * self.L,), dtype=torch.int64, device=dev) for _ in range(2)]
        self._cur = -1                                         # which buffer holds ``draft`` (-1: neither)
        self.draft: Optional[torch.Tensor] = None
        lib = N.lib()
        self.n_ints = N.EL_HDR + n
        self._mb_key, ptr, self.seq = _MailboxPool.take(dev, self.n_ints)      # (a mailbox of an earlier loop: its numbering continues)
        self._mb_ptr = ptr
        self.mailbox = np.ctypeslib.as_array((C.c_int32 * self.n_ints).from_address(ptr.value))
        self._wait = lib.jf_mailbox_wait
        self.timeout_us = int(wait_timeout_s * 1e6)
        self.flags = MultiblockLoop.publish_flags(dev)
        self.version = 0                                       # bumped by compact(): callers cache per-batch tensors against it
        self._c: Optional[N.EngineLoop] = None
        self.cursors_host = [0, 0, 0]

    @property
    def B(self) -> int:
        return int(self.members.size)

    def close(self) -> None:
        if self._mb_ptr is not None:
            if self.device.type == "cuda":                     # a queued commit may still be mailing
                torch.cuda.synchronize(self.device)
            self.mailbox = None
            _MailboxPool.give(self._mb_key, self._mb_ptr, self.seq)
            self._mb_ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_draft(self, draft: torch.Tensor) -> None:
        self.draft = draft.to(device=self.device, dtype=torch.int64).contiguous()
        self._cur = -1

    def next_buffer(self) -> torch.Tensor:
        """Where the step writes the next draft [B, L]; ``advance`` makes it the draft."""
        k = 1 - self._cur if self._cur >= 0 else 0
        self._nxt = k
        return self._buf[k][:self.B * self.L].view(self.B, self.L)

    def commit(self, rows_dev: torch.Tensor, tokens: torch.Tensor, cursors: Optional[torch.Tensor]) -> None:
        """Queue jf_engine_loop_commit behind the step that wrote ``rows_dev`` / ``tokens`` (and the next draft into
        ``next_buffer()``, which becomes ``draft``)."""
        c = self._c
        if c is None or c.rows != rows_dev.data_ptr() or c.tokens != tokens.data_ptr():
            c = self._c = N.EngineLoop(
                B=self.B, L=self.L, kind=self.kind, ring_cap=self.cap, rows=rows_dev.data_ptr(), tokens=tokens.data_ptr(),
                remaining=self.remaining.data_ptr(), kv_start=self.kv_start.data_ptr(), positions=self.positions.data_ptr(),
                slot=self.slot.data_ptr(), ring=self.ring.data_ptr(), ring_len=self.ring_len.data_ptr(),
                cursors=None if cursors is None else cursors.data_ptr(), n_cursors=0 if cursors is None else int(cursors.numel()),
                flags=int(self.flags), mailbox=self._mb_ptr.value)
        self.seq += 1
        N.check(N.lib().jf_engine_loop_commit(C.byref(c), self.seq, _stream(self.device)), "jf_engine_loop_commit")
        self.draft = self._buf[self._nxt][:self.B * self.L].view(self.B, self.L)
        self._cur = self._nxt

    def wait(self):
        """Poll for the record of the last commit: (n, eos, active_next, fallback) int arrays [B]; ``cursors_host`` is refreshed."""
        rc = self._wait(self._mb_ptr, self.seq, self.timeout_us, _stream(self.device))
        if rc:
            N.check(rc, "jf_mailbox_wait")
        m = self.mailbox
        hdr = m[:N.EL_HDR].tolist()
        if hdr[N.EL_STEP_ERROR]:
            N.check(N.JF_E_LAUNCH, f"engine step (row {hdr[N.EL_STEP_ERROR] - 1}: a workgroup of the step waited 2 s for another: launch incomplete)")
        if hdr[N.EL_ERROR]:
            raise RuntimeError(f"engine loop: the committed-token ring of row {hdr[N.EL_ERROR] - 1} is full (capacity {self.cap})")
        cu = self.cursors_host
        for i in range(3):
            cu[i] = (hdr[N.EL_CURSORS + 2 * i] & 0xFFFFFFFF) | (hdr[N.EL_CURSORS + 2 * i + 1] << 32)
        w = m[N.EL_HDR:N.EL_HDR + self.B].copy()
        return w & 0xFFFF, (w >> 16) & 1, (w >> 17) & 1, (w >> 18) & 1

    def compact(self, keep_rows: np.ndarray) -> None:
        """Keep the rows ``keep_rows`` (positions in the current batch, ascending) — once, when a request has left the group."""
        idx = torch.from_numpy(np.asarray(keep_rows, dtype=np.int64)).to(self.device)
        self.members = self.members[keep_rows]
        self.kv_start = self.kv_start[idx].contiguous()
        self.remaining = self.remaining[idx].contiguous()
        self.positions = self.positions[idx].contiguous()
        self.slot = self.slot[idx].contiguous()
        self.draft = self.draft[idx].contiguous()
        self._cur = -1
        self._c = None
        self.version += 1

    def tokens_host(self):
        """(ring [n, cap] int64, ring_len [n]) on the host: every token the rows have committed, by slot."""
        rl = self.ring_len.cpu().numpy()
        return self.ring.cpu().numpy(), rl


# --------------------------------------------------------------------------------------------
# HF single-block step (SB:197-273)
# --------------------------------------------------------------------------------------------
class SingleBlockStepper:
    """State of one jacobi_forward_greedy call on the device: the draft row, accepted_n_gram and the argmax workspace.
    Per iteration: jf_argmax_partial + jf_sb_step and ONE descriptor read-back (the reference: ~12 small launches and three
    host syncs, SB:199-235)."""

    def __init__(self, input_ids: torch.Tensor, device):
        dev = torch.device(device)
        self.device = dev
        n = int(input_ids.shape[-1])
        self.cap = n
        self.out = input_ids.reshape(-1).to(device=dev, dtype=torch.int64).clone()
        self.acc = input_ids.reshape(-1).to(device=dev, dtype=torch.int64).clone()      # SB:145: accepted_n_gram aliases the input
        self.packed = new_packed(n, dev)
        self.desc_dev = torch.zeros((len(N.SB_FIELDS),), dtype=torch.int32, device=dev)
        self.desc_host = torch.zeros((len(N.SB_FIELDS),), dtype=torch.int32, pin_memory=dev.type == "cuda")
        self.L = n
        self.total = 0

    def draft(self) -> torch.Tensor:
        return self.out[:self.L].view(1, self.L)

    def step(self, logits: torch.Tensor, eos_id: Optional[int], kv_before: int) -> dict:
        """logits [L, V] of the forwarded draft -> the iteration's descriptor (dict of N.SB_FIELDS)."""
        L = self.L
        flat = logits.reshape(-1, logits.shape[-1])
        if flat.shape[0] != L:
            raise ValueError(f"expected logits for {L} positions, got {tuple(logits.shape)}")
        hk = STAGE_HOOK
        hk and hk("sb_body", "begin", flat.shape[0] * flat.shape[1] * flat.element_size())
        argmax_partial(flat, self.packed)
        N.check(N.lib().jf_sb_step(_ptr(self.out), L, _ptr(self.packed), -1 if eos_id is None else int(eos_id), self.total,
                                   self.cap, _ptr(self.acc), int(kv_before), _ptr(self.desc_dev), _stream(self.device)),
                "jf_sb_step")
        hk and hk("sb_body", "end", 0)
        self.desc_host.copy_(self.desc_dev, non_blocking=True)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        d = dict(zip(N.SB_FIELDS, self.desc_host.tolist()))
        self.total, self.L = d["total"], d["next_len"]
        return d


# --------------------------------------------------------------------------------------------
# paged-KV index buffers of one batched Jacobi forward (MR:1204-1265, 965-986)
# --------------------------------------------------------------------------------------------
class PagedFill:
    """The reference's ``jacobi_buffers`` (MR:650-686) and their per-forward fill, as one launch: for the paged layout
    (``Config.kv_cache_layout = "paged"``: engine/model_runner.py) and for callers that keep the reference's paged KV cache +
    varlen attention; the default layout (a static cache row per request) does not need it."""

    def __init__(self, max_batch: int, max_block_len: int, max_blocks_per_seq: int, block_size: int, device):
        dev = torch.device(device)
        self.device, self.block_size = dev, int(block_size)
        self.max_batch, self.max_L, self.max_cols = int(max_batch), int(max_block_len), int(max_blocks_per_seq)
        n = self.max_batch * self.max_L
        self.input_ids = torch.zeros((n,), dtype=torch.int64, device=dev)
        self.positions = torch.zeros((n,), dtype=torch.int64, device=dev)
        self.slot_mapping = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.cu_seqlens_q = torch.zeros((self.max_batch + 1,), dtype=torch.int32, device=dev)
        self.cu_seqlens_k = torch.zeros((self.max_batch + 1,), dtype=torch.int32, device=dev)
        self.cache_seqlens = torch.zeros((self.max_batch,), dtype=torch.int32, device=dev)
        self.block_tables = torch.full((self.max_batch, self.max_cols), -1, dtype=torch.int32, device=dev)
        self.seq_len = torch.zeros((self.max_batch,), dtype=torch.int32, device=dev)
        self.err = torch.zeros((1,), dtype=torch.int32, device=dev)

    def fill(self, draft: torch.Tensor, seq_lens: Sequence[int], block_tables: Sequence[Sequence[int]], check: bool = True):
        """draft [B, L] int64; seq_lens = len(seq) per row; block_tables = seq.block_table per row.  Returns the views
        (input_ids, positions, slot_mapping, cu_seqlens_q, cu_seqlens_k, cache_seqlens, block_tables, max_seqlen_k)."""
        B, L = draft.shape
        if L < 2:
            raise ValueError("Draft must have at least 2 tokens (seed + 1 speculative)")              # MR:1144-1145
        if B > self.max_batch or L > self.max_L:
            raise RuntimeError("PagedFill capacity exceeded")
        for i, S in enumerate(seq_lens):
            if S < 1:
                raise ValueError(f"Sequence {i} has invalid length S={S}. Must be >= 1.")               # MR:1222-1223
            if len(block_tables[i]) > self.max_cols:                                                     # MR:1240-1247
                raise RuntimeError(f"Sequence {i} needs {len(block_tables[i])} blocks but buffer only has {self.max_cols}.")
        bt = np.full((B, self.max_cols), -1, dtype=np.int32)
        for i, t in enumerate(block_tables):
            bt[i, :len(t)] = t
        self.block_tables[:B].copy_(torch.from_numpy(bt), non_blocking=True)
        self.seq_len[:B].copy_(torch.tensor(list(seq_lens), dtype=torch.int32), non_blocking=True)
        draft = draft.to(device=self.device, dtype=torch.int64).contiguous()
        self.err.zero_()
        N.check(N.lib().jf_engine_fill(_ptr(draft), B, L, _ptr(self.seq_len), _ptr(self.block_tables), self.max_cols,
                                       self.block_size, _ptr(self.input_ids), _ptr(self.positions), _ptr(self.slot_mapping),
                                       _ptr(self.cu_seqlens_q), _ptr(self.cu_seqlens_k), _ptr(self.cache_seqlens),
                                       _ptr(self.err), _stream(self.device)), "jf_engine_fill")
        if check and int(self.err.item()):
            raise RuntimeError(f"Sequence {int(self.err.item()) - 1}: a draft position has no KV block "
                               "(Cannot allocate blocks for draft tokens)")                            # MR:1190-1191
        n = B * L
        return (self.input_ids[:n], self.positions[:n], self.slot_mapping[:n], self.cu_seqlens_q[:B + 1],
                self.cu_seqlens_k[:B + 1], self.cache_seqlens[:B], self.block_tables[:B],
                max(int(S) - 1 + L for S in seq_lens))

    def fill_device(self, draft: torch.Tensor, seq_len_dev: torch.Tensor, B: int):
        """The same launch for a caller whose lengths and block tables already live on the device (the engine's chunk loop:
        ``seq_len_dev`` [B] int32 = len(seq) per row, ``self.block_tables[:B]`` kept current by the caller): no host copy, no
        read-back — a position without a block is reported in ``self.err`` (checked by the caller once per chunk)."""
        L = draft.shape[1]
        if B > self.max_batch or L > self.max_L:
            raise RuntimeError("PagedFill capacity exceeded")
        N.check(N.lib().jf_engine_fill(_ptr(draft), B, L, _ptr(seq_len_dev), _ptr(self.block_tables), self.max_cols,
                                       self.block_size, _ptr(self.input_ids), _ptr(self.positions), _ptr(self.slot_mapping),
                                       _ptr(self.cu_seqlens_q), _ptr(self.cu_seqlens_k), _ptr(self.cache_seqlens),
                                       _ptr(self.err), _stream(self.device)), "jf_engine_fill")
        n = B * L
        return self.positions[:n], self.slot_mapping[:n], self.cache_seqlens[:B], self.block_tables[:B]


# --------------------------------------------------------------------------------------------
# non-greedy verify (JDN:299-354, 581-639)
# --------------------------------------------------------------------------------------------
def active_filters(sp, vocab_size: int) -> Tuple[int, float]:
    """(top_k, top_p) a request object asks for, as jf_rs_filter takes them: 0 / 0.0 = that stage is off.  The reference's
    _build_target_probs (JDN:110-123) reads both with getattr — SamplingParams has no such fields (sampling_params.py:4-38), so
    they only exist when a caller planted them on the instance — and switches a stage off for top_k None / <= 0 / >= V (JDN:75)
    and top_p None / <= 0 / >= 1 (JDN:92-96)."""
    if sp is None:
        return 0, 0.0
    top_k, top_p = getattr(sp, "top_k", None), getattr(sp, "top_p", None)
    k = int(top_k) if (top_k is not None and 0 < int(top_k) < int(vocab_size)) else 0
    pp = float(top_p) if (top_p is not None and 0.0 < float(top_p) < 1.0) else 0.0
    return k, pp


class RowFilter:
    """jf_rs_filter on the rows jf_rs_probs has just read: one 48-byte record per row (what top-k / top-p make of an id, as a
    function of its probability and its id) instead of the filtered tensor; the steps take the records (``filt``)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.records = torch.zeros((N.RS_FILTER_ROW_BYTES,), dtype=torch.uint8, device=self.device)
        self.ws = torch.zeros((16,), dtype=torch.uint8, device=self.device)
        self._need: dict = {}

    def run(self, flat: torch.Tensor, draft_next: torch.Tensor, temperature: float, top_k: int, top_p: float, p_draft: torch.Tensor,
            row_max: torch.Tensor, row_sumexp: torch.Tensor) -> torch.Tensor:
        """flat [R, V] logits (jf_rs_probs has filled p_draft / row_max / row_sumexp for them) -> the record tensor."""
        R, V = flat.shape
        lib = N.lib()
        dt = _dtype_code(flat)
        need = self._need.get((dt, R, V))
        if need is None:
            need = self._need[(dt, R, V)] = int(lib.jf_rs_filter_workspace_bytes(dt, R, V))
        self.ws = _grown(self.ws, need)
        self.records = _grown(self.records, R * N.RS_FILTER_ROW_BYTES)
        done = _stage("rs_filter", 2 * R * V * flat.element_size())          # the logits twice (pattern counts, tie ids): no tensor is written
        N.check(lib.jf_rs_filter(_ptr(flat), dt, R, V, flat.stride(0) if R > 1 else V, _ptr(draft_next), float(temperature), int(top_k),
                                 float(top_p), _ptr(self.records), _ptr(p_draft), _ptr(row_max), _ptr(row_sumexp),
                                 _ptr(self.ws) if need else None, self.ws.numel() if need else 0, _stream(self.device)), "jf_rs_filter")
        done()
        return self.records

    def expand(self, flat: torch.Tensor, temperature: float) -> torch.Tensor:
        """The dense tensor of the last ``run`` on ``flat`` (the reference's ``probs``), in the logits' dtype."""
        R, V = flat.shape
        out = torch.empty((R, V), dtype=flat.dtype, device=self.device)
        N.check(N.lib().jf_rs_filter_expand(_ptr(flat), _dtype_code(flat), R, V, flat.stride(0) if R > 1 else V, float(temperature),
                                            _ptr(self.records), _ptr(out), _stream(self.device)), "jf_rs_filter_expand")
        return out


def filtered_probs(logits: torch.Tensor, temperature: float, top_k: int, top_p: float, draft_next: Optional[torch.Tensor] = None):
    """_build_target_probs (JDN:110-123) of logits [R, V] through the library: (probs [R, V] in the logits' dtype, p_draft [R],
    records [R, 48] uint8).  For callers that want the tensor the reference builds — and the parity tests; the decoders never
    materialise it."""
    dev = logits.device
    flat = logits if logits.stride(-1) == 1 else logits.contiguous()
    R, V = flat.shape
    dn = torch.zeros((R,), dtype=torch.int64, device=dev) if draft_next is None else draft_next.to(device=dev, dtype=torch.int64).contiguous()
    f32 = lambda: torch.zeros((R,), dtype=torch.float32, device=dev)
    p_draft, row_max, row_sumexp = f32(), f32(), f32()
    packed = new_packed(R, dev)
    lib = N.lib()
    ws = torch.zeros((int(lib.jf_rs_workspace_bytes(R, V)) // 4 + 4,), dtype=torch.float32, device=dev)
    N.check(lib.jf_rs_probs(_ptr(flat), _dtype_code(flat), R, V, flat.stride(0) if R > 1 else V, _ptr(dn), float(temperature), _ptr(p_draft),
                            _ptr(row_max), _ptr(row_sumexp), _ptr(packed), _ptr(ws), ws.numel() * 4, _stream(dev)), "jf_rs_probs")
    rf = RowFilter(dev)
    rec = rf.run(flat, dn, temperature, top_k, top_p, p_draft, row_max, row_sumexp)
    probs = rf.expand(flat, temperature)
    return probs, p_draft, rec[:R * N.RS_FILTER_ROW_BYTES].view(R, N.RS_FILTER_ROW_BYTES).clone()


class RsStepper:
    """Rejection-sampling verify of a batch of rows: jf_rs_probs (softmax-gather + argmax, logits read once) followed by
    jf_rs_step (accept/reject in stream order, bonus draws, next drafts) and one read-back per iteration.  The logits
    dtype selects the arithmetic: bf16 logits get torch's bf16 rounding points (JDN:64-70 never widens them)."""

    def __init__(self, max_rows: int, max_L: int, device, pad_stream, u_stream, bonus_stream):
        dev = torch.device(device)
        self.device = dev
        self.max_rows, self.max_L = int(max_rows), int(max_L)
        n = self.max_rows * self.max_L
        self.packed = new_packed(n, dev)
        f32 = lambda k: torch.zeros((k,), dtype=torch.float32, device=dev)
        self.p_draft, self.row_max, self.row_sumexp = f32(n), f32(n), f32(n)
        self.ws = torch.zeros((n * 16 * 2,), dtype=torch.float32, device=dev)     # grown to jf_rs_workspace_bytes(R, V) on demand
        self.step_ws = torch.zeros((int(N.lib().jf_rs_step_workspace_bytes(self.max_rows)) // 8 + 2,), dtype=torch.float64, device=dev)
        self.committed = torch.zeros((self.max_rows, self.max_L), dtype=torch.int64, device=dev)
        self.next_draft = torch.zeros((self.max_rows, self.max_L), dtype=torch.int64, device=dev)
        self.rows_dev = torch.zeros((self.max_rows, N.RS_ROW_INTS), dtype=torch.int32, device=dev)
        pin = dev.type == "cuda"
        self.rows_host = torch.zeros((self.max_rows, N.RS_ROW_INTS), dtype=torch.int32, pin_memory=pin)
        self.tok_host = torch.zeros((self.max_rows, self.max_L), dtype=torch.int64, pin_memory=pin)
        self.pad_stream = torch.as_tensor(pad_stream).to(device=dev, dtype=torch.int64).contiguous()
        self.u_stream = torch.as_tensor(u_stream).to(device=dev, dtype=torch.float32).contiguous()
        self.bonus_stream = torch.as_tensor(bonus_stream).to(device=dev, dtype=torch.float32).contiguous()
        self.cursors = torch.zeros((3,), dtype=torch.int64, device=dev)          # uniforms, bonus, pads
        self.remaining = torch.zeros((self.max_rows,), dtype=torch.int32, device=dev)
        self._ws_need: dict = {}
        self.filter: Optional[RowFilter] = None

    def _check(self, draft: torch.Tensor, logits: torch.Tensor) -> Tuple[int, int]:
        B, L = draft.shape
        if L < 2:
            raise ValueError("Draft must have at least 2 tokens (seed + 1 speculative)")
        if logits.ndim != 3 or logits.size(0) != B or logits.size(1) != L - 1:
            raise ValueError(f"forward must return logits [B, L-1, vocab], expected [{B}, {L - 1}, *], got {tuple(logits.shape)}")
        if B > self.max_rows or L > self.max_L:
            raise RuntimeError("RsStepper capacity exceeded")
        return int(B), int(L)

    def _enqueue(self, draft: torch.Tensor, logits: torch.Tensor, temperature: float, eos_id: Optional[int], remaining: torch.Tensor,
                 next_draft: torch.Tensor, top_k: int, top_p: float) -> torch.Tensor:
        """jf_rs_probs [+ jf_rs_filter] + jf_rs_step on draft [B, L] / logits [B, L-1, V]; returns the committed-token view."""
        B, L = draft.shape
        dev = self.device
        V = logits.shape[-1]
        flat = logits.reshape(B * (L - 1), V)
        if flat.stride(1) != 1:
            flat = flat.contiguous()
        R = B * (L - 1)
        draft_next = draft[:, 1:].reshape(-1).contiguous()
        lib = N.lib()
        need = self._ws_need.get((R, V))
        if need is None:
            need = self._ws_need[(R, V)] = int(lib.jf_rs_workspace_bytes(R, V))
        self.ws = _grown(self.ws, need)
        done = _stage("rs_probs", R * V * flat.element_size())
        N.check(lib.jf_rs_probs(_ptr(flat), _dtype_code(flat), R, V, flat.stride(0), _ptr(draft_next), float(temperature),
                                _ptr(self.p_draft), _ptr(self.row_max), _ptr(self.row_sumexp), _ptr(self.packed),
                                _ptr(self.ws), self.ws.numel() * 4, _stream(dev)), "jf_rs_probs")
        done()
        filt = None
        if int(top_k) > 0 or float(top_p) > 0.0:
            if self.filter is None:
                self.filter = RowFilter(dev)
            filt = self.filter.run(flat, draft_next, temperature, top_k, top_p, self.p_draft, self.row_max, self.row_sumexp)
        cm = self.committed.view(-1)[:B * L].view(B, L)
        cur = self.cursors
        c_ptr = lambda i: C.c_void_p(cur.data_ptr() + 8 * i)
        done = _stage("rs_step", B * V * flat.element_size())               # <= one rejected row per draft row
        N.check(lib.jf_rs_step(_ptr(flat), _dtype_code(flat), V, flat.stride(0), _ptr(draft), B, L, _ptr(self.p_draft),
                               _ptr(self.row_max), _ptr(self.row_sumexp), _ptr(self.packed), float(temperature),
                               -1 if eos_id is None else int(eos_id), _ptr(remaining),
                               _ptr(self.u_stream), self.u_stream.numel(), c_ptr(0),
                               _ptr(self.bonus_stream), self.bonus_stream.numel(), c_ptr(1),
                               _ptr(self.pad_stream), self.pad_stream.numel(), c_ptr(2),
                               _ptr(cm), _ptr(next_draft), _ptr(self.rows_dev), _ptr(self.step_ws), self.step_ws.numel() * 8,
                               _ptr(filt), _stream(dev)), "jf_rs_step")
        done()
        return cm

    def step(self, draft: torch.Tensor, logits: torch.Tensor, temperature: float, eos_id: Optional[int],
             remaining: Sequence[int], cursors: Sequence[int], top_k: int = 0, top_p: float = 0.0):
        """``top_k`` / ``top_p`` (``active_filters``): when one is active the rows go through jf_rs_filter — the filtered,
        renormalised distribution as a probability tensor in the logits' dtype (JDN:72-123) — and jf_rs_step samples from that."""
        B, L = self._check(draft, logits)
        dev = self.device
        draft = draft.to(device=dev, dtype=torch.int64).contiguous()
        self.remaining[:B].copy_(torch.tensor(list(remaining), dtype=torch.int32), non_blocking=True)
        self.cursors.copy_(torch.tensor(list(cursors), dtype=torch.int64), non_blocking=True)
        nd = self.next_draft.view(-1)[:B * L].view(B, L)
        cm = self._enqueue(draft, logits, temperature, eos_id, self.remaining, nd, top_k, top_p)
        self.rows_host[:B].copy_(self.rows_dev[:B], non_blocking=True)
        th = self.tok_host.view(-1)[:B * L].view(B, L)
        th.copy_(cm, non_blocking=True)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        rows = self.rows_host[:B].numpy()
        if int(rows[0, N.RS_FIELDS.index("rsv")]) != 0:       # the one-launch step gave up on an in-kernel wait (2 s bound) instead of hanging
            self.rows_dev[0, N.RS_FIELDS.index("rsv")] = 0
            N.check(N.JF_E_LAUNCH, "jf_rs_step (a workgroup of the one-launch step waited 2 s for another: launch incomplete)")
        return rows, th.numpy(), nd

    def step_loop(self, loop: "EngineLoop", logits: torch.Tensor, temperature: float, eos_id: Optional[int], top_k: int = 0,
                  top_p: float = 0.0) -> None:
        """The step on ``loop.draft`` with the loop's device arrays and the commit launch behind it; nothing is read back
        (``loop.wait()``).  The stream cursors stay on the device (``self.cursors``; the record reports them)."""
        self._check(loop.draft, logits)
        cm = self._enqueue(loop.draft, logits, temperature, eos_id, loop.remaining, loop.next_buffer(), top_k, top_p)
        loop.commit(self.rows_dev, cm, self.cursors)


# --------------------------------------------------------------------------------------------
# on-policy rollout step (JDO:270-327 verify + JDO:465-477 re-draft)
# --------------------------------------------------------------------------------------------
class OnPolicyStepper:
    """jf_rs_probs + jf_rs_onpolicy_step for one sequence per call: three launches, one read-back."""

    def __init__(self, max_L: int, device, u_stream, m_stream, stop_ids: Sequence[int]):
        dev = torch.device(device)
        self.device, self.max_L = dev, int(max_L)
        n = self.max_L
        self.packed = new_packed(n, dev)
        f32 = lambda k: torch.zeros((k,), dtype=torch.float32, device=dev)
        self.p_draft, self.row_max, self.row_sumexp = f32(n), f32(n), f32(n)
        self.ws = torch.zeros((n * 16 * 2,), dtype=torch.float32, device=dev)     # grown to jf_rs_workspace_bytes(R, V) on demand
        self.step_ws = torch.zeros((int(N.lib().jf_rs_step_workspace_bytes(n)) // 8 + 2,), dtype=torch.float64, device=dev)
        self.out = torch.zeros((2, n), dtype=torch.int64, device=dev)              # committed, redraft
        self.row_dev = torch.zeros((N.OP_ROW_INTS,), dtype=torch.int32, device=dev)
        pin = dev.type == "cuda"
        self.row_host = torch.zeros((N.OP_ROW_INTS,), dtype=torch.int32, pin_memory=pin)
        self.out_host = torch.zeros((2, n), dtype=torch.int64, pin_memory=pin)
        self.u_stream = torch.as_tensor(u_stream).to(device=dev, dtype=torch.float32).contiguous()
        self.m_stream = torch.as_tensor(m_stream).to(device=dev, dtype=torch.float32).contiguous()
        self.stop_ids = torch.tensor([int(x) for x in stop_ids], dtype=torch.int32, device=dev)
        self.cursors = torch.zeros((2,), dtype=torch.int64, device=dev)             # uniforms, multinomial

    def step(self, proposed: torch.Tensor, logits: torch.Tensor, temperature: float, cursors: Sequence[int], top_k: int = 0,
             top_p: float = 0.0):
        """proposed [R] int64, logits [R, V] -> (row dict, committed list, redraft list [R] (valid from n_committed)).
        ``top_k`` / ``top_p`` (``active_filters``): the rows go through jf_rs_filter first (JDO:99-136, the same filters)."""
        R = int(proposed.numel())
        if logits.dim() != 2 or logits.shape[0] != R:
            raise ValueError(f"forward must return logits [1, {R}, vocab], got {tuple(logits.shape)}")     # JDO:392-393
        if R > self.max_L:
            raise RuntimeError("OnPolicyStepper capacity exceeded")
        dev = self.device
        V = logits.shape[-1]
        flat = logits if logits.stride(1) == 1 else logits.contiguous()
        prop = proposed.to(device=dev, dtype=torch.int64).contiguous()
        lib = N.lib()
        self.ws = _grown(self.ws, int(lib.jf_rs_workspace_bytes(R, V)))
        N.check(lib.jf_rs_probs(_ptr(flat), _dtype_code(flat), R, V, flat.stride(0) if R > 1 else V, _ptr(prop),
                                float(temperature), _ptr(self.p_draft), _ptr(self.row_max), _ptr(self.row_sumexp),
                                _ptr(self.packed), _ptr(self.ws), self.ws.numel() * 4, _stream(dev)), "jf_rs_probs")
        filt = None
        if int(top_k) > 0 or float(top_p) > 0.0:
            if getattr(self, "filter", None) is None:
                self.filter = RowFilter(dev)
            filt = self.filter.run(flat, prop, temperature, top_k, top_p, self.p_draft, self.row_max, self.row_sumexp)
        self.cursors.copy_(torch.tensor(list(cursors), dtype=torch.int64), non_blocking=True)
        cur = self.cursors
        c_ptr = lambda i: C.c_void_p(cur.data_ptr() + 8 * i)
        cm, rd = self.out[0], self.out[1]
        N.check(lib.jf_rs_onpolicy_step(_ptr(flat), _dtype_code(flat), V, flat.stride(0) if R > 1 else V, _ptr(prop), R,
                                        _ptr(self.p_draft), _ptr(self.row_max), _ptr(self.row_sumexp), _ptr(self.packed),
                                        float(temperature), _ptr(self.stop_ids), int(self.stop_ids.numel()),
                                        _ptr(self.u_stream), self.u_stream.numel(), c_ptr(0),
                                        _ptr(self.m_stream), self.m_stream.numel(), c_ptr(1),
                                        _ptr(cm), _ptr(rd), _ptr(self.row_dev), _ptr(self.step_ws),
                                        self.step_ws.numel() * 8, _ptr(filt), _stream(dev)), "jf_rs_onpolicy_step")
        self.row_host.copy_(self.row_dev, non_blocking=True)
        self.out_host[:, :R].copy_(self.out[:, :R], non_blocking=True)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        row = dict(zip(N.OP_FIELDS, self.row_host.tolist()))
        n = row["n_committed"]
        return row, self.out_host[0, :n].tolist(), self.out_host[1, :R].tolist()
