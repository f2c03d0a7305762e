"""ctypes binding of the C-ABI library (include/jacobiforcing.h).

The HIP library is the only implementation of the hot path this package has.  If it cannot be
loaded the package fails loudly — there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libjacobiforcing.so"
SRC_DIR = _PKG / "csrc"
INCLUDE_DIR = _PKG.parent / "include"

JF_OK, JF_E_INVALID, JF_E_CAPACITY, JF_E_LAUNCH, JF_E_SHAPE = 0, -1, -2, -3, -4
JF_F32, JF_BF16 = 0, 1
JF_MB_INACTIVE, JF_MB_KEEP = -1, -2


class MbParams(C.Structure):
    _fields_ = [("n", C.c_int32), ("K", C.c_int32), ("spawn_threshold", C.c_int32), ("pool_size", C.c_int32),
                ("eos_id", C.c_int32), ("pad_id", C.c_int32), ("max_iter", C.c_int32), ("max_blocks", C.c_int32),
                ("lookahead_start_ratio", C.c_double)]


class MbDesc(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("B", "T", "done", "error", "iters", "kv_len", "ret_len", "next_token", "kv_src_row", "kv_copy_dst",
                 "kv_copy_len", "events", "accepted", "nspans", "rsv0", "rsv1")]


DESC_INTS = C.sizeof(MbDesc) // 4
DESC_FIELDS = [f[0] for f in MbDesc._fields_]


class MbLoop(C.Structure):
    """jf_mb_loop (include/jacobiforcing.h): the pointers of the loop around the step, filled once."""
    _fields_ = [("states", C.c_void_p), ("state_ints", C.c_int64), ("P", C.c_int32), ("order", C.c_int32),
                ("packed", C.c_void_p), ("packed_cap", C.c_int64), ("desc", C.c_void_p),
                ("input_ids", C.c_void_p), ("positions", C.c_void_p), ("row_prompt", C.c_void_p), ("row_len", C.c_void_p),
                ("row_cand", C.c_void_p), ("row_kv_len", C.c_void_p), ("valid_index", C.c_void_p),
                ("rows_cap", C.c_int32), ("t_cap", C.c_int32), ("t_align", C.c_int32), ("valid_align", C.c_int32),
                ("cand_rows", C.c_int32), ("flags", C.c_int32), ("pad_fill", C.c_int64),
                ("kv_len", C.c_void_p), ("mailbox", C.c_void_p),
                ("drv", C.c_void_p), ("drv_ints", C.c_int64), ("draws", C.c_void_p), ("draw_len", C.c_int32),
                ("max_seq_len", C.c_int32)]


# mailbox header slots / per-prompt driver record / driver block header (include/jacobiforcing.h)
MB_SEQ, MB_RTOT, MB_RMAIN, MB_TPAD, MB_TMAX, MB_NVALID, MB_NVALID_PAD, MB_NDONE, MB_MAXKV, MB_ERROR, MB_ACCEPTED, MB_NCALL_END = range(12)
MB_MAILBOX_HDR, MB_FIN_INTS = 16, 8
MB_PACKED_EXTRA = 65536            # JF_MB_PACKED_ENTRIES(positions) - positions: room for jf_mb_verify's per-chunk result slots
FIN_FIELDS = ["stop", "calls", "iters_total", "new_tokens", "ret_len", "next_token", "iters", "text_off"]
DRV_FIELDS = ["active", "stop", "calls", "iters_total", "new_tokens", "budget", "max_calls", "text_len", "cursor", "fin_ret_len",
              "fin_next", "fin_iters", "fin_off"]
DRV_HDR_INTS = 16
STOP_REASONS = {0: None, 1: "eos", 2: "max_new_tokens", 3: "max_calls", 4: "max_seq_len", 5: "max_seq_len"}
EVT_SPAWN, EVT_SWITCH, EVT_EARLY, EVT_CALL_END, EVT_STOPPED, EVT_FAST, EVT_SLOW_NEXT = 1, 2, 4, 8, 16, 32, 64


def mailbox_ints(P: int) -> int:
    return MB_MAILBOX_HDR + int(P) * (DESC_INTS + MB_FIN_INTS)


class EngineRow(C.Structure):
    _fields_ = [("acc_len", C.c_int32), ("n_new", C.c_int32), ("eos", C.c_int32), ("active_next", C.c_int32),
                ("n_pads", C.c_int32), ("rsv", C.c_int32 * 3)]


ENGINE_ROW_INTS = C.sizeof(EngineRow) // 4
MB_LOOP_PUBLISH_FENCE = 1        # jf_mb_loop.flags (JF_MB_LOOP_PUBLISH_FENCE)
ENGINE_FIELDS = ["acc_len", "n_new", "eos", "active_next", "n_pads", "rsv0", "rsv1", "rsv2"]   # int32 columns of a row record


class RsRow(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n_committed", "eos", "reject_pos", "n_bonus_draws", "n_uniforms", "n_pads",
                                         "active_next", "rsv")]


class RsFilterRow(C.Structure):
    """jf_rs_filter_row (include/jacobiforcing.h): what top-k / top-p make of one row, as a function of (probability, id)."""
    _fields_ = [("sum", C.c_double), ("row_max", C.c_float), ("x_keep", C.c_float), ("cut1", C.c_uint32), ("tie1", C.c_int32),
                ("s1", C.c_float), ("cut2", C.c_uint32), ("tie2", C.c_int32), ("s2", C.c_float), ("flags", C.c_uint32),
                ("rsv", C.c_uint32)]


RS_FILTER_ROW_BYTES = C.sizeof(RsFilterRow)
RS_FILTER_FLAGS_WORD = RsFilterRow.flags.offset // 4
RS_FILT_TOPK, RS_FILT_TOPP = 1, 2
RS_ROW_INTS = C.sizeof(RsRow) // 4
RS_FIELDS = [f[0] for f in RsRow._fields_]


class OpRow(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n_committed", "stop_hit", "reject_pos", "n_bonus_draws", "n_uniforms", "n_redraft",
                                         "redraft_base_lo", "redraft_base_hi")]


class SbDesc(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("raw", "num", "total", "done", "next_token", "kv_len", "next_len", "eos")]


SB_FIELDS = [f[0] for f in SbDesc._fields_]


class EngineLoop(C.Structure):
    """jf_engine_loop (include/jacobiforcing.h): the device arrays of the loop around jf_engine_step / jf_rs_step."""
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("kind", C.c_int32), ("ring_cap", C.c_int32),
                ("rows", C.c_void_p), ("tokens", C.c_void_p), ("remaining", C.c_void_p), ("kv_start", C.c_void_p),
                ("positions", C.c_void_p), ("slot", C.c_void_p), ("ring", C.c_void_p), ("ring_len", C.c_void_p),
                ("cursors", C.c_void_p), ("n_cursors", C.c_int32), ("flags", C.c_int32), ("mailbox", C.c_void_p)]


EL_SEQ, EL_ERROR, EL_STEP_ERROR, EL_CURSORS, EL_HDR = 0, 1, 2, 4, 16
EL_KIND_GREEDY, EL_KIND_SAMPLING = 0, 1
OP_ROW_INTS = C.sizeof(OpRow) // 4
OP_FIELDS = [f[0] for f in OpRow._fields_]

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

_SIGNATURES = {
    "jf_version": (C.c_int, []),
    "jf_timing_arm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jf_last_error": (C.c_char_p, []),
    "jf_device_identity": (C.c_int, [C.c_int, C.c_char_p, _sz]),
    "jf_argmax_partial": (C.c_int, [_vp, C.c_int, _i64, _i64, _i64, _vp, _vp]),
    "jf_argmax_scatter": (C.c_int, [_vp, C.c_int, _i64, _i64, _i64, _vp, _vp, _vp]),
    "jf_argmax_decode": (C.c_int, [_vp, _i64, _vp, _vp]),
    "jf_argmax_rows": (C.c_int, [_vp, C.c_int, _i64, _i64, _i64, _vp, _vp, _vp]),
    "jf_accept_lengths": (C.c_int, [_vp, C.c_int, _vp, _i64, C.c_int, C.c_int, _vp, _vp, _vp]),
    "jf_mb_state_ints": (_i64, [C.POINTER(MbParams)]),
    "jf_mb_max_rows": (_i32, [C.POINTER(MbParams)]),
    "jf_mb_max_tokens": (_i32, [C.POINTER(MbParams)]),
    "jf_mb_begin": (C.c_int, [_vp, _i64, C.c_int, C.POINTER(MbParams), _vp, _vp, _vp, _vp]),
    "jf_mb_pack": (C.c_int, [_vp, _i64, C.c_int, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "jf_mb_step": (C.c_int, [_vp, _i64, C.c_int, _vp, _i64, _vp, _vp]),
    "jf_mb_verify": (C.c_int, [_vp, C.c_int, _i64, _i64, _i64, _vp, _vp, _i64, C.c_int, _vp, _i64, _i64, _i32, _vp,
                               C.POINTER(MbParams), _vp]),
    "jf_mb_read_ret": (C.c_int, [_vp, _i64, C.c_int, _vp, _i32, _vp]),
    "jf_mb_set_fast_path": (C.c_int, [C.c_int]),
    "jf_host_alloc": (C.c_int, [_sz, C.POINTER(C.c_void_p)]),
    "jf_host_free": (C.c_int, [_vp]),
    "jf_mailbox_wait": (C.c_int, [_vp, _i32, _i64, _vp]),
    "jf_mb_loop_begin": (C.c_int, [C.POINTER(MbLoop), _i32, C.POINTER(MbParams), _vp, _vp, _vp]),
    "jf_mb_loop_iterate": (C.c_int, [C.POINTER(MbLoop), _i32, _vp, C.c_int, _i64, _i64, _i64, C.c_int, _i32, _i32,
                                     C.POINTER(MbParams), C.c_int, _vp, _vp, _vp]),
    "jf_mb_loop_pack": (C.c_int, [C.POINTER(MbLoop), _i32, C.POINTER(MbParams), _vp]),
    "jf_kv_append": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i64, _i32, _vp]),
    "jf_rope_kv_append": (C.c_int, [_vp, C.c_int, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp,
                                    _vp, _i64, _vp]),
    "jf_swiglu": (C.c_int, [_vp, C.c_int, _i64, _i64, _vp, _vp]),
    "jf_kv_commit": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, C.c_int, _i32, _i32, _i32, _i64, _i64, _i32, _vp]),
    "jf_engine_step": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "jf_sb_step": (C.c_int, [_vp, C.c_int, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "jf_engine_loop_commit": (C.c_int, [C.POINTER(EngineLoop), _i32, _vp]),
    "jf_engine_fill": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "jf_rs_probs": (C.c_int, [_vp, C.c_int, _i64, _i64, _i64, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "jf_rs_filter_workspace_bytes": (_sz, [C.c_int, _i64, _i64]),
    "jf_rs_filter": (C.c_int, [_vp, C.c_int, _i64, _i64, _i64, _vp, _f32, _i32, C.c_double, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "jf_rs_filter_expand": (C.c_int, [_vp, C.c_int, _i64, _i64, _i64, _f32, _vp, _vp, _vp]),
    "jf_rs_workspace_bytes": (_sz, [_i64, _i64]),
    "jf_rs_step": (C.c_int, [_vp, C.c_int, _i64, _i64, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _f32, _i32, _vp,
                             _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "jf_rs_step_workspace_bytes": (_sz, [_i64]),
    "jf_rs_onpolicy_step": (C.c_int, [_vp, C.c_int, _i64, _i64, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_float, _vp, C.c_int,
                                      _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_LIB = None


class NativeLibraryError(RuntimeError):
    pass


def load(path: os.PathLike | None = None):
    """dlopen the library and attach signatures.  Raises NativeLibraryError when it is missing."""
    p = Path(path) if path is not None else Path(os.environ.get("JF_LIB", LIB_PATH))   # JF_LIB: kernel A/B builds (tools/)
    if not p.exists():
        raise NativeLibraryError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "jacobiforcing_amd has no CPU / eager fallback for the Jacobi loop body.")
    try:
        lib = C.CDLL(str(p))
    except OSError as e:  # e.g. libamdhip64 missing
        raise NativeLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


def lib():
    global _LIB
    if _LIB is None:
        _LIB = load()
    return _LIB


def check(rc: int, what: str = ""):
    """Map C-ABI return codes onto the exception types the reference raises."""
    if rc == JF_OK:
        return
    msg = lib().jf_last_error()
    msg = msg.decode() if isinstance(msg, (bytes, bytearray)) else str(msg)
    text = f"{what}: {msg}" if what else msg
    if rc == JF_E_INVALID:
        raise ValueError(text)
    raise RuntimeError(text)


def raise_state_error(code: int, what: str, aux: int = 0):
    if code == JF_E_INVALID:
        raise ValueError(f"{what}: invalid state inside the Jacobi state machine (shape/assert; see MB:482, MB:631, MB:667)")
    if code == JF_E_SHAPE:
        # what torch raises at MB:482 when a re-surfaced pseudo block (Q3, K >= 3) has k rows and the RA draft has B
        raise RuntimeError(f"{what}: The size of tensor a ({aux >> 16}) must match the size of tensor b ({aux & 0xFFFF}) at "
                           "non-singleton dimension 0 (draft rows vs candidate rows, MB:482)")
    if code == JF_E_CAPACITY:
        raise RuntimeError(f"{what}: fixed capacity exceeded inside the Jacobi state machine (raise max_blocks)")
    if code:
        raise RuntimeError(f"{what}: state machine error {code}")
