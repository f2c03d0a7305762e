#!/usr/bin/env python3
"""In-kernel stamps of the convergence launch INSIDE the decode step of bench.py's model (logits just written by the lm_head
GEMM), next to tools/verify_trace.py (synthetic logits nobody has just written).  Needs the experiment build:

    tools/build_exp.sh vtrace -DJF_EXP_VERIFY_TRACE
    JF_LIB=tools/exp/libjf_exp_vtrace.so python tools/verify_trace_insitu.py [--prompts 64] [--iters 24]

Every traced iteration resets the stamp buffers before the launch (a blocking copy: the launch then starts from an idle
queue, the memory system is as the forward left it) and reads them after it."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from jacobiforcing_amd import _native, ops  # noqa: E402
from jacobiforcing_amd.engine.multiblock_decoder import MultiblockJacobiDecoder  # noqa: E402
from jacobiforcing_amd.modeling.qwen2 import Qwen2Config, Qwen2Model, Qwen2Weights  # noqa: E402
from jacobiforcing_amd.synthetic import ScriptedAcceptance, humaneval_shaped_prompts  # noqa: E402
from jacobiforcing_amd.tuning import enable_tuned_gemms, grid_alignment  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompts", type=int, default=64)
    ap.add_argument("--iters", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--scripted", action="store_true")
    ap.add_argument("--before", choices=["none", "touch-pages", "sweep-other", "reread", "idle-5ms", "idle-50ms"], default="none",
                    help="what runs between the forward and the traced launch: touch-pages = one element per 4 KB of the logits "
                         "(address translations warm, caches as the forward left them); sweep-other = a read-only pass over an "
                         "unrelated 1 GB buffer (the producer's dirty lines are written back before the launch); reread = a "
                         "read-only pass over the logits themselves; idle-N = the GPU idles that long before the launch")
    a = ap.parse_args()
    lib = _native.lib()
    if not hasattr(lib, "jf_exp_read_vtrace"):
        raise SystemExit("needs JF_LIB=tools/exp/libjf_exp_vtrace.so (tools/build_exp.sh vtrace -DJF_EXP_VERIFY_TRACE)")
    dev = torch.device("cuda", 0)
    tuned = enable_tuned_gemms()
    cfg = Qwen2Config.qwen2_5_coder_7b()
    model = Qwen2Model(cfg, Qwen2Weights(cfg, dev, dtype=torch.bfloat16, seed=0))
    prm = ops.MultiblockParams(n=32, K=2, r=0.85, lookahead_start_ratio=0.0, n_gram_pool_size=4, eos_token_id=None,
                               pad_token_id=cfg.pad_token_id)
    P = a.prompts
    vocab_hi = min(151643, cfg.vocab_size - 2)
    prompts = humaneval_shaped_prompts(P, seed=1234, vocab_hi=vocab_hi)
    ta, la = grid_alignment(P, tuned)
    dec = MultiblockJacobiDecoder(model, P, prm, max_seq_len=4096, t_align=ta, logit_align=la)
    if a.scripted:
        dec.logits_hook = ScriptedAcceptance(cfg.vocab_size, robust_pct=82, vocab_hi=vocab_hi)
    rows, steps, flagged, pred, anat = [], [], [], {}, []
    other = torch.zeros(1 << 29, dtype=torch.bfloat16, device=dev) if a.before == "sweep-other" else None
    state = dict(i=0)

    def before(b, flat):
        state["i"] += 1
        if state["i"] > a.warmup:
            torch.cuda.synchronize()
            if a.before == "touch-pages":
                state["sink"] = flat.view(-1)[::2048].float().sum()
            elif a.before == "sweep-other":
                state["sink"] = other.sum()
            elif a.before == "reread":
                state["sink"] = flat.float().amax()
            elif a.before.startswith("idle"):
                __import__("time").sleep(0.005 if a.before == "idle-5ms" else 0.05)
            torch.cuda.synchronize()
            state["flag"] = (b.desc_dev.cpu().numpy().reshape(P, -1)[:, _native.DESC_FIELDS.index("events")] & _native.EVT_SLOW_NEXT) != 0
            lib.jf_exp_reset_vtrace()

    def after(b, flat):
        if state["i"] <= a.warmup:
            return
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (2 + 8 * 256))()
        lib.jf_exp_read_vtrace(buf, 2 + 8 * 256)
        st = np.array(buf[:], dtype=np.uint64)
        ib = (C.c_ulonglong * (2 * 8192))()
        lib.jf_exp_read_vitems(ib, 2 * 8192)
        it = np.array(ib[:], dtype=np.uint64).reshape(-1, 2)
        it = it[it[:, 0] > 0]
        t0 = int(it[:, 0].min())
        dur = (it[:, 1] - it[:, 0]).astype(np.float64) / 100.0
        real = it[dur > 1.0]                                      # list-padding rows leave at once
        ends = [max(int(st[2 + 8 * p + 7]), int(st[2 + 8 * p + 5])) for p in range(P)]
        arrived = [int(st[2 + 8 * p + 2]) for p in range(P)]
        ev = b.desc_dev.cpu().numpy().reshape(P, -1)[:, _native.DESC_FIELDS.index("events")]
        fl = state["flag"]
        arr_us = np.array([(x - t0) / 100.0 for x in arrived])
        live = np.array([x > 0 for x in arrived])
        if (fl & live).any() and (~fl & live).any():
            flagged.append((arr_us[fl & live].mean(), arr_us[~fl & live].mean(), int((fl & live).sum())))
        for p in range(P):
            if arrived[p] > 0:
                pred[(bool(fl[p]), "fast" if ev[p] & _native.EVT_FAST else "other")] = pred.get((bool(fl[p]), "fast" if ev[p] & _native.EVT_FAST else "other"), 0) + 1
                cls = ("call end" if ev[p] & _native.EVT_CALL_END else "fast" if ev[p] & _native.EVT_FAST else "general")
                steps.append((cls, (int(st[2 + 8 * p + 4]) - int(st[2 + 8 * p + 3])) / 100.0, (ends[p] - arrived[p]) / 100.0,
                              (ends[p] - t0) / 100.0 > (max(ends) - t0) / 100.0 - 0.05))
        # launch anatomy (small shapes): when the item workgroups start, how long one lasts, and the last stepper's phases
        last = int(np.argmax(ends))
        ph = [int(st[2 + 8 * last + k]) for k in range(8)]
        anat.append(dict(start_med=float(np.median(it[:, 0] - t0)) / 100.0, start_max=float((it[:, 0] - t0).max()) / 100.0,
                         dur_med=float(np.median(dur[dur > 1.0])) if (dur > 1.0).any() else 0.0, dur_max=float(dur.max()),
                         stepper_start=(ph[0] - t0) / 100.0, image=(ph[1] - t0) / 100.0, seen=(ph[2] - t0) / 100.0, gathered=(ph[3] - t0) / 100.0,
                         stepped=(ph[4] - t0) / 100.0, written=(max(ph[5], ph[6]) - t0) / 100.0, end=(ends[last] - t0) / 100.0))
        rows.append(dict(valid=dec.last_valid_rows, launched=flat.shape[0], items=len(it), real=len(real),
                         items_end=(int(it[:, 1].max()) - t0) / 100.0, last_start=(int(it[:, 0].max()) - t0) / 100.0,
                         arrived=(max(arrived) - t0) / 100.0, end=(max(ends) - t0) / 100.0))

    ops.VERIFY_HOOK = (before, after)
    bench.run_steps(dec, prompts, 0, a.warmup + a.iters, seed=1234)
    ops.VERIFY_HOOK = None
    V = cfg.vocab_size
    print(f"# in situ, {P} prompts per GPU{' (scripted acceptance)' if a.scripted else ''}, before the launch: {a.before}: us after the first item start")
    print("# valid rows  logits rows   MB    items end   last stepper saw its rows   launch end    stream TB/s   whole TB/s")
    for r in rows:
        mb = r["valid"] * V * 2 / 1e6
        print(f"  {r['valid']:7d}   {r['launched']:7d}   {mb:6.1f}   {r['items_end']:7.1f}   {r['arrived']:10.1f}   {r['end']:14.1f}   "
              f"{mb / r['items_end']:10.2f}   {mb / r['end']:8.2f}")
    mbs = np.array([r["valid"] * V * 2 / 1e6 for r in rows]); ie = np.array([r["items_end"] for r in rows]); en = np.array([r["end"] for r in rows])
    print(f"# mean: {mbs.mean():.1f} MB, items end {ie.mean():.1f} us ({mbs.mean() / ie.mean():.2f} TB/s), launch end {en.mean():.1f} us "
          f"({mbs.mean() / en.mean():.2f} TB/s = {mbs.mean() / en.mean() / 8:.3f} of 8 TB/s); tail {np.mean(en - ie):.1f} us")
    if anat:
        m = lambda k: float(np.mean([x[k] for x in anat]))
        print(f"# anatomy (means over the launches, us after the first item start): item workgroups start  median {m('start_med'):.1f} / last {m('start_max'):.1f}; "
              f"one item lasts  median {m('dur_med'):.1f} / longest {m('dur_max'):.1f}")
        print(f"#   the launch's last stepper: starts {m('stepper_start'):.1f}, image in LDS {m('image'):.1f}, sees its last row {m('seen'):.1f}, tokens gathered {m('gathered'):.1f}, "
              f"stepped {m('stepped'):.1f}, written back / descriptor out {m('written'):.1f}, end {m('end'):.1f}")
    print("# steppers by what their step was: count, gathered -> stepped us (mean / max), rows seen -> end us (mean / max), times it was the launch's last")
    for cls in ("fast", "general", "call end"):
        sel = [x for x in steps if x[0] == cls]
        if sel:
            st_ = np.array([x[1] for x in sel]); tl = np.array([x[2] for x in sel])
            print(f"#   {cls:9s} {len(sel):5d}   step {st_.mean():5.1f} / {st_.max():5.1f}   tail {tl.mean():5.1f} / {tl.max():5.1f}   last {sum(x[3] for x in sel)}")
    if flagged:
        f = np.array(flagged)
        print(f"# prompts listed first (EVT_SLOW_NEXT): {f[:, 2].mean():.1f} per launch; they saw their rows at {f[:, 0].mean():.1f} us, the others at {f[:, 1].mean():.1f} us")
        print("# prediction: " + "  ".join(f"flagged={k[0]} step={k[1]}: {v}" for k, v in sorted(pred.items())))


if __name__ == "__main__":
    main()

