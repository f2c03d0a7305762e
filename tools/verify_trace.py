#!/usr/bin/env python3
"""Timeline of the fused verify launch (jf_mb_verify) at 1 / 8 / 64 prompts: synthetic bf16 logits at V = 152064, the state
machines advance with whatever those logits accept.  Prints HIP-event microseconds of the fused launch and of the two
launches it replaces; with the experiment build (-DJF_EXP_VERIFY_TRACE, JF_LIB=tools/exp/libjf_exp_vtrace.so) also the
in-kernel stamps: when the items ran and what each stepper did after its rows arrived.

    python tools/verify_trace.py [--prompts 1 8 64] [--iters 12]

Experiment build (run in the repo root; the .so is git-ignored and travels with gpurun):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DJF_EXP_VERIFY_TRACE -Iinclude \
          -Ijacobiforcing_amd/csrc jacobiforcing_amd/csrc/*.hip -o tools/exp/libjf_exp_vtrace.so
"""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import _native as N, ops  # noqa: E402

V = 152064


def run(P, iters, fused, trace_lib=None, loop=False):
    prm = ops.MultiblockParams(n=32, K=2, r=0.85, n_gram_pool_size=4, eos_token_id=None, pad_token_id=151643)
    batch = ops.MultiblockBatch(P, prm, "cuda")
    batch.fused = fused
    g = torch.Generator(device="cuda").manual_seed(7)
    ids = torch.randint(0, 151000, (P, 32), generator=g, device="cuda")
    lp = None
    if loop:                                                      # the loop API: kv_len on the device, mailbox, pack queued behind
        kvl = torch.zeros(P, dtype=torch.int32, device="cuda")
        lp = ops.MultiblockLoop(batch, kv_len=kvl, t_cap=128, t_align=8, valid_align=8 * P, compact=True, cand_rows=3, order=1,
                                max_seq_len=1 << 20)
        sm = lp.begin(ids, torch.full((P,), 200, dtype=torch.int32))
    else:
        d = batch.begin(ids, torch.full((P,), 200, dtype=torch.int32))
    times, rows_l = [], []
    stamps = None
    mstamps = None
    for it in range(iters):
        if loop:
            if sm.Rtot == 0:
                break
        else:
            pk = batch.pack(d, t_align=8, compact=True, valid_align=8 * P)
            if pk is None:
                break
        nv = lp.valid_index().numel() if loop else batch.valid_index.numel()
        logits = torch.randn(nv, V, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
        _ = torch.zeros(64 << 20, device="cuda").sum()          # push the fresh logits out of the caches a little
        torch.cuda.synchronize()
        if trace_lib is not None and it == iters - 1:
            trace_lib.jf_exp_reset_vtrace()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.VERIFY_HOOK = (lambda *_: a.record(), lambda *_: b.record())
        if loop:
            lp.iterate(logits)
            sm = lp.wait()
            d = sm.d
        else:
            d = batch.verify(logits, compacted=True)
        ops.VERIFY_HOOK = None
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) * 1e3)
        rows_l.append(batch.Nvalid)
        if trace_lib is not None and it == iters - 1:
            buf = (C.c_ulonglong * (2 + 8 * 256))()
            trace_lib.jf_exp_read_vtrace(buf, 2 + 8 * 256)
            stamps = np.array(buf[:], dtype=np.uint64)
            if hasattr(trace_lib, "jf_exp_read_mtrace"):
                mb_ = (C.c_ulonglong * (16 * 256))()
                trace_lib.jf_exp_read_mtrace(mb_, 16 * 256)
                mstamps = np.array(mb_[:], dtype=np.uint64).reshape(256, 16)
            ib = (C.c_ulonglong * (2 * 8192))()
            trace_lib.jf_exp_read_vitems(ib, 2 * 8192)
            it_ = np.array(ib[:], dtype=np.uint64).reshape(-1, 2)
            it_ = it_[it_[:, 0] > 0]
            stamps[0], stamps[1] = it_[:, 0].min(), it_[:, 1].max()
            item_us = (it_[:, 1] - it_[:, 0]).astype(np.float64) / 100.0
            starts = (it_[:, 0] - it_[:, 0].min()).astype(np.float64) / 100.0
            print(f"      item workgroups {len(it_)}: duration mean {item_us.mean():.1f} us (min {item_us.min():.1f}, max {item_us.max():.1f}); "
                  f"starts 0..{starts.max():.1f} us (median {np.median(starts):.1f})")
        done = batch.desc_field(d, "done")
        if done.any():                                            # restart finished calls (rolling, like the decoder)
            kv = np.where(done == 1, batch.desc_field(d, "kv_len"), N.JF_MB_KEEP).astype(np.int32)
            if loop:
                sm = lp.begin(ids, torch.from_numpy(kv))
            else:
                d = batch.begin(ids, torch.from_numpy(kv))
    t = np.array(times[2:])
    r = np.array(rows_l[2:])
    mb = r.mean() * V * 2 / 1e6
    print(f"P={P:3d} {'loop   ' if loop else 'fused  ' if fused else 'unfused'} rows/launch {r.mean():7.1f} ({mb:6.1f} MB)  {t.mean():6.1f} us (min {t.min():6.1f})  "
          f"{mb / t.mean() * 1e3 / 1e3:6.2f} TB/s = {mb / t.mean() / 8:5.3f} of 8 TB/s", flush=True)
    if stamps is not None:
        t0 = int(stamps[0])
        rel = lambda x: (int(x) - t0) / 100.0                     # 100 MHz ticks -> us
        print(f"      items: first start 0.0 us, last end {rel(stamps[1]):.1f} us")
        for p in list(range(min(P, 3))) + ([P - 1] if P > 3 else []):
            s = stamps[2 + 8 * p: 2 + 8 * p + 8]
            print(f"      stepper {p:3d}: start {rel(s[0]):6.1f}  image {rel(s[1]):6.1f}  arrived {rel(s[2]):6.1f}  gathered {rel(s[3]):6.1f}  "
                  f"stepped {rel(s[4]):6.1f}  written(w1-3) {rel(s[5]):6.1f}  desc-out {rel(s[6]):6.1f}  end {rel(s[7]):6.1f}")
        if mstamps is not None:                                     # state machine phases of the last stepper (last visit of each stamp)
            names = {1: "scalars", 2: "accept-scan", 3: "commit", 4: "re-draft", 5: "pool-push", 6: "candidates", 7: "spans-done",
                     8: "spawn/promote", 9: "early-stop", 10: "build_out", 11: "scalars-stored",
                     12: "fast64:accept-scan", 13: "fast64:checks", 14: "fast64:pool-push", 15: "fast64:candidates"}
            q = P - 1
            row = [(rel(mstamps[q, k]), names[k]) for k in names if mstamps[q, k] >= stamps[2 + 8 * q + 3]]
            row.sort()
            print("      machine of stepper %d: " % q + "  ".join(f"{nm} {t:.2f}" for t, nm in row))
        if mstamps is not None:                                     # the straight-line step's stages over all steppers that took it
            rows_ = []
            for q in range(P):
                g_, st_ = stamps[2 + 8 * q + 3], stamps[2 + 8 * q + 4]
                m_ = mstamps[q]
                if all(g_ <= m_[k] <= st_ for k in (12, 13, 14, 15)) and m_[12] <= m_[13] <= m_[14] <= m_[15]:
                    rows_.append([(int(m_[12]) - int(g_)) / 100, (int(m_[13]) - int(m_[12])) / 100, (int(m_[14]) - int(m_[13])) / 100,
                                  (int(m_[15]) - int(m_[14])) / 100, (int(st_) - int(m_[15])) / 100])
            tw = [((int(mstamps[q, 0]) - int(stamps[2 + 8 * q + 3])) / 100, (int(stamps[2 + 8 * q + 4]) - int(mstamps[q, 0])) / 100)
                  for q in range(P) if stamps[2 + 8 * q + 3] <= mstamps[q, 0] <= stamps[2 + 8 * q + 4]]
            if tw:                                                    # -DJF_EXP_STEP_TWICE with JF_EXP_TWICE=1
                tw = np.array(tw)
                print(f"      step run twice through the same instructions ({len(tw)} steppers): first pass {tw[:, 0].mean():.2f} us "
                      f"(max {tw[:, 0].max():.2f}), second pass {tw[:, 1].mean():.2f} us (max {tw[:, 1].max():.2f})")
            if rows_:
                r_ = np.array(rows_).mean(axis=0)
                print(f"      straight-line step ({len(rows_)} steppers, mean us): gathered->accept-scan {r_[0]:.2f}  checks {r_[1]:.2f}  "
                      f"commit+re-draft+pool {r_[2]:.2f}  candidates {r_[3]:.2f}  header+descriptor+kv_len {r_[4]:.2f}")
        ends = np.array([max(rel(stamps[2 + 8 * p + 7]), rel(stamps[2 + 8 * p + 5])) for p in range(P)])
        arr = np.array([rel(stamps[2 + 8 * p + 2]) for p in range(P)])
        print(f"      all steppers: arrived {arr.min():.1f}..{arr.max():.1f} us, end {ends.min():.1f}..{ends.max():.1f} us")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompts", type=int, nargs="+", default=[1, 8, 64])
    ap.add_argument("--iters", type=int, default=12)
    a = ap.parse_args()
    trace_lib = None
    lib = N.lib()
    if hasattr(lib, "jf_exp_read_vtrace"):
        trace_lib = lib
        lib.jf_exp_read_vtrace.argtypes = [C.c_void_p, C.c_int]
        lib.jf_exp_read_vitems.argtypes = [C.c_void_p, C.c_int]
        if hasattr(lib, "jf_exp_read_mtrace"):
            lib.jf_exp_read_mtrace.argtypes = [C.c_void_p, C.c_int]
    for P in a.prompts:
        run(P, a.iters, False)
        run(P, a.iters, True, trace_lib)
        run(P, a.iters, True, trace_lib, loop=True)


if __name__ == "__main__":
    main()
