#!/usr/bin/env python3
"""jf_rs_probs + jf_rs_step at BASELINE config 5's shape (batch 64, block 32, V = 152064): every row rejects somewhere, so
the bonus path (segment sums + inverse-CDF pick + repair) runs for all 64 rows.  Prints the HIP-event time of the step
(all launches of jf_rs_step) and of jf_rs_probs; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split.

    python tools/microbench_rs_step.py [--dtype bf16|f32] [--temperature 1.0] [--batch 64] [--block 32]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--block", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--p-hit", type=float, default=0.6)
    ap.add_argument("--trace", action="store_true", help="in-kernel stamps of the one-launch step (JF_LIB=tools/exp/libjf_exp_rstrace.so)")
    ap.add_argument("--checkpoint-like", action="store_true",
                    help="what a trained Jacobi-Forcing checkpoint gives the step: every row's first 1-6 proposals hold ~0.97 of their position's "
                         "mass (accepted), the next one ~0.02 (rejected; its residual draw almost never collides with it)")
    a = ap.parse_args()
    B, L, V = a.batch, a.block, 152064
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(1)
    logits = (torch.randn(B, L - 1, V, generator=g, device="cuda") * 2).to(dt)
    draft = torch.randint(0, V, (B, L), generator=g, device="cuda")
    # the proposed id of every position holds ~p_hit of the mass at this temperature; uniforms are random -> most rows
    # reject within a few positions and the residual draw collides with the proposed id with probability ~p_hit
    import math
    others = V * math.exp(0.5 * (2.0 / a.temperature) ** 2)            # E[sum exp(x / T)] for x ~ N(0, 2^2)
    boost = lambda p: a.temperature * math.log(p / (1 - p) * others)
    if a.checkpoint_like:
        k = torch.randint(1, 7, (B, 1), generator=g, device="cuda")                # accepted proposals per row
        pos = torch.arange(L - 1, device="cuda").unsqueeze(0)
        val = torch.where(pos < k, torch.tensor(boost(0.97), device="cuda"), torch.tensor(boost(0.02), device="cuda")).to(dt)
        logits.scatter_(2, draft[:, 1:].unsqueeze(-1), val.unsqueeze(-1))
    else:
        logits.scatter_(2, draft[:, 1:].unsqueeze(-1), boost(a.p_hit))
    n = 1 << 16
    gc = torch.Generator().manual_seed(2)                               # the same streams in every process: runs are comparable
    st = ops.RsStepper(B, L, "cuda", torch.randint(0, V, (n,), generator=gc), torch.rand(n, generator=gc), torch.rand(n, generator=gc))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        rows, toks, nd = st.step(draft, logits, a.temperature, None, [L] * B, [0, 0, 0])
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(a.iters):
        st.step(draft, logits, a.temperature, None, [L] * B, [i * 7, i * 3, 0])
    ev[1].record()
    torch.cuda.synchronize()
    import numpy as np
    from jacobiforcing_amd import _native as N
    if a.trace:
        import ctypes as C
        lib = N.lib()
        names = {1: "accept workgroup done", 15: "accept walk decided the last row", 7: "last row handed its uniform (wavefront 3 of the chain)",
                 12: "end workgroup has every row's finish word", 25: "end: scans done", 13: "finished", 17: "last row's bonus workgroup has its uniform",
                 19: "... has walked", 21: "... has finished its row (token, record, next draft's seed and tail)"}
        acc = {k: [] for k in names}
        for i in range(8):
            lib.jf_exp_rs_trace(None, 1)
            st.step(draft, logits, a.temperature, None, [L] * B, [i * 7, i * 3, 0])
            torch.cuda.synchronize()
            buf = (C.c_ulonglong * 32)()
            lib.jf_exp_rs_trace(buf, 0)
            t0 = buf[0]
            for k in names:
                acc[k].append((buf[k] - t0) / 100.0)
        rb = (C.c_ulonglong * 1024)()
        lib.jf_exp_rs_rows(rb)
        r = np.array(rb[:], dtype=np.float64).reshape(8, 128)
        print("# last launch, per row: flag stored by the accept walk / segment 0 saw it / segment 0 stored / interval published by segment 0 / uniform handed out / bonus workgroup has it / bonus token stored (us)")
        for b_ in list(range(0, B, 8)) + [B - 1]:
            print(f"#   row {b_:3d}: " + "  ".join(f"{(r[k, b_] - t0) / 100.0:7.1f}" for k in (0, 1, 4, 2, 3, 5, 6)))
        pub, hand = (r[2, :B] - t0) / 100.0, (r[3, :B] - t0) / 100.0
        lag = hand - np.maximum.accumulate(pub)              # a row's uniform against the moment the last interval in front of it (or its own) was published
        print(f"#   all rows: interval published {pub.min():.1f} .. {pub.max():.1f} (row {int(pub.argmax())}); uniform handed out {hand.min():.1f} .. {hand.max():.1f}; "
              f"hand-out behind the last interval in front: mean {lag.mean():.1f}, max {lag.max():.1f} us (row {int(lag.argmax())})")
        print("#   published: " + " ".join(f"{x:.1f}" for x in pub))
        print("#   handed:    " + " ".join(f"{x:.1f}" for x in hand))
        if hasattr(lib, "jf_exp_rs_phases"):
            pb = (C.c_ulonglong * 16)()
            lib.jf_exp_rs_phases(pb)
            nm = ["table in LDS", "phase A: loads + float64 exps", "reduced over the workgroup", "all 16 partials of the row in", "phase B: probabilities + scans",
                  "sums formed, stores issued", "announced"]
            print("# row 0, segment 0 of the last launch (us): " + ";  ".join(f"{nm[k]} {(pb[k] - t0) / 100.0:.1f}" for k in range(7)))
        print("# in-kernel stamps of rs_step_fused_kernel, us after the accept workgroup started (mean of 8 launches):")
        for k in sorted(names, key=lambda k: np.mean(acc[k])):
            print(f"#   {np.mean(acc[k]):7.1f}  {names[k]}")
    f = N.RS_FIELDS.index
    rej = int((rows[:, f("reject_pos")] >= 0).sum())
    print(f"B={B} L={L} V={V} {a.dtype} T={a.temperature}: {ev[0].elapsed_time(ev[1]) / a.iters * 1e3:.1f} us per probs+step+readback "
          f"({rej} of {B} rows rejected, mean committed {rows[:, f('n_committed')].mean():.2f}, mean draws "
          f"{rows[:, f('n_bonus_draws')][rows[:, f('reject_pos')] >= 0].mean():.2f})", flush=True)
    print(f"algorithmic bytes: probs {B * (L - 1) * V * logits.element_size() / 1e6:.1f} MB, bonus rows {rej * V * logits.element_size() / 1e6:.1f} MB "
          f"(segment sums, read once) + {rej * V * logits.element_size() / 16 / 1e6:.2f} MB (one segment per draw)")


if __name__ == "__main__":
    main()
