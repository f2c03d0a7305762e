#!/usr/bin/env python3
"""The non-greedy step's three stages (jf_rs_probs, jf_rs_filter, jf_rs_step) at batch 64 x block 32 x V = 152 064, bf16, with and without
top_k / top_p: microseconds per stage between HIP events (ops.STAGE_HOOK).  Does the step cost more on probability rows?"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import _native, ops  # noqa: E402

B, L, V = 64, 32, 152064
g = torch.Generator().manual_seed(3)
logits = (torch.randn(B, L - 1, V, generator=g) * 2.5).to(torch.bfloat16).cuda()
heavy = logits.float().argmax(-1).cpu()
draft = torch.randint(0, V, (B, L), generator=g)
pick = torch.rand(B, L - 1, generator=g) < float(sys.argv[1] if len(sys.argv) > 1 else 0.6)
draft[:, 1:] = torch.where(pick, heavy, draft[:, 1:])
n = 64 * B * L
unis = torch.rand(n, generator=g); bonus = torch.rand(n, generator=g); pads = torch.randint(0, V, (n,), generator=g)
for k, p in ((0, 0.0), (50, 0.0), (0, 0.9), (50, 0.9)):
    st = ops.RsStepper(B, L, torch.device("cuda"), pads, unis, bonus)
    acc = {}

    def hook(name, phase, nbytes):
        if phase == "arm":
            return None
        e = torch.cuda.Event(enable_timing=True); e.record()
        acc.setdefault(name, []).append(e)
    for it in range(6):
        if it == 2:
            acc.clear(); ops.STAGE_HOOK = hook
        rows, toks, nd = st.step(draft.cuda(), logits, 0.8, None, [L] * B, [0, 0, 0], k, p)
    torch.cuda.synchronize(); ops.STAGE_HOOK = None
    out = []
    for name, ev in acc.items():
        us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(0, len(ev), 2)]
        out.append(f"{name} {sum(us) / len(us):8.1f} us")
    f = _native.RS_FIELDS.index
    print(f"top_k={k:3d} top_p={p:4.2f}   " + "   ".join(out) + f"   (mean committed {rows[:, f('n_committed')].mean():.1f})", flush=True)
