#!/usr/bin/env python3
"""Does the host ever see the mailbox's sequence word before the tables published in front of it?  jf_mb_loop_begin for P prompts
over and over (every prompt restarts: the pack launch's extra workgroup copies all P descriptors into the mailbox, then stamps
it), the host waits for the stamp and checks the descriptor table at once.  Run several copies side by side to load the link.

    python tools/mailbox_stress.py [--seconds 20] [--prompts 48]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from jacobiforcing_amd import _native as N, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--prompts", type=int, default=48)
    a = ap.parse_args()
    P, n = a.prompts, 16
    prm = ops.MultiblockParams(n=n, K=2, r=0.85, n_gram_pool_size=4, eos_token_id=None, pad_token_id=0)
    batch = ops.MultiblockBatch(P, prm, "cuda")
    kvl = torch.zeros(P, dtype=torch.int32, device="cuda")
    lp = ops.MultiblockLoop(batch, kv_len=kvl, t_cap=64, t_align=1, valid_align=8, compact=True, cand_rows=3, order=1, max_seq_len=1 << 20)
    fB, fT, fkv = (N.DESC_FIELDS.index(k) for k in ("B", "T", "kv_len"))
    g = np.random.default_rng(1)
    bad = rounds = 0
    t0 = time.time()
    while time.time() - t0 < a.seconds:
        ids = torch.from_numpy(g.integers(1, 1000, size=(P, n))).cuda()
        kv = g.integers(5, 500, size=P).astype(np.int32)
        # clear the descriptor slots the host is about to check, so that a table that has not landed yet cannot look right
        lp.mailbox[N.MB_MAILBOX_HDR:N.MB_MAILBOX_HDR + P * N.DESC_INTS] = -7
        s = lp.begin(ids, torch.from_numpy(kv))
        d = s.d
        ok = (d[:, fB] == 1).all() and (d[:, fT] == n).all() and (d[:, fkv] == kv).all() and s.Rtot == P and s.Nvalid == P * n
        bad += 0 if ok else 1
        rounds += 1
        if not ok and bad <= 3:
            print("stale table:", d[:, fB].tolist()[:16], "Rtot", s.Rtot, flush=True)
    print(f"rounds {rounds}  stale {bad}", flush=True)


if __name__ == "__main__":
    main()
