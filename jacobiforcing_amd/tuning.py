"""GEMM solution selection for the decode shapes (stock PyTorch-ROCm TunableOp).

``tunableop_mi355x.csv`` holds, for Qwen2.5-7B's five projection shapes at M = 8..56 (multiples of 8, round 3: the single-block
and batch-1 shapes), 64..512 (multiples of 64) and 1024..4096 rows, which
hipBLASLt / rocBLAS solution was fastest on an MI355X (produced by ``tools/tune_gemms.py``).  At the skinny M of Jacobi
decoding the default heuristics run these weight-streaming GEMMs at ~30 % of HBM bandwidth; the tuned picks are ~1.4x faster.
Nothing here touches the loop body; it only tells PyTorch which library kernel to call."""
from __future__ import annotations

import os
import shutil
import sys
import tempfile
from pathlib import Path

import torch

CSV = Path(__file__).resolve().parent / "tunableop_mi355x.csv"


def enable_tuned_gemms(csv: Path = CSV) -> bool:
    """Load the committed selections (tuning itself stays off).  Returns False when there is nothing to load."""
    if not csv.exists() or not torch.cuda.is_available():
        return False
    try:
        # TunableOp may rewrite its file at exit: give every process a private copy so the repo file is never touched
        # (keyed on the rank as well as the pid: eight ranks of one node start within the same millisecond)
        tmp = Path(tempfile.gettempdir()) / f"jf_tunableop_r{os.environ.get('RANK', '0')}_{os.getpid()}.csv"
        shutil.copyfile(csv, tmp)
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(False)
        torch.cuda.tunable.set_filename(str(tmp), insert_device_ordinal=False)
        return bool(torch.cuda.tunable.read_file(str(tmp)))
    except Exception as e:  # older torch / validator mismatch: fall back to the default heuristics
        print(f"[tuning] tuned GEMM table not loaded: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        try:
            torch.cuda.tunable.enable(False)
        except Exception:
            pass
        return False


def grid_alignment(num_prompts: int, tuned: bool = True):
    """(t_align, logit_align) that keep the GEMM row counts of a P-prompt Jacobi forward on the tuned grid (multiples of 64):
    the forward has Rtot x Tpad rows with Rtot >= P, lm_head one row per draft-carrying position.  Off the grid the library's
    default heuristics are 1.1-1.3x slower — at ONE prompt a forward of 32-56 rows takes 6.4 ms against 5.7 ms at 64 rows
    (profiles/batch1_r03.txt)."""
    if not tuned:
        return 1, 1
    ov = os.environ.get("JF_GRID_ALIGN")                     # "t_align,logit_align": A/B runs (profiles/batch1_align_ab_r04.txt)
    if ov:
        t, l = (int(x) for x in ov.split(","))
        return max(t, 1), max(l, 1)
    P = max(int(num_prompts), 1)
    if P <= 2:
        # one or two prompts: rows on the M = 8..56 part of the table (round 4; padding a 17-24-token row to 64 cost a
        # forward 5.9 ms against 4.9 ms at 16 rows, profiles/batch1_forward_split_r03.txt)
        return 8, 8
    return max(8, 64 // P), max(64, 8 * P)


SMALL_ROW_ALIGN = 8      # forwards of fewer than 64 rows (single-block drafts, L <= 16): the table has every multiple of 8
