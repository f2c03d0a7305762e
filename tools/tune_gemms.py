#!/usr/bin/env python3
"""One-off GEMM selection for the decode shapes of Qwen2.5-7B on MI355X with PyTorch's TunableOp (picks among the
hipBLASLt / rocBLAS solutions; stock PyTorch-ROCm, nothing custom).  Rows M = prompts x padded block length are kept on a
grid by the decoder (t_align=8: multiples of 64 with 8 prompts per GPU, of 512 with 64), so a handful of M values x five
GEMMs cover a decode step.  TUNE_M=512,1024,... adds rows to an existing table (TUNE_OUT = where to write).

    python tools/tune_gemms.py            # writes jacobiforcing_amd/tunableop_mi355x.csv
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
OUT = Path(os.environ.get("TUNE_OUT", ROOT / "jacobiforcing_amd" / "tunableop_mi355x.csv"))   # existing rows are kept
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_VERBOSE"] = "0"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = str(OUT)
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "40")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

sys.path.insert(0, str(ROOT))
from jacobiforcing_amd.modeling.qwen2 import Qwen2Config  # noqa: E402

cfg = Qwen2Config.qwen2_5_coder_7b()
dev = torch.device("cuda")
H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
w = dict(qkv=torch.randn((nq + 2 * nkv) * hd, H, device=dev, dtype=torch.bfloat16) * 0.02,
         bqkv=torch.randn((nq + 2 * nkv) * hd, device=dev, dtype=torch.bfloat16) * 0.02,
         o=torch.randn(H, nq * hd, device=dev, dtype=torch.bfloat16) * 0.02,
         gu=torch.randn(2 * I, H, device=dev, dtype=torch.bfloat16) * 0.02,
         d=torch.randn(H, I, device=dev, dtype=torch.bfloat16) * 0.02,
         lm=torch.randn(V, H, device=dev, dtype=torch.bfloat16) * 0.02)
Ms = [int(m) for m in os.environ.get("TUNE_M", "64,128,192,256,320,384,448,512").split(",")]
for M in Ms:
    x = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
    xi = torch.randn(M, I, device=dev, dtype=torch.bfloat16)
    xa = torch.randn(M, nq * hd, device=dev, dtype=torch.bfloat16)
    only = os.environ.get("TUNE_ONLY", "")          # "lm": lm_head only (its M is the compacted position count, a finer grid)
    for _ in range(2):
        if only != "lm":
            F.linear(x, w["qkv"], w["bqkv"]); F.linear(xa, w["o"]); F.linear(x, w["gu"]); F.linear(xi, w["d"])
        F.linear(x, w["lm"])
    torch.cuda.synchronize()
    print("tuned M =", M, flush=True)
try:
    torch.cuda.tunable.write_file(str(OUT))
except Exception as e:      # some builds only write at exit (to <name>0.csv)
    print("write_file:", type(e).__name__, e)
print("wrote", OUT)
