"""Phase timing of mb_step_kernel in situ (experiment build, -DJF_EXP_MB_TRACE; never the product library).

Runs the bench workload (random-init Qwen2.5-Coder-7B shape, 8 prompts) for a few iterations and after each one reads
prompt 0's shader-clock stamps (100 MHz s_memrealtime) back, printing the mean time between consecutive phases.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DJF_EXP_MB_TRACE -Iinclude -Ijacobiforcing_amd/csrc \
          jacobiforcing_amd/csrc/*.hip -o jacobiforcing_amd/lib/libjf_exp_trace.so
    JF_LIB=$PWD/jacobiforcing_amd/lib/libjf_exp_trace.so python tools/mb_step_trace.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jacobiforcing_amd import _native  # noqa: E402

PHASES = {0: "kernel entry", 1: "load_scalars", 2: "accept scan", 3: "commit accepted", 4: "re-draft", 5: "pool push x2",
          6: "candidates", 7: "span loop end", 8: "spawn/promote", 9: "early-stop check", 10: "build_out", 11: "store_scalars",
          12: "zero packed / exit"}


def main():
    from jacobiforcing_amd import ops
    from jacobiforcing_amd.engine.multiblock_decoder import MultiblockJacobiDecoder
    from jacobiforcing_amd.modeling.qwen2 import Qwen2Config, Qwen2Model, Qwen2Weights
    from jacobiforcing_amd.synthetic import ScriptedAcceptance, humaneval_shaped_prompts
    lib = _native.load()
    if not hasattr(lib, "jf_exp_read_trace"):
        raise SystemExit("this tool needs the experiment build (-DJF_EXP_MB_TRACE, see the header of this file) selected with JF_LIB=...")
    lib.jf_exp_read_trace.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    cfg = Qwen2Config.qwen2_5_coder_7b()
    model = Qwen2Model(cfg, Qwen2Weights(cfg, dev, dtype=torch.bfloat16, seed=0))
    prm = ops.MultiblockParams(n=32, K=2, r=0.85, lookahead_start_ratio=0.0, n_gram_pool_size=4, eos_token_id=None,
                               pad_token_id=cfg.pad_token_id)
    vocab_hi = min(151643, cfg.vocab_size - 2)
    prompts = humaneval_shaped_prompts(8, seed=1234, vocab_hi=vocab_hi)
    dec = MultiblockJacobiDecoder(model, 8, prm, max_seq_len=4096)
    if "--scripted" in sys.argv:
        dec.logits_hook = ScriptedAcceptance(cfg.vocab_size, robust_pct=82, vocab_hi=vocab_hi)
    rows = []
    buf = np.zeros(32, dtype=np.uint64)

    def on_iter(i, d):
        torch.cuda.synchronize()
        assert lib.jf_exp_read_trace(buf.ctypes.data) == 0
        if i >= 8:
            rows.append(buf.copy())

    dec.generate(prompts, max_new_tokens=1 << 30, max_calls=1 << 30, seed=1234, on_iteration=on_iter, max_iterations=48)
    a = np.stack(rows).astype(np.int64)
    ks = sorted(PHASES)
    print("phase                    mean_us   (delta from previous stamp, prompt 0, %d steps)" % len(rows))
    prev = a[:, 0]
    for k in ks[1:]:
        cur = a[:, k]
        ok = cur >= prev          # phases not reached in a step keep an old stamp
        d = (cur - prev)[ok] / 100.0
        print(f"{PHASES[k]:24s} {d.mean() if len(d) else float('nan'):8.2f}   n={ok.sum()}")
        prev = np.where(ok, cur, prev)
    tot = (a[:, 12] - a[:, 0]) / 100.0
    print(f"{'entry -> exit':24s} {tot.mean():8.2f}")


if __name__ == "__main__":
    main()
