#!/usr/bin/env python3
"""The bench's two engine sections alone (bench.py nongreedy_section / engine_greedy_section on a random-init Qwen2.5-Coder-7B-shaped
model), for same-box A/Bs of the chunk loop: JF_ENGINE_LOOP=0/1 (callback contract vs device arrays), with / without the stage timer.

    python tools/engine_sections.py [--no-stage-timer] [--max-tokens 64]
"""
import argparse
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from jacobiforcing_amd.modeling.qwen2 import Qwen2Config, Qwen2Model, Qwen2Weights  # noqa: E402
from jacobiforcing_amd.tuning import enable_tuned_gemms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-stage-timer", action="store_true")
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = Qwen2Config()
    enable_tuned_gemms()
    w = Qwen2Weights(cfg, dev, dtype=torch.bfloat16, seed=0)
    model = Qwen2Model(cfg, w)
    if a.no_stage_timer:
        class _Null:
            done = {}
            def __enter__(self): return self
            def __exit__(self, *e): pass
            def summary(self, *x, **k): return None
            def timing(self, n): return ""
        bench.StageTimer = _Null
    out = {}
    for i in range(a.repeat):
        for name, fn in (("nongreedy", lambda: bench.nongreedy_section(model, cfg, w, True, max_tokens=a.max_tokens)),
                         ("engine_greedy", lambda: bench.engine_greedy_section(model, cfg, w, True, max_tokens=a.max_tokens))):
            r = fn()
            lb = r.get("loop_body") or {}
            line = dict(tok_s=round(r["value"], 1), ms_per_step=round(r["ms_per_step"], 3), iterations=r["iterations"],
                        body_us=lb.get("body_us_median"), gpu_idle_us=lb.get("gpu_idle_us_median"), host_gap_us=lb.get("host_gap_us_median"))
            if name == "nongreedy" and isinstance(r.get("filtered"), dict):
                line["filtered_ms_per_step"] = round(r["filtered"].get("ms_per_step", 0.0), 3)
                line["rs_filter_us"] = (r["filtered"].get("rs_filter") or {}).get("us_per_launch")
            out[f"{name}#{i}"] = line
            print(f"JF_ENGINE_LOOP={os.environ.get('JF_ENGINE_LOOP', '1')} stage_timer={not a.no_stage_timer} {name}#{i}: {json.dumps(line)}", flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
