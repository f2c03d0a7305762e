#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_and_dist.py tests/test_kernels.py tests/test_decoder_e2e.py tests/test_hf_seam.py -m gpu -x -q > gpurun_out/r3_gputest5.log 2>&1
tail -5 gpurun_out/r3_gputest5.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench5.json 2> gpurun_out/r3_bench5.err
tail -c 1500 gpurun_out/r3_bench5.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench5.json").read().strip().splitlines()[-1])
sc=d.get("scripted_acceptance") or {}
print(round(d["value"]), "tok/s", round(d["ms_per_step"],2), "ms/step verify", round(d["roofline"]["us_per_launch"],1), "us frac", round(d["roofline"]["frac"],3),
      "body", round(d["loop_body"]["body_us_per_step"],1), "idle mean", round(d["loop_body"]["gpu_idle_us_per_step"],1), "med", round(d["loop_body"]["gpu_idle_us_median"],1))
print([(s["prompts_per_gpu"], round(s["us_per_launch"],1), round(s["frac"],3), round(s["body_us_per_step"],1), round(s["gpu_idle_us_median"],1)) for s in d.get("roofline_by_shape",{}).get("shapes",[])])
print("scripted", round(sc.get("value",0)), round(sc.get("tokens_per_forward",0),2), sc.get("verified"), sc.get("tokens_checked"), round((sc.get("roofline") or {}).get("frac",0),3))
for k in ("single_block","nongreedy","vs_ar"):
    print(k, json.dumps(d.get(k))[:900])
print("cpu", json.dumps(d.get("cpu_baseline"))[:600])
PY
