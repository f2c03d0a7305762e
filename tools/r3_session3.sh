#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multiblock.py tests/test_decoder_e2e.py tests/test_bench_and_dist.py tests/test_multiblock_fuzz.py -m gpu -x -q > gpurun_out/r3_gputest3.log 2>&1
tail -3 gpurun_out/r3_gputest3.log
JF_LIB=tools/libjf_exp_vtrace.so timeout 600 python tools/verify_trace.py > gpurun_out/r3_vtrace.txt 2>&1
JF_MB_FAST=0 JF_LIB=tools/libjf_exp_vtrace.so timeout 600 python tools/verify_trace.py --prompts 64 > gpurun_out/r3_vtrace_nofast.txt 2>&1
B="--steps 20 --warmup 5 --cpu-baseline-seconds 0"
timeout 600 python bench.py $B > gpurun_out/r3_b3_default.json 2> gpurun_out/r3_b3_default.err
JF_MB_FAST=0 timeout 600 python bench.py $B > gpurun_out/r3_b3_nofast.json 2>/dev/null
python - <<'PY'
import json
for n in ("default","nofast"):
    try:
        d=json.loads(open(f"gpurun_out/r3_b3_{n}.json").read().strip().splitlines()[-1])
        sc=d.get("scripted_acceptance") or {}
        print(n, round(d["value"]), "tok/s", round(d["ms_per_step"],2), "ms/step verify", round(d["roofline"]["us_per_launch"],1), "us frac", round(d["roofline"]["frac"],3),
              "body", round(d["loop_body"]["body_us_per_step"],1), "idle med", round(d["loop_body"]["gpu_idle_us_median"],1),
              [(s["prompts_per_gpu"], round(s["us_per_launch"],1), round(s["frac"],3)) for s in d.get("roofline_by_shape",{}).get("shapes",[])],
              "scripted", round(sc.get("value",0)), round(sc.get("tokens_per_forward",0),2), round((sc.get("roofline") or {}).get("us_per_launch",0),1), round((sc.get("roofline") or {}).get("frac",0),3))
    except Exception as e:
        print(n, "failed", e)
PY
cat gpurun_out/r3_vtrace.txt
grep -A12 "loop" gpurun_out/r3_vtrace_nofast.txt | tail -14
