#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-scripted"
timeout 600 python bench.py $B > gpurun_out/r3_b4_default.json 2> gpurun_out/r3_b4_default.err
JF_LIB=tools/libjf_exp_publish_fence.so timeout 600 python bench.py $B > gpurun_out/r3_b4_publish_fence.json 2>/dev/null
timeout 600 python -m pytest tests/test_decoder_e2e.py tests/test_bench_and_dist.py tests/test_multiblock.py -m gpu -x -q 2>&1 | tail -2
python - <<'PY'
import json
for n in ("default","publish_fence"):
    try:
        d=json.loads(open(f"gpurun_out/r3_b4_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"]), "tok/s", round(d["ms_per_step"],2), "ms/step verify", round(d["roofline"]["us_per_launch"],1), "us frac", round(d["roofline"]["frac"],3),
              "body", round(d["loop_body"]["body_us_per_step"],1), "idle mean", round(d["loop_body"]["gpu_idle_us_per_step"],1), "med", round(d["loop_body"]["gpu_idle_us_median"],1),
              [(s["prompts_per_gpu"], round(s["us_per_launch"],1), round(s["frac"],3), round(s["body_us_per_step"],1), round(s["gpu_idle_us_median"],1)) for s in d.get("roofline_by_shape",{}).get("shapes",[])])
    except Exception as e:
        print(n, "failed", e)
PY
for P in 64 1; do
  rm -rf /tmp/prof_b$P
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b$P -- python $GRAFT_REPO_ROOT/bench.py --prompts-per-gpu $P --steps 16 --warmup 4 --no-shapes --no-scripted --no-prewarm --cpu-baseline-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r3_bubble_bench_$P.log 2>&1)
  python tools/iteration_bubble.py /tmp/prof_b$P > gpurun_out/r3_bubble_$P.txt 2>&1
done
head -3 gpurun_out/r3_bubble_64.txt gpurun_out/r3_bubble_1.txt
