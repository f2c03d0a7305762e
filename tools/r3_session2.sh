#!/bin/bash
# round 3: A/B of the hand-off fences in the convergence launch, then the bubble trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--steps 20 --warmup 5 --no-scripted --cpu-baseline-seconds 0"
timeout 600 python bench.py $B > gpurun_out/r3_ab_default.json 2> gpurun_out/r3_ab_default.err
JF_LIB=tools/libjf_exp_arrive_release.so timeout 600 python bench.py $B > gpurun_out/r3_ab_arrive_release.json 2>/dev/null
JF_LIB=tools/libjf_exp_publish_fence.so timeout 600 python bench.py $B > gpurun_out/r3_ab_publish_fence.json 2>/dev/null
timeout 900 python -m pytest tests/test_multiblock.py tests/test_decoder_e2e.py tests/test_bench_and_dist.py tests/test_multiblock_fuzz.py -m gpu -x -q > gpurun_out/r3_gputest2.log 2>&1
tail -3 gpurun_out/r3_gputest2.log
for P in 64 1; do
  rm -rf /tmp/prof_b$P
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b$P -- python $GRAFT_REPO_ROOT/bench.py --prompts-per-gpu $P --steps 16 --warmup 4 --no-shapes --no-scripted --no-prewarm --cpu-baseline-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r3_bubble_bench_$P.log 2>&1)
  python tools/iteration_bubble.py /tmp/prof_b$P > gpurun_out/r3_bubble_$P.txt 2>&1
done
python - <<'PY'
import json
for n in ("default","arrive_release","publish_fence"):
    try:
        d=json.loads(open(f"gpurun_out/r3_ab_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"],2), "ms/step verify", round(d["roofline"]["us_per_launch"],1), "us frac", round(d["roofline"]["frac"],3),
              "body", round(d["loop_body"]["body_us_per_step"],1), "idle", round(d["loop_body"]["gpu_idle_us_per_step"],1), "median", round(d["loop_body"]["gpu_idle_us_median"],1),
              [(s["prompts_per_gpu"], round(s["us_per_launch"],1), round(s["frac"],3), round(s["gpu_idle_us_median"] or 0,1)) for s in d.get("roofline_by_shape",{}).get("shapes",[])])
    except Exception as e:
        print(n, "failed", e)
PY
cat gpurun_out/r3_bubble_64.txt gpurun_out/r3_bubble_1.txt
