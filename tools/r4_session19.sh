#!/bin/bash
# the bench line with its events attached to the dispatches (final library), and the same command under rocprofv3
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r4zz_bench_n1.json 2> $O/r4zz_bench_n1.err
rm -rf /tmp/prof_head
(cd /tmp && JF_DUMP_LAUNCHES=$GRAFT_REPO_ROOT/$O/r4zz_launches_headline.json timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_head -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-shapes --no-sections --no-scripted --no-prewarm --cpu-baseline-seconds 0 > $GRAFT_REPO_ROOT/$O/r4zz_bench_under_rocprof.json 2>/dev/null)
python tools/verify_per_dispatch.py /tmp/prof_head $O/r4zz_launches_headline.json > $O/r4zz_verify_per_dispatch_headline.txt 2>&1
tail -2 $O/r4zz_verify_per_dispatch_headline.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4zz_bench_n1.json")); ng=d["nongreedy"]
print(round(d["value"]), d["ms_per_step"], d["roofline"]["us_per_launch"], d["roofline"]["frac"], ng["roofline"]["us_per_launch"], ng["roofline"]["frac"], ng["rs_step"]["us_per_launch"], d["vs_ar"]["vs_ar"], d["vs_ar"]["iteration_cost_in_ar_steps"], d["cpu_baseline"]["value"])
for sh in d["roofline_by_shape"]["shapes"]: print(sh["prompts_per_gpu"], sh["us_per_launch"], sh["frac"])
d2=json.load(open("gpurun_out/r4zz_bench_under_rocprof.json")); print("under rocprof: HIP events", d2["roofline"]["us_per_launch"], d2["roofline"]["frac"])
PY
