#!/bin/bash
# round 3: jf_engine_step as one launch — parity suites, then kernel durations against the two launches (rocprofv3 --stats)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_engine_decoder.py tests/test_engine_fuzz.py tests/test_llm_api.py tests/test_kernels.py -m gpu -x -q > gpurun_out/r3g_gputest.log 2>&1
tail -2 gpurun_out/r3g_gputest.log
for F in 1 0; do
  rm -rf /tmp/prof_eng$F
  (cd /tmp && JF_ENGINE_ONE_LAUNCH=$F timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eng$F -- python $GRAFT_REPO_ROOT/tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 --only "jacobi greedy" > /tmp/eng$F.log 2>&1)
  grep "tok/s" /tmp/eng$F.log | head -1
  python - $F <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/prof_eng{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "engine_" in r["Name"]:
        print(f"   one_launch={sys.argv[1]} {r['Name'].split('(')[0][:40]:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done > gpurun_out/r3g_engine_step.txt 2>&1
cat gpurun_out/r3g_engine_step.txt
