#!/bin/bash
# round 3: the one-launch rejection-sampling step — parity suites, then fused vs four launches
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels.py tests/test_engine_fuzz.py tests/test_engine_decoder.py tests/test_llm_api.py -m gpu -x -q > gpurun_out/r3_gputest6.log 2>&1
tail -5 gpurun_out/r3_gputest6.log
for F in 1 0; do
  for DT in bf16 f32; do
    JF_RS_FUSED=$F timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | sed "s/^/fused=$F /"
  done
done > gpurun_out/r3_rs_step.txt
(cd /tmp && for F in 1 0; do rm -rf /tmp/prof_rs$F; JF_RS_FUSED=$F timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rs$F -- python $GRAFT_REPO_ROOT/tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 > /dev/null 2>&1; python - $F <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/prof_rs{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
print(f"fused={sys.argv[1]} rocprofv3 --kernel-trace --stats, kernels of jf_rs_step / jf_rs_probs:")
for r in rows:
    n = r["Name"]
    if "rs_" in n:
        print(f"   {n.split('(')[0][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done) >> gpurun_out/r3_rs_step.txt 2>&1
cat gpurun_out/r3_rs_step.txt
