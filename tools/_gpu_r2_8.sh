set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/microbench_rs.py 1.0 0.8 0.7 > gpurun_out/r2_rs_probs.log 2>&1
rm -f gpurun_out/r2_rs_step.log
for dt in bf16 f32; do for T in 1.0 0.8; do timeout 300 python tools/microbench_rs_step.py --dtype $dt --temperature $T >> gpurun_out/r2_rs_step.log 2>&1; done; done
timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 1.0 --p-hit 0.2 >> gpurun_out/r2_rs_step.log 2>&1
timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 1.0 --p-hit 0.95 >> gpurun_out/r2_rs_step.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rs -- python $GRAFT_REPO_ROOT/tools/microbench_rs_step.py --dtype bf16 --temperature 1.0 > /tmp/prof_rs.log 2>&1
f=$(find /tmp/prof_rs -name "*kernel_stats.csv" | head -1); grep -E "Name|rs_|argmax" "$f" | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/r2_rs_step_kernel_stats.csv
cd $GRAFT_REPO_ROOT
timeout 600 python tools/verify_trace.py > gpurun_out/r2_verify_trace_plain.log 2>&1
JF_LIB=$GRAFT_REPO_ROOT/tools/libjf_exp_vtrace.so timeout 600 python tools/verify_trace.py > gpurun_out/r2_verify_trace_stamps.log 2>&1
