set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5 > gpurun_out/r2_gputest6.log
timeout 600 python tools/verify_trace.py > gpurun_out/r2_verify_trace_plain.log 2>&1
JF_LIB=$GRAFT_REPO_ROOT/tools/libjf_exp_vtrace.so timeout 600 python tools/verify_trace.py > gpurun_out/r2_verify_trace_stamps.log 2>&1
timeout 600 python tools/microbench_rs.py 1.0 0.8 > gpurun_out/r2_rs_probs.log 2>&1
rm -f gpurun_out/r2_rs_step.log
for dt in bf16 f32; do for T in 1.0 0.8; do timeout 300 python tools/microbench_rs_step.py --dtype $dt --temperature $T >> gpurun_out/r2_rs_step.log 2>&1; done; done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rs -- python $GRAFT_REPO_ROOT/tools/microbench_rs_step.py --dtype bf16 --temperature 1.0 > /tmp/prof_rs.log 2>&1
f=$(find /tmp/prof_rs -name "*kernel_stats.csv" | head -1); grep -E "Name|rs_|argmax" "$f" | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/r2_rs_step_kernel_stats.csv
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eng -- python $GRAFT_REPO_ROOT/tools/engine_throughput.py --batch 64 --max-tokens 64 --only jacobi > /tmp/prof_eng.log 2>&1
f=$(find /tmp/prof_eng -name "*kernel_stats.csv" | head -1); grep -E "Name|rs_|engine_|argmax|kv_append|mb_" "$f" | cut -c1-300 > $GRAFT_REPO_ROOT/gpurun_out/r2_engine_kernel_stats.csv
tail -4 /tmp/prof_eng.log
