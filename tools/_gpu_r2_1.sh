set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r2_gputest2.log
timeout 600 python tools/microbench_rs.py > gpurun_out/r2_rs_probs.log 2>&1
rm -f gpurun_out/r2_rs_step.log
for dt in bf16 f32; do for T in 1.0 0.8; do timeout 300 python tools/microbench_rs_step.py --dtype $dt --temperature $T >> gpurun_out/r2_rs_step.log 2>&1; done; done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rs -- python $GRAFT_REPO_ROOT/tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 > /tmp/prof_rs.log 2>&1
f=$(find /tmp/prof_rs -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r2_rs_step_kernel_stats.csv
tail -3 /tmp/prof_rs.log
