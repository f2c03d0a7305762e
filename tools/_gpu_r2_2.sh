set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/r2_gputest3.log
timeout 600 python tools/microbench_rs.py 1.0 0.8 > gpurun_out/r2_rs_probs.log 2>&1
rm -f gpurun_out/r2_rs_step.log
for dt in bf16; do for T in 1.0 0.8; do timeout 300 python tools/microbench_rs_step.py --dtype $dt --temperature $T >> gpurun_out/r2_rs_step.log 2>&1; done; done
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_fused.json 2> gpurun_out/r2_bench_fused.err
JF_FUSED_VERIFY=0 timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 > gpurun_out/r2_bench_unfused.json 2> gpurun_out/r2_bench_unfused.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-scripted > /tmp/prof_b.log 2>&1
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); head -40 "$f" > $GRAFT_REPO_ROOT/gpurun_out/r2_bench_kernel_stats.csv
tail -2 /tmp/prof_b.log
