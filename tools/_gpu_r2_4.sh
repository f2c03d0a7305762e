set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5 > gpurun_out/r2_gputest4.log
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 > gpurun_out/r2_bench_fused.json 2> gpurun_out/r2_bench_fused.err
JF_FUSED_VERIFY=0 timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 > gpurun_out/r2_bench_unfused.json 2> gpurun_out/r2_bench_unfused.err
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-shapes --no-scripted > gpurun_out/r2_bench_fused_b.json 2>> gpurun_out/r2_bench_fused.err
JF_FUSED_VERIFY=0 timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-shapes --no-scripted > gpurun_out/r2_bench_unfused_b.json 2>> gpurun_out/r2_bench_unfused.err
