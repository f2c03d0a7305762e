#!/bin/bash
# round 3, first GPU session: parity suite, bench (resident + host-driven), kernel trace of the timed window
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest1.log 2>&1
tail -5 gpurun_out/r3_gputest1.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err
tail -c 600 gpurun_out/r3_bench1.err
JF_RESIDENT=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --cpu-baseline-seconds 0 > gpurun_out/r3_bench1_hostdriven.json 2> gpurun_out/r3_bench1_hostdriven.err
for P in 64 1; do
  rm -rf /tmp/prof_b$P
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b$P -- python $GRAFT_REPO_ROOT/bench.py --prompts-per-gpu $P --steps 16 --warmup 4 --no-shapes --no-scripted --no-prewarm --cpu-baseline-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r3_bubble_bench_$P.log 2>&1)
  python tools/iteration_bubble.py /tmp/prof_b$P > gpurun_out/r3_bubble_$P.txt 2>&1
done
cat gpurun_out/r3_bubble_64.txt gpurun_out/r3_bubble_1.txt
