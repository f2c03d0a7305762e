#!/bin/bash
export TMPDIR=/tmp
rm -rf /tmp/prof_eng
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eng -- python $GRAFT_REPO_ROOT/tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 --only "jacobi greedy" > $GRAFT_REPO_ROOT/gpurun_out/r3_eng_greedy.log 2>&1)
cp $(find /tmp/prof_eng -name "*kernel_stats.csv" | head -1) gpurun_out/r3_engine_greedy_kernel_stats.csv
tail -2 gpurun_out/r3_eng_greedy.log
