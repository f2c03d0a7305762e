#!/bin/bash
export TMPDIR=/tmp
rm -rf /tmp/prof_eng2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eng2 -- python $GRAFT_REPO_ROOT/tools/_r02/tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 --only "jacobi greedy" > /dev/null 2>&1)
cp $(find /tmp/prof_eng2 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r3_engine_greedy_kernel_stats_r02tree.csv
