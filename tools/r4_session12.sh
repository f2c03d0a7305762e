#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --no-sections --no-scripted > $O/r4y_bench_cpu.json 2> $O/r4y_bench_cpu.err; python -c "
import json; d=json.load(open('$O/r4y_bench_cpu.json')); print(d['value'], d['roofline']['frac']); print(json.dumps(d['cpu_baseline']['verify_kernel'])[:900])"
timeout 2400 bash tools/pmc_insitu_vs_synth.sh > $O/r4y_pmc2.log 2>&1; cat $O/pmc2/summary.txt
