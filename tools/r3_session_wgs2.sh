#!/bin/bash
# round 3: item workgroups x chunks per row of the convergence launch (in-order list, long steps first)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
for CFG in "1024 0" "1024 3000" "1024 4500" "1024 6000" "1536 3000" "1536 6000" "768 6000" "2048 6000"; do
  set -- $CFG
  if [ $2 = 0 ]; then unset JF_ARGMAX_ITEMS; else export JF_ARGMAX_ITEMS=$2; fi
  JF_VERIFY_ITEM_WGS=$1 timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --no-sections --cpu-baseline-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['scripted_acceptance']['roofline']
print('G=$1 items=$2', round(d['value']), 'tok/s verify %.1f us %.3f  scripted %.0f tok/s %.1f us %.3f  body %.1f' % (r['us_per_launch'], r['frac'], d['scripted_acceptance']['value'], s['us_per_launch'], s['frac'], d['loop_body']['body_us_per_step']))"
done | tee gpurun_out/r3x_sweep.txt
for CFG in "1024 0" "1024 6000"; do set -- $CFG; if [ $2 = 0 ]; then unset JF_ARGMAX_ITEMS; else export JF_ARGMAX_ITEMS=$2; fi
for M in "--warmup 30" "--scripted --iters 40"; do echo "== G=$1 items=$2 $M"; JF_VERIFY_ITEM_WGS=$1 JF_LIB=tools/libjf_exp_vtrace.so timeout 400 python tools/verify_trace_insitu.py $M 2>&1 | grep -v amdgpu.ids | tail -7; done; done > gpurun_out/r3x_insitu.txt
cat gpurun_out/r3x_insitu.txt
