#!/bin/bash
# round 3: how many item workgroups walk the list in the convergence launch
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multiblock.py tests/test_multiblock_fuzz.py tests/test_decoder_e2e.py tests/test_bench_and_dist.py -m gpu -x -q > gpurun_out/r3w_gputest.log 2>&1
tail -3 gpurun_out/r3w_gputest.log
for G in 0 1536 1024 768 512 384 256; do
  JF_VERIFY_ITEM_WGS=$G timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --no-sections --cpu-baseline-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['scripted_acceptance']['roofline']
print('G=$G', round(d['value']), 'tok/s verify %.1f us %.3f  scripted %.0f tok/s %.1f us %.3f  body %.1f' % (r['us_per_launch'], r['frac'], d['scripted_acceptance']['value'], s['us_per_launch'], s['frac'], d['loop_body']['body_us_per_step']))"
done | tee gpurun_out/r3w_sweep.txt
for G in 768 512; do for M in "" "--scripted --iters 40"; do echo "== G=$G $M"; JF_VERIFY_ITEM_WGS=$G JF_LIB=tools/libjf_exp_vtrace.so timeout 400 python tools/verify_trace_insitu.py $M 2>&1 | grep -v amdgpu.ids | tail -5; done; done > gpurun_out/r3w_insitu.txt
cat gpurun_out/r3w_insitu.txt
