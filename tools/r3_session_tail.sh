#!/bin/bash
# round 3: remaining vectors of an argmax item in one round of loads (A/B against the serial tail loop on the same box)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py tests/test_multiblock.py tests/test_bench_and_dist.py -m gpu -x -q > gpurun_out/r3z_gputest.log 2>&1
tail -3 gpurun_out/r3z_gputest.log
for i in 1 2 3; do for L in base new; do
  if [ $L = base ]; then export JF_LIB=tools/libjf_exp_base.so; else unset JF_LIB; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-sections --cpu-baseline-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['scripted_acceptance']['roofline']
print('$L', round(d['value']), 'tok/s verify %.1f us %.3f  scripted %.0f tok/s %.1f us %.3f  body %.1f' % (r['us_per_launch'], r['frac'], d['scripted_acceptance']['value'], s['us_per_launch'], s['frac'], d['loop_body']['body_us_per_step']), ' shapes', [(sh['prompts_per_gpu'], round(sh['us_per_launch'],1)) for sh in d['roofline_by_shape']['shapes']])"
done; done | tee gpurun_out/r3z_ab.txt
unset JF_LIB
for L in base new; do if [ $L = base ]; then export JF_LIB=tools/libjf_exp_base.so; else unset JF_LIB; fi; echo "== $L"; timeout 300 python tools/microbench_argmax.py 2>&1 | grep -v amdgpu.ids | tail -14; done > gpurun_out/r3z_microbench.txt
cat gpurun_out/r3z_microbench.txt
