#!/bin/bash
# round 3: result slots as the hand-off of the convergence launch + flat write-back — parity suite, bench, in-kernel stamps
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3t_gputest.log 2>&1
tail -3 gpurun_out/r3t_gputest.log
for i in 1 2; do
  timeout 900 python bench.py --steps 20 --warmup 5 --no-sections --cpu-baseline-seconds 0 2>/dev/null | tail -1 > gpurun_out/r3t_bench_$i.json
  python - gpurun_out/r3t_bench_$i.json $i <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]; s = d["scripted_acceptance"]["roofline"]
print(f"run {sys.argv[2]}: {d['value']:.0f} tok/s  verify {r['us_per_launch']:.1f} us frac {r['frac']:.3f}  scripted {s['us_per_launch']:.1f} us frac {s['frac']:.3f}  body {d['loop_body']['body_us_per_step']:.1f} idle {d['loop_body']['gpu_idle_us_median']:.1f}")
for sh in d["roofline_by_shape"]["shapes"]:
    print(f"    P={sh['prompts_per_gpu']:3d} verify {sh['us_per_launch']:.1f} us frac {sh['frac']:.3f} body {sh['body_us_per_step']:.1f}")
PY
done | tee gpurun_out/r3t_bench.txt
JF_LIB=tools/libjf_exp_vtrace.so timeout 300 python tools/verify_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3t_vtrace.txt
grep -A8 "P= 64 loop\|P=  8 loop\|P=  1 loop" gpurun_out/r3t_vtrace.txt
