#!/bin/bash
# round 3: slots + flat write-back + long steps listed first — parity suite, bench (twice), in-situ stamps
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3u_gputest.log 2>&1
tail -3 gpurun_out/r3u_gputest.log
for i in 1 2; do
  timeout 900 python bench.py --steps 20 --warmup 5 --no-sections --cpu-baseline-seconds 0 2>/dev/null | tail -1 > gpurun_out/r3u_bench_$i.json
  python - gpurun_out/r3u_bench_$i.json $i <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]; s = d["scripted_acceptance"]["roofline"]
print(f"run {sys.argv[2]}: {d['value']:.0f} tok/s  verify {r['us_per_launch']:.1f} us frac {r['frac']:.3f}  scripted {d['scripted_acceptance']['value']:.0f} tok/s {s['us_per_launch']:.1f} us frac {s['frac']:.3f}  body {d['loop_body']['body_us_per_step']:.1f} idle {d['loop_body']['gpu_idle_us_median']:.1f}")
for sh in d["roofline_by_shape"]["shapes"]:
    print(f"    P={sh['prompts_per_gpu']:3d} verify {sh['us_per_launch']:.1f} us frac {sh['frac']:.3f} body {sh['body_us_per_step']:.1f}")
PY
done | tee gpurun_out/r3u_bench.txt
for M in "" "--scripted --iters 40"; do JF_LIB=tools/libjf_exp_vtrace.so timeout 400 python tools/verify_trace_insitu.py $M 2>&1 | grep -v amdgpu.ids | tail -6; done > gpurun_out/r3u_insitu.txt
cat gpurun_out/r3u_insitu.txt
