#!/bin/bash
# round 3: hand-off experiments on the convergence launch — result slots polled directly (no arrival counter, no wait in
# the items) and the cost of the write-back, against the shipped library on the same box
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
JF_LIB=tools/libjf_exp_slots.so timeout 900 python -m pytest tests/test_multiblock.py tests/test_multiblock_fuzz.py tests/test_decoder_e2e.py -m gpu -x -q > gpurun_out/r3s_gputest_slots.log 2>&1
tail -3 gpurun_out/r3s_gputest_slots.log
for i in 1 2 3; do
  for L in base slots; do
    if [ $L = base ]; then unset JF_LIB; else export JF_LIB=tools/libjf_exp_slots.so; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --no-sections --cpu-baseline-seconds 0 2>/dev/null | tail -1 > gpurun_out/r3s_bench_${L}_$i.json
    python - gpurun_out/r3s_bench_${L}_$i.json $L $i <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]; s = d["scripted_acceptance"]["roofline"]
print(f"{sys.argv[2]:6s} run {sys.argv[3]}: {d['value']:.0f} tok/s  verify {r['us_per_launch']:.1f} us frac {r['frac']:.3f}  scripted {s['us_per_launch']:.1f} us frac {s['frac']:.3f}  body {d['loop_body']['body_us_per_step']:.1f}")
PY
  done
done | tee gpurun_out/r3s_ab.txt
unset JF_LIB
for L in vtrace slots_vt nowb_vt; do
  echo "== $L"; JF_LIB=tools/libjf_exp_$L.so timeout 300 python tools/verify_trace.py 2>&1 | grep -v amdgpu.ids | grep -A8 "P= 64 loop\|P=  8 loop\|P=  1 loop"
done > gpurun_out/r3s_vtrace.txt 2>&1
cat gpurun_out/r3s_vtrace.txt
