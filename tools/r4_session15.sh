#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 200 python tools/r4_rs_probe.py > $O/r4A_probe.txt 2>&1; tail -3 $O/r4A_probe.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4A_rs_trace.txt 2>&1; cat $O/r4A_rs_trace.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace --p-hit 0.001 > $O/r4A_rs_trace_nocoll.txt 2>&1; cat $O/r4A_rs_trace_nocoll.txt | grep -v "row  "
for DT in bf16 f32; do timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | head -1; done
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py -m gpu -q -x -p no:cacheprovider -n 6 -k "rs_ or nongreedy or onpolicy or sampl" > $O/r4A_sampling_tests.log 2>&1; tail -3 $O/r4A_sampling_tests.log
JF_FUZZ_SCALE=100 timeout 1500 python -m pytest tests/test_engine_fuzz.py -m gpu -q -p no:cacheprovider -n 12 -k "nongreedy or onpolicy" > $O/r4A_soak100.log 2>&1; tail -3 $O/r4A_soak100.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --no-scripted --cpu-baseline-seconds 0 > $O/r4A_bench_sections.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/r4A_bench_sections.json')); ng=d['nongreedy']; print('nongreedy', ng['value'], ng['roofline']['us_per_launch'], ng['rs_step']['us_per_launch']); print('vs_ar', d['vs_ar']['vs_ar'], d['vs_ar']['iteration_cost_in_ar_steps'])"
