#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 200 python tools/r4_rs_probe.py > $O/r4h_probe.txt 2>&1; tail -3 $O/r4h_probe.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4h_rs_trace.txt 2>&1; cat $O/r4h_rs_trace.txt
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py -m gpu -q -x -p no:cacheprovider -n 6 -k "rs_ or nongreedy or onpolicy or sampl" > $O/r4h_sampling_tests.log 2>&1; tail -3 $O/r4h_sampling_tests.log
JF_FUZZ_SCALE=100 timeout 1500 python -m pytest tests/test_engine_fuzz.py -m gpu -q -p no:cacheprovider -n 12 -k "nongreedy or onpolicy" > $O/r4h_soak100.log 2>&1; tail -3 $O/r4h_soak100.log
for DT in bf16 f32; do for F in 1 0; do JF_RS_FUSED=$F timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/fused=$F /"; done; done > $O/r4h_rs_step.txt; cat $O/r4h_rs_step.txt
