#!/bin/bash
# round 4, session 3: the exact-probability sampling kernels (one-launch step rewritten): parity suites, seed 4555, a short soak,
# microbenchmarks + in-kernel stamps
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py tests/test_llm_api.py -m gpu -q -x -p no:cacheprovider --durations=8 > $O/r4c_sampling_tests.log 2>&1; tail -25 $O/r4c_sampling_tests.log
JF_RS_FUSED=0 timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py -m gpu -q -x -p no:cacheprovider -k "rs_ or nongreedy or onpolicy or sampl" > $O/r4c_sampling_tests_multi.log 2>&1; tail -5 $O/r4c_sampling_tests_multi.log
JF_FUZZ_SCALE=30 timeout 1200 python -m pytest tests/test_engine_fuzz.py -m gpu -q -p no:cacheprovider -n 10 -k "nongreedy or onpolicy" > $O/r4c_soak30.log 2>&1; tail -6 $O/r4c_soak30.log
for DT in bf16 f32; do for F in 1 0; do JF_RS_FUSED=$F timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/fused=$F /"; done; done > $O/r4c_rs_step.txt; cat $O/r4c_rs_step.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4c_rs_trace.txt 2>&1; cat $O/r4c_rs_trace.txt
