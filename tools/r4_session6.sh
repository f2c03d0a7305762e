#!/bin/bash
# round 4, session 6: sampling step with the accept candidates from jf_rs_probs, DPP scans, readlane chain
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 200 python tools/r4_rs_probe.py > $O/r4f_probe.txt 2>&1; tail -4 $O/r4f_probe.txt
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py tests/test_llm_api.py -m gpu -q -x -p no:cacheprovider -n 6 > $O/r4f_sampling_tests.log 2>&1; tail -3 $O/r4f_sampling_tests.log
JF_RS_FUSED=0 timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py -m gpu -q -x -p no:cacheprovider -n 6 -k "rs_ or nongreedy or onpolicy or sampl" > $O/r4f_sampling_tests_multi.log 2>&1; tail -3 $O/r4f_sampling_tests_multi.log
for DT in bf16 f32; do for F in 1 0; do JF_RS_FUSED=$F timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/fused=$F /"; done; done > $O/r4f_rs_step.txt; cat $O/r4f_rs_step.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4f_rs_trace.txt 2>&1; cat $O/r4f_rs_trace.txt
JF_FUZZ_SCALE=100 timeout 1500 python -m pytest tests/test_engine_fuzz.py -m gpu -q -p no:cacheprovider -n 12 > $O/r4f_soak100.log 2>&1; tail -3 $O/r4f_soak100.log
