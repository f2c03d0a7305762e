#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 200 python tools/r4_rs_probe.py > $O/r4j_probe.txt 2>&1; tail -3 $O/r4j_probe.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4j_rs_trace.txt 2>&1; cat $O/r4j_rs_trace.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace --p-hit 0.001 > $O/r4j_rs_trace_nocoll.txt 2>&1; cat $O/r4j_rs_trace_nocoll.txt
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py tests/test_llm_api.py -m gpu -q -x -p no:cacheprovider -n 6 > $O/r4j_sampling_tests.log 2>&1; tail -3 $O/r4j_sampling_tests.log
JF_FUZZ_SCALE=100 timeout 1500 python -m pytest tests/test_engine_fuzz.py -m gpu -q -p no:cacheprovider -n 12 -k "nongreedy or onpolicy" > $O/r4j_soak100.log 2>&1; tail -3 $O/r4j_soak100.log
