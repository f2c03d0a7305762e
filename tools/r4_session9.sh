#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4i_rs_trace.txt 2>&1; cat $O/r4i_rs_trace.txt
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py tests/test_llm_api.py -m gpu -q -x -p no:cacheprovider -n 6 > $O/r4i_sampling_tests.log 2>&1; tail -3 $O/r4i_sampling_tests.log
for DT in bf16 f32; do for F in 1 0; do JF_RS_FUSED=$F timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/fused=$F /"; done; done > $O/r4i_rs_step.txt; cat $O/r4i_rs_step.txt
