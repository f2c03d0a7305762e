#!/bin/bash
# round 4, session 5: sampling step after the two-wavefront walk / merged staging / shared LDS; launch anatomy at 1 / 8 / 64 prompts
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 200 python tools/r4_rs_probe.py > $O/r4e_probe.txt 2>&1; tail -4 $O/r4e_probe.txt
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py tests/test_llm_api.py -m gpu -q -x -p no:cacheprovider -n 6 > $O/r4e_sampling_tests.log 2>&1; tail -3 $O/r4e_sampling_tests.log
for DT in bf16 f32; do for F in 1 0; do JF_RS_FUSED=$F timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/fused=$F /"; done; done > $O/r4e_rs_step.txt; cat $O/r4e_rs_step.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4e_rs_trace.txt 2>&1; cat $O/r4e_rs_trace.txt
JF_FUZZ_SCALE=40 timeout 1500 python -m pytest tests/test_engine_fuzz.py -m gpu -q -p no:cacheprovider -n 12 -k "nongreedy or onpolicy" > $O/r4e_soak40.log 2>&1; tail -3 $O/r4e_soak40.log
for P in 1 8 64; do echo "## tools/verify_trace_insitu.py --prompts $P"; JF_LIB=tools/libjf_exp_vtrace.so timeout 400 python tools/verify_trace_insitu.py --prompts $P --iters 24 2>&1 | grep -v amdgpu.ids | grep "^#"; done > $O/r4e_vtrace_insitu.txt; cat $O/r4e_vtrace_insitu.txt
