#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace --checkpoint-like > $O/r4C_rs_trace_ckpt.txt 2>&1; grep -v amdgpu $O/r4C_rs_trace_ckpt.txt | grep -v "row  " | tail -13
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace --p-hit 0.9 > $O/r4C_rs_trace_hit09.txt 2>&1; grep -v "row  \|amdgpu" $O/r4C_rs_trace_hit09.txt | tail -4
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4C_rs_trace.txt 2>&1; grep -v "row  \|amdgpu" $O/r4C_rs_trace.txt | tail -13
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace --p-hit 0.001 > $O/r4C_rs_trace_nocoll.txt 2>&1; grep -v "row  \|amdgpu" $O/r4C_rs_trace_nocoll.txt | tail -13
for DT in bf16 f32; do timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 --checkpoint-like 2>&1 | grep -v amdgpu.ids | head -1; done
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py -m gpu -q -x -p no:cacheprovider -n 6 -k "rs_ or nongreedy or onpolicy or sampl or timing" 2>&1 | tail -2
JF_FUZZ_SCALE=400 timeout 2400 python -m pytest tests/test_engine_fuzz.py -m gpu -q -p no:cacheprovider -n 12 -k "nongreedy or onpolicy" > $O/r4C_soak400.log 2>&1; tail -2 $O/r4C_soak400.log
