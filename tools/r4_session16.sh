#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4t_rs_trace.txt 2>&1; grep -v "row  \|amdgpu" $O/r4t_rs_trace.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace --p-hit 0.001 > $O/r4t_rs_trace_nocoll.txt 2>&1; grep -v "row  \|amdgpu" $O/r4t_rs_trace_nocoll.txt
