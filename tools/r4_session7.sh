#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4g_rs_trace.txt 2>&1; cat $O/r4g_rs_trace.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace --batch 8 > $O/r4g_rs_trace_b8.txt 2>&1; cat $O/r4g_rs_trace_b8.txt
