#!/bin/bash
# the restructured straight-line step: parity (goldens, fuzz sweeps) and its stage stamps
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_multiblock.py tests/test_multiblock_fuzz.py tests/test_loop_fuzz.py -m gpu -x -q > gpurun_out/r4s14_tests.log 2>&1
tail -3 gpurun_out/r4s14_tests.log
JF_LIB=tools/libjf_exp_vtrace2.so JF_EXP_TWICE=1 timeout 600 python tools/verify_trace.py --prompts 1 8 64 --iters 12 > gpurun_out/r4s14_vtrace_twice.log 2>&1
grep -v "^      stepper\|item workgroups" gpurun_out/r4s14_vtrace_twice.log | tail -40
JF_LIB=tools/libjf_exp_vtrace.so timeout 600 python tools/verify_trace.py --prompts 1 8 64 --iters 12 > gpurun_out/r4s14_vtrace.log 2>&1
grep -v "^      stepper\|item workgroups" gpurun_out/r4s14_vtrace.log | tail -40
