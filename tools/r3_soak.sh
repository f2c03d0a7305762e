#!/bin/bash
# 100x soak of the randomised parity sweeps on the GPU (round 3 kernels: step_fast64, one-launch rs_step, loop API untouched by these sweeps)
export TMPDIR=/tmp
mkdir -p gpurun_out
JF_FUZZ_SCALE=100 timeout 3000 python -m pytest tests/test_engine_fuzz.py tests/test_multiblock_fuzz.py -m gpu -n 12 -q -p no:cacheprovider > gpurun_out/r3_soak100.log 2>&1
tail -15 gpurun_out/r3_soak100.log
PROFILE=1 timeout 600 python tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 --only "jacobi greedy" > gpurun_out/r3_engine_profile_greedy.txt 2>&1
PROFILE=1 timeout 600 python tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 --only "T=0.8" > gpurun_out/r3_engine_profile_ng.txt 2>&1
tail -20 gpurun_out/r3_engine_profile_greedy.txt
