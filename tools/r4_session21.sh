#!/bin/bash
# wider sweep of the sampling paths on the final library (seeds beyond the 100x soak)
mkdir -p gpurun_out
JF_FUZZ_SCALE=400 timeout 2400 python -m pytest tests/test_engine_fuzz.py -m gpu -q -p no:cacheprovider -n 12 -k "nongreedy or onpolicy" > gpurun_out/r4_soak400_sampling.log 2>&1; tail -3 gpurun_out/r4_soak400_sampling.log
for B in 128 96 17; do timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --batch $B 2>&1 | grep -v amdgpu.ids | head -1; done
