#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_multiblock.py tests/test_multiblock_fuzz.py tests/test_decoder_e2e.py tests/test_bench_and_dist.py tests/test_hf_seam.py tests/test_kernels.py -m gpu -x -q -n 4 > gpurun_out/r3_gputest13.log 2>&1; tail -3 gpurun_out/r3_gputest13.log
JF_LIB=tools/libjf_exp_vtrace.so timeout 600 python tools/verify_trace.py > gpurun_out/r3_vtrace13.txt 2>&1
grep -E "^P=|stepper  63|stepper   0|all steppers" gpurun_out/r3_vtrace13.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-sections --cpu-baseline-seconds 0 > gpurun_out/r3_b13.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_b13.json").read().strip().splitlines()[-1])
sc=d["scripted_acceptance"]
print(round(d["value"]), "tok/s", round(d["ms_per_step"],2), "verify", round(d["roofline"]["us_per_launch"],1), round(d["roofline"]["frac"],3), [(s["prompts_per_gpu"], round(s["us_per_launch"],1), round(s["frac"],3)) for s in d["roofline_by_shape"]["shapes"]], "scripted", round(sc["value"]), sc["verified"], round(sc["roofline"]["us_per_launch"],1), round(sc["roofline"]["frac"],3))
PY
