#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hf_seam.py tests/test_decoder_e2e.py tests/test_bench_and_dist.py -m gpu -x -q -n 4 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 > gpurun_out/r3_b15.json 2> gpurun_out/r3_b15.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_b15.json").read().strip().splitlines()[-1])
sc=d["scripted_acceptance"]
print(round(d["value"]), "tok/s", round(d["ms_per_step"],2), "verify", round(d["roofline"]["us_per_launch"],1), round(d["roofline"]["frac"],3), "idle", round(d["loop_body"]["gpu_idle_us_median"],1))
print([(s["prompts_per_gpu"], round(s["us_per_launch"],1), round(s["frac"],3), round(s["gpu_idle_us_median"],1)) for s in d["roofline_by_shape"]["shapes"]], "scripted", round(sc["value"]), sc["verified"])
for k in ("single_block","nongreedy","vs_ar"):
    v=d[k]; print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a not in ("workload","note","roofline","rs_step")})
PY
for P in 1 8; do timeout 600 python bench.py --prompts-per-gpu $P --steps 48 --warmup 8 --no-shapes --no-sections --cpu-baseline-seconds 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sc=d['scripted_acceptance']
print('P=$P:', round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/step verify', round(d['roofline']['us_per_launch'],1), '| scripted', round(sc['value'],1), round(sc['tokens_per_forward'],2), round(sc['ms_per_step'],3))"; done
timeout 600 python -m jacobiforcing_amd.drivers.sb_math500 --synthetic 2 --n 16 --max-new-tokens 192 --csv /tmp/sb.csv 2>&1 | grep "toks/sec"
timeout 600 python -m jacobiforcing_amd.drivers.mr_humaneval --synthetic 4 --batch 1 --max-new-tokens 192 --csv /tmp/mr.csv 2>&1 | grep "toks/sec"
