#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 --no-shapes --cpu-baseline-seconds 0 > $O/r4y_bench_dispatch.json 2> $O/r4y_bench_dispatch.err
JF_VERIFY_EVENTS=bracket timeout 900 python bench.py --steps 20 --warmup 5 --no-shapes --cpu-baseline-seconds 0 > $O/r4y_bench_bracket.json 2> $O/r4y_bench_bracket.err
python - <<'PY'
import json
for n in ("dispatch","bracket"):
    try:
        d=json.load(open(f"gpurun_out/r4y_bench_{n}.json"))
        ng=d["nongreedy"]
        print(n, round(d["value"]), round(d["roofline"]["us_per_launch"],1), round(d["roofline"]["frac"],3), "ng probs", round(ng["roofline"]["us_per_launch"],1), round(ng["roofline"]["frac"],3), "step", round(ng["rs_step"]["us_per_launch"],1), "sb", d["single_block"]["value"], "vs_ar", round(d["vs_ar"]["vs_ar"],2), d["scripted_acceptance"]["verified"])
    except Exception as e:
        print(n, "failed", e); print(open(f"gpurun_out/r4y_bench_{n}.err").read()[-2000:])
PY
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -n 8 2>&1 | tail -3
