#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --no-sections --cpu-baseline-seconds 0 > gpurun_out/r3_b12.json 2>/dev/null
rm -rf /tmp/prof_b12
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b12 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-shapes --no-sections --no-scripted --cpu-baseline-seconds 0 > /dev/null 2>&1)
cp $(find /tmp/prof_b12 -name "*kernel_stats.csv" | head -1) gpurun_out/r3_b12_kernel_stats.csv
python - <<'PY'
import json, csv
d=json.loads(open("gpurun_out/r3_b12.json").read().strip().splitlines()[-1])
print(round(d["value"]), "tok/s", round(d["ms_per_step"],2), "ms/step", "scripted", round(d["scripted_acceptance"]["value"]), round(d["scripted_acceptance"]["ms_per_step"],2), d["scripted_acceptance"]["verified"])
rows=list(csv.DictReader(open("gpurun_out/r3_b12_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f"{r['Name'][:110]:110s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:8.1f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}% avg {float(r['AverageNs'])/1e3:7.1f}")
PY
