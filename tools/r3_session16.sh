#!/bin/bash
export TMPDIR=/tmp
rm -rf /tmp/prof_p1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p1 -- python $GRAFT_REPO_ROOT/bench.py --prompts-per-gpu 1 --steps 48 --warmup 8 --no-shapes --no-sections --no-scripted --no-prewarm --cpu-baseline-seconds 0 > /dev/null 2>&1)
python tools/kernel_classes.py /tmp/prof_p1 | head -12
python - <<'PY'
import csv, glob
f=glob.glob("/tmp/prof_p1/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:8.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}% avg {float(r['AverageNs'])/1e3:7.1f}")
PY
