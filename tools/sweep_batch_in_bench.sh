#!/bin/bash
# tokens/s, step time and the argmax launch's live roofline fraction vs prompts per GPU (bench.py workload otherwise unchanged)
for P in 8 16 32 64; do
  timeout 600 python bench.py --prompts-per-gpu $P --steps 48 --warmup 8 --cpu-baseline-seconds 0 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['scripted_acceptance']
print('P=$P', 'tok/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'argmax_us', round(r['us_per_launch'],1), 'MB', round(r['bytes_per_launch']/1e6,1), 'frac', round(r['frac'],3), '| scripted tok/s', round(s['value'],1), 'ms', round(s['ms_per_step'],2), 'tpf', round(s['tokens_per_forward'],2), 'frac', round(s['roofline']['frac'],3), 'MB', round(s['roofline']['bytes_per_launch']/1e6,1))"
done
