#!/bin/bash
export TMPDIR=/tmp
B="--prompts-per-gpu 1 --steps 48 --warmup 8 --no-shapes --no-sections --cpu-baseline-seconds 0"
for cfg in "8 8" "64 64" "32 64" "16 16"; do set -- $cfg
  timeout 600 python bench.py $B --t-align $1 --logit-align $2 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sc=d['scripted_acceptance']
print('t_align $1 logit_align $2:', round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/step | scripted', round(sc['value'],1), round(sc['tokens_per_forward'],2), round(sc['ms_per_step'],3))"
done
