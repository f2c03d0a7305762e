#!/usr/bin/env python3
"""Table of the per-rank idle-gap / host-gap fields of bench.py lines (tools/session.sh ranks8 -> profiles/idle_gap_8ranks_r05.txt)."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads([ln for ln in open(path) if ln.startswith("{")][0])
    except Exception as e:  # a failed run is part of the record
        print(f"{path}: no JSON line ({type(e).__name__}: {e})")
        continue
    print(f"== {path}: n_gpus {d['n_gpus']} ranks_seen {d['ranks_seen']} devices_distinct {d['devices_distinct']} shared_device {d['shared_device']} "
          f"backend {d['config']['dist_backend']}  prompts/rank {d['config']['prompts_per_gpu']}  {d['value']:.0f} tok/s  {d['ms_per_step']:.2f} ms/step")
    print("   rank  pid      cpus  device            tokens  seconds  verify_us  body_us  gpu_idle med/mean/p95 us     host_gap med/mean/p95 us")
    for x in d["per_rank"]:
        f = lambda v: "   -  " if v is None else f"{v:6.1f}"
        print(f"   {x['rank']:4d}  {x['pid']:<8d} {str(x.get('cpus')):>4}  {x['device']['pci']:<16s} {x['tokens']:7d}  {x['seconds']:7.3f}  {f(x.get('verify_us'))}   "
              f"{f(x.get('body_us'))}   {f(x.get('gpu_idle_us_median'))} {f(x.get('gpu_idle_us'))} {f(x.get('gpu_idle_us_p95'))}      "
              f"{f(x.get('host_gap_us_median'))} {f(x.get('host_gap_us'))} {f(x.get('host_gap_us_p95'))}")
    lb = d.get("loop_body", {}).get("by_rank", {})
    print(f"   over ranks: host_gap median {lb.get('host_gap_us_median')}  p95 {lb.get('host_gap_us_p95')}")
