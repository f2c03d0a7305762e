#!/usr/bin/env python3
"""One-paragraph digest of a bench.py line (tools/session.sh)."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads([ln for ln in open(path) if ln.startswith("{")][-1])
    except Exception as e:
        print(f"{path}: no JSON line ({type(e).__name__}: {e})")
        try:
            print(open(path.replace(".json", ".err")).read()[-1500:])
        except OSError:
            pass
        continue
    r = d.get("roofline") or {}
    lb = d.get("loop_body") or {}
    print(f"{d['value']:.0f} tok/s  {d['ms_per_step']:.2f} ms/step  n_gpus {d['n_gpus']} (ranks_seen {d.get('ranks_seen')}, devices {d.get('devices_distinct')})  "
          f"verify {r.get('us_per_launch', 0):.1f} us = {r.get('frac', 0):.3f}  body {lb.get('body_us_per_step') or 0:.1f} us  "
          f"idle median {lb.get('gpu_idle_us_median') or 0:.1f} us  host gap median {lb.get('host_gap_us_median') or 0:.1f} us")
    ot = r.get("other_timing")
    if ot:
        print(f"   timed the other way ({ot['method'][:44]}...): {ot['us_per_launch']:.1f} us = {ot['frac']:.3f}")
    s = d.get("scripted_acceptance")
    if s:
        sr = s.get("roofline") or {}
        print(f"   scripted: {s['value']:.0f} tok/s  TPF {s['tokens_per_forward']:.2f}  verified {s['verified']}  verify {sr.get('us_per_launch', 0):.1f} us = {sr.get('frac', 0):.3f}")
    for sh in (d.get("roofline_by_shape") or {}).get("shapes", []):
        print(f"   {sh['prompts_per_gpu']:3d} prompts: {sh['bytes_per_launch'] / 1e6:7.1f} MB  {sh['us_per_launch']:6.1f} us = {sh['frac']:.3f}  body {sh['body_us_per_step']:.1f}  idle median {sh['gpu_idle_us_median']:.1f}")
    ng = d.get("nongreedy")
    if ng and "roofline" in ng:
        print(f"   nongreedy: {ng['value']:.0f} tok/s  rs_probs {ng['roofline']['us_per_launch']:.1f} us = {ng['roofline']['frac']:.3f}  rs_step {ng['rs_step']['us_per_launch']:.1f} us")
        f = ng.get("filtered") or {}
        if f.get("rs_filter"):
            print(f"   nongreedy, top_k {f['top_k']} top_p {f['top_p']}: {f['value']:.0f} tok/s  {f['ms_per_step']:.2f} ms/step  rs_filter {f['rs_filter']['us_per_launch']:.0f} us per call")
        elif f:
            print("   nongreedy filtered:", f)
    elif ng:
        print("   nongreedy:", ng)
    if "single_block" in d:
        print(f"   single_block: {d['single_block'].get('value')}   vs_ar: {(d.get('vs_ar') or {}).get('vs_ar')}  iteration cost {(d.get('vs_ar') or {}).get('iteration_cost_in_ar_steps')}")
    tt = d.get("trained_toy") or {}
    if tt.get("decoders"):
        print("   trained toy (measured tokens per forward, == AR " + str(tt.get("verified")) + "): " + "  ".join(f"{k} {v['tokens_per_forward']}" for k, v in tt["decoders"].items()))
    elif tt:
        print("   trained toy:", tt)
    cb = d.get("cpu_baseline")
    if cb:
        print(f"   cpu_baseline: {cb.get('value')} {cb.get('unit')} on {cb.get('cores')} cores ({cb.get('kind')})")
