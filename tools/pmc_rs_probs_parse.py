#!/usr/bin/env python3
"""Parse the two rocprofv3 --pmc passes of tools/session.sh rs_probs (directories A/ and B/ under argv[1])."""
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for d in ("A", "B"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rs_probs_partial_kernel" not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[key].add((d, r["Dispatch_Id"]))
print("rs_probs_partial_kernel<DT (0 = fp32, 1 = bf16), VEC, SCALE>: SQ counters summed over 6 launches per shape, V = 152064")
print("wave-cycle shares: WAIT_ANY = parked on s_waitcnt (memory), WAIT_INST = issue stalls, ACTIVE = issuing; the three are disjoint\n")
for key in sorted(acc):
    c = acc[key]
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    nl = max(len([1 for d, _ in cnt[key] if d == "A"]), 1)
    print(f"{key[0]:48s} workgroups {key[1]:5d}  launches {nl}")
    print(f"    waves/launch {c['SQ_WAVES'] / nl:8.0f}   wave cycles: parked on memory {c['SQ_WAIT_ANY'] / wc:5.1%}  issue-stalled {c['SQ_WAIT_INST_ANY'] / wc:5.1%}  "
          f"issuing {c['SQ_ACTIVE_INST_ANY'] / wc:5.1%}  (VALU {c['SQ_ACTIVE_INST_VALU'] / wc:5.1%}, VMEM {c['SQ_ACTIVE_INST_VMEM'] / wc:5.1%})")
    if c["SQ_INSTS_VALU"]:
        elems = key[1]  # placeholder
        print(f"    per launch: VALU instructions {c['SQ_INSTS_VALU'] / nl:12.0f}  of which transcendental {c['SQ_INSTS_VALU_TRANS_F32'] / nl:12.0f}  "
              f"VMEM reads {c['SQ_INSTS_VMEM_RD'] / nl:10.0f}  SALU {c['SQ_INSTS_SALU'] / nl:10.0f}  VALU per 16-byte load {c['SQ_INSTS_VALU'] / max(c['SQ_INSTS_VMEM_RD'], 1):6.1f}")
        if c["SQ_LEVEL_WAVES"] and c["GRBM_GUI_ACTIVE"]:
            print(f"    mean resident waves (SQ_LEVEL_WAVES / GRBM_GUI_ACTIVE) {c['SQ_LEVEL_WAVES'] / c['GRBM_GUI_ACTIVE']:8.1f}   "
                  f"mean VMEM instructions in flight {c['SQ_INST_LEVEL_VMEM'] / c['GRBM_GUI_ACTIVE']:8.1f}")
