#!/bin/bash
# Where the cycles of the per-prompt state machine go (mb_step_kernel, the two-launch path: same Machine::step as the fused
# launch's steppers, on the HBM state block): SQ / SQC counters, rocprofv3 --pmc passes with --kernel-trace only.
#   gpurun --timeout 900 -- 'bash tools/pmc_mb_step.sh'   ->  gpurun_out/pmc_mb/pmc_mb_step.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_mb
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_LEVEL|DCACHE)[A-Z0-9_]*)" | sort -u > $OUT/counters_available.txt
export JF_FUSED_VERIFY=0
run() { d=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$d -o run -- python tools/verify_trace.py --prompts 64 --iters 8 > $OUT/$d.log 2>&1; }
run A SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run B SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH
run C SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
python - $OUT <<'PY' > $OUT/pmc_mb_step.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for d in "ABC":
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not any(s in k for s in ("mb_step_kernel", "mb_pack_kernel")):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
print("per launch (64 prompts = 64 wavefronts), counters averaged over the launches of each pass")
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print(f"    {c:32s} {acc[k][c] / max(n[k][c], 1):14.1f}   ({n[k][c]} samples)")
PY
for d in A B C; do tail -2 $OUT/$d.log | cut -c1-300 >> $OUT/pmc_mb_step.txt; done
rm -rf $OUT/A $OUT/B $OUT/C
cat $OUT/pmc_mb_step.txt
