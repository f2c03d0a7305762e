#!/bin/bash
# Why does the convergence launch stream 5.9-6.0 TB/s inside the decode step and 7.0 TB/s over synthetic logits between idle
# periods?  The same counters on both, separate rocprofv3 passes (kernel-trace + one --pmc group each):
#   GRBM_GUI_ACTIVE            GPU-busy cycles of the dispatch -> with the kernel's duration: the core clock it ran at
#   TCC_HIT_sum TCC_MISS_sum   L2 lookups of the dispatch
#   TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum    requests to the fabric (HBM / Infinity Cache), how many of them 32-byte
#   TCP_TCC_READ_REQ_sum       vector-L1 -> L2 read requests
#   gpurun --timeout 2400 -- 'bash tools/pmc_insitu_vs_synth.sh'  -> gpurun_out/pmc2/summary.txt (copy to profiles/)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc2
mkdir -p $OUT
i=0
for grp in "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    JF_DUMP_LAUNCHES=$OUT/launches_insitu_$i.json timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/insitu_$i -o run -- \
        python bench.py --steps 12 --warmup 2 --no-scripted --no-shapes --no-sections --cpu-baseline-seconds 0 > $OUT/insitu_$i.log 2>&1
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/synth_$i -o run -- \
        python tools/verify_trace.py --prompts 64 --iters 14 > $OUT/synth_$i.log 2>&1
done
python tools/pmc_insitu_vs_synth_parse.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -rf $OUT/insitu_[0-9] $OUT/synth_[0-9]
