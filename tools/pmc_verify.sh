#!/bin/bash
# HBM traffic of the verify launch (mb_verify_kernel) inside bench.py from the PMC counters (MI355X_MICROARCH.md, HBM
# section): FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (they do not fit one pass), kernel-trace only.
#   gpurun --timeout 1500 -- 'bash tools/pmc_verify.sh'   ->  gpurun_out/pmc/pmc_verify.json  (copy to profiles/)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
    JF_DUMP_LAUNCHES=$OUT/launches_$c.json timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv \
        -d $OUT/$c -o run -- python bench.py --steps 12 --warmup 2 --no-scripted --no-shapes --no-sections --cpu-baseline-seconds 0 > $OUT/bench_$c.log 2>&1
done
python tools/pmc_verify_parse.py $OUT
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
