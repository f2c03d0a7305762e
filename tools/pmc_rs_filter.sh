#!/bin/bash
# HBM traffic of jf_rs_filter's launches (rs_filter_zone_kernel) from the PMC counters, FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
# (MI355X_MICROARCH.md, HBM section), against the algorithmic R x V x 2 bytes of ONE read of the logits; rs_probs_partial_kernel of the same runs
# as the calibration (it reads the logits exactly once).   gpurun -- 'bash tools/pmc_rs_filter.sh'  ->  gpurun_out/pmc_filter/pmc_rs_filter.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_filter
mkdir -p $OUT
for CASE in "peaked 50 0.9" "flat 50 0.9" "flat 0 0.9" "peaked 50 0.0"; do
    T=$(echo $CASE | tr ' .' '__')
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${T}_$c -o run -- python tools/prof_rs_filter_min.py $CASE > $OUT/${T}_$c.log 2>&1
    done
    python tools/pmc_rs_filter_parse.py $OUT/${T}_FETCH_SIZE $OUT/${T}_WRITE_SIZE "$CASE"
    rm -rf $OUT/${T}_FETCH_SIZE $OUT/${T}_WRITE_SIZE
done | tee $OUT/pmc_rs_filter.txt
