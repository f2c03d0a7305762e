#!/bin/bash
# One parametrised GPU session runner (round 5; replaces the numbered r3_/r4_ session scripts).
#   gpurun --timeout 1800 -- 'bash tools/session.sh <name> [args]'
# Every session writes under gpurun_out/<name>/ ; the files profiles/README.md cites are copied from there by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
NAME=${1:-help}; shift || true
O=gpurun_out/$NAME
mkdir -p "$O"
PYT="python -m pytest -q -p no:cacheprovider"

pmc_rs_probs() {      # $1 = label, JF_LIB selects the library: SQ counters of rs_probs_partial_kernel in two --pmc passes
    local L=$1 D=$O/pmc_$1
    mkdir -p $D
    timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY \
        --kernel-trace --output-format csv -d $D/A -o run -- python tools/prof_rs_probs_min.py > $D/a.log 2>&1
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d $D/B -o run -- python tools/prof_rs_probs_min.py > $D/b.log 2>&1
    python tools/pmc_rs_probs_parse.py $D > $O/pmc_rs_probs_$L.txt
    rm -rf $D/A $D/B
    cat $O/pmc_rs_probs_$L.txt
}

case "$NAME" in
rs_probs)             # the softmax-gather stream: parity subset, microbenchmark and SQ counters, shipped library vs round 4's
    timeout 1200 $PYT tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py -m gpu -x -n 6 -k "rs_ or nongreedy or onpolicy or sampl or softmax" 2>&1 | tail -3
    echo "== microbench, shipped library"; timeout 600 python tools/microbench_rs.py 1.0 0.8 2>&1 | grep -v amdgpu.ids | tee $O/microbench_new.txt
    if [ -f tools/libjf_r04.so ]; then
        echo "== microbench, round-4 library"; JF_LIB=tools/libjf_r04.so timeout 600 python tools/microbench_rs.py 1.0 0.8 2>&1 | grep -v amdgpu.ids | tee $O/microbench_r04.txt
        JF_LIB=tools/libjf_r04.so pmc_rs_probs r04
    fi
    pmc_rs_probs new
    ;;
rs_ab)                # microbenchmark of jf_rs_probs over experiment builds: bash tools/session.sh rs_ab rsA rsB ...
    for L in "$@"; do
        echo "== $L"; JF_LIB=tools/libjf_exp_$L.so timeout 600 python tools/microbench_rs.py 1.0 0.8 2>&1 | grep -v amdgpu.ids | grep "R= 1984\|R=  496" | tee $O/microbench_$L.txt
    done
    ;;
ranks8)               # what eight ranks do to one host: the real model, 8 prompts per rank, 1 rank vs 8 ranks sharing the one GPU (gloo)
    COMMON="--steps 40 --warmup 8 --no-scripted --no-shapes --no-sections --cpu-baseline-seconds 0"
    timeout 900 python bench.py --gpus 1 --prompts-per-gpu 8 $COMMON > $O/ranks1.json 2> $O/ranks1.err
    JF_DIST_BACKEND=gloo JF_FORCE_DEVICE=0 timeout 1500 python bench.py --gpus 8 --total-prompts 64 $COMMON > $O/ranks8.json 2> $O/ranks8.err
    JF_PIN_CPUS=0 JF_DIST_BACKEND=gloo JF_FORCE_DEVICE=0 timeout 1500 python bench.py --gpus 8 --total-prompts 64 $COMMON > $O/ranks8_unpinned.json 2> $O/ranks8_unpinned.err
    python tools/idle_gap_table.py $O/ranks1.json $O/ranks8.json $O/ranks8_unpinned.json | tee $O/idle_gap.txt
    ;;
dist_tests)           # the N > 1 line's evidence fields on the GPU box
    timeout 2400 $PYT tests/test_bench_and_dist.py -m gpu -x 2>&1 | tail -5
    ;;
*)
    echo "sessions: rs_probs rs_ab ranks8 dist_tests"; exit 2;;
esac
