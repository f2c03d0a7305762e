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
        echo "== $L"; JF_LIB=tools/exp/libjf_exp_$L.so timeout 600 python tools/microbench_rs.py 1.0 0.8 2>&1 | grep -v amdgpu.ids | grep "R= 1984\|R=  496" | tee $O/microbench_$L.txt
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
engine_loop)          # round 6: the engine decoders' chunk loop on device arrays: parity subset, PROFILE=1 sections, A/B against the callback contract
    timeout 1500 $PYT tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py tests/test_llm_api.py -m gpu -x -n 6 -s 2>&1 | grep -v "amdgpu.ids" | grep "per-position JS\|passed\|failed\|Error\|error" | tail -12
    for MODE in "jacobi greedy" "jacobi T=0.8"; do
        T=$(echo $MODE | tr -d ' =.')
        timeout 400 python tools/engine_throughput.py --only "$MODE" --max-tokens 96 2>&1 | grep "tok/s" | tee $O/tput_$T.txt
        JF_ENGINE_LOOP=0 timeout 400 python tools/engine_throughput.py --only "$MODE" --max-tokens 96 2>&1 | grep "tok/s" | sed 's/^/callbacks: /' | tee $O/tput_callbacks_$T.txt
        PROFILE=1 timeout 400 python tools/engine_throughput.py --only "$MODE" --max-tokens 96 2>&1 | grep "tok/s\|jacobi\.\|overhead" | tee $O/profile_$T.txt
        JF_ENGINE_LOOP=0 PROFILE=1 timeout 400 python tools/engine_throughput.py --only "$MODE" --max-tokens 96 2>&1 | grep "tok/s\|jacobi\.\|overhead" | sed 's/^/callbacks: /' | tee $O/profile_callbacks_$T.txt
    done
    ;;
engine_ab)            # round 6: the bench's engine sections, chunk loop on device arrays vs the callback contract, with / without the stage timer (same box)
    timeout 900 python tools/engine_sections.py 2>&1 | grep "JF_ENGINE_LOOP" | tee $O/loop.txt
    JF_ENGINE_LOOP=0 timeout 900 python tools/engine_sections.py 2>&1 | grep "JF_ENGINE_LOOP" | tee $O/callbacks.txt
    timeout 900 python tools/engine_sections.py --no-stage-timer 2>&1 | grep "JF_ENGINE_LOOP" | tee $O/loop_nostage.txt
    ;;
filter)               # round 6: jf_rs_filter as records (bf16: pattern counts in LDS): parity subset, microbenchmark, the bench's filtered section
    timeout 1500 $PYT tests/test_kernels.py tests/test_engine_decoder.py tests/test_engine_fuzz.py -m gpu -x -n 6 -k "filter or nongreedy or onpolicy or rs_ or sampl or budget" 2>&1 | tail -12
    timeout 600 python tools/microbench_rs_filter.py "$@" 2>&1 | grep -v amdgpu.ids | tee $O/microbench.txt
    timeout 900 python tools/engine_sections.py --repeat 1 2>&1 | grep "JF_ENGINE_LOOP" | tee $O/sections.txt
    ;;
filter_trace)         # round 6: in-kernel stamps of rs_filter_zone_kernel's workgroup 0 (tools/build_exp.sh flttrace -DJF_EXP_FLT_TRACE first) + rows-per-CU A/B (zn8 / zn4 builds)
    JF_LIB=tools/exp/libjf_exp_flttrace.so timeout 300 python tools/rs_filter_trace.py 2>&1 | grep -v amdgpu.ids | tee $O/trace.txt
    for L in zn8 zn4; do [ -f tools/exp/libjf_exp_$L.so ] && { echo "== $L"; JF_LIB=tools/exp/libjf_exp_$L.so timeout 200 python tools/microbench_rs_filter.py 2>&1 | grep "R= 1984"; }; done | tee $O/waves.txt
    ;;
paged)                # round 6: the paged KV layout (Config.kv_cache_layout="paged", the consumer of jf_engine_fill) against the contiguous one: parity tests + engine throughput
    timeout 900 $PYT tests/test_llm_api.py tests/test_engine_decoder.py -m gpu -x -n 4 -k "paged" 2>&1 | tail -3
    for MODE in "jacobi greedy" "jacobi T=0.8" "autoregressive"; do
        timeout 400 python tools/engine_throughput.py --only "$MODE" --max-tokens 96 2>&1 | grep "tok/s" | sed 's/^/contiguous: /'
        JF_KV_LAYOUT=paged timeout 400 python tools/engine_throughput.py --only "$MODE" --max-tokens 96 2>&1 | grep "tok/s" | sed 's/^/paged:      /'
    done | tee $O/throughput.txt
    ;;
gputests)             # the whole GPU suite + smoke
    timeout 2400 $PYT tests -m gpu -n 8 --durations=10 > $O/gputest.log 2>&1; tail -14 $O/gputest.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
    ;;
bench)                # the driver's command (+ extra flags): bash tools/session.sh bench [flags]
    timeout 1500 python bench.py --steps 20 --warmup 5 "$@" > $O/bench.json 2> $O/bench.err
    python tools/bench_brief.py $O/bench.json
    ;;
timing_ab)            # events attached to the dispatch against events recorded around the launch, one box
    timeout 900 python bench.py --steps 20 --warmup 5 --no-shapes --cpu-baseline-seconds 0 > $O/bench_dispatch.json 2> $O/bench_dispatch.err
    JF_VERIFY_EVENTS=bracket timeout 900 python bench.py --steps 20 --warmup 5 --no-shapes --cpu-baseline-seconds 0 > $O/bench_bracket.json 2> $O/bench_bracket.err
    for n in dispatch bracket; do echo "== $n"; python tools/bench_brief.py $O/bench_$n.json; done | tee $O/timing_ab.txt
    ;;
item_wgs)             # how many item workgroups walk the convergence launch's list (JF_VERIFY_ITEM_WGS) x chunks per row (JF_ARGMAX_ITEMS)
    for CFG in "0 0" "1536 0" "1024 0" "768 0" "512 0" "256 0" "1024 3000" "1024 6000"; do
        set -- $CFG
        echo "== item workgroups $1  JF_ARGMAX_ITEMS $2"
        env JF_VERIFY_ITEM_WGS=$1 $( [ $2 != 0 ] && echo JF_ARGMAX_ITEMS=$2 ) timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --no-sections --cpu-baseline-seconds 0 2>/dev/null > $O/b.json
        python tools/bench_brief.py $O/b.json
    done | tee $O/item_wgs.txt
    ;;
mailbox)              # does the host ever see the sequence word before the tables?  12 processes side by side, cheap order vs release fence
    for F in 0 1; do
        echo "== JF_PUBLISH_FENCE=$F"
        pids=""
        for i in $(seq 1 12); do JF_PUBLISH_FENCE=$F timeout 120 python tools/mailbox_stress.py --seconds 25 > /tmp/ms_$i.log 2>&1 & pids="$pids $!"; done
        for p in $pids; do wait $p; done
        cat /tmp/ms_*.log | grep -v amdgpu.ids | grep "rounds\|stale table" | sort | uniq -c | sort -rn | head -8
    done > $O/mailbox.txt 2>&1
    cat $O/mailbox.txt
    ;;
rs_step)              # the sampling step: one launch vs four, both dtypes; per-kernel durations under rocprofv3; [trace] with the rstrace build
    for DT in bf16 f32; do for F in 1 0; do JF_RS_FUSED=$F timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/fused=$F /"; done; done | tee $O/rs_step.txt
    for DT in bf16 f32; do timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 --checkpoint-like 2>&1 | grep -v amdgpu.ids | head -1; done | tee -a $O/rs_step.txt
    timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --p-hit 0.9 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/p-hit 0.9: /" | tee -a $O/rs_step.txt
    if [ "${1:-}" = trace ] && [ -f tools/exp/libjf_exp_rstrace.so ]; then      # tools/build_exp.sh rstrace -DJF_EXP_RS_TRACE
        for A in "" "--p-hit 0.001" "--p-hit 0.9" "--checkpoint-like"; do echo "## --trace $A"; JF_LIB=tools/exp/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace $A 2>&1 | grep -v "amdgpu.ids\|row  "; done | tee -a $O/rs_step.txt
    fi
    rm -rf /tmp/prof_ng
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ng -- python $ROOT/tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 64 --only "T=0.8" > /dev/null 2>&1)
    python tools/kernel_avg.py /tmp/prof_ng rs_ | tee $O/rs_insitu_rocprof.txt
    ;;
engine_step)          # jf_engine_step as one launch against its two launches (rocprofv3 --stats of the engine's greedy Jacobi run)
    for F in 1 0; do
        rm -rf /tmp/prof_eng$F
        (cd /tmp && JF_ENGINE_ONE_LAUNCH=$F timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eng$F -- python $ROOT/tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 --only "jacobi greedy" > /tmp/eng$F.log 2>&1)
        grep "tok/s" /tmp/eng$F.log | head -1
        python tools/kernel_avg.py /tmp/prof_eng$F engine_ | sed "s/^/one_launch=$F /"
    done | tee $O/engine_step.txt
    ;;
rocprof_bench)        # rocprofv3 --kernel-trace --stats of the bench command + per-grid / per-class / per-dispatch views
    rm -rf /tmp/prof_bench /tmp/prof_head
    (cd /tmp && JF_DUMP_LAUNCHES=$ROOT/$O/launches.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $ROOT/bench.py --steps 20 --warmup 5 --no-shapes --no-sections --cpu-baseline-seconds 0 > $ROOT/$O/rocprof_bench.log 2>&1)
    cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/rocprof_bench_kernel_stats.csv
    python tools/verify_by_grid.py /tmp/prof_bench > $O/verify_by_grid.txt 2>&1
    python tools/kernel_classes.py /tmp/prof_bench > $O/kernel_classes.txt 2>&1
    (cd /tmp && JF_DUMP_LAUNCHES=$ROOT/$O/launches_headline.json timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_head -- python $ROOT/bench.py --steps 20 --warmup 5 --no-shapes --no-sections --no-scripted --no-prewarm --cpu-baseline-seconds 0 > /dev/null 2>&1)
    python tools/verify_per_dispatch.py /tmp/prof_head $O/launches_headline.json > $O/verify_per_dispatch_headline.txt 2>&1
    tail -5 $O/verify_per_dispatch_headline.txt; head -30 $O/kernel_classes.txt
    ;;
pmc_verify)           # HBM bytes of the convergence launch from the PMC counters (separate passes): -> gpurun_out/pmc/pmc_verify.json
    timeout 1500 bash tools/pmc_verify.sh 2>&1 | tail -5
    ;;
anatomy)              # in-kernel stamps of the convergence launch inside the decode step (tools/build_exp.sh vtrace -DJF_EXP_VERIFY_TRACE first)
    for P in 1 8 64; do echo "## tools/verify_trace_insitu.py --prompts $P"; JF_LIB=tools/exp/libjf_exp_vtrace.so timeout 400 python tools/verify_trace_insitu.py --prompts $P --iters 24 2>&1 | grep -v amdgpu.ids | grep "^#"; done > $O/vtrace_insitu.txt
    echo "## tools/verify_trace_insitu.py --scripted --iters 40" >> $O/vtrace_insitu.txt; JF_LIB=tools/exp/libjf_exp_vtrace.so timeout 400 python tools/verify_trace_insitu.py --scripted --iters 40 2>&1 | grep "^#" >> $O/vtrace_insitu.txt
    cat $O/vtrace_insitu.txt
    ;;
soak)                 # the randomised parity sweeps at JF_FUZZ_SCALE (default 100)
    SC=${1:-100}
    JF_FUZZ_SCALE=$SC timeout 3000 $PYT tests/test_engine_fuzz.py tests/test_multiblock_fuzz.py -m gpu -n 12 > $O/soak$SC.log 2>&1; tail -4 $O/soak$SC.log
    JF_FUZZ_SCALE=$SC timeout 3000 $PYT tests/test_loop_fuzz.py -m gpu -n 12 > $O/loopsoak$SC.log 2>&1; tail -3 $O/loopsoak$SC.log
    ;;
evidence)             # the round's evidence in one call: suite, smoke, the driver's bench line, RCCL n=1, strong 64, 1 / 8 prompts, profiles
    bash "$0" gputests
    timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; python tools/bench_brief.py $O/bench_n1.json
    JF_DIST_BACKEND=nccl JF_DIST_FORCE_INIT=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 timeout 900 python bench.py --steps 20 --warmup 5 --no-shapes --no-sections --no-scripted --cpu-baseline-seconds 0 > $O/bench_rccl_n1.json 2> $O/bench_rccl_n1.err
    timeout 600 python bench.py --steps 20 --warmup 5 --total-prompts 64 --no-sections --cpu-baseline-seconds 0 > $O/bench_strong64.json 2>/dev/null
    for P in 1 8; do timeout 600 python bench.py --prompts-per-gpu $P --steps 48 --warmup 8 --no-shapes --no-sections --cpu-baseline-seconds 0 > $O/bench_p$P.json 2>/dev/null; done
    bash "$0" rocprof_bench; cp gpurun_out/rocprof_bench/* $O/ 2>/dev/null
    bash "$0" pmc_verify; cp gpurun_out/pmc/pmc_verify.json $O/ 2>/dev/null
    ;;
*)
    grep -E "^[a-z_0-9]+\)" "$0" | sed "s/)  *#/: /; s/^/  /"; exit 2;;
esac
