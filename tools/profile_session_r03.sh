#!/bin/bash
# round 3 evidence session (1 x MI355X): parity suite, smoke, the bench line, rocprofv3 stats / traces, PMC traffic, small shapes.
#   gpurun --timeout 3000 -- 'bash tools/profile_session_r03.sh'   -> gpurun_out/r3e_*  (copied into profiles/ by hand)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r3e_gputest.log 2>&1; tail -3 $O/r3e_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r3e_smoke.log 2>&1; tail -1 $O/r3e_smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r3e_bench_n1.json 2> $O/r3e_bench_n1.err
JF_RESIDENT=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --no-sections --cpu-baseline-seconds 0 > $O/r3e_bench_hostdriven.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --total-prompts 64 --no-sections --cpu-baseline-seconds 0 > $O/r3e_bench_strong64.json 2>/dev/null
for P in 1 8; do timeout 600 python bench.py --prompts-per-gpu $P --steps 48 --warmup 8 --no-shapes --no-sections --cpu-baseline-seconds 0 > $O/r3e_bench_p$P.json 2>/dev/null; done
# rocprofv3 --kernel-trace --stats of the bench command (headline + scripted windows)
rm -rf /tmp/prof_bench
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-shapes --no-sections --cpu-baseline-seconds 0 > $GRAFT_REPO_ROOT/$O/r3e_rocprof_bench.log 2>&1)
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/r3e_rocprof_bench_kernel_stats.csv
python tools/verify_by_grid.py /tmp/prof_bench > $O/r3e_verify_by_grid.txt 2>&1
python tools/kernel_classes.py /tmp/prof_bench > $O/r3e_kernel_classes.txt 2>&1
# iteration bubble (kernel trace of the timed window, no scripted run)
for P in 64 1; do
  rm -rf /tmp/prof_b$P
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b$P -- python $GRAFT_REPO_ROOT/bench.py --prompts-per-gpu $P --steps 16 --warmup 4 --no-shapes --no-scripted --no-sections --no-prewarm --cpu-baseline-seconds 0 > /dev/null 2>&1)
  python tools/iteration_bubble.py /tmp/prof_b$P > $O/r3e_bubble_$P.txt 2>&1
done
# HBM traffic of the convergence launch (PMC, separate passes)
timeout 1500 bash tools/pmc_verify.sh > $O/r3e_pmc_verify.log 2>&1; cp $O/pmc/pmc_verify.json $O/r3e_pmc_verify.json
# in-kernel timeline
JF_LIB=tools/libjf_exp_vtrace.so timeout 600 python tools/verify_trace.py > $O/r3e_vtrace.txt 2>&1
for M in "--warmup 30" "--scripted --iters 40"; do echo "## tools/verify_trace_insitu.py $M"; JF_LIB=tools/libjf_exp_vtrace.so timeout 400 python tools/verify_trace_insitu.py $M 2>&1 | grep -v amdgpu.ids; done > $O/r3e_vtrace_insitu.txt
# batch-1 drivers
timeout 600 python -m jacobiforcing_amd.drivers.ar_baseline --synthetic 2 --max-new-tokens 256 > $O/r3e_ar.txt 2>&1
timeout 600 python -m jacobiforcing_amd.drivers.sb_math500 --synthetic 4 --n 16 --max-new-tokens 256 --csv /tmp/sb.csv > $O/r3e_sb.txt 2>&1
timeout 600 python -m jacobiforcing_amd.drivers.mr_humaneval --synthetic 8 --batch 1 --max-new-tokens 256 --csv /tmp/mr.csv > $O/r3e_mr.txt 2>&1
timeout 900 python tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 > $O/r3e_engine.txt 2>&1
for M in "jacobi greedy" "T=0.8"; do PROFILE=1 timeout 600 python tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 --only "$M" 2>&1 | grep -v amdgpu.ids; done > $O/r3e_engine_profile.txt
JF_FUZZ_SCALE=100 timeout 1500 python -m pytest tests/test_engine_fuzz.py tests/test_multiblock_fuzz.py -m gpu -n 12 -q -p no:cacheprovider > $O/r3e_soak100.log 2>&1; tail -4 $O/r3e_soak100.log
JF_FUZZ_SCALE=100 timeout 900 python -m pytest tests/test_loop_fuzz.py -m gpu -n 12 -q -p no:cacheprovider > $O/r3e_loopsoak.log 2>&1; tail -2 $O/r3e_loopsoak.log
for DT in bf16 f32; do for F in 1 0; do JF_RS_FUSED=$F timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/fused=$F /"; done; done > $O/r3e_rs_step.txt
tail -3 $O/r3e_gputest.log; tail -1 $O/r3e_smoke.log; ls -la $O | grep r3e_
