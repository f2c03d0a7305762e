#!/bin/bash
# round 4 evidence session (1 x MI355X): parity suite, smoke, the bench line, rocprofv3 stats, PMC traffic, launch anatomy, soaks.
#   gpurun --timeout 5400 -- 'bash tools/profile_session_r04.sh'   -> gpurun_out/r4z_*  (copied into profiles/ by hand)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -n 8 --durations=10 > $O/r4z_gputest.log 2>&1; tail -16 $O/r4z_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r4z_smoke.log 2>&1; tail -1 $O/r4z_smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r4z_bench_n1.json 2> $O/r4z_bench_n1.err
JF_DIST_BACKEND=nccl JF_DIST_FORCE_INIT=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 timeout 900 python bench.py --steps 20 --warmup 5 --no-shapes --no-sections --no-scripted --cpu-baseline-seconds 0 > $O/r4z_bench_rccl_n1.json 2> $O/r4z_bench_rccl_n1.err
timeout 600 python bench.py --steps 20 --warmup 5 --total-prompts 64 --no-sections --cpu-baseline-seconds 0 > $O/r4z_bench_strong64.json 2>/dev/null
for P in 1 8; do timeout 600 python bench.py --prompts-per-gpu $P --steps 48 --warmup 8 --no-shapes --no-sections --cpu-baseline-seconds 0 > $O/r4z_bench_p$P.json 2>/dev/null; done
# rocprofv3 --kernel-trace --stats of the bench command (headline + scripted windows)
rm -rf /tmp/prof_bench
(cd /tmp && JF_DUMP_LAUNCHES=$GRAFT_REPO_ROOT/$O/r4z_launches.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-shapes --no-sections --cpu-baseline-seconds 0 > $GRAFT_REPO_ROOT/$O/r4z_rocprof_bench.log 2>&1)
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/r4z_rocprof_bench_kernel_stats.csv
python tools/verify_by_grid.py /tmp/prof_bench > $O/r4z_verify_by_grid.txt 2>&1
python tools/kernel_classes.py /tmp/prof_bench > $O/r4z_kernel_classes.txt 2>&1
# headline window alone under the profiler (no scripted run): per-dispatch durations joined to the launch bytes
rm -rf /tmp/prof_head
(cd /tmp && JF_DUMP_LAUNCHES=$GRAFT_REPO_ROOT/$O/r4z_launches_headline.json timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_head -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-shapes --no-sections --no-scripted --no-prewarm --cpu-baseline-seconds 0 > /dev/null 2>&1)
python tools/verify_per_dispatch.py /tmp/prof_head $O/r4z_launches_headline.json > $O/r4z_verify_per_dispatch_headline.txt 2>&1
# HBM traffic of the convergence launch (PMC, separate passes)
timeout 1500 bash tools/pmc_verify.sh > $O/r4z_pmc_verify.log 2>&1; cp $O/pmc/pmc_verify.json $O/r4z_pmc_verify.json
# in-kernel timeline and anatomy
for P in 1 8 64; do echo "## tools/verify_trace_insitu.py --prompts $P"; JF_LIB=tools/libjf_exp_vtrace.so timeout 400 python tools/verify_trace_insitu.py --prompts $P --iters 24 2>&1 | grep -v amdgpu.ids | grep "^#"; done > $O/r4z_vtrace_insitu.txt
echo "## tools/verify_trace_insitu.py --scripted --iters 40" >> $O/r4z_vtrace_insitu.txt; JF_LIB=tools/libjf_exp_vtrace.so timeout 400 python tools/verify_trace_insitu.py --scripted --iters 40 2>&1 | grep "^#" >> $O/r4z_vtrace_insitu.txt
# the straight-line step's stages, and the step run twice through the same instructions (instruction fetch vs issue time)
JF_LIB=tools/libjf_exp_vtrace.so timeout 400 python tools/verify_trace.py --prompts 1 8 64 --iters 12 2>&1 | grep -v "amdgpu.ids\|^      stepper\|item workgroups" > $O/r4z_step_stages.txt
echo "## -DJF_EXP_STEP_TWICE, JF_EXP_TWICE=1" >> $O/r4z_step_stages.txt
JF_LIB=tools/libjf_exp_vtrace2.so JF_EXP_TWICE=1 timeout 400 python tools/verify_trace.py --prompts 1 8 64 --iters 12 2>&1 | grep -v "amdgpu.ids\|^      stepper\|item workgroups" >> $O/r4z_step_stages.txt
# sampling step
for DT in bf16 f32; do for F in 1 0; do JF_RS_FUSED=$F timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/fused=$F /"; done; done > $O/r4z_rs_step.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace 2>&1 | grep -v amdgpu.ids >> $O/r4z_rs_step.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace --p-hit 0.001 2>&1 | grep -v amdgpu.ids >> $O/r4z_rs_step.txt
timeout 300 python tools/microbench_rs.py > $O/r4z_rs_probs.txt 2>&1
timeout 900 python tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 > $O/r4z_engine.txt 2>&1
# the non-greedy engine loop under the profiler: the sampling kernels' durations inside the decode step
rm -rf /tmp/prof_ng
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ng -- python $GRAFT_REPO_ROOT/tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 64 --only "T=0.8" > /dev/null 2>&1)
python - > $O/r4z_rs_insitu_rocprof.txt <<'PYEOF'
import csv, glob
f = glob.glob("/tmp/prof_ng/**/*kernel_stats.csv", recursive=True)[0]
print("# rocprofv3 --kernel-trace --stats -- python tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 64 --only T=0.8")
print("# (engine non-greedy Jacobi decoding, 64 requests x block 32, V = 152064, bf16): this package's sampling kernels inside the decode step")
for r in csv.DictReader(open(f)):
    if "rs_" in r["Name"]:
        print(f"{r['Name'].split('(')[0][:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}  max {float(r['MaxNs']) / 1e3:8.1f}")
PYEOF
# soaks
JF_FUZZ_SCALE=100 timeout 1500 python -m pytest tests/test_engine_fuzz.py tests/test_multiblock_fuzz.py -m gpu -n 12 -q -p no:cacheprovider > $O/r4z_soak100.log 2>&1; tail -4 $O/r4z_soak100.log
JF_FUZZ_SCALE=100 timeout 1500 python -m pytest tests/test_loop_fuzz.py -m gpu -n 12 -q -p no:cacheprovider > $O/r4z_loopsoak.log 2>&1; tail -3 $O/r4z_loopsoak.log
tail -3 $O/r4z_gputest.log; tail -1 $O/r4z_smoke.log; ls -la $O | grep r4z_
