#!/bin/bash
# round 4, session 1 (1 x MI355X): GPU suite (new: runaway goldens, RCCL single rank, checkpoint headline), smoke, bench line,
# batch-1 alignment A/B, in-kernel stamps of the convergence launch at 1 / 8 / 64 prompts, in-kernel stamps of jf_rs_step.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/r4a_gputest.log 2>&1; tail -5 $O/r4a_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r4a_smoke.log 2>&1; tail -1 $O/r4a_smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r4a_bench_n1.json 2> $O/r4a_bench_n1.err; tail -c 600 $O/r4a_bench_n1.err
timeout 900 python tools/r4_align_ab.py > $O/r4a_align_ab.txt 2>&1; cat $O/r4a_align_ab.txt | grep align=
for P in 1 8 64; do echo "## tools/verify_trace_insitu.py --prompts $P"; JF_LIB=tools/libjf_exp_vtrace.so timeout 400 python tools/verify_trace_insitu.py --prompts $P --iters 24 2>&1 | grep -v amdgpu.ids; done > $O/r4a_vtrace_insitu.txt
JF_LIB=tools/libjf_exp_rstrace.so timeout 300 python tools/microbench_rs_step.py --dtype bf16 --temperature 0.8 --trace > $O/r4a_rs_trace.txt 2>&1
for DT in bf16 f32; do timeout 300 python tools/microbench_rs_step.py --dtype $DT --temperature 0.8 2>&1 | grep -v amdgpu.ids | head -3; done > $O/r4a_rs_step.txt
tail -3 $O/r4a_gputest.log; ls -la $O | grep r4a_
