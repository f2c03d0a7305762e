#!/bin/bash
# The round-2 evidence session: bench line, rocprofv3 kernel stats + trace of the same command, PMC traffic of the verify
# launch, batch-1 / batch-8 / strong-scaling lines, the drivers, the engine path.  Results under gpurun_out/prof_r02/;
# the summaries that matter are copied into profiles/ by hand.
#   gpurun --timeout 4800 -- 'bash tools/profile_session_r02.sh'
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r02
mkdir -p $O
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|Counter)|SQ_|TCC_|GRBM|VALU|Occupancy|MemUnit|FETCH|WRITE" | head -400 > $O/counters_available.txt)
# 1. the bench line as the driver runs it
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_r02_n1.json 2> $O/bench_r02_n1.err
# 2. kernel stats + trace of the same command
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-scripted --cpu-baseline-seconds 0 > $O/bench_under_rocprof.json 2> /tmp/prof_b.err)
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); python - "$f" > $O/rocprof_bench_r02_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
print(",".join(rows[0]))
for r in rows[1:]:
    r[0] = r[0][:140]
    print(",".join('"%s"' % x if i == 0 else x for i, x in enumerate(r)))
PY
python tools/verify_by_grid.py /tmp/prof_b > $O/rocprof_bench_r02_verify_by_grid.txt 2>&1
# 3. HBM traffic of the verify launch
bash tools/pmc_verify.sh > $O/pmc_verify.log 2>&1; cp gpurun_out/pmc/pmc_verify.json $O/ 2>/dev/null
# 4. batch 1 (config 2 / 3 literal) and the drivers
timeout 600 python bench.py --prompts-per-gpu 1 --steps 48 --warmup 8 --cpu-baseline-seconds 0 --no-shapes > $O/bench_b1.json 2>> $O/bench_r02_n1.err
timeout 600 python bench.py --prompts-per-gpu 8 --steps 48 --warmup 8 --cpu-baseline-seconds 0 --no-shapes > $O/bench_b8.json 2>> $O/bench_r02_n1.err
timeout 600 python bench.py --total-prompts 64 --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-shapes > $O/bench_strong64.json 2>> $O/bench_r02_n1.err
(timeout 600 python -m jacobiforcing_amd.drivers.sb_math500 --synthetic 4 --n 16 --max-new-tokens 256 --csv /tmp/sb.csv; tail -5 /tmp/sb.csv) > $O/driver_sb_n16.txt 2>&1
(timeout 600 python -m jacobiforcing_amd.drivers.ar_baseline --synthetic 2 --max-new-tokens 256) > $O/driver_ar.txt 2>&1
(timeout 900 python -m jacobiforcing_amd.drivers.mr_humaneval --synthetic 8 --batch 1 --max-new-tokens 256 --csv /tmp/mr.csv; tail -12 /tmp/mr.csv) > $O/driver_mr_b1.txt 2>&1
# 5. engine path
timeout 1200 python tools/engine_throughput.py --batch 64 --max-tokens 96 > $O/engine_throughput_r02.txt 2>&1
