set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r02b
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4 > $O/gputest.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_r02_n1.json 2> $O/bench.err
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
bash tools/pmc_verify.sh > $O/pmc_verify.log 2>&1; cp gpurun_out/pmc/pmc_verify.json $O/ 2>/dev/null
JF_FUSED_VERIFY=0 timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 > $O/bench_unfused.json 2>> $O/bench.err
timeout 600 python bench.py --prompts-per-gpu 1 --steps 48 --warmup 8 --cpu-baseline-seconds 0 --no-shapes > $O/bench_b1.json 2>> $O/bench.err
timeout 600 python bench.py --prompts-per-gpu 8 --steps 48 --warmup 8 --cpu-baseline-seconds 0 --no-shapes > $O/bench_b8.json 2>> $O/bench.err
timeout 600 python bench.py --total-prompts 64 --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-shapes > $O/bench_strong64.json 2>> $O/bench.err
