set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rev in 0 1 0 1; do
JF_ARGMAX_REVERSE=$rev timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-shapes > gpurun_out/r2_bench_rev$rev.json 2>> gpurun_out/r2_bench_rev.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2_bench_rev$rev.json") if l.startswith("{")][0])
print("reverse=$rev", round(d["ms_per_step"],2), "ms/step  verify us", round(d["roofline"]["us_per_launch"],1), "frac", round(d["roofline"]["frac"],3), "| scripted us", round(d["scripted_acceptance"]["roofline"]["us_per_launch"],1), "frac", round(d["scripted_acceptance"]["roofline"]["frac"],3))
PY
done > gpurun_out/r2_reverse_ab.txt 2>&1
JF_ARGMAX_REVERSE=1 timeout 600 python -m pytest tests/test_kernels.py tests/test_multiblock.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3 >> gpurun_out/r2_reverse_ab.txt
