set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5 > gpurun_out/r2_gputest8.log
bash tools/pmc_rs_probs.sh > gpurun_out/r2_pmc_rs.log 2>&1
timeout 600 python tools/microbench_rs.py 1.0 0.8 0.7 > gpurun_out/r2_rs_probs.log 2>&1
