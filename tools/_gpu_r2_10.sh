set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5 > gpurun_out/r2_gputest9.log
