set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/verify_trace.py > gpurun_out/r2_verify_trace_plain.log 2>&1
JF_LIB=$GRAFT_REPO_ROOT/tools/libjf_exp_vtrace.so timeout 600 python tools/verify_trace.py > gpurun_out/r2_verify_trace_stamps.log 2>&1
timeout 900 python -m pytest tests/test_multiblock.py tests/test_multiblock_fuzz.py tests/test_hf_seam.py tests/test_decoder_e2e.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5 > gpurun_out/r2_gputest5.log
