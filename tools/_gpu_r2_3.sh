cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/verify_trace.py > gpurun_out/r2_verify_trace_plain.log 2>&1
JF_LIB=$GRAFT_REPO_ROOT/tools/libjf_exp_vtrace.so timeout 600 python tools/verify_trace.py > gpurun_out/r2_verify_trace_stamps.log 2>&1
