#!/bin/bash
# round 4, session 2: the whole GPU suite (no -x)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -n 6 > $O/r4b_gputest.log 2>&1; tail -15 $O/r4b_gputest.log
