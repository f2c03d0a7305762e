#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x -p no:cacheprovider -k "masked_argmax or timing_events or segment_counts" > gpurun_out/r4x_segtests.log 2>&1; tail -25 gpurun_out/r4x_segtests.log
