#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x -p no:cacheprovider -k "segment_counts" > gpurun_out/r4x_segtests.log 2>&1; tail -5 gpurun_out/r4x_segtests.log
