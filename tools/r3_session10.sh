#!/bin/bash
export TMPDIR=/tmp
cd tools/_r02 && timeout 600 python tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 2>&1 | grep tok/s | sed 's/^/r02 tree: /'
cd $GRAFT_REPO_ROOT && timeout 600 python tools/engine_throughput.py --batch 64 --block-len 32 --max-tokens 96 2>&1 | grep tok/s | sed 's/^/r03 tree: /'
