#!/bin/bash
# what the driver runs at round end, as it runs it (serial pytest, smoke, bench)
mkdir -p gpurun_out
O=gpurun_out
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $O/r4_driver_gputest.log 2>&1; tail -6 $O/r4_driver_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r4_driver_bench.log 2>&1; grep real $O/r4_driver_bench.log; grep -c '"metric"' $O/r4_driver_bench.log
