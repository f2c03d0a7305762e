#!/bin/bash
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
for F in 1 0; do
  JF_MB_FAST=$F bash tools/pmc_mb_step.sh > /dev/null 2>&1
  cp gpurun_out/pmc_mb/pmc_mb_step.txt gpurun_out/r3_pmc_mb_step_fast$F.txt
done
grep -A26 "^mb_step_kernel" gpurun_out/r3_pmc_mb_step_fast1.txt | grep -E "SQ_INSTS|SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY"
echo ----
grep -A26 "^mb_step_kernel" gpurun_out/r3_pmc_mb_step_fast0.txt | grep -E "SQ_INSTS|SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY"
timeout 300 python - <<'PY'
import torch, time
from jacobiforcing_amd import ops
for M in (1024, 2560, 4096):
    gu = torch.randn(M, 2*18944, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.swiglu(gu)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.swiglu(gu)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    mb = M * 18944 * 6 / 1e6
    print(f"swiglu M={M}: {us:.1f} us for {mb:.0f} MB = {mb/us/1e3*1e3/1e3:.2f} TB/s = {mb/us/8e3*1e3/1e3:.3f} of 8 TB/s")
PY
