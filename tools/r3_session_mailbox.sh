#!/bin/bash
# round 3: does the host ever see the mailbox's sequence word before the tables in front of it?  12 processes side by side.
mkdir -p gpurun_out
export TMPDIR=/tmp
for L in shipped pubfence; do   # (the first session ran the then-shipped plain stores as 'shipped': 61 000 stale tables of 66 000 rounds per process)
  if [ $L = shipped ]; then unset JF_LIB; else export JF_LIB=tools/libjf_exp_pubfence.so; fi
  echo "== $L"
  pids=""
  for i in $(seq 1 12); do timeout 120 python tools/mailbox_stress.py --seconds 25 > /tmp/ms_$i.log 2>&1 & pids="$pids $!"; done
  for p in $pids; do wait $p; done
  cat /tmp/ms_*.log | grep -v amdgpu.ids | grep "rounds\|stale table" | sort | uniq -c | sort -rn | head -8
done > gpurun_out/r3k_mailbox.txt 2>&1
cat gpurun_out/r3k_mailbox.txt
