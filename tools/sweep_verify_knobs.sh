#!/bin/bash
# Launch-shape knobs of the fused verify launch inside bench.py (64 prompts per GPU): per-wavefront vs 4-wave workgroup items,
# non-temporal vs plain loads, item count, row order.  -> profiles/verify_knobs_r02.txt
#   gpurun --timeout 3000 -- 'bash tools/sweep_verify_knobs.sh > gpurun_out/verify_knobs.txt 2>&1'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {
  env "$@" timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-shapes > /tmp/b.json 2>/dev/null
  python - "$*" <<'PY'
import json,sys
d=json.loads([l for l in open("/tmp/b.json") if l.startswith("{")][0])
print(f"{sys.argv[1]:44s} {d['ms_per_step']:6.2f} ms/step  verify {d['roofline']['us_per_launch']:6.1f} us frac {d['roofline']['frac']:.3f} | scripted {d['scripted_acceptance']['roofline']['us_per_launch']:6.1f} us frac {d['scripted_acceptance']['roofline']['frac']:.3f}")
PY
}
for i in 1 2; do
run JF_X=0
run JF_ARGMAX_WAVE=1
run JF_ARGMAX_NT=0
run JF_ARGMAX_REVERSE=1
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=512
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=768
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=1024
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=1280
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=1536
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=2048
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=3072
done
