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
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=512
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=768
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=1024
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=1280
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=1536
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=2048
run JF_ARGMAX_WAVE=0 JF_ARGMAX_ITEMS=3072
done
