for cfg in "" "JF_ARGMAX_ITEMS=512" "JF_ARGMAX_ITEMS=768" "JF_ARGMAX_ITEMS=1536" "JF_ARGMAX_ITEMS=2048" "JF_ARGMAX_ITEMS=4096" "JF_ARGMAX_WAVE=1" "JF_ARGMAX_WAVE=1 JF_ARGMAX_ITEMS=2048" "JF_ARGMAX_UNROLL=4" "JF_ARGMAX_UNROLL=16"; do
  r=$(env $cfg timeout 300 python bench.py --steps 48 --warmup 8 --no-scripted --cpu-baseline-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['us_per_launch'],2), round(d['ms_per_step'],3))")
  echo "$cfg => $r"
done
