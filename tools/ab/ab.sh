# usage: ab.sh <lib.so> <label> [env...]
cp bark_amd/csrc/libbarkhip.so /tmp/orig.so
for spec in "$@"; do
  lib=${spec%%:*}; rest=${spec#*:}; label=${rest%%:*}; envs=${rest#*:}; [ "$envs" = "$rest" ] && envs=""
  cp $lib bark_amd/csrc/libbarkhip.so
  env BARK_BENCH_NOCHECK=1 $envs timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); f=r['roofline']; print('$label', round(r['value'],1), f['ms_per_step'], round(f['panel_kernel']['executed_tflops'],2))" || echo "$label failed"
done
cp /tmp/orig.so bark_amd/csrc/libbarkhip.so
