cp bark_amd/csrc/libbarkhip.so /tmp/orig.so
for spec in "$@"; do
  lib=${spec%%:*}; label=${spec#*:}
  cp $lib bark_amd/csrc/libbarkhip.so
  BARK_BENCH_NOCHECK=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --batch 16 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); g=r['roofline']['gram_kernel']; print('$label', round(g['ms'],4), 'ms', round(g['achieved_GBs'],1), 'GB/s')" || echo "$label failed"
done
cp /tmp/orig.so bark_amd/csrc/libbarkhip.so
