# usage: env.sh "LABEL:VAR=VAL ..." ...   (same library, different environment switches)
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  env BARK_BENCH_NOCHECK=1 $envs timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 ${BENCH_ARGS:-} 2>gpurun_out/err_$label.log | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$label', round(r['value'],1), round(r['ms_per_step'],2), r['roofline']['ms_per_step'])" || { echo "$label failed"; tail -3 gpurun_out/err_$label.log; }
done
