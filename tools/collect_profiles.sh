#!/bin/bash
# usage (on the GPU box, repo root): tools/collect_profiles.sh [rNN]  -> gpurun_out/prof_rNN/*  (copy what is wanted into profiles/rNN/)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${1:-r05}
mkdir -p $OUT
export PYTHONPATH=$ROOT
python3 bench.py > $OUT/bench_c3_final.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
for spec in "c3 4096 256 2" "b64 4096 64 3" "b16 4096 16 3" "b8 4096 8 3" "b1 4096 1 5" "c2 1024 1 10" "c5 16384 1 2" "c5_posterior 16384 1 1 10000" "n64 64 256 10" "n128 128 256 10" "n256 256 256 10" "n512 512 256 10" "n768 768 256 10"; do
  set -- $spec; label=$1; shift
  tools/kernel_trace.sh $label product "$@" > /dev/null || exit 1
  cp gpurun_out/trace_$label.txt $OUT/trace_$label.txt
done
tools/hbm_counters.sh 4096 256 1 > $OUT/hbm_counters.log || exit 1
cp gpurun_out/hbm_counters.json $OUT/bench_c3_hbm_counters.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_api
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt_api -o t -- python3 $ROOT/tools/profile_api_surface.py > /dev/null 2>&1 || { echo api trace failed; exit 1; }
python3 $ROOT/tools/ab/kstats.py /tmp/kt_api > $OUT/api_surface_kernels.txt
rm -rf /tmp/ks
# bench.py without RANK / WORLD_SIZE is a LAUNCHER that starts its worker as a child process; under the profiler that
# child would be spawned from a process whose preloaded profiler library may already have initialised the GPU.  With
# the rank variables set, the program after `--` is the worker itself (bench.py main(): worker path, no new process).
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o t -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-configs > $OUT/bench_under_rocprof.json 2>/dev/null || { echo stats run failed; exit 1; }
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/bench_c3_kernel_stats_final.csv
unset RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT
cd $ROOT
for spec in "c3 4096 256 1" "b1 4096 1 3" "c2 1024 1 5"; do
  set -- $spec
  tools/sq_counters.sh $1 $2 $3 $4 > $OUT/sq_$1.log || exit 1
  cp gpurun_out/sq_counters_$1.json $OUT/sq_counters_$1.json
done
ls -la $OUT
