#!/bin/bash
# usage: tools/hbm_counters.sh [N] [B] [steps]  -> gpurun_out/hbm_counters.json
# HBM byte counters of the dense MLL sweep: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one
# pass on gfx950), counter collection only (no trace domains), the interpreter directly after `--`.
ROOT=$(pwd)
N=${1:-4096}; B=${2:-256}; STEPS=${3:-1}
export PYTHONPATH=$ROOT
# counter collection serialises the dispatches: the device-side hand-over of chain-bound sweeps (bark_device_wait) would wait
# for a row launch that cannot run beside diag_kernel — event joins for these passes
export BARK_NO_DEVICE_WAIT=1
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o t -- python3 $ROOT/tools/profile_mll.py $N $B $STEPS > /dev/null 2>&1 || { echo "pmc pass $c failed"; exit 1; }
done
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/hbm_counters.json $N $B 50 $((STEPS + 1)) FETCH_SIZE=/tmp/pmc_FETCH_SIZE WRITE_SIZE=/tmp/pmc_WRITE_SIZE
