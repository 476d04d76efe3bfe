#!/bin/bash
# usage: tools/sq_counters.sh <label> [N] [B] [steps]  -> gpurun_out/sq_counters_<label>.json
# SQ / GRBM counters of the dense MLL sweep, one --pmc pass (8 SQ slots + GRBM; no trace domains beside it), the
# interpreter directly after `--` (tools/profile_mll.py is the worker itself: no launcher hop under the profiler).
ROOT=$(pwd)
label=$1; N=${2:-4096}; B=${3:-256}; STEPS=${4:-1}
export PYTHONPATH=$ROOT
# counter collection serialises the dispatches: the device-side hand-over of chain-bound sweeps (bark_device_wait) would wait
# for a row launch that cannot run beside diag_kernel — event joins for these passes
export BARK_NO_DEVICE_WAIT=1
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq_$label
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d /tmp/pmc_sq_$label -o t -- python3 $ROOT/tools/profile_mll.py $N $B $STEPS > /dev/null 2>&1 || { echo "sq pmc pass failed"; exit 1; }
python3 $ROOT/tools/sq_summary.py /tmp/pmc_sq_$label $ROOT/gpurun_out/sq_counters_$label.json $N $B $((STEPS + 1))
