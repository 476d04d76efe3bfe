#!/bin/bash
# usage: tools/kernel_trace.sh <label> <lib.so|product> [profile_mll args...]  -> gpurun_out/trace_<label>.txt (per-kernel n / avg / total)
ROOT=$(pwd)
label=$1; lib=$2; shift 2
[ "$lib" != "product" ] && export BARK_LIB_PATH=$ROOT/$lib
export PYTHONPATH=$ROOT
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$label
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/kt_$label -o t -- python3 $ROOT/tools/profile_mll.py "$@" > $ROOT/gpurun_out/trace_$label.txt 2>/dev/null || { echo "trace $label failed"; exit 1; }
python3 $ROOT/tools/ab/kstats.py /tmp/kt_$label >> $ROOT/gpurun_out/trace_$label.txt
cat $ROOT/gpurun_out/trace_$label.txt
