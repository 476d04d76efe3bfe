#!/bin/bash
# kernel timeline + per-step periods of the last sweep of a profiled run (rocprofv3 --kernel-trace; run on the GPU box from the repo root)
# usage: tools/ab/timeline.sh <label> <lib.so|product> N B steps skip count
ROOT=$(pwd); label=$1; lib=$2; N=$3; B=$4; steps=$5; skip=$6; count=$7
[ "$lib" != "product" ] && export BARK_LIB_PATH=$ROOT/$lib
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$label
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tl_$label -o t -- python3 $ROOT/tools/profile_mll.py $N $B $steps > /dev/null 2>&1 || { echo trace failed; exit 1; }
python3 $ROOT/tools/ab/ktimeline.py /tmp/tl_$label $skip $count
python3 $ROOT/tools/ab/ksteps_period.py /tmp/tl_$label 4
