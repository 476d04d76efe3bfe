#!/bin/bash
# per-block-step periods of the last sweep of a profiled run (every step), one line per step:  tools/ab/steps.sh <label> <lib.so|product> N B steps [C]
ROOT=$(pwd); label=$1; lib=$2; shift 2
[ "$lib" != "product" ] && export BARK_LIB_PATH=$ROOT/$lib
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_$label
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/st_$label -o t -- python3 $ROOT/tools/profile_mll.py "$@" > /dev/null 2>&1 || { echo trace failed; exit 1; }
python3 $ROOT/tools/ab/ksteps_period.py /tmp/st_$label 1
python3 $ROOT/tools/ab/kstats.py /tmp/st_$label
