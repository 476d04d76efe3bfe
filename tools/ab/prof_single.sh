# usage: prof_single.sh [N]  — kernel trace of a single-matrix MLL sweep (tools/profile_c5.py), per-kernel averages
export PYTHONPATH=$PWD
ROOT=$PWD
N=${1:-16384}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/single_$N
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/single_$N -o t -- python3 $ROOT/tools/profile_c5.py $N > /dev/null 2>&1 || exit 1
python3 $ROOT/tools/ab/kstats.py /tmp/single_$N
