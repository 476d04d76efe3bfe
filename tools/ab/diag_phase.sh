# usage: diag_phase.sh <lib.so>... — diag_kernel average duration in a lone-matrix sweep (N=4096) per library variant
cp bark_amd/csrc/libbarkhip.so /tmp/orig.so
export PYTHONPATH=$PWD
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  cp $ROOT/$lib $ROOT/bark_amd/csrc/libbarkhip.so
  rm -rf /tmp/dp_$name
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/dp_$name -o t -- python3 $ROOT/tools/profile_c5.py 4096 > /dev/null 2>&1
  echo "== $name"; python3 $ROOT/tools/ab/kstats.py /tmp/dp_$name diag_kernel
done
cp /tmp/orig.so $ROOT/bark_amd/csrc/libbarkhip.so
