# usage: swap_prof.sh <lib.so>...  — kernel-trace of tools/profile_swap.py per library variant, summary to stdout
cp bark_amd/csrc/libbarkhip.so /tmp/orig.so
export PYTHONPATH=$PWD
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  cp $ROOT/$lib $ROOT/bark_amd/csrc/libbarkhip.so
  rm -rf /tmp/swapab_$name
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/swapab_$name -o t -- python3 $ROOT/tools/profile_swap.py 4096 100 > /dev/null 2>&1 || exit 1
  echo "== $name"
  python3 $ROOT/tools/ab/kstats.py /tmp/swapab_$name skinny colsum small_kernel reduce_shares left_factor rank_update expand
done
cp /tmp/orig.so $ROOT/bark_amd/csrc/libbarkhip.so
