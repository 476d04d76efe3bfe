# usage: swap.sh <lib.so>...   — tools/profile_swap.py with each library variant swapped in
cp bark_amd/csrc/libbarkhip.so /tmp/orig.so
export PYTHONPATH=$PWD
for lib in "$@"; do
  cp $lib bark_amd/csrc/libbarkhip.so
  echo "== $lib"
  timeout -k 10 200 python tools/profile_swap.py 4096 100 2>/dev/null
done
cp /tmp/orig.so bark_amd/csrc/libbarkhip.so
