# usage: single.sh <lib.so>...  — single-matrix timings (tools/time_refactor.py 4096, validate c5-like MLL) per library variant
cp bark_amd/csrc/libbarkhip.so /tmp/orig.so
export PYTHONPATH=$PWD
for lib in "$@"; do
  cp $lib bark_amd/csrc/libbarkhip.so
  echo "== $lib"
  timeout -k 10 200 python tools/time_refactor.py 4096 2>/dev/null | head -2
  timeout -k 10 200 python tools/time_refactor.py 16384 2>/dev/null | head -2
done
cp /tmp/orig.so bark_amd/csrc/libbarkhip.so
