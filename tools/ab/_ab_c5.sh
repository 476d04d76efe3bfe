set -e
for cfg in "16384 1 3" "16384 1 2 10000" "4096 1 9" "4096 8 7" "4096 16 7" "8192 2 5" "8192 8 3"; do
  for lib in product las1; do
    if [ $lib = product ]; then unset BARK_LIB_PATH; else export BARK_LIB_PATH=$PWD/tools/ab/$lib.so; fi
    python tools/profile_mll.py $cfg 2>&1 | tail -1
  done
done
