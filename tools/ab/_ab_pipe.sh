set -e
for cfg in "4096 96 5" "4096 128 5" "4096 160 5" "4096 192 3" "2048 64 7" "2048 128 7" "2048 192 7" "1024 128 9" "1024 512 9" "8192 32 3" "8192 64 3" "8192 128 2" "512 512 9" "3000 100 5"; do
  for lib in product nopipe; do
    if [ $lib = product ]; then unset BARK_LIB_PATH; else export BARK_LIB_PATH=$PWD/tools/ab/$lib.so; fi
    python tools/profile_mll.py $cfg 2>&1 | tail -1
  done
done
