#!/bin/bash
# several library variants over several shapes, one process per (variant, shape), variants interleaved per shape:
#   LIBS="product tools/ab/a.so tools/ab/b.so" tools/ab/multi_ab.sh "4096 16 5" "16384 1 3" ...   -> one line per shape: ms per variant
cd "$(dirname "$0")/../.."
for shape in "$@"; do
  line="$shape :"
  for lib in $LIBS; do
    if [ "$lib" = product ]; then unset BARK_LIB_PATH; else export BARK_LIB_PATH=$PWD/$lib; fi
    out=$(timeout -k 10 180 python3 tools/profile_mll.py $shape 2>&1 | tail -1)
    ms=$(echo "$out" | sed -n 's/.*: \([0-9.]*\) ms per call.*/\1/p')
    [ -n "$DIGEST" ] && ms="$ms/$(echo "$out" | sed -n 's/.* mll=\([0-9a-f]*\).*/\1/p')"   # DIGEST=1: + hash of the MLL vector
    line="$line  $(basename $lib .so)=${ms:-FAIL}"
  done
  echo "$line"
done
