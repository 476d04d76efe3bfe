#!/bin/bash
# product library against a variant, one process per (variant, shape):  tools/ab/lib_ab.sh tools/ab/<variant>.so "4096 1 7" "1024 1 9" ...
cd "$(dirname "$0")/../.."
var=$1; shift
for shape in "$@"; do
  for lib in product $var; do
    if [ "$lib" = product ]; then unset BARK_LIB_PATH; else export BARK_LIB_PATH=$PWD/$lib; fi
    timeout -k 10 120 python3 tools/profile_mll.py $shape 2>&1 | tail -1
  done
done
