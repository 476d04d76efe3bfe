#!/bin/bash
# usage: tools/ab/build_variant.sh <name> [extra hipcc flags...]   ->  tools/ab/<name>.so  (git-ignored; travels with gpurun)
# Builds a variant of the product library from the CURRENT sources with extra -D flags (tuning constants only);
# to compare with an older revision:  GIT_REV=<rev> tools/ab/build_variant.sh old
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
src=$ROOT/bark_amd/csrc
if [ -n "$GIT_REV" ]; then
  tmp=$(mktemp -d); mkdir -p $tmp/bark_amd/csrc $tmp/include
  for f in $(git -C $ROOT ls-tree --name-only $GIT_REV bark_amd/csrc/ include/); do git -C $ROOT show $GIT_REV:$f > $tmp/$f; done
  src=$tmp/bark_amd/csrc
fi
objs=""
for f in $(cd $src && ls *.cpp *.hip); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -x hip -c $src/$f -o /tmp/abv_${name}_$f.o &
  objs="$objs /tmp/abv_${name}_$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o $ROOT/tools/ab/$name.so $objs -ldl
echo built tools/ab/$name.so
