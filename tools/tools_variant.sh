#!/bin/bash
# usage: tools_variant.sh <name> [git-rev] [extra hipcc flags] — dev: build a variant of the library into
# build/ab/lib<name>.so (from a git revision's csrc when given, else the working tree), for A/B timing on one box
# with PHYLONIUM_AMD_LIB (tools_ab.sh).
set -e
NAME=$1; REV=${2:-WORK}; EXTRA=$3
ROOT=$(cd $(dirname $0)/.. && pwd)
TMP=$(mktemp -d)
mkdir -p $TMP/phylonium_amd $TMP/include
if [ "$REV" = "WORK" ]; then
  cp -r $ROOT/phylonium_amd/csrc $ROOT/phylonium_amd/host $TMP/phylonium_amd/; cp $ROOT/include/*.h $TMP/include/
else
  git -C $ROOT archive $REV phylonium_amd/csrc phylonium_amd/host include | tar -x -C $TMP
fi
rm -f $TMP/phylonium_amd/csrc/*.o
make -s -j6 -C $TMP/phylonium_amd/csrc ../libphylonium_amd.so FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $EXTRA" >/dev/null
mkdir -p $ROOT/build/ab
cp $TMP/phylonium_amd/libphylonium_amd.so $ROOT/build/ab/lib$NAME.so
rm -rf $TMP
echo built build/ab/lib$NAME.so
