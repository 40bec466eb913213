#!/bin/bash
# Build a variant of the product library beside it: tools/build_variant.sh <name> [extra hipcc flags...] [--base <git rev>]
# -> bayesian_torch_amd/libbtorch_hip_<name>.so (git-ignored; select with BT_LIB_PATH). Used for A/B kernel measurements on one box.
set -e
name=$1; shift
rev=""
flags=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--base" ]; then rev=$2; shift 2; else flags+=("$1"); shift; fi
done
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/btvar.XXXX)
mkdir -p $tmp/pkg $tmp/include
if [ -n "$rev" ]; then
  git -C $root archive $rev bayesian_torch_amd/csrc include | tar -x -C $tmp
  mv $tmp/bayesian_torch_amd/csrc $tmp/pkg/csrc
else
  cp -r $root/bayesian_torch_amd/csrc $tmp/pkg/csrc; cp $root/include/*.h $tmp/include/
fi
rm -f $tmp/pkg/csrc/*.o
make -C $tmp/pkg/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DBT_STAMPS=0 ${flags[*]}" > $tmp/build.log 2>&1 || { tail -20 $tmp/build.log; exit 1; }
cp $tmp/pkg/libbtorch_hip.so $root/bayesian_torch_amd/libbtorch_hip_$name.so
rm -rf $tmp
echo built bayesian_torch_amd/libbtorch_hip_$name.so
