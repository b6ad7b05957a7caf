#!/bin/bash
# A/B builds of libcsnet_hip.so with extra -D flags (selected at run time through SOD100K_HIP_LIB); the outputs land in
# gpurun_variants/ (git-ignored, shipped to the GPU box).
# usage: tools/build_variant.sh <name> <file.hip[,file.hip...]> [-DFLAG ...]
# Only the listed sources are recompiled with the flags (in parallel); every other object is the product build's own
# (sod100k_amd/csrc/build/*.o, same compiler flags: run _native.build() first -- this script does).
set -e
name=$1; files=$2; shift; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/gpurun_variants" /tmp/csn_obj
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from sod100k_amd import _native; _native.build()"
cd "$ROOT/sod100k_amd/csrc"
SRCS=$(python3 -c "import sys; sys.path.insert(0, '$ROOT'); from sod100k_amd import _native; print(' '.join(_native.SOURCES))")
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
objs=""
for s in $SRCS; do
  if [[ ",$files," == *",$s,"* ]]; then
    o=/tmp/csn_obj/${s%.hip}.$name.o
    $CC "$@" -c $s -o $o &
  else
    o=build/${s%.hip}.o
  fi
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$ROOT/gpurun_variants/lib_$name.so" $objs build/csn_build_id.o
echo built gpurun_variants/lib_$name.so
