#!/bin/bash
# A/B builds of libcsnet_hip.so with extra -D flags (selected at run time through SOD100K_HIP_LIB); the outputs land in
# gpurun_variants/ (git-ignored, shipped to the GPU box).
# usage: tools/build_variant.sh <name> <file.hip[,file.hip...]> [-DFLAG ...]
# Only the listed sources are recompiled with the flags; the other objects are built once (cached in /tmp/csn_obj, keyed by
# the source's mtime) and linked in.
set -e
name=$1; files=$2; shift; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/gpurun_variants" /tmp/csn_obj
cd "$ROOT/sod100k_amd/csrc"
SRCS="csn_plan.hip k_misc.hip k_goct_pw.hip k_ms.hip k_train.hip k_wgrad.hip k_wgrad_c3.hip k_wgrad_bf.hip k_goct_c3.hip k_csf.hip k_pw4.hip k_c3q.hip k_pwq.hip k_ilb.hip"
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
objs=""
for s in $SRCS; do
  if [[ ",$files," == *",$s,"* ]]; then
    o=/tmp/csn_obj/${s%.hip}.$name.o
    $CC "$@" -c $s -o $o
  else
    o=/tmp/csn_obj/${s%.hip}.o
    if [ ! -f $o ] || [ $s -nt $o ] || [ -n "$(find . -maxdepth 1 \( -name '*.h' -o -name '*.inl' \) -newer $o)" ]; then $CC -c $s -o $o; fi
  fi
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$ROOT/gpurun_variants/lib_$name.so" $objs
echo built gpurun_variants/lib_$name.so
