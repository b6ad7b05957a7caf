#!/bin/bash
# A/B builds of libcsnet_hip.so with extra -D flags (selected at run time through SOD100K_HIP_LIB); the outputs land in
# gpurun_variants/ (git-ignored, shipped to the GPU box).   usage: tools/build_variant.sh <name> [-DFLAG ...]
set -e
name=$1; shift
mkdir -p gpurun_variants
cd sod100k_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result "$@" \
  -o ../../gpurun_variants/lib_$name.so csn_plan.hip k_misc.hip k_goct_pw.hip k_ms.hip k_train.hip k_wgrad.hip k_wgrad_c3.hip k_goct_c3.hip k_csf.hip k_ilb.hip
echo built gpurun_variants/lib_$name.so
