#!/bin/bash
# channels per step of the third input / low branch in the high-only pw4 forms (CSFHead.fuse branch 0)
mkdir -p gpurun_out/r3r
O=gpurun_out/r3r
run() { tag=$1; shift; env "$@" timeout 200 python tools/unit_table.py --tag $tag --json $O/$tag.json > $O/$tag.txt 2>&1; echo "$tag $(grep -h 'oct_fuse.fuse \|stage2.3.conv1x1\|fuse1x1' $O/$tag.txt | awk '{printf "%s ", $3}') $(tail -1 $O/$tag.txt | cut -c1-70)"; }
run xb4 A=1
run xb1 SOD100K_HIP_LIB=gpurun_variants/lib_xb1.so
run xb2 SOD100K_HIP_LIB=gpurun_variants/lib_xb2.so
run lbh4 SOD100K_HIP_LIB=gpurun_variants/lib_lbh4.so
run xb4b A=1
