#!/bin/bash
# max-pooled copies written by the depthwise pair (default) vs pool2_kernel launches
mkdir -p gpurun_out/r3ac
O=gpurun_out/r3ac
run() { tag=$1; shift; env "$@" timeout 200 python tools/unit_table.py --tag $tag --quiet --json $O/$tag.json > $O/$tag.txt 2>&1; tail -1 $O/$tag.txt | cut -c1-200; }
run fused A=1
run nofuse CSN_NO_MP_FUSE=1
run fused2 A=1
run nofuse2 CSN_NO_MP_FUSE=1
