#!/bin/bash
# pw4 tile width on the final tree
mkdir -p gpurun_out/r3ad
O=gpurun_out/r3ad
run() { tag=$1; shift; env "$@" timeout 200 python tools/unit_table.py --tag $tag --json $O/$tag.json > $O/$tag.txt 2>&1; tail -1 $O/$tag.txt | cut -c1-110; }
run w16 A=1
run w32 CSN_PW4_TWL=5
run w64 CSN_PW4_TWL=6
run w16b A=1
