#!/bin/bash
# concurrent batch slices on the final tree
mkdir -p gpurun_out/r3aa
O=gpurun_out/r3aa
run() { tag=$1; shift; env "$@" timeout 200 python tools/unit_table.py --tag $tag --quiet --json $O/$tag.json > $O/$tag.txt 2>&1; tail -1 $O/$tag.txt | cut -c1-90; }
run base A=1
run slices CSN_SLICE_LANES=1
run base2 A=1
run slices2 CSN_SLICE_LANES=1
