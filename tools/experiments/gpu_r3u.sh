#!/bin/bash
# accumulator budget of the two-output pw4 groups (more, smaller M groups)
mkdir -p gpurun_out/r3u
O=gpurun_out/r3u
run() { tag=$1; shift; env "$@" timeout 200 python tools/unit_table.py --tag $tag --json $O/$tag.json > $O/$tag.txt 2>&1; echo "$tag $(grep -h 'stage[12].[123].conv1x1' $O/$tag.txt | awk '{printf "%s ", $3}') $(tail -1 $O/$tag.txt | cut -c1-70)"; }
run b100 A=1
run b80 CSN_PW4_BUDGET2=80
run b64 CSN_PW4_BUDGET2=64
run b48 CSN_PW4_BUDGET2=48
run b100b A=1
