#!/bin/bash
# A/B inside one lease: split-output planning of the two-branch 1x1 units, row-strided MSBlock kernel
mkdir -p gpurun_out/r3n
O=gpurun_out/r3n
run() { tag=$1; shift; env "$@" timeout 200 python tools/unit_table.py --tag $tag --json $O/$tag.json > $O/$tag.txt 2>&1; tail -1 $O/$tag.txt; }
run base CSN_MS_ROWS=0
run msr1 CSN_MS_ROWS=1
run msr2 CSN_MS_ROWS=2
run split1 CSN_MS_ROWS=0 CSN_PW4_SPLIT_OUT=1
run split2 CSN_MS_ROWS=0 CSN_PW4_SPLIT_OUT=2
run split1b CSN_MS_ROWS=0 CSN_PW4_SPLIT_OUT=1 CSN_PW4_HI_BUDGET=164
run split2b CSN_MS_ROWS=0 CSN_PW4_SPLIT_OUT=2 CSN_PW4_HI_BUDGET=164
run base2 CSN_MS_ROWS=0
grep -h "ms.convs" $O/base.txt $O/msr1.txt $O/msr2.txt
