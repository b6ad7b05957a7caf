#!/bin/bash
mkdir -p gpurun_out/r3af
O=gpurun_out/r3af
run() { tag=$1; shift; env "$@" timeout 100 python tools/unit_table.py --tag $tag --quiet --json $O/$tag.json > $O/$tag.txt 2>&1; tail -1 $O/$tag.txt | cut -c1-170; }
run base A=1
run c3q4 CSN_C3Q_TWL=4
run c3q5 CSN_C3Q_TWL=5
run dw56 CSN_DW2_LDS_KB=56
run base2 A=1
