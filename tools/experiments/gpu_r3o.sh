#!/bin/bash
# sweep of the row-strided MSBlock kernel: rows per lane, pixels per lane, double buffering (one lease)
mkdir -p gpurun_out/r3o
O=gpurun_out/r3o
run() { tag=$1; shift; env "$@" timeout 200 python tools/unit_table.py --tag $tag --json $O/$tag.json > $O/$tag.txt 2>&1; echo "$tag $(grep -h 'ms.convs' $O/$tag.txt | awk '{printf "%s ", $3}') $(tail -1 $O/$tag.txt | cut -c1-60)"; }
run base CSN_MS_ROWS=0
for R in 4 6 8; do for PX in 4 2; do for DB in 0 1; do
  run r${R}p${PX}d${DB} CSN_MS_ROWS=2 CSN_MS_R=$R CSN_MS_PX=$PX CSN_MS_DB=$DB
done; done; done
