#!/bin/bash
# HBM counter passes (FETCH_SIZE / WRITE_SIZE, separate runs) over three train steps of each storage mode
# -> gpurun_out/$1/{cal,train}_{fetch,write}/ ; tools/pmc_train.py turns them into profiles/<tag>_pmc_train.json
# usage: gpu_pmc_train.sh <outdir> <tag> [x2 | unpruned]   (unpruned: the expand-2 training net -> profiles/<tag>_pmc_train_unpruned.json)
out=$PWD/gpurun_out/$1; mkdir -p $out
R=$PWD
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 1 --train-net ${3:-x2}"
cd /tmp && export TMPDIR=/tmp
( timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -o c -- $R/tools/probes/fetch_cal ) > $out/cal_fetch.log 2>&1
( timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -o c -- $R/tools/probes/fetch_cal ) > $out/cal_write.log 2>&1
( timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/train_fetch -o f -- $CMD ) > $out/train_fetch.log 2>&1
( timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/train_write -o f -- $CMD ) > $out/train_write.log 2>&1
cd $R
python tools/pmc_train.py $out ${2:-r3} ${3:-x2} 2>&1 | tee $out/pmc_train.txt
