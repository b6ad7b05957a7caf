#!/bin/bash
# HBM counter passes (FETCH_SIZE / WRITE_SIZE, separate runs as gpurun requires) of the calibration probe and of one eval
# forward -> gpurun_out/$1/{cal,fwd}_{fetch,write}/ ; tools/pmc_hbm.py turns them into profiles/<tag>_pmc_hbm.json
out=$PWD/gpurun_out/$1; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
( timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -o c -- $R/tools/probes/fetch_cal ) > $out/cal_fetch.log 2>&1
( timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -o c -- $R/tools/probes/fetch_cal ) > $out/cal_write.log 2>&1
( timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fwd_fetch -o f -- python $R/tools/unit_table.py --steps 2 --iters 1 --quiet ) > $out/fwd_fetch.log 2>&1
( timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/fwd_write -o f -- python $R/tools/unit_table.py --steps 2 --iters 1 --quiet ) > $out/fwd_write.log 2>&1
cd $R
python tools/pmc_hbm.py $out ${2:-r3} 2>&1 | tee $out/pmc_hbm.txt
