#!/bin/bash
mkdir -p gpurun_out/r3c
O=$PWD/gpurun_out/r3c
./tools/probes/mfma4_probe > $O/mfma4.txt 2>&1; cat $O/mfma4.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
( timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/tools/unit_table.py --steps 5 --iters 1 --quiet ) > $O/trace.log 2>&1
cd $R
python tools/trace_forward.py $O/trace > $O/forward.txt 2>&1; tail -60 $O/forward.txt
