#!/bin/bash
# kernel trace of tools/unit_table.py -> gpurun_out/$1/trace ; prints the replayed forward (-3) and the serialised one (-1)
O=$PWD/gpurun_out/$1; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
( timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/tools/unit_table.py --steps 5 --iters 1 --quiet ) > $O/trace.log 2>&1
cd $R
python tools/trace_forward.py $O/trace -3 > $O/replay.txt 2>&1
python tools/trace_forward.py $O/trace -1 > $O/serial.txt 2>&1
cat $O/replay.txt
