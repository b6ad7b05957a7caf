#!/bin/bash
# First GPU session: parity tests, smoke, bench at several sub-batch sizes, rocprof kernel stats.
mkdir -p gpurun_out
R=$PWD
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log
( timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' ) > gpurun_out/smoke.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench_sb0.log 2>&1
for sb in 8 16 32; do
  ( timeout 300 python bench.py --steps 20 --warmup 5 --sub-batch $sb --no-cpu-baseline ) > gpurun_out/bench_sb$sb.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1 -o run1 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $R/gpurun_out/rocprof1.log 2>&1
cd $R
find gpurun_out/prof1 -name "*stats*" | head
rocminfo | grep -E "Marketing|Compute Unit|Max Clock" | head -8 > gpurun_out/rocminfo.txt
cat gpurun_out/pytest_gpu.log | tail -15
cat gpurun_out/smoke.log | tail -3
tail -c 600 gpurun_out/bench_sb0.log
