#!/bin/bash
# One GPU lease, a list of steps (round 5: replaces the one-off gpu_r3*/gpu_r4* scripts that used to pile up under tools/experiments/).
#   gpurun --timeout T -- 'bash tools/gpu_lease.sh <outdir> <step> [<step> ...]'
# Everything a step prints lands in gpurun_out/<outdir>/<step-name>.log (merged back by gpurun); a one-line summary goes to stdout.
# steps:
#   probe:<name>                   tools/probes/<name> (binary built in this container; travels with the tree)
#   tests[:<pytest -k expression>] GPU tests (whole suite without an expression)
#   unit[:<tag>[:ENV=V,ENV=V]]     tools/unit_table.py (per-unit table of the eval forward), optional environment switches
#   bench[:<tag>[:ENV=V,...[:bench args,comma separated]]]   python bench.py (default arguments = the driver's line)
#   eval[:<tag>[:ENV=V,...]]       bench.py, eval forward only (no train points, no CSF, no CPU baseline, no batch-1 latency)
#   sh:<file>                      any other script of tools/ (kept for the round-end collection)
O=$PWD/gpurun_out/$1; mkdir -p $O; shift
R=$PWD
EV="--train-steps 0 --csf-batch 0 --no-cpu-baseline --no-latency-b1"
for spec in "$@"; do
  IFS=: read -r kind a b c <<< "$spec"
  E=(X=1); [ -n "$b" ] && E+=(${b//,/ })
  case $kind in
    probe) ( timeout 120 tools/probes/$a ) > $O/probe_$a.log 2>&1; echo "== probe $a"; tail -n 40 $O/probe_$a.log ;;
    tests) if [ -n "$a" ]; then ( timeout 1500 python -m pytest tests -m gpu -q -s -k "$a" 2>&1 | tail -60 ) > $O/tests.log
           else ( timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 ) > $O/tests.log; fi
           echo "== tests $a"; grep -E "passed|failed|error|worst|bce|unpruned" $O/tests.log | tail -12 ;;
    unit) ( env "${E[@]}" timeout 300 python tools/unit_table.py --json $O/unit_${a:-base}.json ) > $O/unit_${a:-base}.log 2>&1
          echo "== unit ${a:-base} ${b}"; tail -n 3 $O/unit_${a:-base}.log ;;
    bench) ( env "${E[@]}" timeout 900 python bench.py ${c//,/ } ) > $O/bench_${a:-full}.json 2> $O/bench_${a:-full}.err
           echo "== bench ${a:-full} ${b}"; python tools/bench_line.py $O/bench_${a:-full}.json ;;
    eval) ( env "${E[@]}" timeout 300 python bench.py $EV ) > $O/eval_${a:-base}.json 2> $O/eval_${a:-base}.err
          echo "== eval ${a:-base} ${b}"; python tools/bench_line.py $O/eval_${a:-base}.json ;;
    sh) bash tools/$a ;;
    *) echo "unknown step $spec" ;;
  esac
done
