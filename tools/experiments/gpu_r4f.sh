#!/bin/bash
# round 4, lease F: train-step tests (verbose), bench (train only), kernel trace
O=$PWD/gpurun_out/${1:-r4f}; mkdir -p $O; R=$PWD
( timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_replay.py -m gpu -v -x -k "units_local or train_step_bf16 or train_golden or train_step_gradients or overlap2 or loss_goes_down or train_replay or batch256 or unpruned" 2>&1 | grep -v "^  File \"/usr" ) > $O/pytest_v.log 2>&1
grep -E "PASSED|FAILED|ERROR|Fatal|fault|Abort|passed|failed|Error" $O/pytest_v.log | cut -c1-150 | head -40
B="python bench.py --steps 5 --warmup 2 --train-steps 6 --csf-batch 0 --no-cpu-baseline --no-latency-b1 --event-steps 0 --profile-iters 1"
( timeout 300 $B ) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("eval", d["value"], "train fp32", d["train_step"]["ms_per_step"], "bf16", d["train_step_bf16"]["ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 3 ) > $O/trace.log 2>&1
cd $R
python tools/train_step_breakdown.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 0 > $O/train_step_kernels.md 2>&1
cp $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv 2>/dev/null; gzip -f $O/kernel_trace.csv
rm -rf $O/trace
grep -A70 "## bf16" $O/train_step_kernels.md | head -75
