#!/bin/bash
# round 4, lease B: probe of the bf16 weight-gradient kernels (1x1 + 3x3), the train-step tests, a train-only bench and its kernel trace
O=$PWD/gpurun_out/${1:-r4b}; mkdir -p $O; R=$PWD
( timeout 300 tools/probes/wgbf_probe ) > $O/wgbf_probe.log 2>&1; grep -v "^own\|^pool" $O/wgbf_probe.log | tail -14
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -q -x -k "units_local or train_step_bf16 or train_step_gradients or train_golden or unpruned or loss_goes_down or overlap2" --durations=8 2>&1 | tail -25 ) > $O/pytest_sel.log; tail -14 $O/pytest_sel.log
( timeout 600 python bench.py --steps 5 --warmup 2 --train-steps 5 --csf-batch 0 --no-cpu-baseline --no-latency-b1 --event-steps 0 --profile-iters 1 ) > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("eval", d["value"], d["ms_per_step"])
print("train fp32", d["train_step"]["ms_per_step"], "bf16", d["train_step_bf16"]["ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 3 ) > $O/trace.log 2>&1
cd $R
python tools/train_step_breakdown.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 0 > $O/train_step_kernels.md 2>&1
cp $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv 2>/dev/null; gzip -f $O/kernel_trace.csv
rm -rf $O/trace
grep -A60 "## bf16" $O/train_step_kernels.md | head -75
