#!/bin/bash
# round 4, lease A: the bf16 weight-gradient kernel on its own (lane maps, timing), the whole GPU suite (new path tests included),
# a short bench line and a kernel trace of the train step
O=$PWD/gpurun_out/r4a; mkdir -p $O; R=$PWD
( timeout 300 tools/probes/wgbf_probe ) > $O/wgbf_probe.log 2>&1; tail -20 $O/wgbf_probe.log
( timeout 1500 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -60 ) > $O/pytest_gpu.log; tail -30 $O/pytest_gpu.log
( timeout 600 python bench.py --steps 20 --warmup 5 --train-steps 5 --csf-batch 0 --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("eval", d["value"], d["ms_per_step"], "pw4", d["roofline"]["us_per_launch"], d["roofline"]["frac"])
print("train fp32", d["train_step"]["ms_per_step"], "bf16", d["train_step_bf16"]["ms_per_step"])
print("b1", d["latency_b1"]["median_ms"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 3 ) > $O/trace.log 2>&1
cd $R
python tools/train_step_breakdown.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/train_step_kernels.md 2>&1
rm -rf $O/trace
head -70 $O/train_step_kernels.md
