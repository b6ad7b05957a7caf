#!/bin/bash
# train step: GPU tests of the train path, timing of both storage modes, kernel breakdown of one step -> gpurun_out/$1/
O=$PWD/gpurun_out/${1:-r3t}; mkdir -p $O
R=$PWD
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replay.py -m gpu -x -q -k "train" 2>&1 | tail -6 ) > $O/pytest_train.log
tail -2 $O/pytest_train.log
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --train-steps 10 ) > $O/bench_t.json 2> $O/bench_t.err
python - $O/bench_t.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
for k in ('train_step','train_step_bf16'):
    print(k, d[k]['ms_per_step'], 'ms', d[k]['value'], 'img/s', d[k]['roofline']['frac'])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --profile-iters 1 --train-steps 3 ) > $O/trace.log 2>&1
cd $R
python tools/train_step_breakdown.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/train_step_kernels.md 2>&1
cat $O/train_step_kernels.md | head -80
