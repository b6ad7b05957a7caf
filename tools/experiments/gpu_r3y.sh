#!/bin/bash
# after the bf16-sized workspace layout: train tests (incl. batch 256 in both storage modes), bf16 / fp32 step timing
O=$PWD/gpurun_out/${1:-r3y}; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replay.py -m gpu -x -q -k "train or bf16" 2>&1 | tail -6 ) > $O/pytest_train.log
tail -3 $O/pytest_train.log
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --train-steps 10 ) > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(d['value'])
for k in ('train_step','train_step_bf16'):
    print(k, d[k]['ms_per_step'], 'ms', d[k]['value'], 'img/s', 'loss', d[k]['loss'])
PY
