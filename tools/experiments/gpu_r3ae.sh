#!/bin/bash
# row-wide tiles by default: eval GPU tests, forward time, train-step time
O=$PWD/gpurun_out/r3ae; mkdir -p $O
( timeout 120 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replay.py -m gpu -x -q -k "not train and not bf16 and not batch256" 2>&1 | tail -2 )
timeout 100 python tools/unit_table.py --tag w64 --quiet --json $O/w64.json > $O/w64.txt 2>&1; tail -1 $O/w64.txt | cut -c1-100
( timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --train-steps 10 --profile-iters 1 ) > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
for k in ('train_step','train_step_bf16'):
    print(k, d[k]['ms_per_step'], 'ms', d[k]['value'], 'img/s')
PY
