#!/bin/bash
# train step A/B: BatchNorm backward apply fused into the depthwise backward (default) vs two passes (CSN_BN_BWD_FUSE=0)
O=$PWD/gpurun_out/${1:-r3v}; mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --csf-batch 0 --no-latency-b1 --event-steps 0 --train-steps 10"
show() { python - $1 <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
for k in ('train_step','train_step_bf16'):
    print(sys.argv[1].split('/')[-1], k, d[k]['ms_per_step'], 'ms', d[k]['value'], 'img/s')
PY
}
( timeout 600 $B ) > $O/fused.json 2> $O/fused.err; show $O/fused.json
( CSN_BN_BWD_FUSE=0 timeout 600 $B ) > $O/split.json 2> $O/split.err; show $O/split.json
( timeout 600 $B ) > $O/fused2.json 2> $O/fused2.err; show $O/fused2.json
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replay.py -m gpu -x -q -k "train" 2>&1 | tail -6 ) > $O/pytest_train.log
tail -3 $O/pytest_train.log
