#!/bin/bash
mkdir -p gpurun_out/r3k
O=gpurun_out/r3k
CSN_C3Q_KSPLIT=0 timeout 200 python tools/unit_table.py --tag noksplit --json $O/noks.json > $O/noks.txt 2>&1; tail -1 $O/noks.txt
timeout 200 python tools/unit_table.py --tag ksplit --json $O/ks.json > $O/ks.txt 2>&1; tail -1 $O/ks.txt
grep -E "stage[0234].0.conv1x1" $O/noks.txt $O/ks.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replay.py tests/test_gpu_rccl.py -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.log
tail -3 $O/pytest.log
