#!/bin/bash
mkdir -p gpurun_out/r3i
O=gpurun_out/r3i
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_gpu_golden or test_gpu_x1 or test_gpu_unit_probes or test_gpu_op_goldens or test_gpu_vs_oracle_shapes or test_gpu_full_size" 2>&1 | tail -8 ) > $O/pytest.log
tail -2 $O/pytest.log
CSN_MS_QUAD=0 timeout 200 python tools/unit_table.py --tag msold --json $O/msold.json > $O/msold.txt 2>&1; tail -1 $O/msold.txt; grep "ms.convs" $O/msold.txt
timeout 200 python tools/unit_table.py --tag msq --json $O/t.json > $O/t.txt 2>&1; tail -1 $O/t.txt; grep "ms.convs" $O/t.txt
CSN_SUB_BATCH=32 CSN_SLICE_LANES=1 timeout 200 python tools/unit_table.py --tag "slices_sb32" --quiet 2>&1 | tail -1
