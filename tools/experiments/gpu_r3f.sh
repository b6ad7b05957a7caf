#!/bin/bash
mkdir -p gpurun_out/r3f
O=gpurun_out/r3f
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_gpu_golden or test_gpu_x1 or test_gpu_unit_probes or test_gpu_op_goldens or test_gpu_vs_oracle_shapes or test_gpu_full_size" 2>&1 | tail -8 ) > $O/pytest.log
tail -2 $O/pytest.log
timeout 200 python tools/unit_table.py --tag r3f --json $O/t.json > $O/t.txt 2>&1; tail -1 $O/t.txt
CSN_PW4_NOSPLIT=1 timeout 200 python tools/unit_table.py --tag nosplit --json $O/nosplit.json > $O/nosplit.txt 2>&1; tail -1 $O/nosplit.txt
CSN_OVERLAP=0 timeout 200 python tools/unit_table.py --tag nolanes --quiet 2>&1 | tail -1
