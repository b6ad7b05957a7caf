#!/bin/bash
# A/B of the LDS fill (8 loads in flight vs the plain loop) and of the fuse out2 route, one lease
mkdir -p gpurun_out/r3m
O=gpurun_out/r3m
timeout 200 python tools/unit_table.py --tag new --json $O/new.json > $O/new.txt 2>&1; tail -1 $O/new.txt
SOD100K_HIP_LIB=gpurun_variants/lib_fill1.so timeout 200 python tools/unit_table.py --tag fill1 --json $O/fill1.json > $O/fill1.txt 2>&1; tail -1 $O/fill1.txt
CSN_PW4_NOQ=1 timeout 200 python tools/unit_table.py --tag noq --json $O/noq.json > $O/noq.txt 2>&1; tail -1 $O/noq.txt
timeout 200 python tools/unit_table.py --tag new2 --json $O/new2.json > $O/new2.txt 2>&1; tail -1 $O/new2.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.log
tail -3 $O/pytest.log
