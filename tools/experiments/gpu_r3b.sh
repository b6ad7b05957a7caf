#!/bin/bash
# round-3 call B: c3q parity + A/B
mkdir -p gpurun_out/r3b
O=gpurun_out/r3b
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_gpu_golden or test_gpu_x1 or test_gpu_unit_probes or test_gpu_op_goldens or test_gpu_vs_oracle_shapes or test_gpu_full_size or test_gpu_train_forward_vs_oracle or test_gpu_train_units_local" 2>&1 | tail -8 ) > $O/pytest.log
tail -3 $O/pytest.log
CSN_C3Q=0 timeout 200 python tools/unit_table.py --tag noc3q --quiet --json $O/noc3q.json 2>&1 | tail -1
timeout 200 python tools/unit_table.py --tag c3q --json $O/c3q.json > $O/c3q.txt 2>&1; tail -1 $O/c3q.txt
for n in 2 3 5 7; do
  CSN_C3Q_NT=$n timeout 200 python tools/unit_table.py --tag c3q_nt$n --json $O/nt$n.json > $O/nt$n.txt 2>&1; tail -1 $O/nt$n.txt
done
CSN_PW4_TWL=5 timeout 200 python tools/unit_table.py --tag twl5 --quiet --json $O/twl5.json 2>&1 | tail -1
CSN_PW4_TWL=3 timeout 200 python tools/unit_table.py --tag twl3 --quiet --json $O/twl3.json 2>&1 | tail -1
grep -E "c3q|pool2|goct_c3" $O/c3q.txt
