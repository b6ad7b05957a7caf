#!/bin/bash
# round-3 call A: parity of the pw4 path + per-unit A/B tables (base vs pw4 vs build / launch variants)
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_gpu_golden or test_gpu_x1 or test_gpu_unit_probes or test_gpu_op_goldens or test_gpu_vs_oracle_shapes or test_gpu_full_size" 2>&1 | tail -8 ) > $O/pytest.log
tail -3 $O/pytest.log
CSN_PW4=0 timeout 200 python tools/unit_table.py --tag base --json $O/base.json > $O/base.txt 2>&1; tail -1 $O/base.txt
timeout 200 python tools/unit_table.py --tag pw4 --json $O/pw4.json > $O/pw4.txt 2>&1; tail -1 $O/pw4.txt
for v in hb8 hb2 lb1 occ3 hb8lb3; do
  SOD100K_HIP_LIB=$PWD/gpurun_variants/lib_$v.so timeout 200 python tools/unit_table.py --tag $v --json $O/$v.json > $O/$v.txt 2>&1; tail -1 $O/$v.txt
done
for g in 768 1024 4096; do
  CSN_PW4_GRID=$g timeout 200 python tools/unit_table.py --tag grid$g --quiet --json $O/grid$g.json 2>&1 | tail -1
done
for t in 3 5; do
  CSN_PW4_TWL=$t timeout 200 python tools/unit_table.py --tag twl$t --quiet --json $O/twl$t.json 2>&1 | tail -1
done
CSN_OVERLAP=0 timeout 200 python tools/unit_table.py --tag nolanes --quiet 2>&1 | tail -1
grep -E "stage1|stage2" $O/base.txt | head -12
grep -E "stage1|stage2" $O/pw4.txt | head -12
