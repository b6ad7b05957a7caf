#!/bin/bash
# A/B of the round-4 tile rules: flat pw4 / c3q tiles (CSN_PW4_FLAT) and the depthwise kernels' lanes per row (CSN_DW_LX)
O=$PWD/gpurun_out/${1:-r4j}; mkdir -p $O
B="--no-cpu-baseline --csf-batch 0 --no-latency-b1 --train-net x2"
run() { name=$1; shift; ( env "$@" timeout 400 python bench.py $B ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "eval %.0f img/s %.4f ms | fp32 %.2f ms | bf16 %.2f ms" % (j["value"], j["ms_per_step"], j["train_step"]["ms_per_step"], j["train_step_bf16"]["ms_per_step"]))
PY
}
run new X=1
run noflat CSN_PW4_FLAT=0
run oldlx CSN_DW_LX=0
run new2 X=1
( timeout 900 python -m pytest tests -m gpu -q -x -k "golden or train or paths" 2>&1 | tail -8 ) > $O/pytest.log; tail -4 $O/pytest.log
