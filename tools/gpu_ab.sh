#!/bin/bash
# A/B bench runs on one lease: tools/gpu_ab.sh <outdir> name[:variant-lib][:ENV=V,...][:bench,args] ...   (variant libs: tools/build_variant.sh)
O=$PWD/gpurun_out/$1; mkdir -p $O; shift
B="--no-cpu-baseline --csf-batch 0 --no-latency-b1 --train-net x2"
for spec in "$@"; do
  IFS=: read -r name lib envs extra <<< "$spec"
  E=(X=1)
  [ -n "$lib" ] && E+=(SOD100K_HIP_LIB=$PWD/gpurun_variants/lib_$lib.so)
  [ -n "$envs" ] && E+=(${envs//,/ })
  ( env "${E[@]}" timeout 400 python bench.py $B ${extra//,/ } ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "eval %.0f img/s %.4f ms | fp32 %.2f ms | bf16 %.2f ms" % (j["value"], j["ms_per_step"], j["train_step"]["ms_per_step"], j["train_step_bf16"]["ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
