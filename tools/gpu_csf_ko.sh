#!/bin/bash
# Knock-out builds of csf_gemm_kernel (gpurun_variants/ko_*.so, built locally with -DCSF_KO_*): head time at batch 32.
mkdir -p gpurun_out/csf
for v in ${KO_LIST:-"" gpurun_variants/ko_ALL3.so gpurun_variants/ko_ALL4.so gpurun_variants/ko_DSREAD.so}; do
  if [ -n "$v" ]; then export SOD100K_HIP_LIB=$PWD/$v; else unset SOD100K_HIP_LIB; fi
  echo "== ${v:-baseline}"
  timeout 200 python bench.py --steps 2 --warmup 1 --train-steps 0 --no-cpu-baseline --profile-iters 1 --csf-steps 5 2>&1 | grep -o '"ms_head_hip": [0-9.]*\|"error".*' | head -2
done 2>&1 | tee gpurun_out/csf/ko.log
