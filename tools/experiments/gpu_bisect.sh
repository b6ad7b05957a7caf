#!/bin/bash
for v in "X=1" "CSN_C3Q=0" "CSN_MS_QUAD=0" "CSN_C3Q=0 CSN_MS_QUAD=0"; do
  echo "== $v"
  ( env $v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "well_conditioned" 2>&1 | grep -E "passed|failed|gradients further" | cut -c1-400 )
done
