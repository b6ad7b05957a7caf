#!/bin/bash
# the whole GPU test suite -> gpurun_out/$1/pytest_gpu.log
O=gpurun_out/${1:-t}; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
