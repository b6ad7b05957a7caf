#!/bin/bash
O=$PWD/gpurun_out/${1:-r4i}; mkdir -p $O; R=$PWD
( timeout 300 tools/probes/wgbf_probe ) > $O/wgbf_probe.log 2>&1; grep "^ms\|probe:" $O/wgbf_probe.log
bash tools/experiments/gpu_r4f.sh $1
