#!/bin/bash
# Round profile on the MI355X box: kernel trace + stats of the default bench command, and the two HBM
# PMC passes (separate runs, as gpurun requires).  Outputs land in gpurun_out/round/ (scratch); run
# tools/summarize_profile.py afterwards to write the tracked summaries under profiles/.
mkdir -p gpurun_out/round
R=$PWD
( timeout 900 python bench.py ) > gpurun_out/round/bench.json 2> gpurun_out/round/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/round/trace $R/gpurun_out/round/fetch $R/gpurun_out/round/write $R/gpurun_out/round/fetchcal
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/round/trace -o trace -- python $R/bench.py --no-cpu-baseline ) > $R/gpurun_out/round/trace.log 2>&1
( timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/round/fetch -o fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-iters 1 --train-steps 0 --csf-batch 0 ) > $R/gpurun_out/round/fetch.log 2>&1
( timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/round/write -o write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-iters 1 --train-steps 0 --csf-batch 0 ) > $R/gpurun_out/round/write.log 2>&1
( timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/round/fetchcal -o fetchcal -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-iters 1 --train-steps 0 --csf-batch 0 --no-fuse-cls ) > $R/gpurun_out/round/fetchcal.log 2>&1
# CSF+Res2Net head (BASELINE config 5): its own kernel trace and one SQ counter pass (matrix-pipe busy, wave-cycle split)
rm -rf $R/gpurun_out/round/csf_trace $R/gpurun_out/round/csf_pmc
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/round/csf_trace -o trace -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-iters 1 --train-steps 0 ) > $R/gpurun_out/round/csf_trace.log 2>&1
( timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/round/csf_pmc -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-iters 1 --train-steps 0 --csf-steps 2 ) > $R/gpurun_out/round/csf_pmc.log 2>&1
cd $R
ls gpurun_out/round gpurun_out/round/trace | head -20
tail -c 400 gpurun_out/round/bench.json
