#!/bin/bash
mkdir -p gpurun_out/r3ab
O=gpurun_out/r3ab
timeout 200 python tools/unit_table.py --tag b64 --json $O/b64.json > $O/b64.txt 2>&1; tail -1 $O/b64.txt | cut -c1-80
timeout 200 python tools/unit_table.py --batch 32 --tag b32 --json $O/b32.json > $O/b32.txt 2>&1; tail -1 $O/b32.txt | cut -c1-80
