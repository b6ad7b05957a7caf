#!/bin/bash
# wider low-only pw4 forms for CSFHead.fuse's branches 1 / 2 ((0,10) / (0,8): one M group instead of two)
mkdir -p gpurun_out/r3q
O=gpurun_out/r3q
run() { tag=$1; shift; env "$@" timeout 200 python tools/unit_table.py --tag $tag --json $O/$tag.json > $O/$tag.txt 2>&1; echo "$tag $(grep -h 'oct_fuse.fuse ' $O/$tag.txt | awk '{printf "%s ", $3}') $(tail -1 $O/$tag.txt | cut -c1-70)"; }
run new A=1
run nosplit CSN_PW4_NOSPLIT=1
run new2 A=1
python - <<'PY'
import json
for t in ('new','nosplit'):
    d=json.load(open(f'gpurun_out/r3q/{t}.json'))
    print(t, json.dumps(d)[:300])
PY
