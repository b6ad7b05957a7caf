#!/bin/bash
# eval forward at batches around 64: does a launch's time follow (items / wave slots) rounded UP (profiles/r3_notes.md "rounds x wave time")?
O=$PWD/gpurun_out/${1:-r4p}; mkdir -p $O; shift
for b in "$@"; do
  ( timeout 300 python bench.py --batch $b --no-cpu-baseline --csf-batch 0 --no-latency-b1 --train-steps 0 ) > $O/b$b.json 2> $O/b$b.err
  python - $O/b$b.json $b <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pk = j["roofline"]["per_kernel"]
print("B=%s %.0f img/s %.4f ms  %.2f us/img | " % (sys.argv[2], j["value"], j["ms_per_step"], 1e3 * j["ms_per_step"] / int(sys.argv[2])) + " ".join("%s %.3f" % (k.split("_")[0], v["ms"]) for k, v in pk.items()))
PY
done
