#!/bin/bash
# A/B sweep of the eval forward on the GPU box: one bench.py line per variant into gpurun_out/$1/<tag>.json
# usage: tools/gpu_ab.sh <outdir-tag> "<TAG ENV=VAL ...>" ...
out=gpurun_out/$1; shift
mkdir -p $out
for spec in "$@"; do
  tag=${spec%% *}; envs=${spec#* }; [ "$envs" == "$spec" ] && envs=""
  env $envs python bench.py --steps 30 --warmup 5 --train-steps 0 --csf-batch 0 --no-cpu-baseline --event-steps 30 \
      > $out/$tag.json 2> $out/$tag.err || echo "FAILED $tag" >> $out/failed.txt
  python - "$out/$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    pk=d['roofline']['per_kernel']
    print(f"{sys.argv[2]:28s} {d['value']:9.1f} img/s  {d['ms_per_step']:.3f} ms  " + "  ".join(f"{k.split('_kernel')[0]}={v['ms']:.3f}" for k,v in pk.items()), flush=True)
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
done
