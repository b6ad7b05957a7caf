#!/bin/bash
# Kernel trace of the CSF+Res2Net point (bench.py --csf-batch 32, eval forward only) -> gpurun_out/<out>/csf_trace_<tag>.txt:
# every launch of the LAST head forward in order (kernel, blocks, microseconds).  usage: bash tools/gpu_csf_trace.sh <out> <tag> [ENV=V ...]
O=$PWD/gpurun_out/$1; mkdir -p $O; R=$PWD; TAG=${2:-new}; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $O/csftrace_$TAG
( env "$@" X=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/csftrace_$TAG -o t -- python $R/bench.py --steps 2 --warmup 1 --train-steps 0 --no-cpu-baseline --no-latency-b1 --event-steps 0 --profile-iters 1 --csf-steps 3 ) > $O/csftrace_$TAG.log 2>&1
cd $R
python - $O/csftrace_$TAG $O/csf_trace_$TAG.txt <<'PY'
import csv, glob, os, re, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]) for r in rows]
# the head of the last forward: from the first csf_resize after the last backbone kernel to the end
idx = [i for i, n in enumerate(names) if n.startswith("csf_cls")]
end = idx[-1] + 2 if idx else len(rows)
start = end
while start > 0 and names[start - 1].startswith("csf_") and not names[start - 1].startswith("csf_bn_act"):
    start -= 1
tot = 0.0
fam = {}
with open(sys.argv[2], "w") as fo:
    for i in range(start, min(end, len(rows))):
        r = rows[i]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += us
        k = names[i].split("<")[0]
        fam[k] = fam.get(k, 0.0) + us
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        fo.write(f"{i - start:3d} {names[i][:40]:40s} {g // 256:7d} blocks {us:9.1f} us\n")
    fo.write(f"head: {tot / 1e3:.3f} ms; " + ", ".join(f"{k} {v / 1e3:.3f}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])) + "\n")
print(open(sys.argv[2]).read().splitlines()[-1])
PY
rm -rf $O/csftrace_$TAG
