#!/usr/bin/env python3
"""Sums of the SQ counters by kernel family over the LAST bf16 train step of tools/gpu_pmc_sq_train.sh's passes (steps are delimited by
bce_logits_kernel dispatches).  usage: pmc_sq_summary.py gpurun_out/<dir>"""
import collections
import csv
import glob
import os
import re
import sys

src = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
names = []
for d in sorted(glob.glob(os.path.join(src, "sq_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        disp = collections.OrderedDict()
        for r in rows:
            k = int(r["Dispatch_Id"])
            e = disp.setdefault(k, {"name": r["Kernel_Name"], "c": {}})
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ds = [disp[k] for k in sorted(disp)]
        idx = [i for i, e in enumerate(ds) if e["name"].startswith("bce_logits")]
        steps = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if any("csn_bf16" in e["name"] or "wgrad_bf16" in e["name"] or "pwq16" in e["name"] for e in ds[a:b])]
        if not steps:
            continue
        a, b = steps[-1]
        for e in ds[a:b]:
            fam = re.sub(r"^void ", "", e["name"].split("(")[0]).split("<")[0]
            for cn, v in e["c"].items():
                tot[fam][cn] += v
                if cn not in names:
                    names.append(cn)
            tot[fam]["_launches_" + os.path.basename(d)] += 1
fams = sorted(tot, key=lambda f: -tot[f].get("SQ_BUSY_CYCLES", 0.0))
print("family launches | busy Mcyc | waves K | wave-cycles/busy (waves resident per SE-slot) | VALU/wave | MFMA/wave | SALU/wave | VMEM_RD/wave | VMEM_WR/wave | active_any/wave_cyc | wait_any/wave_cyc | active_valu/busy | mfma_busy/busy")
for f in fams[:28]:
    c = tot[f]
    w = max(c.get("SQ_WAVES", 0.0), 1.0)
    busy = max(c.get("SQ_BUSY_CYCLES", 0.0), 1.0)
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    print("%-30s %4d | %8.2f | %8.1f | %6.2f | %7.0f | %6.0f | %6.0f | %6.1f | %6.1f | %.3f | %.3f | %.3f | %.3f" % (
        f[:30], c.get("_launches_sq_a", 0), busy / 1e6, w / 1e3, wc / busy, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_MFMA", 0) / w,
        c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_VMEM_RD", 0) / w, c.get("SQ_INSTS_VMEM_WR", 0) / w,
        c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_VALU", 0) / busy,
        c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy))
