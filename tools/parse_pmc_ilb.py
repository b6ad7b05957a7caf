#!/usr/bin/env python3
"""Per-kernel sums of the counter passes written by tools/gpu_pmc_ilb.sh (last whole forward of each pass)."""
import collections, csv, glob, os, sys
out = sys.argv[1]
tab = collections.OrderedDict()
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        disp = collections.OrderedDict()
        for r in rows:
            disp.setdefault(r["Dispatch_Id"], [r["Kernel_Name"], {}])[1][r["Counter_Name"]] = float(r["Counter_Value"])
        seq = list(disp.values())
        ends = [i for i, (n, _) in enumerate(seq) if "bilinear_up2" in n]
        if len(ends) >= 2:
            seq = seq[ends[-2] + 1: ends[-1] + 1]
        for n, cs in seq:
            key = n.replace("void ", "").split("(")[0]
            t = tab.setdefault(key, collections.OrderedDict(launches=0))
            for c, v in cs.items():
                t[c] = t.get(c, 0.0) + v
        for key in {n.replace("void ", "").split("(")[0] for n, _ in seq}:
            pass
for k, t in tab.items():
    print(k)
    print("   " + "  ".join(f"{c}={v:.4g}" for c, v in t.items() if c != "launches"))
