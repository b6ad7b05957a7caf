#!/usr/bin/env python3
"""Per-kernel time of ONE train step from a rocprofv3 --kernel-trace csv: the last fp32 step and the last bf16 step
(steps are delimited by bce_logits_kernel launches).  usage: train_step_breakdown.py <trace_kernel_trace.csv> [rows [launch list]]
Third argument: a file that receives every launch of the last bf16 step in order (kernel, blocks x / y / z, microseconds)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("bce_logits")]


def step_table(a, b):
    tot = collections.OrderedDict()
    for r in rows[a:b]:
        n = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        c = tot.setdefault(n, [0, 0.0])
        c[0] += 1
        c[1] += d
    return tot


def is16(tab):
    return any("csn_bf16" in k for k in tab)


steps = [step_table(idx[i], idx[i + 1]) for i in range(len(idx) - 1)]
f32 = [s for s in steps if not is16(s)]
b16 = [s for s in steps if is16(s)]
for name, tab in (("fp32", f32[-1] if f32 else None), ("bf16", b16[-1] if b16 else None)):
    if tab is None:
        continue
    print(f"## {name}: {sum(v[1] for v in tab.values()) / 1e3:.2f} ms of kernel time in one step")
    top = 32 if len(sys.argv) < 3 else int(sys.argv[2])      # second argument: rows per table (0 = every kernel)
    for k, v in sorted(tab.items(), key=lambda kv: -kv[1][1])[:top or None]:
        print(f"{v[1] / 1e3:8.2f} ms {v[0]:4d}  {k[:90]}")

if len(sys.argv) > 3 and b16:
    last = max(i for i in range(len(idx) - 1) if is16(steps[i]))
    with open(sys.argv[3], "w") as fo:
        for r in rows[idx[last]:idx[last + 1]]:
            n = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0])
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            g = [int(r.get("Grid_Size_" + a, 0) or 0) // max(1, int(r.get("Workgroup_Size_" + a, 1) or 1)) for a in "XYZ"]
            fo.write(f"{n[:56]:56s} {g[0]:7d} {g[1]:5d} {g[2]:4d} {d:9.1f}\n")
