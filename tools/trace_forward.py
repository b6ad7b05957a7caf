#!/usr/bin/env python3
"""Per-dispatch list of ONE eval forward from a rocprofv3 kernel trace (csv).

usage: tools/trace_forward.py <dir-or-csv> [index]
Takes the dispatches between two consecutive bilinear_up2 launches (= one whole forward; index -1 = the last one, which
under tools/unit_table.py is the serialised profiling pass, -3 = a hipGraph replay), prints start offset, gap to the end
of everything before it, duration (us), and the wall time from the first start to the last end (overlapping stream lanes
included)."""
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0]


def main():
    p = sys.argv[1]
    if os.path.isdir(p):
        c = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))
        if not c:
            raise SystemExit("no *kernel_trace.csv under " + p)
        p = c[-1]
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "bilinear_up2" in r["Kernel_Name"]]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    a, b = ends[which - 1] + 1, ends[which] + 1
    sel = rows[a:b]
    t0 = int(sel[0]["Start_Timestamp"])
    tot = 0
    agg = {}
    prev_end = t0
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d = (e - s) / 1e3
        gap = (s - prev_end) / 1e3
        prev_end = max(prev_end, e)
        tot += d
        k = short(r["Kernel_Name"])
        agg[k.split("<")[0]] = agg.get(k.split("<")[0], 0.0) + d
        grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
        wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
        print(f"{(s - t0) / 1e3:9.1f} gap {gap:7.1f} +{d:8.1f} us  {k[:60]:60s} grid {grid} wg {wg} vgpr {r.get('VGPR_Count', r.get('Arch_VGPR_Count', '?'))} lds {r.get('LDS_Block_Size', '?')}")
    wall = (max(int(r["End_Timestamp"]) for r in sel) - t0) / 1e3
    print(f"sum {tot:.1f} us, wall {wall:.1f} us, " + "  ".join(f"{k}={v:.1f}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
