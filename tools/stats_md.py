#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats csv -> a markdown table for profiles/.  usage: tools/stats_md.py <dir> <out.md> "<title>" [bench.json]"""
import csv
import glob
import json
import os
import sys


def main():
    src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
    f = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    with open(dst, "w") as o:
        o.write(f"# {title}\n\n")
        if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
            try:
                b = json.loads([l for l in open(sys.argv[4]) if l.startswith("{")][-1])
                r = b.get("roofline") or {}
                o.write(f"bench line of the same run: {b['value']} img/s, {b['ms_per_step']} ms/step; dominant kernel {r.get('kernel')} at "
                        f"{r.get('us_per_launch')} us/launch (HIP events, serialised), {r.get('launches_per_step')} launches/step\n\n")
            except Exception as e:
                o.write(f"(bench line not readable: {e})\n\n")
        fam = {}
        for r in rows:
            k = r["Name"].replace("void ", "").split("<")[0].split("(")[0]
            f_ = fam.setdefault(k, [0, 0.0])
            f_[0] += int(r["Calls"]); f_[1] += float(r["TotalDurationNs"])
        o.write("By kernel family (template instantiations merged; avg us = what `roofline.us_per_launch` averages over):\n\n")
        o.write("| kernel family | calls | total ms | avg us |\n|---|---|---|---|\n")
        for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
            o.write(f"| {k} | {c} | {t / 1e6:.2f} | {t / c / 1e3:.2f} |\n")
        o.write("\nBy instantiation:\n\n")
        o.write("| kernel | calls | total ms | avg us | % | min us | max us |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            name = r["Name"].replace("|", "/")[:70]
            o.write(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{float(r['Percentage']):.2f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} |\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
