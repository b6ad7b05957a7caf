#!/usr/bin/env python3
"""profiles/r4_unit_table.md from the end-of-round collection (tools/gpu_final_r4.sh -> gpurun_out/r4/):

* eval forward, per launch group of every unit: kernel, microseconds (HIP events after every launch, serialised, tools/unit_table.py),
  algorithmic MB (unit inputs read once + outputs written once), algorithmic GB/s and the fraction of 8 TB/s;
* one bf16 (and one fp32) train step, per kernel family: launches, ms (rocprofv3 --kernel-trace), HBM bytes by the counters
  (profiles/r4_pmc_train.json: FETCH_SIZE / WRITE_SIZE passes of the same tree) and the rate they imply.

usage: python tools/r4_tables.py [gpurun_out/r4] [profiles/r4_pmc_train.json]"""
import collections
import csv
import gzip
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 8000.0


def train_tables(trace_gz):
    rows = list(csv.DictReader(gzip.open(trace_gz, "rt")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("bce_logits")]
    out = {}
    for a, b in zip(idx[:-1], idx[1:]):
        fam = collections.OrderedDict()
        inst = set()
        for r in rows[a:b]:
            n = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0])
            k = n.split("<")[0]
            f = fam.setdefault(k, [0, 0.0])
            f[0] += 1
            f[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            inst.add(n)
        mode = "bf16" if any("csn_bf16" in n or "wgrad_bf16" in n for n in inst) else "fp32"
        out[mode] = fam          # the last step of each mode wins
    return out


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r4")
    pmc = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r4_pmc_train.json")
    lines = ["# Round-4 tables: eval forward per unit, train step per kernel family (MI355X)", ""]
    ut = os.path.join(src, "unit_table.json")
    if os.path.exists(ut):
        u = json.load(open(ut))
        lines += [f"## Eval forward, batch 64 x 3x224x224 fp32 (hipGraph replay median {u['median_ms']:.3f} ms = "
                  f"{64 / u['median_ms'] * 1e3:.0f} img/s; per-launch times serialised)", "",
                  "| unit | kernel | us | algorithmic MB | GB/s | of 8 TB/s |", "|---|---|---|---|---|---|"]
        tot_us = tot_mb = 0.0
        for r in u["units"]:
            gb = r.get("GBps") or 0.0
            lines.append(f"| {r['unit']} | {r['kernel']} | {r['ms'] * 1e3:.1f} | {r['alg_MB']:.1f} | {gb:.0f} | {gb / PEAK:.3f} |")
            tot_us += r["ms"] * 1e3
            tot_mb += r["alg_MB"]
        lines += [f"| **sum** | | {tot_us:.0f} | {tot_mb:.0f} | {tot_mb / tot_us * 1e3:.0f} | {tot_mb / tot_us * 1e3 / PEAK:.3f} |", ""]
        lines += ["By kernel (same run):", "", "| kernel | launches | ms |", "|---|---|---|"]
        for k, v in sorted(u["kernels"].items(), key=lambda kv: -kv[1]["ms"]):
            lines.append(f"| {k} | {v['launches']} | {v['ms']:.3f} |")
        lines.append("")
    tg = os.path.join(src, "train_kernel_trace.csv.gz")
    if os.path.exists(tg):
        tabs = train_tables(tg)
        pj = json.load(open(pmc)) if os.path.exists(pmc) else {}
        for mode in ("bf16", "fp32"):
            if mode not in tabs:
                continue
            fam = tabs[mode]
            tot = sum(v[1] for v in fam.values())
            byk = pj.get(mode, {}).get("by_kernel", {})
            hb = pj.get(mode, {}).get("hbm_bytes_per_step")
            lines += [f"## Train step, batch 256, {mode} storage: {tot:.2f} ms of kernel time, {sum(v[0] for v in fam.values())} launches"
                      + (f", {hb / 1e9:.1f} GB of HBM traffic by the counters = {hb / tot / 1e6:.0f} GB/s" if hb else ""), "",
                      "| kernel family | launches | ms | counter GB (read + written) | GB/s | of 8 TB/s |", "|---|---|---|---|---|---|"]
            for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
                b = byk.get(k)
                if b:
                    gbs = (b["read_bytes"] + b["write_bytes"]) / 1e9
                    rate = gbs / v[1] * 1e3 if v[1] > 0 else 0.0
                    lines.append(f"| {k} | {v[0]} | {v[1]:.3f} | {b['read_bytes'] / 1e9:.2f} + {b['write_bytes'] / 1e9:.2f} | {rate:.0f} | {rate / PEAK:.3f} |")
                else:
                    lines.append(f"| {k} | {v[0]} | {v[1]:.3f} | | | |")
            lines.append("")
    dst = os.path.join(ROOT, "profiles", "r4_unit_table.md")
    open(dst, "w").write("\n".join(lines) + "\n")
    print("wrote", dst, len(lines), "lines")


if __name__ == "__main__":
    main()
