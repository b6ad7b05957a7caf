#!/usr/bin/env python3
"""profiles/rN_unit_table.md from an end-of-round collection (tools/gpu_final.sh -> gpurun_out/rN/):

* eval forward, per launch group of every unit: kernel, microseconds (HIP events after every launch, serialised, tools/unit_table.py),
  algorithmic MB (unit inputs read once + outputs written once), algorithmic GB/s and the fraction of 8 TB/s;
* one bf16 (and one fp32) train step, per kernel family: launches, ms (rocprofv3 --kernel-trace), HBM bytes by the counters
  (profiles/rN_pmc_train.json: FETCH_SIZE / WRITE_SIZE passes of the same tree) and the rate they imply;
* round 5: per unit also the convolution FLOPs (2 x MAC, from the channel plan) and, where the arithmetic intensity is above
  15 FLOP/B (CSFHead.fuse, fuse1x1, the 3x3 units, MSBlocks), the fraction of the fp32 matrix peak (157.3 TFLOP/s) -- for those
  units the matrix pipe, not HBM, is the honest yardstick.

usage: python tools/round_tables.py <round, e.g. r5> [gpurun_out/<round>] [profiles/<round>_pmc_train.json]"""
import collections
import csv
import gzip
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 8000.0
MFMA_PEAK = 157.3e12


def unit_flops(batch=64, size=224, executed=False):
    """unit name -> convolution FLOPs (2 x MAC) of one batch, from the x2 channel plan (host side only: no device)."""
    sys.path.insert(0, ROOT)
    from sod100k_amd.model import csnet as M
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.build_model(predefine=man)
    arena = m._ensure_arena()
    units, acts, names = m.describe(arena.offsets)
    out = {}
    for u, name in zip(units, names):
        px = lambda a: (size >> acts[a][1]) ** 2
        mac = 0
        if u.kind == 1:      # gOctConv: every (input branch i, output branch j) block at the resolution the conv runs at
            for j in range(int(u.n_out)):
                if u.cout[j] == 0:
                    continue
                for i in range(int(u.n_in)):
                    if u.cin[i] == 0:
                        continue
                    pj = px(u.out_act[j])
                    pi = px(u.in_act[i]) // (4 if u.stride == 2 else 1)
                    # the reference convolves at the LOWER of the two resolutions; pw4_kernel (executed = True, 1x1 units) contracts
                    # the interpolated inputs at the OUTPUT resolution (one contraction per output pixel, k_pw4.hip)
                    res = pj if (executed and u.ksize == 1) else min(pi, pj)
                    mac += res * u.cin[i] * u.cout[j] * u.ksize * u.ksize
        elif u.kind == 2:    # depthwise 3x3
            for j in range(int(u.n_out)):
                if u.cout[j]:
                    mac += px(u.out_act[j]) * u.cout[j] * 9
        elif u.kind == 3:    # MSBlock: every output channel sees all input channels through one dilated 3x3
            mac += px(u.out_act[0]) * u.cin[0] * u.cout[0] * 9
        elif u.kind == 4:    # cls_layer
            mac += px(u.in_act[0]) * u.cin[0]
        out[name] = 2 * mac * batch
    return out


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
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r5"
    src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", rnd)
    pmc = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", f"{rnd}_pmc_train.json")
    lines = [f"# Round-{rnd[1:]} tables: eval forward per unit, train step per kernel family (MI355X)", ""]
    ut = os.path.join(src, "unit_table.json")
    if os.path.exists(ut):
        u = json.load(open(ut))
        lines += [f"## Eval forward, batch 64 x 3x224x224 fp32 (hipGraph replay median {u['median_ms']:.3f} ms = "
                  f"{64 / u['median_ms'] * 1e3:.0f} img/s; per-launch times serialised)", "",
                  "A launch group that covers several units (the depthwise pair, a whole ILBlock on ilb_kernel) is one row: the bytes / FLOPs",
                  "of ALL its units, under the name of its first unit.", "",
                  "GFLOP = 2 x MAC the launch group executes (pw4_kernel contracts interpolated inputs at the output resolution: more than the",
                  "reference's count for CSFHead.fuse / fuse1x1; every other kernel executes the reference's count, 63.5 GFLOP per batch in all).", "",
                  "| unit | kernel | us | algorithmic MB | GB/s | of 8 TB/s | GFLOP executed | FLOP/B | of 157.3 TF (AI > 15) |", "|---|---|---|---|---|---|---|---|---|"]
        tot_us = tot_mb = 0.0
        try:
            fl = unit_flops()
            flx = unit_flops(executed=True)
        except Exception as e:     # (the table must not depend on the model import)
            print("no FLOP columns:", e)
            fl, flx = {}, {}
        names = list(fl)
        for r in u["units"]:
            gb = r.get("GBps") or 0.0
            ex = r["kernel"] == "pw4_kernel"     # what the kernel EXECUTES counts for its matrix-pipe fraction
            f = (flx if ex else fl).get(r["unit"], 0)
            # units folded into this launch group: the ones that follow it in the plan and have no row of their own
            if names and r["unit"] in names:
                have = {q["unit"] for q in u["units"]}
                k = names.index(r["unit"]) + 1
                while k < len(names) and names[k] not in have:
                    f += fl[names[k]]     # (depthwise units: the same in both counts)
                    k += 1
            ai = f / (r["alg_MB"] * 1e6) if r["alg_MB"] else 0.0
            mf = f / (r["ms"] * 1e-3) / MFMA_PEAK if r["ms"] > 0 else 0.0
            lines.append(f"| {r['unit']} | {r['kernel']} | {r['ms'] * 1e3:.1f} | {r['alg_MB']:.1f} | {gb:.0f} | {gb / PEAK:.3f} | "
                         f"{f / 1e9:.2f} | {ai:.1f} | {('%.3f' % mf) if ai > 15 else ''} |")
            tot_us += r["ms"] * 1e3
            tot_mb += r["alg_MB"]
        lines += [f"| **sum** | | {tot_us:.0f} | {tot_mb:.0f} | {tot_mb / tot_us * 1e3:.0f} | {tot_mb / tot_us * 1e3 / PEAK:.3f} | "
                  f"{sum(fl.values()) / 1e9:.1f} | | |", ""]
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
    dst = os.path.join(ROOT, "profiles", f"{rnd}_unit_table.md")
    open(dst, "w").write("\n".join(lines) + "\n")
    print("wrote", dst, len(lines), "lines")


if __name__ == "__main__":
    main()
