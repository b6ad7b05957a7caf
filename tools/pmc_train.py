#!/usr/bin/env python3
"""HBM traffic of ONE train step (fp32 and bf16 storage) from rocprofv3 FETCH_SIZE / WRITE_SIZE passes over
`bench.py --train-steps N` (steps are delimited by bce_logits_kernel launches, as in tools/train_step_breakdown.py),
calibrated on tools/probes/fetch_cal like tools/pmc_hbm.py.

usage: tools/pmc_train.py <dir> <tag> [x2 | unpruned]
<dir> holds cal_fetch/ cal_write/ train_fetch/ train_write/ (tools/gpu_pmc_train.sh).  Writes profiles/<tag>_pmc_train.json."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dispatches(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    disp = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            disp[int(r["Dispatch_Id"])] = (re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]), float(r["Counter_Value"]))
    return [disp[k] for k in sorted(disp)]


def steps(seq):
    idx = [i for i, (n, _) in enumerate(seq) if n.startswith("bce_logits")]
    out = []
    for a, b in zip(idx[:-1], idx[1:]):
        out.append(seq[a:b])
    return out


def main():
    src, tag = sys.argv[1], sys.argv[2]
    net = sys.argv[3] if len(sys.argv) > 3 else "x2"
    sfx = "" if net == "x2" else "_" + net
    GiB = float(1 << 30)
    cf = [v for n, v in dispatches(os.path.join(src, "cal_fetch"), "FETCH_SIZE") if "stream_kernel" in n or "stride2" in n]
    cw = [v for n, v in dispatches(os.path.join(src, "cal_write"), "WRITE_SIZE") if "stream_kernel" in n or "stride2" in n]
    # fetch_cal streams 1 GiB per launch: read 4 / 8 / 16 B per lane, stride-2 pairs, write 4 / 8 / 16 (counters in KiB)
    f_rd = [GiB / (v * 1024) for v in cf[:4]]
    f_wr = [GiB / (v * 1024) for v in cw[4:7]]
    k_rd, k_wr = sum(f_rd) / len(f_rd), sum(f_wr) / len(f_wr)
    print("calibration: bytes per counted byte, reads", [round(v, 3) for v in f_rd], "writes", [round(v, 3) for v in f_wr])
    fe = steps(dispatches(os.path.join(src, "train_fetch"), "FETCH_SIZE"))
    wr = steps(dispatches(os.path.join(src, "train_write"), "WRITE_SIZE"))
    res = {}
    for name, want16 in (("fp32", False), ("bf16", True)):
        sel_f = [s for s in fe if any("csn_bf16" in n for n, _ in s) == want16]
        sel_w = [s for s in wr if any("csn_bf16" in n for n, _ in s) == want16]
        if not sel_f or not sel_w:
            continue
        sf, sw = sel_f[-1], sel_w[-1]
        assert [n for n, _ in sf] == [n for n, _ in sw], "dispatch order differs between the passes"
        fam = collections.OrderedDict()
        for (n, f), (_, w) in zip(sf, sw):
            k = n.split("<")[0]
            a = fam.setdefault(k, dict(launches=0, read_bytes=0.0, write_bytes=0.0))
            a["launches"] += 1
            a["read_bytes"] += f * 1024 * k_rd
            a["write_bytes"] += w * 1024 * k_wr
        rd = sum(a["read_bytes"] for a in fam.values())
        wb = sum(a["write_bytes"] for a in fam.values())
        top = sorted(fam.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"]))
        res[name] = {"launches": len(sf), "read_bytes": int(rd), "write_bytes": int(wb), "hbm_bytes_per_step": int(rd + wb),
                     "by_kernel": {k: {"launches": v["launches"], "read_bytes": int(v["read_bytes"]), "write_bytes": int(v["write_bytes"])}
                                   for k, v in top}}
        print(f"## {name}: {len(sf)} launches, {rd / 1e9:.2f} GB read + {wb / 1e9:.2f} GB written = {(rd + wb) / 1e9:.2f} GB per step")
        for k, v in top[:16]:
            print(f"  {v['launches']:4d}  {v['read_bytes'] / 1e9:7.2f} + {v['write_bytes'] / 1e9:6.2f} GB  {k}")
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = ""
    sys.path.insert(0, ROOT)
    from sod100k_amd import _native as N
    res["_kernel_sources_sha16"] = N.sources_sha16()
    res["_source"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --train-net {net} (x2: csnet-L-x2 at batch 256; "
                      "unpruned: the expand-2 training net, fp32 at batch 64, bf16 at 256), "
                      "last step of each storage mode between two bce_logits_kernel launches; calibrated with tools/probes/fetch_cal "
                      f"(x{k_rd:.3f} reads, x{k_wr:.3f} writes); tree {head}")
    out = os.path.join(ROOT, "profiles", f"{tag}_pmc_train{sfx}.json")
    json.dump(res, open(out, "w"), indent=1)
    json.dump(res, open(os.path.join(ROOT, "profiles", f"pmc_train{sfx}_latest.json"), "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
