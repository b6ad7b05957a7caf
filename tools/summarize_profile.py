#!/usr/bin/env python3
"""Turn the scratch rocprofv3 output of tools/gpu_profile_round.sh (gpurun_out/round/) into the tracked
summaries under profiles/: kernel stats of the default bench command, HBM traffic per kernel from the two PMC
passes, and profiles/pmc_latest.json (read by bench.py for roofline.traffic).

HBM bytes: FETCH_SIZE / WRITE_SIZE are reported in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950
FETCH_SIZE reports exactly 1/2 of a wide (16 B/lane) coalesced stream, other widths must be calibrated on a
known byte count in the kernel's own access pattern.  Calibration used here:
  * depthwise kernels (b128 loads): factor 2.0 (checks against in+halo bytes within 3 %);
  * goct_pw_kernel (dword buffer loads): factor from its cls_layer launch, which reads exactly
    79 x 112 x 112 x 4 B x 64 images once (no re-reads, no halo) -- taken from a third pass with
    `--no-fuse-cls`, because the default plan evaluates cls_layer inside CSFHead.fuse1x1's epilogue;
  * WRITE_SIZE: factor 1.0 (cls launch writes 112 x 112 x 4 B x 64: matches within 2 %).
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r1"


def short(n):
    return n.replace("void ", "").split("<")[0].split("(")[0]


def per_dispatch(name, counter):
    rows = list(csv.DictReader(open(os.path.join(SRC, name, f"{name}_counter_collection.csv"))))
    disp = collections.OrderedDict()
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        disp[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), float(r["Counter_Value"]))
    seq = list(disp.values())
    ends = [i for i, (n, _) in enumerate(seq) if n == "bilinear_up2_kernel"]
    return seq[ends[-2] + 1: ends[-1] + 1]      # one whole forward


def main():
    os.makedirs(DST, exist_ok=True)
    bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
    stats = list(csv.DictReader(open(os.path.join(SRC, "trace", "trace_kernel_stats.csv"))))
    fetch = per_dispatch("fetch", "FETCH_SIZE")
    write = per_dispatch("write", "WRITE_SIZE")
    assert [n for n, _ in fetch] == [n for n, _ in write]
    B = bench["config"]["batch_per_gpu"]
    cls_read = 79 * 112 * 112 * 4 * B
    cal = per_dispatch("fetchcal", "FETCH_SIZE")     # same command with --no-fuse-cls: the last pw launch is cls_layer
    pw_idx = [i for i, (n, _) in enumerate(cal) if n == "goct_pw_kernel"]
    c_pw = cls_read / (cal[pw_idx[-1]][1] * 1024)
    factor = collections.defaultdict(lambda: 2.0, {"goct_pw_kernel": c_pw, "msblock_kernel": c_pw})
    agg = collections.OrderedDict()
    for (n, f), (_, w) in zip(fetch, write):
        a = agg.setdefault(n, dict(launches=0, rd=0.0, wr=0.0))
        a["launches"] += 1
        a["rd"] += f * 1024 * factor[n]
        a["wr"] += w * 1024
    pmc = {n: dict(launches_per_forward=a["launches"], hbm_read_bytes_per_launch=int(a["rd"] / a["launches"]),
                   hbm_write_bytes_per_launch=int(a["wr"] / a["launches"]),
                   hbm_bytes_per_launch=int((a["rd"] + a["wr"]) / a["launches"]), fetch_factor=round(factor[n], 3))
           for n, a in agg.items()}
    json.dump(pmc, open(os.path.join(DST, "pmc_latest.json"), "w"), indent=1)
    json.dump(pmc, open(os.path.join(DST, f"{TAG}_pmc_hbm.json"), "w"), indent=1)
    with open(os.path.join(DST, f"{TAG}_kernel_stats.md"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   ({TAG}, MI355X)\n\n")
        f.write(f"bench line of the same build: {bench['value']} img/s, {bench['ms_per_step']} ms/step, "
                f"dominant kernel {bench['roofline']['kernel']} at {bench['roofline']['us_per_launch']} us/launch "
                f"(HIP events) \n\n| kernel | calls | total ms | avg us | % | min us | max us |\n|---|---|---|---|---|---|---|\n")
        for r in stats:
            f.write(f"| {r['Name'][:60]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | "
                    f"{float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |\n")
        f.write("\n## HBM traffic per launch (PMC passes FETCH_SIZE / WRITE_SIZE, calibrated, see tools/summarize_profile.py)\n\n"
                "| kernel | launches/forward | read MB | write MB | algorithmic MB (bench) |\n|---|---|---|---|---|\n")
        pk = bench["roofline"]["per_kernel"]
        for n, a in pmc.items():
            alg = ""
            if n in pk and pk[n].get("alg_GBps"):
                alg = f"{pk[n]['alg_GBps'] * pk[n]['us_per_launch'] * 1e-3:.1f}"
            f.write(f"| {n} | {a['launches_per_forward']} | {a['hbm_read_bytes_per_launch']/1e6:.1f} | "
                    f"{a['hbm_write_bytes_per_launch']/1e6:.1f} | {alg} |\n")
    json.dump(bench, open(os.path.join(DST, f"{TAG}_bench.json"), "w"), indent=1)
    print(open(os.path.join(DST, f"{TAG}_kernel_stats.md")).read())
    csf_summary(bench)


def csf_summary(bench):
    """profiles/<tag>_csf_head.md: one CSF+Res2Net head forward launch by launch (kernel trace) with the SQ counters of
    the same launches (separate PMC pass).  Matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES (= 32 x MFMAs issued)
    / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
    tr = os.path.join(SRC, "csf_trace", "trace_kernel_trace.csv")
    if not os.path.exists(tr) or "csf_res2net" not in bench:
        return
    rows = [r for r in csv.DictReader(open(tr)) if short(r["Kernel_Name"]).startswith("csf_")
            and short(r["Kernel_Name"]) != "csf_bn_act_kernel"]          # the backbone's pass, not part of the head
    seqs, cur = [], []
    for r in rows:      # a forward ends with the resize that follows csf_cls_kernel (earlier resizes feed the fuse GEMMs)
        cur.append(r)
        if short(r["Kernel_Name"]) == "csf_resize_kernel" and len(cur) > 1 and short(cur[-2]["Kernel_Name"]) == "csf_cls_kernel":
            seqs.append(cur)
            cur = []
    last = [r for r in seqs[-1] if short(r["Kernel_Name"]) != "csf_prep_kernel"]
    pmc_rows = list(csv.DictReader(open(os.path.join(SRC, "csf_pmc", "pmc_counter_collection.csv"))))
    disp = collections.OrderedDict()
    for r in pmc_rows:
        n = short(r["Kernel_Name"])
        if not n.startswith("csf_") or n in ("csf_prep_kernel", "csf_bn_act_kernel"):
            continue
        disp.setdefault(r["Dispatch_Id"], dict(name=n))[r["Counter_Name"]] = float(r["Counter_Value"])
    pseq, cur = [], []
    for d in disp.values():
        cur.append(d)
        if d["name"] == "csf_resize_kernel" and len(cur) > 1 and cur[-2]["name"] == "csf_cls_kernel":
            pseq.append(cur)
            cur = []
    plast = pseq[-1] if pseq else []
    c = bench["csf_res2net"]
    with open(os.path.join(DST, f"{TAG}_csf_head.md"), "w") as f:
        f.write(f"# CSF+Res2Net decoder head, one forward ({c['workload']})\n\n")
        f.write(f"bench: {c['value']} img/s whole network, head {c['ms_head_hip']} ms = {c['head_roofline']['achieved']} TFLOP/s "
                f"= {c['head_roofline']['frac']} of the fp32 matrix peak, backbone (MIOpen) {c['ms_backbone_miopen']} ms\n\n"
                "| # | kernel | blocks | us | matrix pipe | wait_any | wait_inst | active | LDS conflict |\n|---|---|---|---|---|---|---|---|---|\n")
        tot = collections.Counter()
        for i, r in enumerate(last):
            n = short(r["Kernel_Name"])
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            tot[n] += us
            d = plast[i] if i < len(plast) and plast[i]["name"] == n else {}
            wc = d.get("SQ_WAVE_CYCLES", 0) or 1
            gui = d.get("GRBM_GUI_ACTIVE", 0)
            mf = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 1024) if gui else 0
            blocks = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * int(r["Grid_Size_Y"])
            cols = (f"{mf:.2f} | {d.get('SQ_WAIT_ANY', 0) / wc:.2f} | {d.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | "
                    f"{d.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | "
                    f"{d.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, d.get('SQ_LDS_IDX_ACTIVE', 0)):.2f}") if d else " | | | | "
            f.write(f"| {i} | {n} | {blocks} | {us:.1f} | {cols} |\n")
        f.write("\n| kernel | us per forward |\n|---|---|\n")
        for n, us in tot.most_common():
            f.write(f"| {n} | {us:.0f} |\n")
        f.write(f"| total | {sum(tot.values()):.0f} |\n")
    print(open(os.path.join(DST, f"{TAG}_csf_head.md")).read()[:1500])


if __name__ == "__main__":
    main()
