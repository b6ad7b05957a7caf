#!/usr/bin/env python3
"""Per-dispatch counters of ONE eval forward, merged over the rocprofv3 --pmc passes under <dir>/pmc_*/ (same command per
pass, so the dispatch order of the forward is identical).  usage: tools/pmc_forward.py <dir> [kernel-substring ...]"""
import collections
import csv
import glob
import os
import sys


def forward_of(f):
    rows = list(csv.DictReader(open(f)))
    disp = collections.OrderedDict()
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], {}, r.get("Grid_Size", "?"), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?"))])
        d[1][r["Counter_Name"]] = float(r["Counter_Value"])
    seq = [disp[k] for k in sorted(disp)]
    ends = [i for i, d in enumerate(seq) if "bilinear_up2" in d[0]]
    return seq[ends[-2] + 1: ends[-1] + 1]


def main():
    out = sys.argv[1]
    filt = sys.argv[2:]
    merged = None
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            continue
        seq = forward_of(fs[0])
        if merged is None:
            merged = [[n, dict(c), g, v] for n, c, g, v in seq]
        else:
            assert len(seq) == len(merged), (len(seq), len(merged))
            for m, s in zip(merged, seq):
                assert m[0] == s[0]
                m[1].update(s[1])
    for i, (n, c, g, v) in enumerate(merged):
        k = n.replace("void ", "").split("(")[0]
        if filt and not any(f in k for f in filt):
            continue
        w = c.get("SQ_WAVES", 0) or 1
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        line = f"{i:3d} {k[:34]:34s} waves {int(w):7d}"
        for key, lab in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_MFMA", "mfma"), ("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_SMEM", "smem"),
                         ("SQ_INSTS_LDS", "lds"), ("SQ_INSTS_VMEM_RD", "vrd"), ("SQ_INSTS_VMEM_WR", "vwr")):
            if key in c:
                line += f" {lab}/w {c[key] / w:7.0f}"
        for key, lab in (("SQ_WAIT_ANY", "wait"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_ANY", "act"),
                         ("SQ_ACTIVE_INST_VALU", "act_valu"), ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy"), ("SQ_ACTIVE_INST_LDS", "act_lds"),
                         ("SQ_ACTIVE_INST_VMEM", "act_vmem")):
            if key in c:
                line += f" {lab} {c[key] / wc:5.2f}"
        if "SQ_BUSY_CYCLES" in c:
            line += f" busy_cyc {c['SQ_BUSY_CYCLES']:.3g} wave_cyc/w {wc / w:.0f}"
        if "GRBM_GUI_ACTIVE" in c:
            line += f" gui {c['GRBM_GUI_ACTIVE']:.0f}"
        print(line)


if __name__ == "__main__":
    main()
