#!/usr/bin/env python3
"""HBM traffic per kernel of ONE eval forward from rocprofv3 FETCH_SIZE / WRITE_SIZE passes, calibrated on tools/probes/fetch_cal
(MI355X_MICROARCH.md, HBM section: the counters are per-access-width; calibrate on a known byte count in the kernel's pattern).

usage: tools/pmc_hbm.py <dir> <tag> [--units unit_table.json]
<dir> holds cal_fetch/ cal_write/ fwd_fetch/ fwd_write/ (tools/gpu_pmc_hbm.sh).  Writes profiles/<tag>_pmc_hbm.json,
profiles/pmc_latest.json and prints a table."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dispatches(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            disp[int(r["Dispatch_Id"])] = (r["Kernel_Name"].replace("void ", "").split("(")[0], float(r["Counter_Value"]))
    return [disp[k] for k in sorted(disp)]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    GiB = float(1 << 30)
    cf = [v for n, v in dispatches(os.path.join(src, "cal_fetch"), "FETCH_SIZE") if "stream_kernel" in n or "stride2" in n]
    cw = [v for n, v in dispatches(os.path.join(src, "cal_write"), "WRITE_SIZE") if "stream_kernel" in n or "stride2" in n]
    # launch order of fetch_cal: read 4 / 8 / 16 B per lane, stride-2 dword pairs, write 4 / 8 / 16  (counters are in KiB)
    f_rd = {4: GiB / (cf[0] * 1024), 8: GiB / (cf[1] * 1024), 16: GiB / (cf[2] * 1024), "pair": GiB / (cf[3] * 1024)}
    f_wr = {4: GiB / (cw[4] * 1024), 8: GiB / (cw[5] * 1024), 16: GiB / (cw[6] * 1024)}
    print("calibration: bytes per FETCH_SIZE KiB-unit x1024:", {k: round(v, 3) for k, v in f_rd.items()},
          " WRITE_SIZE:", {k: round(v, 3) for k, v in f_wr.items()})
    # dominant access width of every kernel family (bytes per lane of the loads / stores that carry the traffic)
    width = {"pw4_kernel": (8, 8), "c3q_kernel": (8, 8), "dw3x3x2_bn_prelu_kernel": (16, 16), "dw3x3x2_fast_kernel": (16, 16), "ilb_kernel": (8, 16), "hz_kernel": (8, 8), "msr_kernel": (16, 16),
             "msblock_kernel": (4, 4), "goct_pw_kernel": (4, 4), "goct_c3_kernel": (4, 4), "pool2_kernel": (16, 8),
             "bilinear_up2_kernel": (4, 4)}

    def fwd(name, counter):
        seq = dispatches(os.path.join(src, name), counter)
        ends = [i for i, (n, _) in enumerate(seq) if "bilinear_up2" in n]
        return seq[ends[-2] + 1: ends[-1] + 1]
    fe, wr = fwd("fwd_fetch", "FETCH_SIZE"), fwd("fwd_write", "WRITE_SIZE")
    assert [n for n, _ in fe] == [n for n, _ in wr], "dispatch order differs between the passes"
    agg = collections.OrderedDict()
    rows = []
    for (n, f), (_, w) in zip(fe, wr):
        k = n.split("<")[0]
        wl, ws = width.get(k, (4, 4))
        rd_b, wr_b = f * 1024 * f_rd[wl], w * 1024 * f_wr[ws]
        rows.append((n, rd_b, wr_b))
        a = agg.setdefault(k, dict(launches=0, rd=0.0, wr=0.0))
        a["launches"] += 1; a["rd"] += rd_b; a["wr"] += wr_b
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = "?"
    sys.path.insert(0, ROOT)
    from sod100k_amd import _native as N
    out = {"_kernel_sources_sha16": N.sources_sha16(),
           "_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/unit_table.py, one eval forward, batch 64; "
                      f"calibrated with tools/probes/fetch_cal; tree {head}",
           "_calibration": {"read_bytes_per_counted_byte": {str(k): round(v, 4) for k, v in f_rd.items()},
                            "write_bytes_per_counted_byte": {str(k): round(v, 4) for k, v in f_wr.items()}}}
    for k, a in agg.items():
        out[k] = dict(launches_per_forward=a["launches"], hbm_read_bytes_per_launch=int(a["rd"] / a["launches"]),
                      hbm_write_bytes_per_launch=int(a["wr"] / a["launches"]),
                      hbm_bytes_per_launch=int((a["rd"] + a["wr"]) / a["launches"]),
                      hbm_bytes_per_forward=int(a["rd"] + a["wr"]), access_width_B=list(width.get(k, (4, 4))))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
    units = None
    if "--units" in sys.argv:
        units = json.load(open(sys.argv[sys.argv.index("--units") + 1]))["units"]
    print(f"{'kernel':28s} {'launches':>8s} {'read MB':>10s} {'write MB':>10s} {'total MB':>10s}")
    for k, a in agg.items():
        print(f"{k:28s} {a['launches']:8d} {a['rd'] / 1e6:10.1f} {a['wr'] / 1e6:10.1f} {(a['rd'] + a['wr']) / 1e6:10.1f}")
    print(f"{'whole forward':28s} {len(rows):8d} {sum(r[1] for r in rows) / 1e6:10.1f} {sum(r[2] for r in rows) / 1e6:10.1f} "
          f"{sum(r[1] + r[2] for r in rows) / 1e6:10.1f}")
    with open(os.path.join(src, "per_dispatch.txt"), "w") as fo:
        for n, r, w in rows:
            fo.write(f"{n[:40]:40s} read {r / 1e6:9.1f} MB  write {w / 1e6:9.1f} MB\n")


if __name__ == "__main__":
    main()
