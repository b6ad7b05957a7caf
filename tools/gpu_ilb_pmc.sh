#!/bin/bash
# SQ counter passes over tools/probes/ilb_bench_pmc (ilb_kernel at the stage-3.1 and stage-4.1 geometries, no stamps): per dispatch
# instruction counts and where the waves wait -> gpurun_out/$1/ilb_pmc.txt
out=$PWD/gpurun_out/$1; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  ( timeout 120 rocprofv3 --pmc "$@" --output-format csv -d $out/ilbpmc_$name -o $name -- $R/tools/probes/ilb_bench_pmc two ) > $out/ilbpmc_$name.log 2>&1
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD
run b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES
cd $R
python - $out <<'PY' | tee $out/ilb_pmc.txt
import collections, csv, glob, os, sys
src = sys.argv[1]
disp = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(src, "ilbpmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:])
        disp.setdefault(k, {})[r["Counter_Name"]] = disp.setdefault(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# the 12 launches of a geometry are consecutive: print the last launch of each kernel instantiation / grid
last = collections.OrderedDict()
for (d, n), c in disp.items():
    last[(n, c.get("SQ_WAVES"))] = c
for (n, w), c in last.items():
    w = max(w or 1, 1)
    busy = max(c.get("SQ_BUSY_CYCLES", 1), 1); wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
    print(n, "waves %d" % w)
    for k in sorted(c):
        if k in ("SQ_WAVES",): continue
        print("   %-28s %14.0f  per wave %10.1f   / busy %.3f   / wave-cycles %.3f" % (k, c[k], c[k] / w, c[k] / busy, c[k] / wc))
PY
