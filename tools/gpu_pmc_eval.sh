#!/bin/bash
# SQ / TA / TCP counter passes over the eval forward (batch 64): per kernel family, the LARGEST dispatch (most waves) of the last
# forward -> gpurun_out/$1/pmc_eval.txt.  Counters only (no tracing), separate runs per pass.
out=$PWD/gpurun_out/$1; mkdir -p $out
R=$PWD
CMD="env CSN_SLICE_LANES=0 python $R/bench.py --steps 2 --warmup 1 --train-steps 0 --csf-batch 0 --no-cpu-baseline --no-latency-b1 --event-steps 0 --profile-iters 1"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  ( timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pe_$name -o $name -- $CMD ) > $out/pe_$name.log 2>&1
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run b SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU
if [ -n "$PMC_EVAL_ALL" ]; then   # (the memory-side passes: mid-round only, profiles/r5_pmc_eval_mid_round.txt)
run c SQ_WAVES TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_TOTAL_CYCLES
run d SQ_WAVES TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_LFIFO_STALL_CYCLES TCP_TCC_READ_REQ
fi
run e SQ_WAVES SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
cd $R
python - $out <<'PY' | tee $out/pmc_eval.txt
import collections, csv, glob, os, re, sys
src = sys.argv[1]
best = {}
for d in sorted(glob.glob(os.path.join(src, "pe_*"))):
    if not os.path.isdir(d): continue
    disp = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            e = disp.setdefault(k, {"name": re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]).split("<")[0], "grid": int(r.get("Grid_Size", 0) or 0), "c": {}})
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    fam = {}
    for k in sorted(disp):
        e = disp[k]
        key = e["name"]
        if key not in fam or e["grid"] >= fam[key]["grid"]: fam[key] = e     # the largest (and latest) dispatch of the family
    for key, e in fam.items():
        b = best.setdefault(key, {"grid": e["grid"], "c": {}})
        b["c"].update(e["c"])
for key in sorted(best, key=lambda k: -best[k]["c"].get("SQ_WAVE_CYCLES", 0)):
    c = best[key]["c"]
    w = max(c.get("SQ_WAVES", 1), 1); wc = max(c.get("SQ_WAVE_CYCLES", 1), 1); busy = max(c.get("SQ_BUSY_CYCLES", 1), 1)
    print("%s grid %d waves(counter) %d" % (key, best[key]["grid"], w))
    for k in sorted(c):
        print("   %-30s %14.0f  per wave %10.1f  / wave-cycles %.3f  / busy %.3f" % (k, c[k], c[k] / w, c[k] / wc, c[k] / busy))
PY
