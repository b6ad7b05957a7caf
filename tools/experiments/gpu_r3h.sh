#!/bin/bash
mkdir -p gpurun_out/r3h
O=gpurun_out/r3h
timeout 200 python tools/unit_table.py --tag base --quiet 2>&1 | tail -1
for sb in 32 22 16; do
  CSN_SUB_BATCH=$sb CSN_SLICE_LANES=1 timeout 200 python tools/unit_table.py --tag "slices_sb$sb" --quiet 2>&1 | tail -1
  CSN_SUB_BATCH=$sb CSN_SLICE_LANES=0 timeout 200 python tools/unit_table.py --tag "seq_sb$sb" --quiet 2>&1 | tail -1
done
# numerics: sliced-concurrent output equals the whole-batch output
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from sod100k_amd.model import csnet as M
from sod100k_amd.checkpoint import load_manifest_state_dict
man = "sod100k_amd/data/csnet-L-x2.json"
def run(sb, lanes):
    os.environ["CSN_SUB_BATCH"] = str(sb); os.environ["CSN_SLICE_LANES"] = str(lanes)
    m = M.build_model(predefine=man); m.load_state_dict(load_manifest_state_dict(man)); m = m.cuda().eval()
    x = torch.randn(64, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        ys = [m(x).clone() for _ in range(4)]     # eager, capture, replay, replay
    torch.cuda.synchronize()
    return ys
a = run(0, 0); b = run(32, 1); c = run(22, 1)
print("whole vs 2 concurrent slices: max diff", max(float((p - q).abs().max()) for p, q in zip(a, b)), " 3 slices:", max(float((p - q).abs().max()) for p, q in zip(a, c)))
PY
