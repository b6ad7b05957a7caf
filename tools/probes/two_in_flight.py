"""Ceiling of keeping TWO batches of 64 in flight (two plans with their own workspace / output, two streams, steps alternate)
against the bench's one batch in flight: does the latency-bound tail of one batch (stages 3-4, head) hide under the other's
bandwidth-bound stages 0-2?  Measurement only (round 6); outputs are checked bit-identical to the sequential run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sod100k_amd.model import csnet as M
from sod100k_amd.checkpoint import load_manifest_state_dict

dev = torch.device("cuda:0")
man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
B, K = 64, 100
g = torch.Generator().manual_seed(0)
xs = [torch.randn(B, 3, 224, 224, generator=g).to(dev) for _ in range(2)]


def build(slice_lanes):
    ms, es, ys = [], [], []
    for i in range(2):
        m = M.build_model(predefine=man); m.load_state_dict(load_manifest_state_dict(man)); m = m.to(dev).eval()
        e = m.engine_for(xs[i], slice_lanes=slice_lanes); e.refresh(m._arena.flat)
        y = torch.empty(B, 1, 224, 224, device=dev)
        for _ in range(4):
            e.forward(xs[i], out=y)
        ms.append(m); es.append(e); ys.append(y)
    torch.cuda.synchronize()
    return ms, es, ys


for sl in (False, True):
    ms, es, ys = build(sl)
    ref = [y.clone() for y in ys]
    # one in flight: alternate the two plans on ONE stream
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        es[k % 2].forward(xs[k % 2], out=ys[k % 2])
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    st = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for rounds in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            with torch.cuda.stream(st[k % 2]):
                es[k % 2].forward(xs[k % 2], out=ys[k % 2])
        torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    same = all(torch.equal(a, b) for a, b in zip(ref, ys))
    print(f"slice_lanes={sl}: one batch in flight {t1 / K * 1e3:.3f} ms/step = {B * K / t1:.0f} img/s | two in flight "
          f"{t2 / K * 1e3:.3f} ms/step = {B * K / t2:.0f} img/s | outputs identical: {same}", flush=True)
    del ms, es, ys
