"""Per-unit timing table of the eval forward (batch 64, 3x224x224, csnet-L-x2) on the current device.

usage: python tools/unit_table.py [--batch 64] [--iters 10] [--steps 30] [--tag NAME] [--json out.json]
Environment switches of sod100k_amd.engine (CSN_PW4, CSN_TILED3, ...) and SOD100K_HIP_LIB (variant builds) apply.
Prints one line per launch group: unit name, kernel, ms (HIP events after every launch, serialised), algorithmic MB,
GB/s; then the graph-replayed whole-forward time (median of --steps HIP-event timed steps).
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--tag", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--quiet", action="store_true", help="only the summary line")
    args = ap.parse_args()
    from sod100k_amd.model import csnet as M
    from sod100k_amd.checkpoint import load_manifest_state_dict
    dev = torch.device("cuda", 0)
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    model = M.build_model(predefine=man)
    model.load_state_dict(load_manifest_state_dict(man))
    model = model.to(dev).eval()
    x = torch.randn(args.batch, 3, args.size, args.size, generator=torch.Generator().manual_seed(0)).to(dev)
    eng = model.engine_for(x)
    eng.refresh(model._arena.flat)
    y = torch.empty(args.batch, 1, args.size, args.size, device=dev)
    for _ in range(5):
        eng.forward(x, out=y)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in evs:
        a.record(); eng.forward(x, out=y); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    med = statistics.median(ts)
    # per-unit times: the same kernels launched once per whole batch on one stream (at batch >= 32 the product's default plan runs
    # two half-batch slices side by side on two stream lanes: the replay median above is that plan's)
    if getattr(eng, "slice_lanes", False):
        eng = model.engine_for(x, slice_lanes=False)
        eng.refresh(model._arena.flat)
        for _ in range(3):
            eng.forward(x, out=y)
    ms, names, nbytes = eng.profile(x, iters=args.iters)
    ks = eng.kernel_stats()
    rows = []
    for u, (t, n, nb) in enumerate(zip(ms, names, nbytes)):
        if t <= 0:
            continue
        rows.append(dict(unit=eng.unit_names[u], kernel=n, ms=round(t, 4), alg_MB=round(nb / 1e6, 1),
                         GBps=round(nb / (t * 1e-3) / 1e9, 1) if nb else None))
    if not args.quiet:
        for r in rows:
            print(f"{r['unit']:24s} {r['kernel']:26s} {r['ms'] * 1e3:8.1f} us {r['alg_MB']:8.1f} MB {r['GBps'] or 0:8.1f} GB/s")
    ksum = "  ".join(f"{k.replace('_kernel', '')}={v[0]:.3f}" for k, v in sorted(ks.items(), key=lambda kv: -kv[1][0]))
    print(f"[{args.tag}] replay median {med:.3f} ms = {args.batch / med * 1e3:.0f} img/s (min {ts[0]:.3f}) | serialised sum "
          f"{sum(ms):.3f} ms | {ksum}", flush=True)
    if args.json:
        json.dump(dict(tag=args.tag, median_ms=med, min_ms=ts[0], units=rows,
                       kernels={k: dict(ms=v[0], launches=v[1]) for k, v in ks.items()}), open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
