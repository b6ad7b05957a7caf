"""One-off GPU run (not a test): eval forward of the shipped x2 / x1 nets at geometries the suite does not walk -- wide, tall, 320-wide
(two depthwise tiles per row), batch sizes around the slice threshold -- against the oracle.  usage: python tools/probes/shape_fuzz.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as P
from sod100k_amd import _native as N
lib, dev = N.load(), torch.device("cuda", 0)
worst = 0.0
for man in ("csnet-L-x2.json", "csnet-L-x1.json"):
    m, sd = P.make_model(lib, os.path.join(ROOT, "sod100k_amd", "data", man), dev)
    for shape in [(1, 336, 224), (2, 16, 400), (1, 320, 320), (5, 32, 32), (33, 64, 96), (32, 48, 48), (3, 112, 528), (2, 272, 80), (1, 16, 16), (40, 16, 32)]:
        x = torch.from_numpy(np.random.default_rng(7).standard_normal((shape[0], 3) + shape[1:]).astype(np.float32))
        y = m(x.to(dev)).cpu()
        ref = P.oracle_forward(os.path.join(ROOT, "sod100k_amd", "data", man), sd, x)
        e = float((y - ref).abs().max()); worst = max(worst, e)
        eng = m.engine_for(x.to(dev))
        print(man, shape, "max-abs %.2e" % e, "slices" if eng.slice_lanes else "", flush=True)
        assert e <= 1e-4, (man, shape, e)
print("worst", worst)
