"""How many host threads should the CHECKER (the torch oracle on the GPU box's CPU) run on?  Times the two kinds of oracle work
the GPU suite does -- a tiny train step with autograd (B=2, 48x48; fp32 and fp64) and a no-grad train-mode forward of 64
images at 224x224 -- under torch.set_num_threads(n).  Used once in round 6 to pick tests/conftest.py's cap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle import csnet_oracle as O, inputs as I

man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
cfg = O.load_layer_config_json(man)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch default threads", torch.get_num_threads(),
      "load", os.getloadavg(), flush=True)
default = torch.get_num_threads()
x = torch.from_numpy(I.randn_batch(31, 2, 48, 48)); t = torch.from_numpy(I.binary_target(32, 2, 48, 48))
kw = dict(expandflop=1.0, flops_weight=3.0, batchsize=2, lr=0.0, wd=0.0)
xb = torch.from_numpy(I.randn_batch(70, 64))
for n in [default, 64, 32, 16, 8, 4, 2]:
    if n > default:
        continue
    torch.set_num_threads(n)
    sd = O.load_weights(man)
    t0 = time.time(); O.train_step(cfg, {k: v.clone() for k, v in sd.items()}, x, t, **kw); t1 = time.time()
    O.train_step(cfg, {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, x.double(), t.double(), **kw)
    t2 = time.time()
    with torch.no_grad():
        O.csnet_forward(cfg, sd, xb, training=True)
    t3 = time.time()
    print(f"threads {n:4d}: tiny step fp32 {t1 - t0:6.2f} s  fp64 {t2 - t1:6.2f} s | 64-image train-mode forward {t3 - t2:6.2f} s", flush=True)
