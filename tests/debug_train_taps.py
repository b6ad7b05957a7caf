"""Manual debugging aid (not collected by pytest): per-unit error of the train-mode forward vs the oracle."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parity_cases as P
from oracle import csnet_oracle as O, inputs as I
from sod100k_amd import _native as N

dev = torch.device("cuda", 0)
man = os.path.join(P.ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
B, size = 4, 96
m, sd = P.make_model(N.load(), man, dev)
m.train()
x = torch.from_numpy(I.randn_batch(10, B, size, size))
y = m(x.to(dev)).cpu()
eng = m.engine_for(x.to(dev))
cfg = O.load_layer_config_json(man)
taps = {}
sd_ref = {k: v.clone() for k, v in sd.items()}
with torch.no_grad():
    ref = O.csnet_forward(cfg, sd_ref, x, training=True, taps=taps)
units, acts, names = m.describe(m._arena.offsets)
for u, name in zip(units, names):
    if name == "cls_layer":
        continue
    for j in range(3):
        a = u.out_act[j]
        if a < 0 or taps[name][j] is None:
            continue
        g = eng.activation(a).cpu()
        r = taps[name][j]
        bnp = f"{name}.bns.{j}" if f"{name}.bns.{j}.running_var" in sd_ref else f"{name}.bn"
        rv = (sd_ref[bnp + ".running_var"] - 0.9 * sd[bnp + ".running_var"]) / 0.1
        print(f"{name:32s} br{j} absmax {r.abs().max():9.3f} err {(g - r).abs().max():.2e}  min batch var {rv.min():.2e}")
print("logits err", (y - ref).abs().max().item())
