"""Parity checks shared by the CPU-emulation tests (``-m "not gpu"``) and the MI355X tests (``-m gpu``).

Every function takes the bound C-ABI library (``tests/emu/libcsnet_emu.so`` or ``libcsnet_hip.so``) and a
torch device and drives the kernels ONLY through the C ABI (sod100k_amd.engine.Engine / CSNet.forward).
The oracle (oracle/csnet_oracle.py) and the committed goldens are the checkers.
"""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import csnet_oracle as O, inputs as I
from sod100k_amd import _native as N
from sod100k_amd.engine import Engine
from sod100k_amd.model import csnet as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 1e-4          # north_star: saliency logits within 1e-4 max-abs of the reference CPU forward (fp32)
UNIT_TOL = 2e-5     # a single unit on O(1) data: fp32 accumulation-order noise only


def make_model(lib, manifest, device, sub_batch=0):
    sd = O.load_weights(manifest)
    m = M.build_model(predefine=manifest)
    m.load_state_dict(sd)
    m = m.to(device).eval()
    m._lib = lib if device.type == "cpu" else None      # GPU: the product's own loader (libcsnet_hip.so)
    if device.type == "cuda":
        assert lib is N.load()
    m._sub_batch = sub_batch
    return m, sd


def oracle_forward(manifest, sd, x, taps=None):
    with torch.no_grad():
        return O.csnet_forward(O.load_layer_config_json(manifest), sd, x, taps=taps)


def check_golden_logits(lib, device, manifest, golden_file, x, tol=TOL):
    m, _ = make_model(lib, manifest, device)
    y = m(x.to(device)).cpu()
    g = torch.from_numpy(np.load(os.path.join(GOLD, golden_file)))
    assert y.shape == g.shape
    err = (y - g).abs().max().item()
    assert err <= tol, f"{golden_file}: max-abs {err:.3e} > {tol}"
    return err


def check_unit_probes(lib, device, manifest):
    """G3: every unit's output against the reference's probes (localises a mismatch to a kernel)."""
    m, _ = make_model(lib, manifest, device)
    x = torch.from_numpy(I.randn_batch(0, 2)).to(device)
    eng = m.engine_for(x)
    eng.set_option(N.OPT_FUSE_DW, 0)      # materialise every unit's output for the probes
    eng.set_option(N.OPT_FUSE_CLS, 0)
    m(x)
    units, acts, names = m.describe(m._arena.offsets)
    probes = json.load(open(os.path.join(GOLD, "g3_unit_probes_x2.json")))
    worst = 0.0
    for u, name in zip(units, names):
        if name == "cls_layer":
            continue
        for j in range(N.MAX_BRANCH):
            a = u.out_act[j]
            if a < 0:
                continue
            pr = probes[name][j]
            got = eng.activation(a).cpu().numpy()
            assert list(got.shape) == pr["shape"], (name, j)
            flat = got.reshape(-1)
            s = flat[I.probe_indices(flat.size)]
            ref = np.array(pr["samples"], dtype=np.float32)
            scale = max(1.0, pr["absmax"])
            err = float(np.abs(s - ref).max()) / scale
            assert err <= UNIT_TOL, f"{name} branch {j}: rel err {err:.3e}"
            l2 = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
            assert abs(l2 - pr["l2"]) <= 1e-4 * max(1.0, pr["l2"]), (name, j, l2, pr["l2"])
            worst = max(worst, err)
    return worst


def check_ilb_vs_unit_kernels(lib, device, manifest, B, H, W, seed=3, min_blocks=1, env=None):
    """Whole-ILBlock launches (k_ilb.hip, round 5) against the unit kernels they replace (CSN_ILB=0, read at plan creation): every
    block output (conv3x3_2) to UNIT_TOL relative, the logits of both to the oracle.  Returns (fused units, worst block deviation,
    logit deviation from the oracle)."""
    x = torch.from_numpy(I.randn_batch(seed, B, H, W))
    saved = {k: os.environ.get(k) for k in ["CSN_ILB"] + list(env or {})}
    try:
        os.environ.update(env or {})
        os.environ["CSN_ILB"] = "1"
        m1, sd = make_model(lib, manifest, device)
        y1 = m1(x.to(device)).cpu()
        e1 = m1.engine_for(x.to(device))
        names = [e1.lib.csn_unit_kernel_name(e1.plan, u).decode() for u in range(e1.n_units)]
        os.environ["CSN_ILB"] = "0"
        m0, _ = make_model(lib, manifest, device)
        y0 = m0(x.to(device)).cpu()
        e0 = m0.engine_for(x.to(device))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    nfused = sum(n == "ilb_kernel" for n in names)
    assert nfused >= 3 * min_blocks and nfused % 3 == 0, (nfused, names)
    units, acts, unames = m1.describe(m1._arena.offsets)
    worst = 0.0
    for u, name in zip(units, unames):
        if not name.endswith(".conv3x3_2"):
            continue
        for j in range(N.MAX_BRANCH):
            a = u.out_act[j]
            if a < 0:
                continue
            # a block output read ONLY by the stride-2 unit behind it is never stored at full resolution (that unit reads the 2x2
            # averages the block delivers: pool_skip): nothing to compare -- the blocks downstream cover it
            readers = [v for v in units if any(v.in_act[i] == a and v.cin[i] > 0 for i in range(int(v.n_in)))]
            if readers and all(v.kind == N.UNIT_GOCT and v.stride == 2 for v in readers):
                continue
            g1, g0 = e1.activation(a).cpu(), e0.activation(a).cpu()
            err = float((g1 - g0).abs().max()) / max(1.0, float(g0.abs().max()))
            assert err <= UNIT_TOL, f"{name} branch {j}: ilb_kernel vs unit kernels {err:.3e}"
            worst = max(worst, err)
    ref = oracle_forward(manifest, sd, x)
    err_o = float((y1 - ref).abs().max())
    assert err_o <= TOL and float((y0 - ref).abs().max()) <= TOL, err_o
    return nfused, worst, err_o


def check_hz_vs_pw4(lib, device, manifest, B, H, W, seed=7, env=None, fuse_cls=True):
    """hz_kernel (k_head.hip, round 6: the high output of CSFHead.fuse / fuse1x1 with the low -> high terms convolved at the LOW
    resolution, interpolated per output channel from LDS -- the reference's own order, csnet.py:702-707) against pw4_kernel's
    high-only form (CSN_HZ=0: interpolated INPUTS contracted at the output resolution): the outputs of both units to UNIT_TOL
    relative, the logits of both routes to the oracle.  `env`: geometry switches of the kernel (band rows, tiles per group,
    waves per block).  Returns (launches on hz_kernel, worst unit deviation, logit deviation from the oracle)."""
    x = torch.from_numpy(I.randn_batch(seed, B, H, W))
    saved = {k: os.environ.get(k) for k in ["CSN_HZ"] + list(env or {})}
    try:
        os.environ.update(env or {})
        out = {}
        for hz in ("1", "0"):
            os.environ["CSN_HZ"] = hz
            m, sd = make_model(lib, manifest, device)
            xd = x.to(device)
            eng = m.engine_for(xd)
            if not fuse_cls:
                eng.set_option(N.OPT_FUSE_CLS, 0)      # fuse1x1's 79 channels are stored (rows-stored form of the kernel)
            y = m(xd).cpu()
            eng.profile(xd, iters=1)
            census = {k: v[1] for k, v in eng.kernel_stats().items()}
            out[hz] = (m, eng, y, census)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    m1, e1, y1, c1 = out["1"]
    m0, e0, y0, c0 = out["0"]
    nhz = c1.get("hz_kernel", 0)
    assert nhz == 2 and c0.get("hz_kernel", 0) == 0 and c0["pw4_kernel"] == c1["pw4_kernel"] + 2, (c1, c0)
    units, acts, unames = m1.describe(m1._arena.offsets)
    worst = 0.0
    for u, name in zip(units, unames):
        if name not in ("oct_fuse.fuse", "oct_fuse.fuse1x1") or (name == "oct_fuse.fuse1x1" and fuse_cls):
            continue
        a = u.out_act[0]
        g1, g0 = e1.activation(a).cpu(), e0.activation(a).cpu()
        err = float((g1 - g0).abs().max()) / max(1.0, float(g0.abs().max()))
        assert err <= UNIT_TOL, f"{name}: hz_kernel vs pw4_kernel {err:.3e}"
        worst = max(worst, err)
    ref = oracle_forward(manifest, sd, x)
    err_o = float((y1 - ref).abs().max())
    assert err_o <= TOL and float((y0 - ref).abs().max()) <= TOL, err_o
    return nhz, worst, err_o


def check_lane_exchange_vs_loaded_halos(lib, device, manifest, B, H, W, seed=5):
    """Eval forward with the depthwise kernels' halo columns taken from the neighbouring lanes (CSN_DW_XL=1, the default: power-of-two
    lane groups per row, DPP moves) against the round-4 geometry with loaded halo columns (CSN_DW_XL=0; both read at plan creation).
    The arithmetic per output pixel is the same and no value depends on the tiling: the logits are BIT-IDENTICAL."""
    x = torch.from_numpy(I.randn_batch(seed, B, H, W))
    keys = ("CSN_DW_XL", "CSN_C3Q_HL")   # ... and c3q_kernel's halo-lane tiles against its 64-quad tiles with loaded edge columns
    saved = {k: os.environ.get(k) for k in keys}
    ys = {}
    try:
        for v in ("1", "0"):
            for k in keys:
                os.environ[k] = v
            m, sd = make_model(lib, manifest, device)
            ys[v] = m(x.to(device)).cpu()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert torch.equal(ys["1"], ys["0"]), float((ys["1"] - ys["0"]).abs().max())
    ref = oracle_forward(manifest, sd, x)
    err = float((ys["1"] - ref).abs().max())
    assert err <= TOL, err
    return err


def check_vs_oracle(lib, device, manifest, x, sub_batch=0, tol=TOL):
    m, sd = make_model(lib, manifest, device, sub_batch=sub_batch)
    y = m(x.to(device)).cpu()
    ref = oracle_forward(manifest, sd, x)
    err = (y - ref).abs().max().item()
    assert err <= tol, f"max-abs {err:.3e} > {tol}"
    return y, err


# ---------------------------------------------------------------------------------------------------
# single-unit plans for the op-level micro-goldens (G4)
# ---------------------------------------------------------------------------------------------------
class _Arena:
    def __init__(self):
        self.chunks, self.off, self.n = [], {}, 0

    def add(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        self.off[name] = self.n
        pad = (-arr.size) % 4
        self.chunks.append(np.concatenate([arr, np.zeros(pad, np.float32)]))
        self.n += arr.size + pad
        return self.off[name]

    def tensor(self, device):
        return torch.from_numpy(np.concatenate(self.chunks)).to(device)


def _bn(u, j, ar, arrs, tag, bn_prefix, prelu_key):
    u.bn[j].weight = ar.add(f"{j}w", arrs[f"{tag}/sd/{bn_prefix}.weight"])
    u.bn[j].bias = ar.add(f"{j}b", arrs[f"{tag}/sd/{bn_prefix}.bias"])
    u.bn[j].running_mean = ar.add(f"{j}m", arrs[f"{tag}/sd/{bn_prefix}.running_mean"])
    u.bn[j].running_var = ar.add(f"{j}v", arrs[f"{tag}/sd/{bn_prefix}.running_var"])
    u.bn[j].prelu = ar.add(f"{j}a", arrs[f"{tag}/sd/{prelu_key}"])


def run_g4_case(lib, device, tag, meta, arrs):
    """Returns list of (got, expected) per output branch."""
    kind = meta["kind"]
    ar = _Arena()
    acts = [(1, 0)]
    if kind == "cbr":
        cin, cout, k, stride = meta["cin"], meta["cout"], meta["k"], meta["stride"]
        xs = [arrs[f"{tag}/x{i}"] for i in range(len(cin))]
        B, _, H, W = xs[0].shape
        u = N.new_unit(N.UNIT_GOCT)
        u.n_in, u.n_out, u.ksize, u.stride = len(cin), len(cout), k, stride
        u.w_off[0] = ar.add("w", arrs[f"{tag}/sd/conv.weight"])
        in_ids, out_ids = [], []
        for i, c in enumerate(cin):
            acts.append((c, i)); in_ids.append(len(acts) - 1)
            u.cin[i], u.in_act[i] = c, in_ids[-1]
        base = 1 if stride == 2 else 0
        for j, c in enumerate(cout):
            acts.append((c, base + j)); out_ids.append(len(acts) - 1)
            u.cout[j], u.out_act[j] = c, out_ids[-1]
            _bn(u, j, ar, arrs, tag, f"bns.{j}", f"prelus.{j}.weight")
        eng = Engine(lib, [u], acts, B, H, W, device)
        for i, a in enumerate(in_ids):
            eng.activation(a).copy_(torch.from_numpy(xs[i]))
        eng.refresh(ar.tensor(device))
        eng.forward(torch.zeros(B, 1, H, W, device=device))
        return [(eng.activation(a).cpu().numpy(), arrs[f"{tag}/y{j}"]) for j, a in enumerate(out_ids)]
    if kind == "dw":
        ch = meta["ch"]
        xs = [arrs[f"{tag}/x{i}"] for i in range(len(ch))]
        B, _, H, W = xs[0].shape
        u = N.new_unit(N.UNIT_DW)
        u.n_in = u.n_out = len(ch)
        in_ids, out_ids = [], []
        for i, c in enumerate(ch):
            acts.append((c, i)); in_ids.append(len(acts) - 1)
            acts.append((c, i)); out_ids.append(len(acts) - 1)
            u.cin[i] = u.cout[i] = c
            u.in_act[i], u.out_act[i] = in_ids[-1], out_ids[-1]
            u.w_off[i] = ar.add(f"w{i}", arrs[f"{tag}/sd/convs.{i}.weight"])
            _bn(u, i, ar, arrs, tag, f"bns.{i}", f"prelus.{i}.weight")
        eng = Engine(lib, [u], acts, B, H, W, device)
        for i, a in enumerate(in_ids):
            eng.activation(a).copy_(torch.from_numpy(xs[i]))
        eng.refresh(ar.tensor(device))
        eng.forward(torch.zeros(B, 1, H, W, device=device))
        return [(eng.activation(a).cpu().numpy(), arrs[f"{tag}/y{j}"]) for j, a in enumerate(out_ids)]
    if kind == "ms":
        x = arrs[f"{tag}/x0"]
        B, cin, H, W = x.shape
        dil = meta["dil"]
        u = N.new_unit(N.UNIT_MS)
        u.n_in = u.n_out = 1
        u.cin[0], u.cout[0] = cin, sum(dil)
        acts.append((cin, 0)); acts.append((sum(dil), 0))
        u.in_act[0], u.out_act[0] = 1, 2
        for d, c in enumerate(dil):
            u.dil_ch[d] = c
            if c:
                u.w_off[d] = ar.add(f"w{d}", arrs[f"{tag}/sd/msconv.{d}.weight"])
        _bn(u, 0, ar, arrs, tag, "bn", "prelu.weight")
        eng = Engine(lib, [u], acts, B, H, W, device)
        eng.activation(1).copy_(torch.from_numpy(x))
        eng.refresh(ar.tensor(device))
        eng.forward(torch.zeros(B, 1, H, W, device=device))
        return [(eng.activation(2).cpu().numpy(), arrs[f"{tag}/y0"])]
    if kind == "cls":
        x = arrs[f"{tag}/x0"]
        B, cin, h, w = x.shape
        u = N.new_unit(N.UNIT_CLS)
        u.n_in = u.n_out = 1
        u.cin[0], u.cout[0] = cin, 1
        acts.append((cin, 1))
        u.in_act[0] = 1
        u.w_off[0] = ar.add("w", arrs[f"{tag}/w"])
        u.bias_off = ar.add("b", arrs[f"{tag}/b"])
        eng = Engine(lib, [u], acts, B, 2 * h, 2 * w, device)
        eng.activation(1).copy_(torch.from_numpy(x))
        eng.refresh(ar.tensor(device))
        y = eng.forward(torch.zeros(B, 1, 2 * h, 2 * w, device=device))
        return [(y.cpu().numpy(), arrs[f"{tag}/y0"])]
    raise ValueError(kind)


def check_g4(lib, device):
    meta = json.load(open(os.path.join(GOLD, "g4_ops_meta.json")))
    arrs = np.load(os.path.join(GOLD, "g4_ops.npz"))
    report = {}
    for tag, mt in meta.items():
        worst = 0.0
        for got, exp in run_g4_case(lib, device, tag, mt, arrs):
            assert got.shape == exp.shape, (tag, got.shape, exp.shape)
            scale = max(1.0, float(np.abs(exp).max()))
            err = float(np.abs(got - exp).max()) / scale
            assert err <= UNIT_TOL, f"{tag}: rel err {err:.3e}"
            worst = max(worst, err)
        report[tag] = worst
    return report


# ---------------------------------------------------------------------------------------------------
# train-mode forward (batch-stat BN, running-stat update, dynamic-weight-decay penalty)
# ---------------------------------------------------------------------------------------------------
def check_train_forward(lib, device, manifest, B=4, size=64, expandflop=2, seed=10):
    """Train-mode forward vs the oracle on the same seeded batch: logits, every BN's running stats after the
    step, num_batches_tracked and get_flops()."""
    m, sd = make_model(lib, manifest, device)
    m.train()
    x = torch.from_numpy(I.randn_batch(seed, B, size, size))
    m.set_batchsize(B)
    m.clear_flops()
    m.flops_hook(expandflop)
    y = m(x.to(device)).cpu()
    pen = float(m.get_flops())

    cfg = O.load_layer_config_json(manifest)
    sd_ref = {k: v.clone() for k, v in sd.items()}
    taps = {}
    with torch.no_grad():
        ref = O.csnet_forward(cfg, sd_ref, x, training=True, taps=taps)
    pen_ref = float(O.gap_penalty(sd_ref, taps, O.flop_weights(cfg, expandflop), B))
    err = (y - ref).abs().max().item()
    # Batch statistics make the comparison ill-conditioned: the shipped (pruned) checkpoint has channels whose batch
    # variance is ~0, where BN multiplies accumulation-order noise by 1/sqrt(eps) = 316 (tests/debug_train_taps.py
    # prints the per-unit growth).  Tolerance: 5e-5 of the logit range instead of the eval path's absolute 1e-4.
    tol = 5e-5 * max(1.0, ref.abs().max().item())
    assert err <= tol, f"train-mode logits: max-abs {err:.3e} > {tol:.3e}"
    assert abs(pen - pen_ref) <= 1e-5 * max(1.0, abs(pen_ref)) + 1e-7, (pen, pen_ref)
    got = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    worst = 0.0
    for k, v in sd_ref.items():
        if k.endswith("num_batches_tracked"):
            assert int(got[k]) == int(v), k
        elif k.endswith("running_mean") or k.endswith("running_var"):
            e = ((got[k] - v).abs() / (1.0 + v.abs())).max().item()
            assert e <= 1e-5, f"{k}: {e:.3e}"
            worst = max(worst, e)
        else:
            assert torch.equal(got[k], v), k      # forward must not touch learnable parameters
    # eval after train must pick the updated running statistics up again
    m.eval()
    y2 = m(x.to(device)).cpu()
    with torch.no_grad():
        ref2 = O.csnet_forward(cfg, sd_ref, x)
    assert (y2 - ref2).abs().max().item() <= TOL
    return err, pen, pen_ref, worst


def check_train_golden(lib, device, manifest, idx):
    """G5: penalty and BN running statistics after ONE train-mode forward of the reference itself
    (tests/golden/g5_g7_train_step.json; x = randn_batch(10, 4), 224x224)."""
    rec = json.load(open(os.path.join(GOLD, "g5_g7_train_step.json")))[idx]
    ef = 2 if rec["expandflop"] is None else rec["expandflop"]
    m, _ = make_model(lib, manifest, device)
    m.train()
    m.set_batchsize(4)
    m.clear_flops()
    m.flops_hook(ef)
    m(torch.from_numpy(I.randn_batch(10, 4)).to(device))
    pen = float(m.get_flops())
    assert abs(pen - rec["penalty"][0]) <= 1e-5 * max(1.0, abs(rec["penalty"][0])) + 1e-7, (pen, rec["penalty"][0])
    got = m.state_dict()
    for n, v in rec["bn_after_rank0"].items():
        s = float(got[n].double().sum())
        assert abs(s - v["sum"]) <= 1e-5 * max(abs(v["sum"]), 1.0), (n, s, v["sum"])
    return pen


# ---------------------------------------------------------------------------------------------------
# train step: hand-written backward kernels vs autograd through the oracle
# ---------------------------------------------------------------------------------------------------
def bce_and_grad(lib, y, t):
    """csn_bce_with_logits through the C ABI: (fp64 device scalar mean loss, dy)."""
    dy = torch.empty_like(y)
    loss = torch.zeros(1, dtype=torch.float64, device=y.device)
    stream = torch.cuda.current_stream(y.device).cuda_stream if y.is_cuda else 0
    N.check(lib, lib.csn_bce_with_logits(y.data_ptr(), t.data_ptr(), dy.data_ptr(), y.numel(), loss.data_ptr(), stream),
            "csn_bce_with_logits")
    return loss, dy


def grad_errors(model, flat, ref_grads):
    """Per-parameter relative L2 error of the flat gradient arena against a dict of reference gradients."""
    offs = model._arena.offsets
    out = {}
    for name, p in model.named_parameters():
        g = flat[offs[name]:offs[name] + p.numel()].view(p.shape).detach().cpu().double()
        r = ref_grads[name].double()
        out[name] = (float((g - r).norm()), float(r.norm()))
    return out


def check_train_step(lib, device, manifest, B=2, size=32, expandflop=1.0, flops_weight=3.0, seed=10, rel=2e-3):
    m, sd = make_model(lib, manifest, device)
    m.train()
    m.set_batchsize(B)
    m.clear_flops()
    m.flops_hook(expandflop)
    x = torch.from_numpy(I.randn_batch(seed, B, size, size))
    t = torch.from_numpy(I.binary_target(seed + 1, B, size, size))
    xd, td = x.to(device), t.to(device)
    y, pen = m._train_forward_raw(xd)
    loss, dy = bce_and_grad(m._lib or N.load(), y, td)
    flat = m._train_backward_raw(xd, dy, flops_weight / B)

    cfg = O.load_layer_config_json(manifest)
    kw = dict(expandflop=expandflop, flops_weight=flops_weight, batchsize=B, lr=0.0, wd=0.0)
    r = O.train_step(cfg, {k: v.clone() for k, v in sd.items()}, x, t, **kw)
    assert abs(float(loss) - r["loss_bce"]) <= 1e-5 * max(1.0, abs(r["loss_bce"])), (float(loss), r["loss_bce"])
    assert abs(float(pen) / B - r["penalty"]) <= 1e-5 * max(1.0, abs(r["penalty"])), (float(pen) / B, r["penalty"])
    # Gradients through 57 batch-normalised layers are ill-conditioned where the batch variance of a channel is ~0
    # (check_train_forward): the fp32 oracle itself is up to ~2e-2 away from an fp64 run of the same step on single
    # tensors.  Judge every tensor against the fp64 run, allowing the fp32 oracle's own deviation on that tensor.
    r64 = O.train_step(cfg, {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()},
                       x.double(), t.double(), **kw)
    errs = grad_errors(m, flat, r64["grads"])
    gmax = max(n for _, n in errs.values())
    bad, worst = {}, 0.0
    for k, (e, n) in errs.items():
        ref = float((r["grads"][k].double() - r64["grads"][k]).norm())
        lim = rel * n + 3.0 * ref + 1e-6 * gmax
        worst = max(worst, e / (n + 1e-6 * gmax))
        if e > lim:
            bad[k] = (e, n, ref)
    assert not bad, f"{len(bad)} of {len(errs)} gradients off, e.g. {list(bad.items())[:6]}"
    return worst, float(loss), float(pen) / B


def check_bn_bwd_fusion_bit_identical(lib, device, manifest, B=2, size=32, act_dtype="fp32", seed=23):
    """The one-pass depthwise kernels (input activation formed on load from the producer's z, dz formed on load in the
    backward kernel) against the stored scheme (CSN_BN_BWD_FUSE=0 switches both off: y written by bn_apply_gap_kernel, dz by
    bn_bwd_apply_kernel): the same arithmetic per element; what differs in fp32 is the summation order / precision of the
    per-tile partial sums (see below)."""
    flats = []
    for fuse in ("1", "0"):
        os.environ["CSN_BN_BWD_FUSE"] = fuse
        try:
            m, _ = make_model(lib, manifest, device)
            m.set_train_act_dtype(act_dtype)
            m.train(); m.set_batchsize(B); m.clear_flops(); m.flops_hook(1.0)
            x = torch.from_numpy(I.randn_batch(seed, B, size, size)).to(device)
            t = torch.from_numpy(I.binary_target(seed + 1, B, size, size)).to(device)
            y, pen = m._train_forward_raw(x)
            loss, dy = bce_and_grad(m._lib or N.load(), y, t)
            flats.append(m._train_backward_raw(x, dy, 3.0 / B).cpu().clone())
        finally:
            del os.environ["CSN_BN_BWD_FUSE"]
    a, b = flats
    if act_dtype == "fp32":
        # the one difference in fp32: the |GAP| table of a never-stored activation is summed per tile by its consumer instead
        # of per plane by bn_apply_gap_kernel (fp64 partial sums in another order -> the float table may differ in its last
        # bit, which reaches the BatchNorm weight gradients through the penalty term)
        # ... and the BatchNorm-backward sums a consumer takes for its producer are fp32 per lane (<= 64 terms) before the fp64
        # block / slab reduction, where bn_bwd_reduce_kernel is fp64 throughout: mean(dbn), mean(dbn * xhat) differ by ~1e-7
        # relative, which the 57 normalised layers in front amplify -- the sharp test is check_train_units_local
        rel = float((a.double() - b.double()).norm() / b.double().norm())
        assert rel <= 2e-2, rel
        return rel
    rel = float((a.double() - b.double()).norm() / b.double().norm())
    assert rel < 5e-2, rel
    return rel


def check_train_golden_step(lib, device, manifest, idx):
    """G5: gradients and parameters after ONE full train step of the reference itself (B=4, 224x224, seeds 10/11,
    FLOPS.WEIGHT 3, Adam lr 1e-4 / wd 5e-3 with the two parameter groups)."""
    from sod100k_amd.tools.train import FusedTrainer
    rec = json.load(open(os.path.join(GOLD, "g5_g7_train_step.json")))[idx]
    ef = 2 if rec["expandflop"] is None else rec["expandflop"]
    m, _ = make_model(lib, manifest, device)
    m.train()
    m.set_batchsize(4)
    m.clear_flops()
    m.flops_hook(ef)
    tr = FusedTrainer(m, lr=1e-4, weight_decay=5e-3, flops_weight=3.0, batchsize=4, lib=lib)
    x = torch.from_numpy(I.randn_batch(10, 4)).to(device)
    t = torch.from_numpy(I.binary_target(11, 4)).to(device)
    loss, pen = tr.step(x, t)
    assert abs(float(loss) - rec["bce"][0]) <= 1e-5, (float(loss), rec["bce"][0])
    assert abs(float(pen) - rec["penalty"][0]) <= 1e-5 * max(1.0, abs(rec["penalty"][0])), (float(pen), rec["penalty"][0])
    offs = m._arena.offsets
    gmax = max(rec["grad_l2"].values())
    shapes = {k: v.shape for k, v in m.state_dict().items()}
    errs = {}
    for n, v in rec["grad_l2"].items():
        g = float(tr.grad[offs[n]:offs[n] + int(np.prod(shapes[n]))].double().norm())
        errs[n] = abs(g - v) / (v + 1e-6 * gmax)
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    # gradient NORMS of all 419 tensors; the tail is the first block (57 batch-normalised layers of backward
    # through channels whose batch variance is ~0, see check_train_forward)
    assert top[0][1] <= 1e-2 and sum(e > 2e-3 for e in errs.values()) <= 4, top
    worst = top[0][1]
    print("largest grad-norm deviations:", [(k, f"{e:.1e}") for k, e in top])
    sd = m.state_dict()
    for n, v in rec["param_after"].items():
        s = float(sd[n].double().sum())
        # Adam's first step moves every element by lr * g / (|g| + eps') ~ lr * sign(g): an element whose gradient is
        # rounding noise may take the other sign, which shifts the sum by 2 * lr.  Allow 1 % of the elements to.
        flips = max(2.0, 0.01 * sd[n].numel())
        assert abs(s - v["sum"]) <= 2e-4 * flips + 1e-5 * abs(v["sum"]), (n, s, v["sum"])
    for n, v in rec["bn_after_rank0"].items():
        s = float(sd[n].double().sum())
        assert abs(s - v["sum"]) <= 1e-5 * max(abs(v["sum"]), 1.0), (n, s, v["sum"])
    return worst


def check_autograd_seam(lib, device, manifest, B=2, size=32):
    """`loss.backward(); optimizer.step()` over the autograd Function == the fused flat path."""
    from sod100k_amd.tools.train import reference_style_step
    m, _ = make_model(lib, manifest, device)
    m.train()
    m.set_batchsize(B)
    m.clear_flops()
    m.flops_hook(1.0)
    x = torch.from_numpy(I.randn_batch(3, B, size, size)).to(device)
    t = torch.from_numpy(I.binary_target(4, B, size, size)).to(device)
    m._ensure_arena()
    y, pen = m._train_forward_raw(x)
    _, dy = bce_and_grad(m._lib or N.load(), y, t)
    flat = m._train_backward_raw(x, dy, 3.0 / B).clone()
    # rewind the BN buffers the first forward advanced, then the reference-style step with lr 0
    m2, _ = make_model(lib, manifest, device)
    m2.train(); m2.set_batchsize(B); m2.clear_flops(); m2.flops_hook(1.0)
    m2._ensure_arena()
    opt = torch.optim.SGD(m2.parameters(), lr=0.0)
    reference_style_step(m2, opt, x, t, 3.0)
    # ... and with x.requires_grad the seam also returns autograd's x.grad (CSN_OPT_INPUT_GRAD: a plan with one more gradient buffer)
    m3, sd3 = make_model(lib, manifest, device)
    check_input_gradient_seam(m3, O.load_layer_config_json(manifest), sd3, x, t, flops_weight=3.0)
    offs = m2._arena.offsets
    for name, p in m2.named_parameters():
        assert p.grad is not None, name
        ref = flat[offs[name]:offs[name] + p.numel()].view(p.shape)
        # torch's BCE gradient differs from csn_bce_with_logits in the last bit: compare per tensor, not per element
        assert (p.grad - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-7, name     # fp32 sums of ~1e5 terms fed by a dy that differs in its last bit


def check_input_gradient_seam(m, cfg, sd, x, t, flops_weight=3.0):
    """SURVEY 8(b)'s `dx` of csn_backward: `model(x)` with x.requires_grad in train mode -> loss.backward() -> x.grad, against
    autograd through the oracle in fp64 with the fp32 oracle's own distance from it as the yardstick (fp32 storage; the gradient
    w.r.t. the image through ~60 batch-normalised layers in bf16 storage is noise in ANY implementation -- the bf16-emulating
    oracle sits 1.0-1.4 relative L2 from the fp32 one -- so bf16 is checked unit-locally: check_train_units_local(input_grad=True))."""
    B = x.shape[0]
    m.train(); m.set_batchsize(B); m.clear_flops()
    if flops_weight:
        m.flops_hook(1.0)
    xg = x.clone().requires_grad_(True)
    y = m(xg)
    loss = F.binary_cross_entropy_with_logits(y, t)
    if flops_weight:
        loss = loss + flops_weight * m.get_flops()
    loss.backward()
    m.clear_flops()
    assert xg.grad is not None and xg.grad.shape == x.shape and xg.grad.dtype == torch.float32
    assert not getattr(m, "_want_input_grad", False)          # plain steps go back to the plan without the extra buffer
    kw = dict(expandflop=1.0, flops_weight=flops_weight, batchsize=B, lr=0.0, wd=0.0, use_penalty=bool(flops_weight))
    refs = {}
    for dt in (torch.float32, torch.float64):
        xo = x.detach().cpu().to(dt).requires_grad_(True)
        O.train_step(cfg, {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}, xo, t.detach().cpu().to(dt), **kw)
        refs[dt] = xo.grad.double()
    n64 = float(refs[torch.float64].norm())
    mine = float((xg.grad.cpu().double() - refs[torch.float64]).norm()) / n64
    ref32 = float((refs[torch.float32] - refs[torch.float64]).norm()) / n64
    print(f"input gradient through the autograd seam: rel-L2 vs fp64 {mine:.2e} (fp32 oracle: {ref32:.2e})")
    assert n64 > 0 and mine <= max(1e-3, 3.0 * ref32), (mine, ref32)
    return mine, ref32


def check_pre_post(lib, device, manifest):
    """a13: device-side normalise/pack and sigmoid->uint8 either side of the forward, against numpy and G9."""
    from sod100k_amd import engine as E
    rng = np.random.default_rng(3)
    img = rng.random((2, 32, 48, 3), dtype=np.float32)
    got = E.normalize_nchw(lib, torch.from_numpy(img).to(device)).cpu().numpy()
    mean = np.array([0.485, 0.456, 0.406], np.float32); std = np.array([0.229, 0.224, 0.225], np.float32)
    ref = np.transpose((img - mean) / std, (0, 3, 1, 2))
    assert np.abs(got - ref).max() <= 1e-6
    m, _ = make_model(lib, manifest, device)
    y = m(torch.from_numpy(I.image_like()).to(device))
    u8 = E.saliency_u8(lib, y)[0, 0].cpu().numpy()
    g9 = np.load(os.path.join(GOLD, "g9_uint8_x2_image.npy"))
    assert u8.shape == g9.shape and u8.dtype == np.uint8
    assert np.abs(u8.astype(int) - g9.astype(int)).max() <= 1 and (u8 != g9).mean() < 5e-3


def check_std_conv_network(lib, device, random_state):
    """``build_model()`` with its default ``basic_split=[1]``: every ILBlock's first conv is the Conv2dX100 ``std_conv``
    of csnet.py:751-754 (x100 weights; the stride-2 blocks are REAL strided 3x3 convolutions, no avg-pool).  Eval forward,
    train-mode forward and every gradient against the oracle (penalty off: the reference's hook is ill-defined on the
    tensor output of a std_conv module)."""
    from sod100k_amd.model import csnet as M
    m = M.build_model(save_path="/tmp")
    assert m.stage1[0].conv1x1.std_conv and m.stage2[0].conv1x1.stride == 2
    sd = random_state(m, 2)
    for k in sd:                                   # x100 convolutions: keep the activations O(1)
        if k.endswith("conv1x1.conv.weight"):
            sd[k] = sd[k] * 0.01
    m.load_state_dict(sd)
    m = m.to(device)
    if device.type == "cpu":
        m._lib = lib
    cfg = O.init_layers(20, [1])
    x = torch.from_numpy(I.randn_batch(8, 2, 64, 64))
    t = torch.from_numpy(I.binary_target(9, 2, 64, 64))
    m.eval()
    with torch.no_grad():
        ref = O.csnet_forward(cfg, {k: v.clone() for k, v in sd.items()}, x)
    y = m(x.to(device)).cpu()
    assert (y - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    m.train(); m.set_batchsize(2); m.clear_flops()
    xd = x.to(device)
    yt, _ = m._train_forward_raw(xd)
    loss, dy = bce_and_grad(lib, yt, t.to(device))
    flat = m._train_backward_raw(xd, dy, 0.0)
    kw = dict(expandflop=1.0, flops_weight=0.0, batchsize=2, lr=0.0, wd=0.0)
    r32 = O.train_step(cfg, {k: v.clone() for k, v in sd.items()}, x, t, **kw)
    r64 = O.train_step(cfg, {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()},
                       x.double(), t.double(), **kw)
    assert abs(float(loss) - r64["loss_bce"]) <= 1e-5
    mine = np.array([e / (n + 1e-12) for e, n in grad_errors(m, flat, r64["grads"]).values()])
    ref = np.array([float((r32["grads"][k].double() - g).norm() / (g.norm() + 1e-12)) for k, g in r64["grads"].items()])
    assert np.median(mine) <= 2 * np.median(ref) + 1e-6 and mine.max() <= 3 * ref.max() + 1e-5, (
        np.median(mine), np.median(ref), mine.max(), ref.max())
    # x.grad: here the first unit is a Conv2dX100 std_conv (its input gradient = the x100 transposed taps on the image)
    check_input_gradient_seam(m, cfg, sd, xd, t.to(device), flops_weight=0.0)


def well_conditioned_state(manifest, seed=0):
    """The shipped architecture with a WELL-CONDITIONED synthetic state: BN gamma in [0.5, 1.5], running variance in
    [0.5, 1.5], conv weights scaled so that every unit's output stays O(1), PReLU slopes in [0.1, 0.4].  The shipped
    checkpoint has ~50 channels with |gamma| or variance ~1e-20 (decayed by the dynamic weight decay) that make single
    gradients jump on the last bit of a BN output; on this state the tight bound must hold."""
    sd = O.load_weights(manifest)
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            out[k] = v.clone()
        elif (".bns." in k or ".bn." in k) and k.endswith("weight"):
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif (".bns." in k or ".bn." in k) and k.endswith("bias"):
            out[k] = 0.2 * torch.randn(v.shape, generator=g)
        elif k.endswith("running_var"):
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif "prelu" in k:
            out[k] = 0.1 + 0.3 * torch.rand(v.shape, generator=g)
        elif k.endswith("weight") and v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            scale = (1.0 / fan_in) ** 0.5
            if v.shape[1] == 1 or ".msconv." in k:          # Conv2dX100 units: the kernel multiplies by 100
                scale /= 100.0
            out[k] = scale * torch.randn(v.shape, generator=g)
        else:
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
    return out


def _train_step_vs_fp64(lib, device, manifest, B, size, seed):
    """One train step on the well-conditioned state: (relative L2 of all gradients vs an fp64 run of the oracle, the fp32
    oracle's own distance from that run, number of tensors further away than twice the fp32 oracle + 1e-4)."""
    sd = well_conditioned_state(manifest)
    m = M.build_model(predefine=manifest)
    m.load_state_dict(sd)
    m = m.to(device)
    m._lib = lib if device.type == "cpu" else None
    m.train(); m.set_batchsize(B); m.clear_flops(); m.flops_hook(1.0)
    x = torch.from_numpy(I.randn_batch(seed, B, size, size))
    t = torch.from_numpy(I.binary_target(seed + 1, B, size, size))
    xd, td = x.to(device), t.to(device)
    y, pen = m._train_forward_raw(xd)
    loss, dy = bce_and_grad(m._lib or N.load(), y, td)
    flat = m._train_backward_raw(xd, dy, 3.0 / B)
    cfg = O.load_layer_config_json(manifest)
    kw = dict(expandflop=1.0, flops_weight=3.0, batchsize=B, lr=0.0, wd=0.0)
    r = O.train_step(cfg, {k: v.clone() for k, v in sd.items()}, x, t, **kw)
    r64 = O.train_step(cfg, {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, x.double(),
                       t.double(), **kw)
    assert abs(float(loss) - r64["loss_bce"]) <= 1e-5 * max(1.0, abs(r64["loss_bce"])), (float(loss), r64["loss_bce"])
    assert abs(float(pen) / B - r64["penalty"]) <= 1e-5 * max(1.0, abs(r64["penalty"]))
    errs = grad_errors(m, flat, r64["grads"])
    gmax = max(n for _, n in errs.values())
    num = den = ref2 = 0.0
    bad = 0
    for k, (e, n) in errs.items():
        d32 = float((r["grads"][k].double() - r64["grads"][k]).norm())
        num += e * e; den += n * n; ref2 += d32 * d32
        if e > 2.0 * d32 + 1e-4 * max(n, 1e-3 * gmax):
            bad += 1
    return (num / den) ** 0.5, (ref2 / den) ** 0.5, bad


def check_train_step_well_conditioned(lib, device, manifest, B=2, size=64, seeds=(31, 41, 51, 61, 71, 81, 91, 101),
                                      local_seeds=None):
    """One train step on the well-conditioned state, judged against an fp64 run of the oracle with the fp32 ORACLE's own
    distance from that run as the yardstick (57 batch-normalised layers amplify fp32 rounding whatever the conditioning of
    the parameters, so an absolute bound is unreachable for any fp32 implementation).
    The distance is BIMODAL over inputs for any fp32 implementation: ~1e-4 when nothing discrete happens, ~1e-2 when a
    1e-7 difference flips one max-pool argmax or one PReLU branch somewhere (a finite event in the backward pass).  Measured
    at size 32, relative L2 over all gradients, kernels / fp32 oracle per seed -- round-2 kernels: 7.0e-5/4.9e-5, 7.3e-3/7.3e-3,
    2.4e-4/4.5e-5, 1.7e-3/1.7e-3, 3.8e-5/5.4e-3, 1.0e-2/1.1e-4, 7.2e-5/6.2e-5, 9.6e-5/5.2e-5; round-3 kernels: 6.7e-3/4.9e-5,
    7.6e-3/7.3e-3, 2.1e-4/4.5e-5, 2.0e-3/1.7e-3, 4.2e-5/5.4e-3, 7.8e-5/1.1e-4, 8.5e-5/6.2e-5, 8.3e-3/5.2e-5: each implementation
    (the oracle included) has its events on three or four of the eight seeds, on different ones, while every unit agrees
    with the oracle to 4e-7 on the tensors around it (check_train_units_local, the sharp test).  Hence several seeds and
    the QUIET seeds as the yardstick: on at least three of the eight seeds the kernels must be within 1.5x of the fp32
    oracle's own distance AND have fewer than 3 % of the tensors further from fp64 than twice the fp32 oracle (+1e-4); on
    every seed the distance stays below 0.05 (an event, not a wrong gradient) and every unit passes the unit-local check.  A median would not do: with events on half
    of the seeds it sits between the two modes (round-3 kernels at size 64: sorted ratios 0.34, 0.99, 1.00, 1.13, 1.68, 3.2,
    7.3, 13; with split-K in the 3x3 kernel at size 32: 0.90, 0.99, 1.01, 1.04, 3.7, 5.1, 100, 157)."""
    res = [_train_step_vs_fp64(lib, device, manifest, B, size, s_) for s_ in seeds]
    print("well-conditioned step, per seed (kernels vs fp64, fp32 oracle vs fp64, tensors past 2x):",
          ", ".join(f"{r[0]:.1e}/{r[1]:.1e}/{r[2]}" for r in res))
    quiet = sum(1 for r in res if r[0] <= 1.5 * r[1] + 1e-5 and r[2] <= 12)          # 419 gradient tensors
    assert quiet >= min(3, len(res)), f"kernels further from fp64 than the fp32 oracle on almost every seed: {res}"
    # an event, not a wrong gradient: round-4 kernels at size 64 reach 2.2e-2 on their worst seed (the fp32 oracle 1.8e-2 on its own)
    assert max(r[0] for r in res) <= 0.05, res
    # The sharp, deterministic gate on EVERY seed (a <10 % error in a rarely taken path -- pwq, virt_cons, the fused
    # BN apply of dw3x3_bwd -- must not hide behind the "event" allowance above): each unit's forward, dz, dx and parameter
    # gradients re-computed by the oracle from the tensors the device itself produced around that unit (no amplification through
    # depth, the oracle following the device's argmax / PReLU branch only inside its 1e-5 band): 2e-5 / 2e-4 relative L2.
    for s_ in (seeds if local_seeds is None else local_seeds):
        check_train_units_local(lib, device, manifest, B=B, size=min(size, 32), act_dtype="fp32", state="well", seed=s_)
    rel = sorted(r[0] for r in res)
    rel32 = sorted(r[1] for r in res)
    return 0.5 * (rel[(len(rel) - 1) // 2] + rel[len(rel) // 2]), 0.5 * (rel32[(len(rel32) - 1) // 2] + rel32[len(rel32) // 2])


def check_train_step_bf16(lib, device, manifest, B=2, size=32, state="well"):
    """BASELINE config 3's dtype: one train step with bfloat16 activation storage (CSN_OPT_TRAIN_BF16).  Judged against
    (a) the oracle with the SAME storage points rounded through bf16 (oracle.bf16_activations) and (b) the plain fp32
    oracle, with the emulated-bf16 oracle's own distance from fp32 as the yardstick: the kernels may not be further from
    the fp32 step than twice what bf16 storage itself costs.  SURVEY 8(c): loss / gradient norms at ~1e-2 relative."""
    sd = well_conditioned_state(manifest) if state == "well" else O.load_weights(manifest)
    m = M.build_model(predefine=manifest)
    m.load_state_dict(sd)
    m = m.to(device)
    m._lib = lib if device.type == "cpu" else None
    m.set_train_act_dtype("bf16")
    m.train(); m.set_batchsize(B); m.clear_flops(); m.flops_hook(1.0)
    x = torch.from_numpy(I.randn_batch(41, B, size, size))
    t = torch.from_numpy(I.binary_target(42, B, size, size))
    xd, td = x.to(device), t.to(device)
    y, pen = m._train_forward_raw(xd)
    loss, dy = bce_and_grad(m._lib or N.load(), y, td)
    flat = m._train_backward_raw(xd, dy, 3.0 / B)
    cfg = O.load_layer_config_json(manifest)
    kw = dict(expandflop=1.0, flops_weight=3.0, batchsize=B, lr=0.0, wd=0.0)
    r32 = O.train_step(cfg, {k: v.clone() for k, v in sd.items()}, x, t, **kw)
    sd16 = {k: v.clone() for k, v in sd.items()}
    r16 = O.train_step(cfg, sd16, x, t, act_dtype="bf16", **kw)
    # logits: bf16 storage through ~60 layers
    ref_out = float((r16["out"] - r32["out"]).abs().max())
    got_out = float((y.cpu() - r32["out"]).abs().max())
    assert got_out <= 2.0 * ref_out + 1e-3, (got_out, ref_out)
    assert float((y.cpu() - r16["out"]).abs().max()) <= 2.0 * ref_out + 1e-3
    assert abs(float(loss) - r16["loss_bce"]) <= 1e-2 * max(1.0, abs(r16["loss_bce"])), (float(loss), r16["loss_bce"])
    assert abs(float(loss) - r32["loss_bce"]) <= 2e-2 * max(1.0, abs(r32["loss_bce"])), (float(loss), r32["loss_bce"])
    assert abs(float(pen) / B - r16["penalty"]) <= 5e-3 * max(1.0, abs(r16["penalty"])), (float(pen) / B, r16["penalty"])
    assert abs(float(pen) / B - r32["penalty"]) <= 1e-2 * max(1.0, abs(r32["penalty"])), (float(pen) / B, r32["penalty"])
    # BN running statistics (taken from the stored bf16 z)
    got = m.state_dict()
    worst_rs = 0.0
    for k, v in sd16.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            worst_rs = max(worst_rs, ((got[k].cpu() - v).abs() / (1 + v.abs())).max().item())
    # deep layers: 8 samples per channel at this size, each rounded to 8 bits; the worst channel moves with the summation
    # order inside the contraction kernels (round 2: 1.6e-2, round 3: 2.2e-2)
    assert worst_rs <= 3e-2, worst_rs
    # Gradients are NOT compared at whole-step level: through the 60 batch-normalised layers a storage rounding is amplified
    # by ~5e4 (the fp32 oracle is 3e-3 away from fp64; two bf16 runs that differ in one rounding point are O(1) apart in the
    # early stages).  check_train_units_local judges every unit's backward on the device's own upstream gradients instead.
    gn = float(flat.double().norm())
    assert np.isfinite(gn) and gn > 0
    return dict(running_stats=worst_rs, logits=got_out, logits_ref=ref_out, loss=float(loss), loss32=r32["loss_bce"],
                loss16=r16["loss_bce"], grad_norm=gn)


# ---------------------------------------------------------------------------------------------------
# unit-local checks of the train step: every unit's forward and backward judged on the tensors the device itself
# produced for its inputs / upstream gradients, so that errors do not travel through the 60 batch-normalised layers
# (whole-step comparisons amplify a storage rounding by ~5e4 -- the fp32 oracle is 3e-3 from fp64, bf16 storage O(1))
# ---------------------------------------------------------------------------------------------------
def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def off_centre_state(manifest, seed=0, offset=50.0):
    """The well-conditioned state with BADLY CENTRED depthwise channels: in every ILBlock of stages 1-3 the first four
    channels of both branches get a BatchNorm bias of `offset` in the unit in front (conv1x1 -> conv3x3_1, conv3x3_1 -> conv3x3_2:
    their activation is offset + O(1)) and nine equal positive taps in the depthwise unit behind it, so that unit's raw output z has
    |batch mean| = 7-11 x its standard deviation (measured; a larger offset does not raise it: the zero padding's border pixels see
    four or six taps of nine and carry the variance).  The fused depthwise backward forms dz = g sel - (B z + A) with the mean folded
    into A: B z and A cancel to ~10 % of their size there -- the case the re-association is worst at on this network."""
    sd = well_conditioned_state(manifest, seed)
    for k in list(sd):
        for a, b in ((".conv1x1.bns.", ".conv3x3_1.convs."), (".conv3x3_1.bns.", ".conv3x3_2.convs.")):
            if a in k and k.endswith(".bias") and k.split(".")[0] in ("stage1", "stage2", "stage3"):
                j = k.split(a)[1].split(".")[0]
                wk = k.split(a)[0] + b + j + ".weight"
                if wk in sd and sd[k].numel() >= 4:
                    sd[k][:4] = offset
                    sd[wk][:4] = 0.01 / 3.0     # (x100 in the kernel: nine taps of 1/3)
    return sd


def check_train_units_local(lib, device, manifest, B=2, size=64, act_dtype="bf16", state="shipped", flops_weight=3.0,
                            tol_fwd=None, tol_bwd=None, seed=51, net=None, input_grad=False):
    """Returns the worst relative L2 deviations {z, act, dz, dx, dparam} over all units.  net = (model, layer_config, state_dict):
    a prebuilt network (e.g. the pruned one) instead of the manifest's.  input_grad: the plan also forms the gradient w.r.t. the
    image batch (CSN_OPT_INPUT_GRAD; autograd's x.grad), checked as the first unit's dx."""
    global _LOCAL_NET, _LOCAL_INPUT_GRAD
    _LOCAL_NET = net
    _LOCAL_INPUT_GRAD = bool(input_grad)
    import contextlib
    bf16 = act_dtype == "bf16"
    # the depthwise backward forms dz on load and skips the BatchNorm backward's apply pass; CSN_DEBUG_DZ (read at plan creation)
    # lets that pass run AFTER the fused kernel so that the dz probes below exist -- the kernel under test is the product's
    os.environ["CSN_DEBUG_DZ"] = "1"
    try:
        return _check_train_units_local(lib, device, manifest, B, size, act_dtype, state, flops_weight, tol_fwd, tol_bwd, seed)
    finally:
        del os.environ["CSN_DEBUG_DZ"]


_LOCAL_NET = None
_LOCAL_INPUT_GRAD = False


def train_backward_probes(lib, device, manifest, B, size, act_dtype, env, seed=51, state="shipped", flops_weight=3.0):
    """One train-mode forward + backward with the switches of ``env`` set while the plan is created; returns the flat gradient and
    every stored input gradient {(activation, consumer slot): tensor} -- for A/B checks between two forms of one kernel."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        sd = well_conditioned_state(manifest) if state == "well" else O.load_weights(manifest)
        m = M.build_model(predefine=manifest)
        m.load_state_dict(sd)
        m = m.to(device)
        m._lib = lib if device.type == "cpu" else None
        m.set_train_act_dtype(act_dtype)
        m.train(); m.set_batchsize(B); m.clear_flops(); m.flops_hook(1.0)
        hw = size if isinstance(size, tuple) else (size, size)
        xd = torch.from_numpy(I.randn_batch(seed, B, hw[0], hw[1])).to(device)
        td = torch.from_numpy(I.binary_target(seed + 1, B, hw[0], hw[1])).to(device)
        y, pen = m._train_forward_raw(xd)
        eng = m.engine_for(xd, train=True)
        units, acts, names = m.describe(m._arena.offsets)
        loss, dy = bce_and_grad(m._lib or N.load(), y, td)
        flat = m._train_backward_raw(xd, dy, flops_weight / B).cpu()
        G = {}
        for a in range(1, len(acts)):
            for s_ in range(eng.n_consumers(a)):
                G[(a, s_)] = eng.train_probe(a, f"grad{s_}").cpu()
        return flat, G
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def slim_network(manifest, tmp_path, thres=0.01):
    """(model, layer_config, state_dict) of the prune-and-finetune result of the shipped weights (zero-channel branches / dilations)."""
    m = M.build_model(predefine=manifest)
    m.load_state_dict(O.load_weights(manifest))
    new_cfg, mask = M.finetune_model(m, save_path=str(tmp_path), base_layer_config=M.load_layer_config(manifest), thres=thres)
    slim = M.build_model_with_weight(new_cfg, m, mask)
    return slim, new_cfg, {k: v.clone() for k, v in slim.state_dict().items()}


def _check_train_units_local(lib, device, manifest, B, size, act_dtype, state, flops_weight, tol_fwd, tol_bwd, seed):
    import contextlib
    bf16 = act_dtype == "bf16"
    # bf16: one rounding of the output (2^-9 rms) + rare flips of rounded inputs; fp32: summation order only
    tol_fwd = tol_fwd if tol_fwd is not None else (2e-3 if bf16 else 2e-5)
    tol_bwd = tol_bwd if tol_bwd is not None else (3e-2 if bf16 else 2e-4)
    if _LOCAL_NET is not None:
        m, net_cfg, sd = _LOCAL_NET
    else:
        sd = (well_conditioned_state(manifest) if state == "well" else off_centre_state(manifest) if state == "offcentre"
              else O.load_weights(manifest))
        m = M.build_model(predefine=manifest)
        m.load_state_dict(sd)
        net_cfg = None
    m = m.to(device)
    m._lib = lib if device.type == "cpu" else None
    m.set_train_act_dtype(act_dtype)
    m.train(); m.set_batchsize(B); m.clear_flops(); m.flops_hook(1.0)
    hw = size if isinstance(size, tuple) else (size, size)   # (height, width)
    x = torch.from_numpy(I.randn_batch(seed, B, hw[0], hw[1]))
    t = torch.from_numpy(I.binary_target(seed + 1, B, hw[0], hw[1]))
    xd, td = x.to(device), t.to(device)
    m._want_input_grad = _LOCAL_INPUT_GRAD        # (the autograd seam sets this when x.requires_grad)
    y, pen = m._train_forward_raw(xd)
    eng = m.engine_for(xd, train=True)
    assert eng.input_grad == _LOCAL_INPUT_GRAD
    units, acts, names = m.describe(m._arena.offsets)
    n_acts = len(acts)
    A = {0: (eng.train_probe(0, "act").cpu() if bf16 else x.clone())}
    Z = {}
    for a in range(1, n_acts):
        A[a] = eng.train_probe(a, "act").cpu()
        Z[a] = eng.train_probe(a, "z").cpu()
    loss, dy = bce_and_grad(m._lib or N.load(), y, td)
    flat = m._train_backward_raw(xd, dy, flops_weight / B).cpu()
    DZ = {a: eng.train_probe(a, "dz").cpu() for a in range(1, n_acts)}
    G = {}
    for a in range(1, n_acts):
        for s in range(eng.n_consumers(a)):
            G[(a, s)] = eng.train_probe(a, f"grad{s}").cpu()
    if _LOCAL_INPUT_GRAD:
        assert eng.n_consumers(0) == 1
        G[(0, 0)] = eng.train_probe(0, "grad0").cpu()
    cfg = net_cfg if net_cfg is not None else O.load_layer_config_json(manifest)
    blocks = {b["name"]: b for b in O.block_table(cfg)}
    cfg3 = cfg[len(blocks):len(blocks) + 3]
    offs = m._arena.offsets
    pshape = {k: p.shape for k, p in m.named_parameters()}
    flop_tab = m._flop_tab
    worst = dict(z=0.0, act=0.0, dz=0.0, dx=0.0, dparam=0.0)
    bad = []

    def note(kind, name, got, ref, tol, floor=0.0):
        # relative L2, with an absolute floor for tensors that are (nearly) zero
        got, ref = got.detach(), ref.detach()
        e = float((got.double() - ref.double()).norm())
        n = float(ref.double().norm())
        r = e / (n + floor + 1e-30)
        worst[kind] = max(worst[kind], r)
        if r > tol:
            bad.append((kind, name, r, n))

    ctx = O.bf16_activations() if bf16 else contextlib.nullcontext()
    if bf16 and not O.DW_IN_STORED:
        # An activation whose only consumer is a depthwise unit is never stored: the consumer forms it on load from the stored z
        # in fp32.  The "act" probe (CSN_DEBUG_DZ materialises it) went through a bf16 store; what the consumer saw is this:
        for ui, (u, name) in enumerate(zip(units, names)):
            if not (name.endswith(".conv1x1") and name[:-len(".conv1x1")] in blocks) and not name.endswith(".conv3x3_1"):
                continue
            for j in range(int(u.n_out)):
                if u.cout[j] == 0:
                    continue
                a = int(u.out_act[j])
                z = Z[a].float()
                yb = F.batch_norm(z, None, None, sd[f"{name}.bns.{j}.weight"], sd[f"{name}.bns.{j}.bias"], True, 0.0, 1e-5)
                A[a] = F.prelu(yb, sd[f"{name}.prelus.{j}.weight"])
    kink_units = []
    for ui, (u, name) in enumerate(zip(units, names)):
        n_in, n_out = int(u.n_in), int(u.n_out)
        pkeys = [k for k in pshape if k.startswith(name + ".")]
        outs = [j for j in range(n_out) if u.cout[j] > 0] if name != "cls_layer" else [0]

        def local(flips):
            """The unit recomputed by the oracle from the device's own inputs / upstream gradients; ``flips``: PReLU elements
            (per BatchNorm call) that take the other branch -- see O.PRELU_FLIP."""
            # fp32 storage: the reference in fp64 (round 4) -- an fp32 reference has its own summation noise, which on heavily
            # cancelling sums (a PReLU slope's gradient on the shipped checkpoint: +-1 terms, result 0.03) is as large as the bound
            # and made the verdict depend on the last bits of the upstream tensors; against fp64 the deviation IS the device's error
            # (not for the stride-2 units: their max-pool takes avg-pooled values, whose fp64 / fp32 roundings break ties differently
            # -- a discrete event, 6e-2 on dx of stage4.0 -- so those keep the fp32 reference)
            s2 = name.endswith(".conv1x1") and name[:-len(".conv1x1")] in blocks and blocks[name[:-len(".conv1x1")]]["stride"] == 2
            rdt = torch.float32 if (bf16 or s2) else torch.float64
            xs = []
            for i in range(n_in):
                if u.cin[i] > 0:
                    xs.append(A[int(u.in_act[i])].to(rdt).clone().requires_grad_(int(u.in_act[i]) > 0 or _LOCAL_INPUT_GRAD))
                else:
                    xs.append(None)
            loc = {k: (v.to(rdt) if v.is_floating_point() else v).clone() for k, v in sd.items() if k.startswith(name + ".")}
            for k in pkeys:
                loc[k].requires_grad_(True)
            O.Z_CAPTURE = []
            O.PRELU_Y = []
            O.PRELU_FLIP = flips
            try:
                with ctx:
                    if name == "cls_layer":
                        out = O._st(F.conv2d(xs[0], loc["cls_layer.weight"], loc["cls_layer.bias"]))
                        ys = [F.interpolate(out, x.shape[2:], mode="bilinear", align_corners=False)]
                    elif name.endswith(".conv1x1"):
                        blk = blocks[name[:-len(".conv1x1")]]
                        a_in, _ = O._alphas(blk["inlist"]); a_out, _ = O._alphas(blk["outlist"])
                        k = 3 if (blk["first"] or blk["stride"] == 2) else 1
                        ys = O.goct_cbr(xs, loc, name, a_in, a_out, k, blk["stride"], True, y_stored=O.DW_IN_STORED)
                    elif ".conv3x3_" in name:
                        ys = O.simplified_cbr(xs, loc, name, True, y_stored=O.DW_IN_STORED or name.endswith(".conv3x3_2"))
                    elif name == "oct_fuse.fuse":
                        ys = O.goct_cbr(xs, loc, name, O._alphas(cfg3[0][0])[0], O._alphas(cfg3[1][0])[0], 1, 1, True)
                    elif name.startswith("oct_fuse.ms.convs."):
                        ys = [O.ms_block(xs[0], loc, name, cfg3[1][2][int(name.rsplit(".", 1)[1])], True)]
                    elif name == "oct_fuse.fuse1x1":
                        ys = O.goct_cbr(xs, loc, name, O._alphas(cfg3[1][1])[0], [1], 1, 1, True)
                    else:
                        raise AssertionError(f"unit {name}?")
                    if isinstance(ys, torch.Tensor):
                        ys = [ys]
                    zs = list(O.Z_CAPTURE)
                    pre = list(O.PRELU_Y)
                    for z in zs:
                        z.retain_grad()
                    live = [yy for yy in ys if yy is not None]
                    assert len(live) == len(outs), (name, len(live), len(outs))
                    total = None
                    for q, j in enumerate(outs):
                        if name == "cls_layer":
                            up = dy.cpu().to(rdt)
                        else:
                            a = int(u.out_act[j])
                            up = (G[(a, 0)] + (G[(a, 1)] if (a, 1) in G else 0.0)).to(rdt)
                            w = flop_tab[ui * N.MAX_BRANCH + j]
                            if w != 0.0:   # Oct_bn_hook (csnet.py:391-410): 0.5 w sum |mean_hw y| gamma^2, y detached; /batchsize
                                gam = loc[[k for k in pkeys if k.endswith(f"bns.{j}.weight")][0]]
                                gap = live[q].detach().mean(dim=(2, 3)).abs().sum(0)
                                term = (flops_weight / B) * 0.5 * w * (gap * gam * gam).sum()
                                total = term if total is None else total + term
                        term = (live[q] * up).sum()
                        total = term if total is None else total + term
                    total.backward()
            finally:
                O.Z_CAPTURE = None
                O.PRELU_Y = None
                O.PRELU_FLIP = None
            return xs, loc, zs, live, pre

        xs, loc, zs, live, pre = local(None)
        if name != "cls_layer" and not bf16:
            # PReLU kink: an element whose pre-activation is zero to within rounding may take either branch on the device.
            # Where the device's dz says it took the other one, the oracle follows (only inside that band).
            flips = {}
            for q, j in enumerate(outs):
                d = (DZ[int(u.out_act[j])].double() - zs[q].grad.double()).abs()
                band = pre[q].abs() <= 1e-5 * float(pre[q].abs().max())
                m = band & (d > 20 * tol_bwd * float(zs[q].grad.double().pow(2).mean().sqrt()))
                if bool(m.any()):
                    flips[q] = m
            if flips:
                kink_units.append((name, {q: int(m.sum()) for q, m in flips.items()}))
                xs, loc, zs, live, pre = local(flips)
        for q, j in enumerate(outs):
            if name == "cls_layer":
                note("act", name, y.cpu(), live[q], tol_fwd)
            else:
                a = int(u.out_act[j])
                note("z", f"{name}[{j}]", Z[a], zs[q].detach(), tol_fwd)
                note("act", f"{name}[{j}]", A[a], live[q].detach(), tol_fwd)
        gmax = max(float(loc[k].grad.norm()) if loc[k].grad is not None else 0.0 for k in pkeys)
        for k in pkeys:
            g = flat[offs[k]:offs[k] + int(np.prod(pshape[k]))].view(pshape[k])
            ref = loc[k].grad if loc[k].grad is not None else torch.zeros_like(loc[k])
            note("dparam", k, g, ref, tol_bwd, floor=1e-3 * gmax)
        if name != "cls_layer":
            for q, j in enumerate(outs):
                note("dz", f"{name}[{j}]", DZ[int(u.out_act[j])], zs[q].grad, tol_bwd)
        for i in range(n_in):
            a = int(u.in_act[i])
            if u.cin[i] > 0 and (a > 0 or _LOCAL_INPUT_GRAD):
                note("dx", f"{name}<-{i}", G[(a, eng.unit_in_slot(ui, i))], xs[i].grad, tol_bwd)
    assert not bad, f"{len(bad)} unit-local deviations over tolerance, worst first: {sorted(bad, key=lambda b: -b[2])[:8]}"
    assert sum(sum(v.values()) for _, v in kink_units) <= 16, f"too many PReLU kink elements re-branched: {kink_units}"
    if kink_units:
        worst["kink_elements"] = kink_units
    return worst


# ------------------------------------------------------------------------------------------------------------------
# Paths the shipped x2 configuration never takes: the un-pruned training network and the pruned slim one
# ------------------------------------------------------------------------------------------------------------------
def random_state(m, seed):
    """O(1) activations for a freshly built network (its initialisers make every unit's output ~1e2)."""
    g = torch.Generator().manual_seed(seed)
    sd = m.state_dict()
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            out[k] = v.clone()
        elif k.endswith("running_var"):
            out[k] = torch.rand(v.shape, generator=g) + 0.5
        elif k.endswith("running_mean") or k.endswith("bias"):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif "bn" in k and k.endswith("weight"):
            out[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif "prelu" in k:
            out[k] = 0.25 + 0.05 * torch.randn(v.shape, generator=g)
        else:
            fan = max(1, int(np.prod(v.shape[1:])))
            out[k] = torch.randn(v.shape, generator=g) * (0.5 / fan ** 0.5) * (0.01 if "convs" in k or "msconv" in k else 1.0)
    return out


def unpruned_network(expand, width, seed):
    """(model, layer_config, state_dict) of the un-pruned `expand` network the reference's recipe trains
    (csnet-L-x2_train.yml:9-18: basic_split [0.5, 0.5], expand 2.0 -> width 40, 788,631 parameters) with an O(1) random state:
    the `net` argument of check_train_units_local -- the only sharp gate for the wide-channel (M-group / row-chunk) paths of the
    bf16 matrix kernels (wgrad_bf16, pwq16, c3q16), VERDICT r4 weak #1."""
    m = M.build_model(basic_split=[0.5, 0.5], expand=expand, save_path="/tmp")
    sd = random_state(m, seed)
    m.load_state_dict(sd)
    return m, O.init_layers(width, [0.5, 0.5]), {k: v.clone() for k, v in sd.items()}


def check_unpruned(lib, device, expand, width, B=2, size=32, seed=4, train=True, act_dtype="fp32"):
    """basic_split [0.5, 0.5] at `expand` (csnet-L-x2_train.yml:9-18, init_layers csnet.py:414-518): eval forward and one
    train step (fp64 run of the oracle as the truth, the fp32 oracle's own deviation as the yardstick: random state)."""
    m = M.build_model(basic_split=[0.5, 0.5], expand=expand, save_path="/tmp")
    sd = random_state(m, seed)
    m.load_state_dict(sd)
    m = m.to(device)
    if device.type == "cpu":
        m._lib = lib
    cfg = O.init_layers(width, [0.5, 0.5])
    x = torch.from_numpy(I.randn_batch(5, B, size, size))
    t = torch.from_numpy(I.binary_target(6, B, size, size))
    m.eval()
    with torch.no_grad():
        ref = O.csnet_forward(cfg, {k: v.clone() for k, v in sd.items()}, x)
    y = m(x.to(device)).cpu()
    err = (y - ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, ref.abs().max().item()), err
    if not train:
        return err, None, None
    m.train(); m.set_batchsize(B); m.clear_flops(); m.flops_hook(1.0)
    if act_dtype == "bf16":
        m.set_train_act_dtype("bf16")
    xd = x.to(device)
    yt, pen = m._train_forward_raw(xd)
    loss, dy = bce_and_grad(lib, yt, t.to(device))
    flat = m._train_backward_raw(xd, dy, 3.0 / B)
    kw = dict(expandflop=1.0, flops_weight=3.0, batchsize=B, lr=0.0, wd=0.0)
    r32 = O.train_step(cfg, {k: v.clone() for k, v in sd.items()}, x, t, **kw)
    r64 = O.train_step(cfg, {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()},
                       x.double(), t.double(), **kw)
    mine = np.array([e / (n + 1e-12) for e, n in grad_errors(m, flat, r64["grads"]).values()])
    yard = np.array([float((r32["grads"][k].double() - g).norm() / (g.norm() + 1e-12)) for k, g in r64["grads"].items()])
    if act_dtype == "bf16":
        # storage rounding through ~60 normalised layers with random unit-gain weights: O(1e-2) on the scalars; the gradients of
        # ANY bf16-storage run are O(1) from the fp64 run (measured: the oracle's own bf16 emulation 1.13 median / 1.85 worst,
        # the kernels 1.09 / 2.01) -- the oracle's emulation is the yardstick, the unit-local checks are the sharp test
        r16 = O.train_step(cfg, {k: v.clone() for k, v in sd.items()}, x, t, act_dtype="bf16", **kw)
        yard = np.array([float((r16["grads"][k].double() - g).norm() / (g.norm() + 1e-12)) for k, g in r64["grads"].items()])
        assert abs(float(loss) - r64["loss_bce"]) <= 3e-2 * max(1.0, abs(r64["loss_bce"]))
        assert abs(float(pen) / B - r64["penalty"]) <= 3e-2 * max(1.0, abs(r64["penalty"]))
        assert np.isfinite(mine).all() and np.median(mine) <= 1.5 * np.median(yard) + 0.05 and mine.max() <= 3 * yard.max(), (
            np.median(mine), np.median(yard), mine.max(), yard.max())
    else:
        assert abs(float(loss) - r64["loss_bce"]) <= 1e-5
        assert abs(float(pen) / B - r64["penalty"]) <= 1e-5 * max(1.0, abs(r64["penalty"]))
        assert np.median(mine) <= 3 * np.median(yard) and mine.max() <= 3 * yard.max(), (np.median(mine), np.median(yard),
                                                                                         mine.max(), yard.max())
    return err, float(np.median(mine)), float(np.median(yard))


def check_slim_network(lib, device, manifest, tmp_path, thres=0.01, B=2, H=32, W=48):
    """The prune-and-finetune result (finetune_model / build_model_with_weight; it has an output branch with ZERO channels)
    through the kernels: eval forward against the oracle (its train step: check_train_units_local(net=slim_network(...)))."""
    m = M.build_model(predefine=manifest)
    m.load_state_dict(O.load_weights(manifest))
    new_cfg, mask = M.finetune_model(m, save_path=str(tmp_path), base_layer_config=M.load_layer_config(manifest), thres=thres)
    slim = M.build_model_with_weight(new_cfg, m, mask).eval()
    sd = {k: v.clone() for k, v in slim.state_dict().items()}
    slim = slim.to(device)
    if device.type == "cpu":
        slim._lib = lib
    x = torch.from_numpy(I.randn_batch(2, B, H, W))
    with torch.no_grad():
        ref = O.csnet_forward(new_cfg, sd, x)
    y = slim(x.to(device)).cpu()
    err = (y - ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, ref.abs().max().item()), err
    return err


def oracle_adam_trajectory(cfg, sd, x, t, steps, lr, wd, eps, batchsize, flops_weight=0.0, act_dtype=None):
    """BCE of `steps` iterations of train.py:203-216 on FIXED data, by the oracle (Adam state carried along)."""
    sd = {k: v.clone() for k, v in sd.items()}
    st = None
    out = []
    for _ in range(steps):
        r = O.train_step(cfg, sd, x, t, expandflop=1.0, flops_weight=flops_weight, batchsize=batchsize, lr=lr, wd=wd, eps=eps,
                         adam_state=st, use_penalty=flops_weight != 0.0, act_dtype=act_dtype)
        st = r["adam_state"]
        out.append(r["loss_bce"])
    return out
