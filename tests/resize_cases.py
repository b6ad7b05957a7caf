"""Checks of the device resizes (test.py:76-85,94-96) shared by the emulator and the MI355X tests.  skimage is
not available, but what its resize(order=1, mode='reflect', anti_aliasing=False) executes is: scipy.ndimage.zoom(image,
zoom, order=1, mode='mirror', grid_mode=True) (skimage/transform/_warps.py maps 'reflect' to ndimage's 'mirror') -- and
scipy IS here, so that call is the checker: half-pixel-centre bilinear, coordinates outside the image mirrored about the
edge pixel centre (F.interpolate clamps instead: same interior, different border rows / columns when upsampling).  Plus
properties that hold for any correct bilinear resize."""
import numpy as np
import scipy.ndimage as ndi
import torch
import torch.nn.functional as F

from sod100k_amd import engine as E

def sk_resize(x, ho, wo):
    """skimage.transform.resize(x, (ho, wo), order=1, mode='reflect', anti_aliasing=False) of a [planes][h][w] tensor."""
    a = x.numpy().astype(np.float64)
    z = ndi.zoom(a, (1.0, ho / a.shape[1], wo / a.shape[2]), order=1, mode="mirror", grid_mode=True)
    return torch.from_numpy(z.astype(np.float32))


MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


def check_resize_bilinear(lib, dev):
    g = torch.Generator().manual_seed(0)
    for (hi, wi, ho, wo) in [(224, 224, 300, 400), (300, 400, 224, 224), (37, 52, 48, 64), (48, 64, 37, 52), (16, 16, 16, 16),
                             (5, 7, 224, 224), (224, 224, 5, 7)]:
        x = torch.rand(3, hi, wi, generator=g)
        got = E.resize_bilinear(lib, x.to(dev), ho, wo).cpu()
        ref = sk_resize(x, ho, wo)
        assert got.shape == ref.shape
        # fp32 source coordinates and lerp form against scipy's fp64 coordinates: rounding only (measured 2.4e-5)
        assert (got - ref).abs().max().item() <= 5e-5, (hi, wi, ho, wo)
        tref = F.interpolate(x[None], size=(ho, wo), mode="bilinear", align_corners=False)[0]
        my, mx = max(2, -(-ho // hi) + 1), max(2, -(-wo // wi) + 1)     # the mirrored band is half an input pixel wide
        if ho > 2 * my and wo > 2 * mx:                                  # the interior is F.interpolate's rule as well
            assert (got - tref)[:, my:-my, mx:-mx].abs().max().item() <= 5e-5
    x = torch.rand(2, 31, 45, generator=g)
    assert torch.equal(E.resize_bilinear(lib, x.to(dev), 31, 45).cpu(), x)                 # identity
    c = torch.full((1, 20, 30), 0.37)
    assert (E.resize_bilinear(lib, c.to(dev), 57, 41).cpu() - 0.37).abs().max().item() <= 1e-7   # constants
    # an affine ramp is reproduced exactly wherever no border clamp is involved (upsampling interior)
    yy, xx = torch.meshgrid(torch.arange(20.), torch.arange(30.), indexing="ij")
    ramp = (0.5 * yy - 0.25 * xx + 3.0)[None]
    up = E.resize_bilinear(lib, ramp.to(dev), 40, 60).cpu()[0]
    oy, ox = torch.meshgrid((torch.arange(40.) + 0.5) / 2 - 0.5, (torch.arange(60.) + 0.5) / 2 - 0.5, indexing="ij")
    want = 0.5 * oy - 0.25 * ox + 3.0
    assert (up - want)[2:-2, 2:-2].abs().max().item() <= 1e-5


def check_pre_post(lib, dev):
    g = torch.Generator().manual_seed(1)
    for (h, w, H, W) in [(150, 200, 224, 224), (224, 224, 224, 224), (100, 140, 112, 144), (333, 500, 224, 224)]:
        img = torch.rand(2, h, w, 3, generator=g)
        got = E.resize_normalize_nchw(lib, img.to(dev), H, W).cpu()
        ref = (sk_resize(img.permute(0, 3, 1, 2).reshape(6, h, w), H, W).view(2, 3, H, W) - MEAN) / STD
        assert got.shape == (2, 3, H, W)
        # 1 / std (x 4.4) amplifies the rounding of the fp32 source coordinates against scipy's fp64 ones (2.4e-5)
        assert (got - ref).abs().max().item() <= 2.5e-4, (h, w, H, W)
        logits = torch.randn(H, W, generator=g) * 3
        u8 = E.saliency_resize_u8(lib, logits.to(dev), h, w).cpu().numpy()
        p = sk_resize(torch.sigmoid(logits)[None], h, w)[0]
        want = (p.numpy() * 255).astype(np.uint8)
        assert u8.shape == want.shape and u8.dtype == np.uint8
        d = np.abs(u8.astype(int) - want.astype(int))
        assert d.max() <= 1 and (d != 0).mean() < 5e-3, (h, w, d.max(), (d != 0).mean())    # truncation of p * 255 +- 1 ulp
