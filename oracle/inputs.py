"""Deterministic synthetic inputs shared by the golden generator, the tests and bench.py.

TEST INFRASTRUCTURE (see oracle/csnet_oracle.py header).  Everything is regenerated from
numpy seeds on both sides, so the inputs themselves never need to be stored.
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float64)   # test.py:68
STD = np.array([0.229, 0.224, 0.225], dtype=np.float64)    # test.py:69


def randn_batch(seed: int, b: int, h: int = 224, w: int = 224) -> np.ndarray:
    """numpy.random.default_rng(seed).standard_normal((b,3,h,w), float32)."""
    return np.random.default_rng(seed).standard_normal((b, 3, h, w), dtype=np.float32)


def binary_target(seed: int, b: int, h: int = 224, w: int = 224) -> np.ndarray:
    return (np.random.default_rng(seed).random((b, 1, h, w)) > 0.5).astype(np.float32)


def image_like(h: int = 224, w: int = 224) -> np.ndarray:
    """A smooth RGB picture in [0,1], normalised like test.py:86 -> (1,3,h,w) float32."""
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    r = 0.5 + 0.5 * np.sin(6.0 * xx + 2.0 * yy)
    g = np.exp(-((xx - 0.4) ** 2 + (yy - 0.55) ** 2) / 0.05)
    b = 0.5 + 0.5 * np.cos(9.0 * yy * xx + 1.0)
    img = np.stack([r, g, b], axis=-1)                       # H,W,3 float64 in [0,1]
    img = np.transpose((img - MEAN) / STD, (2, 0, 1))
    return img[None].astype(np.float32)


def probe_indices(n: int, k: int = 32) -> np.ndarray:
    return np.linspace(0, n - 1, k).astype(np.int64)


def probe(t) -> dict:
    """Summary of one tensor used by the per-unit probes (G3)."""
    a = np.asarray(t, dtype=np.float32).reshape(-1)
    a64 = a.astype(np.float64)
    return dict(shape=list(np.asarray(t).shape), mean=float(a64.mean()), absmax=float(np.abs(a64).max()),
                l2=float(np.sqrt((a64 * a64).sum())), samples=[float(v) for v in a[probe_indices(a.size)]])
