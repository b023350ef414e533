"""Synthetic stenosed-channel cases of the reference's shape.

There is no dataset in the reference (`.gitignore:1-16`), so tests and bench.py draw
cases from here: an OpenFOAM-like table of cell centres `x,y,z,p,Umean` (the input of
`pixels_coords_testing.py:151`), and model-ready grids.  Host-side numpy only.
"""
from __future__ import annotations

import numpy as np

Y_HALF = 2e-3
Z_LEN = 50e-3


def half_width(z: np.ndarray, seed: int):
    rng = np.random.default_rng(seed)
    s = rng.uniform(0.1, 0.7)
    z0 = rng.uniform(15e-3, 35e-3)
    w = rng.uniform(2e-3, 8e-3)
    return Y_HALF * (1.0 - s * np.exp(-((z - z0) / w) ** 2))


def scattered_case(seed: int, n_z: int = 101, n_y: int = 21, jitter: float = 0.0) -> np.ndarray:
    """(n_z*n_y, 5) fp64 rows x,y,z,p,Umean: n_z slices at z = k*Z_LEN/(n_z-1), n_y cell centres
    across the local opening.  `jitter` (fraction of the spacing) scatters the points."""
    rng = np.random.default_rng(1000 + seed)
    z = np.linspace(0.0, Z_LEN, n_z)
    h = half_width(z, seed)
    t = (np.arange(n_y) + 0.5) / n_y * 2.0 - 1.0            # cell centres in (-1,1)
    Y = (h[:, None] * t[None, :])
    Z = np.repeat(z[:, None], n_y, axis=1)
    if jitter > 0:
        Y = Y + jitter * (2 * h[:, None] / n_y) * rng.uniform(-1, 1, Y.shape)
        Zj = Z + jitter * (Z_LEN / (n_z - 1)) * rng.uniform(-0.01, 0.01, Z.shape)
        Z = np.clip(Zj, 0.0, Z_LEN)
    q = 1.0
    umax = 1.5 * q / (2 * h / Y_HALF)
    U = umax[:, None] * (1.0 - (Y / h[:, None]) ** 2)
    dpdz = (Y_HALF / h) ** 3
    p_line = np.cumsum(dpdz[::-1])[::-1]
    p_line = p_line / p_line.max() * 0.9 - 0.02            # a few negative values near the outlet (clamped by :182)
    P = np.repeat(p_line[:, None], n_y, axis=1) + 0.01 * rng.standard_normal(Y.shape) * 0.1
    arr = np.zeros((n_z * n_y, 5))
    arr[:, 1], arr[:, 2], arr[:, 3], arr[:, 4] = Y.ravel(), Z.ravel(), P.ravel(), np.abs(U).ravel()
    return arr


def model_case(seed: int, height: int = 64, width: int = 512):
    """Model-ready grids without the preprocess: (sdf, u, p) each (H,W) fp32.  `sdf` is the
    analytic wall distance in the reference's grid units (dx = 4/H, negative... sign as
    `utils.py:93`: fluid phi=-1, solid phi=+1 -> distance negative inside the fluid)."""
    y = -Y_HALF + (np.arange(height) + 0.5) * (2 * Y_HALF / height)
    z = (np.arange(width) + 0.5) * (Z_LEN / width)
    h = half_width(z, seed)
    d_phys = np.abs(y)[:, None] - h[None, :]                # <0 in the fluid
    sdf = d_phys / (2 * Y_HALF) * 4.0                       # reference units: channel height = 4
    fluid = d_phys < 0
    u = np.where(fluid, 1.5 / (h[None, :] / Y_HALF) * (1 - (y[:, None] / h[None, :]) ** 2), 0.0)
    dpdz = (Y_HALF / h) ** 3
    p_line = np.cumsum(dpdz[::-1])[::-1]
    p_line = p_line / p_line.max()
    p = np.where(fluid, p_line[None, :], 0.0)
    return sdf.astype(np.float32), np.clip(u / 4.0, 0, 1).astype(np.float32), p.astype(np.float32)


def model_batch(seeds, height: int = 64, width: int = 512):
    """Stacked NHWC fp32 batches (x, u, p) with a trailing channel axis."""
    xs, us, ps = zip(*(model_case(s, height, width) for s in seeds))
    f = lambda a: np.stack(a)[..., None].astype(np.float32)
    return f(xs), f(us), f(ps)
