"""numpy fp64 restatement of the reference's scattered-point -> pixel-grid resample
(TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows /root/reference/pixels_coords_testing.py:151-227
(`process_csv_with_pixel_coordinates_nearest`), whose nearest-neighbour search is
`scipy.interpolate.griddata(method='nearest')` = `cKDTree(points).query(xi)`
(scipy 1.15.3 in the build container; unpinned in the reference).  The published
algorithm of that query for k=1, p=2 is: squared Euclidean distance
`(y-yq)*(y-yq) + (z-zq)*(z-zq)` in fp64 without fused multiply-add, smallest wins.
Between source points at EXACTLY equal distance the tree returns whichever its
traversal meets first (depends on libstdc++ `nth_element` inside the build), which has
no closed form; this restatement and the HIP kernel take the LOWEST INDEX.  Pinned
against the real function by tests/golden/resample_*.npz; `tie_mask` reports the
pixels where the choice is ambiguous so tests can assert mismatches occur only there.
"""
from __future__ import annotations

import numpy as np

Y_RANGE = 4e-3    # pixels_coords_testing.py:156
Z_RANGE = 50e-3   # :157
N_SLICES = 101    # :188
TOL_1 = 1e-5      # :192
TOL_2 = 1e-4      # :197
P_CONVERT = 1072.033  # :225


def pixel_centres(height: int, width: int):
    """:164-168"""
    dy, dz = Y_RANGE / height, Z_RANGE / width
    y_final = -0.002 + (np.arange(height) + 0.5) * dy
    z_final = (np.arange(width) + 0.5) * dz
    return y_final, z_final


def wall_bounds(Y_all, Z_all, z_final):
    """:188-211 -- y_min(z), y_max(z) at 101 slices, two tolerances, np.interp to every column."""
    z_samples = np.linspace(0, Z_RANGE, N_SLICES)
    y_min = np.full(N_SLICES, np.nan)
    y_max = np.full(N_SLICES, np.nan)
    for s, zv in enumerate(z_samples):
        d = np.abs(Z_all - zv)
        idx = np.where(d < TOL_1)[0]
        if len(idx) == 0:
            idx = np.where(d < TOL_2)[0]
        if len(idx) > 0:
            y_min[s] = np.min(Y_all[idx])
            y_max[s] = np.max(Y_all[idx])
    valid = ~np.isnan(y_min) & ~np.isnan(y_max)
    return (np.interp(z_final, z_samples[valid], y_min[valid]),
            np.interp(z_final, z_samples[valid], y_max[valid]))


def nearest_index(Y_all, Z_all, y_final, z_final, chunk: int = 2048, return_ties: bool = False):
    """:214-215 -- index of the nearest source point for every pixel centre (row-major i*W+j)."""
    H, W = len(y_final), len(z_final)
    qy = np.repeat(y_final, W)
    qz = np.tile(z_final, H)
    out = np.empty(H * W, dtype=np.int64)
    ties = np.zeros(H * W, dtype=bool)
    for s in range(0, H * W, chunk):
        dy = Y_all[None, :] - qy[s:s + chunk, None]
        dz = Z_all[None, :] - qz[s:s + chunk, None]
        d2 = dy * dy + dz * dz
        out[s:s + chunk] = np.argmin(d2, axis=1)          # first minimum = lowest index
        if return_ties:
            ties[s:s + chunk] = (d2 == d2.min(axis=1, keepdims=True)).sum(axis=1) > 1
    return (out, ties) if return_ties else out


def resample(arr: np.ndarray, height: int = 64, width: int = 512, return_ties: bool = False, chunk: int = 2048):
    """arr: (N,5) fp64 rows x,y,z,p,Umean  ->  (H*W,5) fp64 rows
    y_center,z_center,p_center,Umean_center,p_convert (the reference's output CSV)."""
    arr = np.array(arr, dtype=np.float64, copy=True)
    y_final, z_final = pixel_centres(height, width)
    Y_all, Z_all = arr[:, 1], arr[:, 2]
    P_all = arr[:, 3]
    P_all[P_all < 0] = 0                                     # :182
    U_all = arr[:, 4]
    y_lo, y_hi = wall_bounds(Y_all, Z_all, z_final)
    res = nearest_index(Y_all, Z_all, y_final, z_final, chunk=chunk, return_ties=return_ties)
    idx, ties = res if return_ties else (res, None)
    P = P_all[idx].reshape(height, width)
    U = U_all[idx].reshape(height, width)
    outside = (y_final[:, None] < y_lo[None, :]) | (y_final[:, None] > y_hi[None, :])   # :218-221
    P[outside] = 0.0
    U[outside] = 0.0
    out = np.empty((height * width, 5))
    out[:, 0] = np.repeat(y_final, width)
    out[:, 1] = np.tile(z_final, height)
    out[:, 2] = P.ravel()
    out[:, 3] = U.ravel()
    out[:, 4] = out[:, 2] * P_CONVERT                        # :225
    if return_ties:
        # a tie only matters where the tied points differ in value and the pixel is not masked
        return out, ties & ~outside.ravel()
    return out
