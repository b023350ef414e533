"""Dataset ingest for the entry scripts: processed CSV folder -> model-ready arrays.

Counterpart of `utils.load_data` (/root/reference/utils.py:64-107): per case, columns `p_convert` and
`Umean_center` of the 64x512-row CSV become the P and U grids, the sign mask pts = -1 where p>0 else +1
(:93), DeltaP = mean(p[:,0]) - mean(p[:,-1]) (:97) and the signed distance of that mask with cell size
4/64 (:100-103), which is the model input.  Files are read in `os.listdir` order, as the reference does.

The reference's SDF is `skfmm.distance` (scikit-fmm, a fast-marching Eikonal solver; absent here and
unpinned in the reference).  `signed_distance` restates its contract -- distance to the zero contour that
lies half-way between cells of opposite sign, negative where phi<0 -- with an exact Euclidean distance
transform instead of fast marching; the two agree to the first-order discretisation error of FMM
(PARITY UNPINNED; a GPU SDF is the next item in SURVEY.md §8(f)).
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np


def signed_distance(phi: np.ndarray, dx: float) -> np.ndarray:
    from scipy import ndimage
    inside = phi < 0
    if inside.all() or (~inside).all():
        # skfmm raises when there is no zero contour; keep that behaviour
        raise ValueError("the array phi contains no zero contour (no zero level set)")
    d_out = ndimage.distance_transform_edt(~inside)     # for cells with phi>0: distance (cells) to nearest phi<0 cell
    d_in = ndimage.distance_transform_edt(inside)
    d = np.where(inside, -(d_in - 0.5), d_out - 0.5)    # the contour sits half a cell from either centre
    return d * dx


def grids_from_table(p_convert: np.ndarray, umean: np.ndarray, height: int = 64, width: int = 512):
    p = np.asarray(p_convert, dtype=np.float64).reshape(height, width)
    u = np.asarray(umean, dtype=np.float64).reshape(height, width)
    pts = np.where(p > 0, -1.0, 1.0)
    delta_p = float(np.mean(p[:, 0]) - np.mean(p[:, -1]))
    sdf = signed_distance(pts, 4.0 / height)
    return sdf, p, u, delta_p


def load_data(processed_folder: str, height: int = 64, width: int = 512):
    """-> bnd_array, sflow_P_array, sflow_U_array, DeltaP_array (lists, one entry per CSV), utils.py:107"""
    import pandas as pd
    bnd, ps, us, dps = [], [], [], []
    for name in os.listdir(processed_folder):
        path = os.path.join(processed_folder, name)
        if not os.path.isfile(path):
            continue
        df = pd.read_csv(path)
        sdf, p, u, dp = grids_from_table(df["p_convert"].to_numpy(), df["Umean_center"].to_numpy(), height, width)
        bnd.append(sdf)
        ps.append(p)
        us.append(u)
        dps.append(dp)
    return bnd, ps, us, dps
