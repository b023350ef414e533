"""Dataset ingest for the entry scripts: processed CSV folder -> model-ready arrays.

Counterpart of `utils.load_data` (/root/reference/utils.py:64-107): per case, columns `p_convert` and
`Umean_center` of the 64x512-row CSV become the P and U grids, the sign mask pts = -1 where p>0 else +1
(:93), DeltaP = mean(p[:,0]) - mean(p[:,-1]) (:97) and the signed distance of that mask with cell size
4/64 (:100-103), which is the model input.  Files are read in `os.listdir` order, as the reference does.

The per-case arithmetic runs on the GPU (`ops.dataset_grids`, csrc/dataset.hip): sign mask, DeltaP in
numpy's summation order, and the signed distance.  The reference's SDF is `skfmm.distance` (scikit-fmm, a
fast-marching Eikonal solver; absent here and unpinned in the reference): the kernel keeps its contract --
distance to the zero contour half-way between cells of opposite sign, negative where phi<0 -- with an exact
Euclidean distance transform (PARITY UNPINNED; checked bit-for-bit against oracle/dataset.py).
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np


def signed_distance(phi: np.ndarray, dx: float) -> np.ndarray:
    """skfmm.distance(phi, dx) for a +-1 mask (utils.py:100-103), on the GPU."""
    from . import ops
    phi = np.asarray(phi, dtype=np.float64)
    sdf, _dp = ops.dataset_grids(np.where(phi < 0, 1.0, -1.0)[None], cell=dx)   # p > 0  <=>  phi < 0
    return sdf[0]


def grids_from_table(p_convert: np.ndarray, umean: np.ndarray, height: int = 64, width: int = 512):
    from . import ops
    p = np.asarray(p_convert, dtype=np.float64).reshape(height, width)
    u = np.asarray(umean, dtype=np.float64).reshape(height, width)
    sdf, dp = ops.dataset_grids(p[None], cell=4.0 / height)
    return sdf[0], p, u, float(dp[0])


def load_data(processed_folder: str, height: int = 64, width: int = 512):
    """-> bnd_array, sflow_P_array, sflow_U_array, DeltaP_array (lists, one entry per CSV), utils.py:107"""
    import pandas as pd
    from . import ops
    ps, us = [], []
    for name in os.listdir(processed_folder):
        path = os.path.join(processed_folder, name)
        if not os.path.isfile(path):
            continue
        df = pd.read_csv(path)
        ps.append(np.asarray(df["p_convert"].to_numpy(), dtype=np.float64).reshape(height, width))
        us.append(np.asarray(df["Umean_center"].to_numpy(), dtype=np.float64).reshape(height, width))
    if not ps:
        return [], [], [], []
    sdf, dp = ops.dataset_grids(np.stack(ps), cell=4.0 / height)     # all cases in one launch
    return [sdf[i] for i in range(len(ps))], ps, us, [float(v) for v in dp]
