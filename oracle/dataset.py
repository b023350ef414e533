"""CPU restatement of the per-case arithmetic of `utils.load_data` (/root/reference/utils.py:64-107).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): the product path is
cfd_amd.ops.dataset_grids on the GPU.

  pts = -1 where p > 0 else +1                       utils.py:89-93
  DeltaP = mean(p[:, 0]) - mean(p[:, -1])            utils.py:97
  sdf = skfmm.distance(pts, dx=4/H)                  utils.py:100-103

PARITY UNPINNED for the SDF: scikit-fmm (third party, version unpinned in the reference: no requirements file) is
not installed in the build image and the reference holds no fixture for it.  skfmm.distance solves |grad d| = 1 by
fast marching from the zero contour of phi, which for phi in {-1,+1} lies half-way between cells of opposite sign;
this restatement keeps that contract (distance to that contour, negative where phi < 0) with the exact Euclidean
distance transform, so it differs from skfmm by fast marching's first-order discretisation error.
"""
import numpy as np


def signed_distance(phi: np.ndarray, dx: float) -> np.ndarray:
    from scipy import ndimage
    inside = phi < 0
    if inside.all() or (~inside).all():
        raise ValueError("the array phi contains no zero contour (no zero level set)")   # as skfmm does
    d_out = ndimage.distance_transform_edt(~inside)     # cells with phi>0: distance (cells) to the nearest phi<0 cell
    d_in = ndimage.distance_transform_edt(inside)
    d = np.where(inside, -(d_in - 0.5), d_out - 0.5)    # the contour sits half a cell from either centre
    return d * dx


def grids_from_table(p_convert: np.ndarray, umean: np.ndarray, height: int = 64, width: int = 512):
    p = np.asarray(p_convert, dtype=np.float64).reshape(height, width)
    u = np.asarray(umean, dtype=np.float64).reshape(height, width)
    pts = np.where(p > 0, -1.0, 1.0)                                    # utils.py:93
    delta_p = float(np.mean(p[:, 0]) - np.mean(p[:, -1]))              # utils.py:97
    sdf = signed_distance(pts, 4.0 / height)                            # utils.py:100-103
    return sdf, p, u, delta_p
