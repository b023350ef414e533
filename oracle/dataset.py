"""CPU restatement of the per-case arithmetic of `utils.load_data` (/root/reference/utils.py:64-107).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): the product path is
cfd_amd.ops.dataset_grids on the GPU.

  pts = -1 where p > 0 else +1                       utils.py:89-93
  DeltaP = mean(p[:, 0]) - mean(p[:, -1])            utils.py:97
  sdf = skfmm.distance(pts, dx=4/H)                  utils.py:100-103

PARITY UNPINNED for the SDF: scikit-fmm (third party, version unpinned in the reference: no requirements file) is
not installed in the build image and the reference holds no fixture for it.  `signed_distance` is a restatement of
scikit-fmm's published fast-marching scheme (oracle/sdf_fmm.c: frozen band by per-axis linear interpolation combined
harmonically, order-2 upwind updates, heap of narrow cells; its header lists every detail and what is uncertain),
checked here against analytic shapes and hand-computed values of the scheme only.
`signed_distance_edt` (the exact Euclidean transform round 1 shipped) is kept as an independent yardstick: fast
marching must stay within its discretisation error of it.
"""
import ctypes

import numpy as np

_lib = None


def _sdf_lib():
    global _lib
    if _lib is None:
        from . import build as B
        _lib = ctypes.CDLL(B.build())
        _lib.sdf_fmm_distance_batch.restype = ctypes.c_int
        _lib.sdf_fmm_distance_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                ctypes.c_void_p]
    return _lib


def signed_distance(phi: np.ndarray, dx: float) -> np.ndarray:
    """skfmm.distance(phi, dx) restated (order 2, no mask): phi (H, W) or (ncase, H, W) fp64"""
    phi = np.ascontiguousarray(phi, dtype=np.float64)
    batched = phi.ndim == 3
    p3 = phi if batched else phi[None]
    out = np.empty_like(p3)
    rc = _sdf_lib().sdf_fmm_distance_batch(p3.ctypes.data, p3.shape[0], p3.shape[1], p3.shape[2], float(dx), out.ctypes.data)
    if rc == 2:
        raise ValueError("the array phi contains no zero contour (no zero level set)")   # skfmm/pfmm.py
    if rc:
        raise MemoryError("oracle sdf_fmm")
    return out if batched else out[0]


def signed_distance_edt(phi: np.ndarray, dx: float) -> np.ndarray:
    """exact Euclidean distance to the contour half a cell from either centre (NOT skfmm's scheme: a yardstick)"""
    from scipy import ndimage
    inside = phi < 0
    if inside.all() or (~inside).all():
        raise ValueError("the array phi contains no zero contour (no zero level set)")
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
