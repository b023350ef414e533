"""Generate tests/golden/resample_*.npz by running the REAL reference function
`process_csv_with_pixel_coordinates_nearest` (/root/reference/pixels_coords_testing.py:151-227)
in the build container.  Run once, by hand:  python tests/golden/make_resample_golden.py

The reference module imports `cv2` and `skfmm` at top level (:8,:10), which are not
installed; the function under test uses neither, so two empty placeholder modules are
registered before the import.  The module-level globals `height`,
`width` exist only under `__main__` (:360-361) and are set here.  Only inputs and outputs
(data) are stored -- no reference source.
"""
import os
import sys
import tempfile
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from cfd_amd import synthetic  # noqa: E402

for name in ("cv2", "skfmm"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, "/root/reference")
import matplotlib  # noqa: E402
matplotlib.use("Agg")
import pixels_coords_testing as ref  # noqa: E402


def run_reference(arr: np.ndarray, height: int, width: int) -> np.ndarray:
    ref.height, ref.width = height, width
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "in.csv"), os.path.join(d, "out.csv")
        # repr-exact CSV round trip: float_format %.17g keeps every fp64 bit
        pd.DataFrame(arr, columns=["x", "y", "z", "p", "Umean"]).to_csv(src, index=False, float_format="%.17g")
        ref.process_csv_with_pixel_coordinates_nearest(src, dst)
        # the reference writes with pandas' default repr (shortest round-trip) -> exact on read
        out = pd.read_csv(dst, float_precision="round_trip").to_numpy()
        # what the reference actually computed on: its own pd.read_csv(src) (:174), whose default
        # float parser is not bit-exact -- so the fixture stores the PARSED table as the input
        back = pd.read_csv(src).to_numpy()
    return back, out


def tie_case(height: int, width: int) -> np.ndarray:
    """Source points placed symmetrically (exact in fp64) about pixel centres so that the
    two nearest points are at EXACTLY the same distance and carry different values."""
    dy, dz = 4e-3 / height, 50e-3 / width
    y = -0.002 + (np.arange(height) + 0.5) * dy
    z = (np.arange(width) + 0.5) * dz
    rows = []
    delta = 2.0 ** -12
    rng = np.random.default_rng(7)
    for j in range(0, width, 2):
        for i in range(0, height, 2):
            zc = z[j]
            lo, hi = zc - delta, zc + delta
            assert (hi - zc) == (zc - lo)
            rows.append([0.0, y[i], lo, rng.uniform(0.1, 1), rng.uniform(0.1, 1)])
            rows.append([0.0, y[i], hi, rng.uniform(0.1, 1), rng.uniform(0.1, 1)])
    # slice points so the wall detection finds an open channel everywhere
    for zv in np.linspace(0, 50e-3, 101):
        rows.append([0.0, -0.0019, zv, 0.5, 0.5])
        rows.append([0.0, 0.0019, zv, 0.5, 0.5])
    return np.array(rows)


def main():
    cases = {
        "lattice_64x512": (synthetic.scattered_case(0), 64, 512),
        "lattice2_64x512": (synthetic.scattered_case(3), 64, 512),
        "jitter_64x512": (synthetic.scattered_case(1, jitter=0.3), 64, 512),
        "sparse_32x128": (synthetic.scattered_case(2, n_z=51, n_y=9, jitter=0.2), 32, 128),   # exercises the 1e-4 tolerance / NaN slices
        "upscaled_128x1024": (synthetic.scattered_case(4, jitter=0.1), 128, 1024),
        "ties_16x64": (tie_case(16, 64), 16, 64),
    }
    for name, (arr, h, w) in cases.items():
        arr, out = run_reference(arr.copy(), h, w)
        np.savez_compressed(os.path.join(HERE, f"resample_{name}.npz"), points=arr, height=h, width=w,
                            out=out.astype(np.float64))
        print(name, arr.shape, out.shape)


if __name__ == "__main__":
    main()
