"""GPU fast-marching SDF (cfd_dataset_grids): wall time by number of cases, next to the C oracle on one host core."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cfd_amd import ops, synthetic
from oracle import dataset as OD
H, W = 64, 512
base = []
for s in range(64):
    _sdf, _u, p = synthetic.model_case(s, H, W)
    base.append(p.astype(np.float64))
for n in (1, 16, 256, 2048, 8192):
    p = np.stack([base[i % 64] for i in range(n)])
    ops.dataset_grids(p[:1], 4.0 / H)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sdf, dp = ops.dataset_grids(p, 4.0 / H)
    t = time.perf_counter() - t0
    print(f"{n:5d} cases: {t*1e3:9.1f} ms  = {t/n*1e3:8.3f} ms/case (incl. H2D/D2H)", flush=True)
t0 = time.perf_counter()
ref = OD.signed_distance(np.where(np.stack(base) > 0, -1.0, 1.0), 4.0 / H)
print(f"oracle (C, 1 core): {(time.perf_counter()-t0)/64*1e3:.3f} ms/case")
sdf, _ = ops.dataset_grids(np.stack(base), 4.0 / H)
print("bit-exact vs oracle:", np.array_equal(sdf.view(np.uint64), ref.view(np.uint64)))
