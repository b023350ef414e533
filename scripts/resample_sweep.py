"""Resample cell-density sweep (run on the GPU box): time of one cfd_resample_nearest call over 256 synthetic cases
for several CFD_TUNE_RESAMPLE_DENSITY values; results are identical for every value (checked)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfd_amd import ops, synthetic, _lib
cases = [synthetic.scattered_case(i, jitter=0.2) for i in range(256)]
job = ops.ResampleJob(cases, 64, 512)
ref = None
for dens in (-1, 2, 3, 4, 6, 8, 12, 16, 24):
    ops.set_tuning(_lib.TUNE_RESAMPLE_DENSITY, dens)
    out = job.run(); torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), dens
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): job.run()
    e1.record(); torch.cuda.synchronize()
    print(f"density {dens:3d} (pts/cell {dens/4 if dens>0 else 1.5}): {e0.elapsed_time(e1)/20*1e3:8.1f} us per call", flush=True)
