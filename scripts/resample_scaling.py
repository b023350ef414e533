"""Resample throughput against the number of cases per call (run on the GPU box): one cfd_resample_nearest call over
N synthetic cases (101 x 21 jittered points -> 64 x 512), HIP events over 10 calls.  Shows how much of the 256-case
bench figure is the fixed chain of launches rather than per-case work."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfd_amd import ops, synthetic
base = [synthetic.scattered_case(i, jitter=0.2) for i in range(256)]
for n in (64, 256, 1024, 4096):
    cases = [base[i % 256] for i in range(n)]
    job = ops.ResampleJob(cases, 64, 512)
    job.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): job.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    npts = sum(len(c) for c in cases)
    byts = 40.0 * (npts + n * 64 * 512)
    print(f"cases {n:5d}: {us:9.1f} us per call, {us/n:6.3f} us per case, {n*64*512/us/1e3:7.2f} G queries/s, "
          f"{byts/us/1e6:6.2f} TB/s algorithmic = {byts/us/1e6/8.0:.3f} of HBM peak", flush=True)
    del job
