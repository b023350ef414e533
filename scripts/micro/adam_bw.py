#!/usr/bin/env python
"""HBM rate of cfd_adam_step on a U-generator-sized range (128 M parameters, bf16 shadow): 30 B per parameter.
    python scripts/micro/adam_bw.py   (round 3: one vector per thread 737 us = 5.46 TB/s; bounded grids were slower)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import ops
n = 128 * 1024 * 1024
w = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 1e-3
m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda"); wc = torch.empty(n, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.adam_step(w, g, m, v, wc, 1e-4, 0.5, 0.999, 1e-7, 1.0)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record(); ops.adam_step(w, g, m, v, wc, 1e-4, 0.5, 0.999, 1e-7, 1.0); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev)[5]
print(f"CFD_ADAM_GRID={os.environ.get('CFD_ADAM_GRID', 'default')}: {t * 1e3:.1f} us, {n * 30 / t / 1e9:.2f} TB/s")
