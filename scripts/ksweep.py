#!/usr/bin/env python
"""Per-tile overhead of the 256x256 kernel: time one conv shape at several Cin (K tiles per tile = 16 * Cin / 64) with the
pixel and channel tiling fixed, and fit t = tiles_per_CU * (a + b * ktiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfd_amd import ops

B, H, W, cout = 128, 16, 128, 512
rows = []
for cin in (64, 128, 256, 512, 1024):
    x = torch.randn((B, H, W, cin), device="cuda").to(torch.bfloat16)
    w = (torch.randn((cout, 4, 4, cin), device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty((B, H // 2, W // 2, cout), device="cuda", dtype=torch.bfloat16)
    ops.set_tuning(ops.TUNE_IGEMM256, 1)
    for _ in range(3):
        ops.conv_forward(x, w, None, ops.CONV_K4S2, cout, out=y, split_k=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        ops.conv_forward(x, w, None, ops.CONV_K4S2, cout, out=y, split_k=1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tiles = (B * (H // 2) * (W // 2) // 256) * (cout // 256)
    kt = 16 * cin // 64
    rows.append((kt, us, tiles))
    print(f"cin {cin:5d} ktiles/tile {kt:4d} tiles {tiles} {us:9.1f} us  per tile-wave {us / (tiles / 256):8.2f} us  TF {2.0*B*(H//2)*(W//2)*cout*16*cin/us/1e6:7.1f}")
(k1, t1, n1), (k2, t2, _) = rows[1], rows[-1]
b = (t2 - t1) / (k2 - k1) / (n1 / 256)
a = t1 / (n1 / 256) - b * k1
print(f"fit: per K tile {b:.3f} us, per tile overhead {a:.2f} us = {a / b:.1f} K tiles")
