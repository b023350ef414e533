"""Time of the first-layer kernels alone (run on the GPU box): cfd_conv_first for U e1 / P e1 / D c1 and cfd_conv3_pool2
(DeltaP g1), HIP events over back-to-back launches.  CFD_AMD_LIB=<variant .so> times another build of the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfd_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B in (128, 16):
    for name, cout, two, dt in (("U e1", 256, False, torch.bfloat16), ("P e1", 128, False, torch.bfloat16),
                                ("D c1", 128, True, torch.bfloat16), ("P e1 f32", 128, False, torch.float32)):
        x = torch.rand((B, 64, 512, 1), device="cuda") - 0.5
        x2 = (torch.rand((B, 64, 512, 1), device="cuda") - 0.5) if two else None
        w = (torch.rand((cout, 4, 4, 2 if two else 1), device="cuda") - 0.5) * 0.1
        bias = torch.zeros(cout, device="cuda")
        y = torch.empty((B, 32, 256, cout), dtype=dt, device="cuda")
        y2 = torch.empty((B, 32, 256, cout), dtype=dt, device="cuda")
        for twice in (False, True):
            us = timeit(lambda: ops.conv_first(x, x2, w, bias, cout, 4, 2, ops.ACT_LEAKY, dt, out=y,
                                               out2=y2 if twice else None, act2=ops.ACT_RELU))
            byts = y.numel() * y.element_size() * (2 if twice else 1)
            print(f"B={B:3d} {name:9s} outputs={2 if twice else 1}: {us:8.1f} us  {byts/us/1e6:5.2f} TB/s written", flush=True)
