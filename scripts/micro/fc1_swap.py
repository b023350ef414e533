"""fc1 (Dense 262144 -> 256 at batch 128): y = x W^T as the library runs it, against the swapped product y^T = W x^T
(the weights on the 256-row m side of the 256-row kernels, the batch on the n side: no half-empty m tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
B, K, N = 128, 262144, 256
for dt in (torch.float32, torch.bfloat16):
    x = (torch.rand((B, K), device="cuda") - 0.5).to(dt)
    w = ((torch.rand((N, K), device="cuda") - 0.5) * 0.01).to(dt)
    y = torch.empty((B, N), dtype=torch.float32, device="cuda")
    yt = torch.empty((N, B), dtype=torch.float32, device="cuda")
    a = timeit(lambda: ops.conv_forward(x, w.view(N, 1, K), None, ops.DENSE, N, out=y))
    b = timeit(lambda: ops.conv_forward(w, x.view(B, 1, K), None, ops.DENSE, B, out=yt))
    err = float((y - yt.t()).abs().max() / y.abs().max())
    print(f"{dt}: x W^T {a:7.1f} us   W x^T {b:7.1f} us   max rel diff {err:.2e}", flush=True)
