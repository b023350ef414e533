"""Pure-write / copy bandwidth yardstick (torch fill_, copy_) for tensors the size of the first-layer outputs."""
import torch
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (134, 268, 537, 1074):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
    t = timeit(lambda: a.fill_(1.0)); print(f"fill  {mb:5d} MB: {t:7.1f} us {mb/t/1e0*1e-6*1e6/1e6:5.2f} TB/s")
    t = timeit(lambda: b.copy_(a)); print(f"copy  {mb:5d} MB: {t:7.1f} us {2*mb/t:5.2f} TB/s (read+write)")
