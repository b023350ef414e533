#!/usr/bin/env python
"""PCIe-inclusive rate of the default bench workload (BASELINE configs[1]: P + DeltaP inference, fp32, B = 128)
through the Keras-surface boundary: host numpy in (pageable memory, as the reference's callers hold it,
U/valid.py:28-29) -> model.predict -> host numpy out.  bench.py's `value` is the HBM-resident rate; this number goes
to DESIGN.md section 5 beside it and is never reported as `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfd_amd import keras_api, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x, _, _ = synthetic.model_batch(list(range(B)))
x = np.ascontiguousarray(x, dtype=np.float32)
os.environ["CFD_AMD_DTYPE"] = DT = sys.argv[2] if len(sys.argv) > 2 else "fp32"      # the default workload computes in fp32
p = keras_api.define_generator_P((64, 512, 1))
d = keras_api.define_cnn_model((64, 512, 1))


def step():
    yp = p.predict(x)
    yd = d.predict(x)
    return yp, yd


for _ in range(3):
    step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    yp, yd = step()
dt = (time.perf_counter() - t0) / n
print(f"P+DeltaP {DT} predict, host numpy in/out, B={B}: {dt*1e3:.2f} ms per batch = {B/dt:.0f} samples/s "
      f"(out {yp.shape} {yd.shape}, finite={bool(np.isfinite(yp).all() and np.isfinite(yd).all())})", flush=True)

# the copies alone (pageable host memory both ways), to separate PCIe from compute
xd = torch.from_numpy(x).cuda()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    a = torch.from_numpy(x).cuda()
    b = torch.from_numpy(x).cuda()
    h = a.cpu().numpy()
dt2 = (time.perf_counter() - t0) / n
mb = 3 * x.nbytes / 1e6
print(f"copies alone (2 x H2D + 1 x D2H of {x.nbytes/1e6:.1f} MB, pageable): {dt2*1e3:.2f} ms per batch "
      f"= {mb/1e3/dt2:.1f} GB/s", flush=True)
