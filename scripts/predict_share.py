"""BASELINE configs[2]'s per-GPU share (1250 geometries, bf16) through the Keras-surface boundary in ONE predict call per
net: numpy in -> U, P, DeltaP predict (chunked inside, keras_api.PREDICT_CHUNK) -> numpy out.  PCIe-inclusive; prints one
JSON line.  Run on the GPU box: python scripts/predict_share.py [n_cases]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CFD_AMD_DTYPE", "bf16")
import numpy as np
import torch

from cfd_amd import keras_api as K, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
base, _, _ = synthetic.model_batch(range(50))
x = np.concatenate([base] * ((n + 49) // 50))[:n]
nets = {"U": K.define_generator_U((64, 512, 1)), "P": K.define_generator_P((64, 512, 1)),
        "DeltaP": K.define_cnn_model((64, 512, 1))}
for m in nets.values():
    m.predict(x[:K.PREDICT_CHUNK])            # warm-up: workspaces, plans
torch.cuda.synchronize()
out, t = {}, {}
t0 = time.perf_counter()
for k, m in nets.items():
    t1 = time.perf_counter()
    out[k] = m.predict(x)
    t[k] = round(time.perf_counter() - t1, 4)
dt = time.perf_counter() - t0
assert out["U"].shape == (n, 64, 512, 1) and out["DeltaP"].shape == (n, 1) and all(np.isfinite(v).all() for v in out.values())
# the first 50 cases repeat every 50: the DeltaP CNN is deterministic at inference -> identical rows across chunks
assert n < 100 or np.array_equal(out["DeltaP"][:50], out["DeltaP"][50:100])
print(json.dumps({"cases": n, "dtype": os.environ["CFD_AMD_DTYPE"], "seconds": round(dt, 4), "per_net_s": t,
                  "samples_per_s_pcie_inclusive": round(n / dt, 1),
                  "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 2)}))
