#!/usr/bin/env python
"""Per-case predict latency through the Keras-surface boundary (the reference's valid.py loop, utils.py:55-58):
eager launches vs the captured HIP graph, batch 1, host numpy in / numpy out."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfd_amd import keras_api, synthetic

x, _, _ = synthetic.model_batch([0])
for name, ctor in (("U", keras_api.define_generator_U), ("P", keras_api.define_generator_P), ("DeltaP", keras_api.define_cnn_model)):
    m = ctor((64, 512, 1))
    for mode in ("0", "1"):
        os.environ["CFD_AMD_GRAPHS"] = mode
        for _ in range(3):
            y = m.predict(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            y = m.predict(x)
        dt = (time.perf_counter() - t0) / n
        print(f"{name:7s} graphs={mode}  {dt*1e3:7.3f} ms per predict (B=1)  out {y.shape} finite={np.isfinite(y).all()}", flush=True)
    if name != "DeltaP":
        os.environ["CFD_AMD_GRAPHS"] = "1"
        a, b = m.predict(x), m.predict(x)
        print(f"        two graph replays differ (fresh dropout): {bool((a != b).any())}")
