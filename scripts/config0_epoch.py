#!/usr/bin/env python
"""BASELINE configs[0] end to end: "U/train.py surrogate, 1 epoch fwd+bwd on synthetic 100-case processed grids".

Writes 100 synthetic raw cases (scattered OpenFOAM-like tables), runs the preprocess step on them
(pixels_coords_testing.process_all_csvs_with_pixel_coordinates: GPU resample of all cases in one launch), then the
training script's driver (`cfd_amd.cli.train("U", ["-n", "1", "-t", folder])`: load_data incl. the GPU SDF, 90/10 split,
90 batch-1 GAN steps, 10 validation steps, checkpoints), each phase timed on the wall clock.  A measurement tool:
tests/ hold the parity checks, and the CPU figure to set beside it is `bench.py --workload train_u_fp32`'s
`cpu_baseline` (the oracle's batch-1 U training steps per second on the host cores).

    python scripts/config0_epoch.py [--cases 100] [--dtype fp32|bf16]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=100)
ap.add_argument("--dtype", default="fp32")
args = ap.parse_args()
os.environ["CFD_AMD_DTYPE"] = args.dtype

import torch  # noqa: E402
import pixels_coords_testing as pct  # noqa: E402
from cfd_amd import cli, synthetic  # noqa: E402

out = {"config": "configs[0]: U/train.py, 1 epoch, batch 1", "cases": args.cases, "dtype": args.dtype}
with tempfile.TemporaryDirectory() as tmp:
    folder = os.path.join(tmp, "cases", "processed")
    os.makedirs(folder)
    t0 = time.perf_counter()
    for s in range(args.cases):
        arr = synthetic.scattered_case(s, jitter=0.2)
        pd.DataFrame(arr, columns=["x", "y", "z", "p", "Umean"]).to_csv(os.path.join(folder, f"case_{s:03d}.csv"), index=False)
    out["write_raw_csv_s"] = round(time.perf_counter() - t0, 3)
    pct.height, pct.width = 64, 512
    sink = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sink):
        pct.process_all_csvs_with_pixel_coordinates(folder)
    torch.cuda.synchronize()
    out["preprocess_resample_s"] = round(time.perf_counter() - t0, 3)      # CSV parse + GPU resample + CSV write
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sink):
            hist = cli.train("U", ["-n", "1", "-t", folder + "/"])
        torch.cuda.synchronize()
        out["train_script_s"] = round(time.perf_counter() - t0, 3)        # model build, load_data, epoch, validation, checkpoints
    finally:
        os.chdir(cwd)
    log = sink.getvalue()
    ep = [ln for ln in log.splitlines() if ln.startswith("epoch time:")]
    out["epoch_s"] = round(float(ep[0].split(":")[1]), 3) if ep else None  # the reference's own "epoch time" print (U/train.py:139)
    n_train = args.cases - int(np.ceil(0.1 * args.cases))
    out["train_steps"] = n_train
    out["steps_per_s"] = round(n_train / out["epoch_s"], 2) if ep else None
    out["losses"] = {k: round(float(v[0]), 5) for k, v in hist.items() if v}
    assert all(np.isfinite(v) for v in out["losses"].values()), out["losses"]

print(json.dumps(out), flush=True)
