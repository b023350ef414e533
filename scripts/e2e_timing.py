#!/usr/bin/env python
"""Wall-clock of the reference's entry points end to end on a synthetic processed folder (the drop-in user's view):
<field>/train.py -n 1, then valid.py and test.py, for U, P and DeltaP.  A measurement tool (tests/ hold the parity checks).

    python scripts/e2e_timing.py [--cases 100] [--dtype fp32|bf16]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=100)
ap.add_argument("--dtype", default="fp32")
args = ap.parse_args()
os.environ["CFD_AMD_DTYPE"] = args.dtype

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch  # noqa: E402
import pixels_coords_testing as pct  # noqa: E402
from cfd_amd import cli, keras_api as K, synthetic  # noqa: E402

out = {"cases": args.cases, "dtype": args.dtype}
with tempfile.TemporaryDirectory() as tmp:
    folder = os.path.join(tmp, "cases", "processed")
    os.makedirs(folder)
    for s in range(args.cases):
        arr = synthetic.scattered_case(s, jitter=0.2)
        pd.DataFrame(arr, columns=["x", "y", "z", "p", "Umean"]).to_csv(os.path.join(folder, f"case_{s:03d}.csv"), index=False)
    pct.height, pct.width = 64, 512
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        pct.process_all_csvs_with_pixel_coordinates(folder)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        for field in ("U", "P", "DeltaP"):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(sink):
                cli.train(field, ["-n", "1", "-t", folder + "/"])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            d = os.path.join(tmp, field, os.path.basename(os.path.normpath(cli.FIELDS[field]["default_dir"])))
            g = K.load_model(os.path.join(d, "model_ckpt", "model_G.keras"), compile=False)
            t2 = time.perf_counter()
            with contextlib.redirect_stdout(sink):
                cli.evaluate(field, folder + "/", model=g, out_dir=os.path.join(tmp, "ev_" + field), mode="valid")
            t3 = time.perf_counter()
            with contextlib.redirect_stdout(sink):
                cli.evaluate(field, folder + "/", model=g, out_dir=os.path.join(tmp, "et_" + field), mode="test", train_folder=folder + "/")
            t4 = time.perf_counter()
            out[field] = {"train_script_s": round(t1 - t0, 2), "load_model_s": round(t2 - t1, 2), "valid_s": round(t3 - t2, 2),
                          "test_s": round(t4 - t3, 2)}
    finally:
        os.chdir(cwd)
    ep = [float(ln.split(":")[1]) for ln in sink.getvalue().splitlines() if ln.startswith("epoch time:")]
    out["epoch_s"] = [round(e, 3) for e in ep]
print(json.dumps(out), flush=True)
