#!/usr/bin/env python
"""Per-launch times of the MFMA conv launches INSIDE a bench workload (ops.KernelTimer: HIP events on the launch stream),
to set beside scripts/layer_bench.py's stand-alone figures for the same layers.

    python scripts/net_layer_times.py [--workload infer_all_bf16|train_u_bf16|infer_pd_fp32] [--batch B] [--steps N]

One line per launch position in the step (median over the steps): kind, GFLOP, microseconds, TFLOP/s, fraction of peak.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cfd_amd import ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="infer_all_bf16")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--sleep", type=int, default=8_000_000, help="on-stream delay (spin cycles) ahead of each step; 0 = none")
    a = ap.parse_args()
    wl = bench.WORKLOADS[a.workload]
    B = a.batch or wl["batch"]
    tdt = torch.float32 if wl["dtype"] == "f32" else torch.bfloat16
    xs, us, _ = synthetic.model_batch(list(range(B)))
    x = torch.from_numpy(xs).cuda()
    if wl.get("train") and wl["nets"] == ["DeltaP"]:
        job = bench.CnnJob(tdt)
        y = torch.full((B, 1), 0.1, device="cuda")
        step = lambda s: job.step(x, y, s, 0)
    elif wl.get("train"):
        job = bench.TrainJob(tdt)
        y = torch.from_numpy(us).cuda()
        step = lambda s: job.step(x, y, s, 0)
    else:
        models = bench.build_nets(wl["nets"], tdt)
        step = lambda s: bench.run_step(models, x, s, 0)
    for s in range(3):
        step(s)
    torch.cuda.synchronize()
    with ops.KernelTimer(max_launches=4096) as kt:
        for s in range(a.steps):
            if a.sleep:
                torch.cuda._sleep(a.sleep)
            step(100 + s)
    torch.cuda.synchronize()
    n = len(kt.records) // a.steps
    peak = bench.PEAK_TFLOPS[wl["dtype"]]
    tot_us = tot_fl = 0.0
    for i in range(n):
        kind, fl = kt.records[i][0], kt.records[i][1]
        us_ = sorted(kt.records[i + k * n][2].elapsed_time(kt.records[i + k * n][3]) * 1e3 for k in range(a.steps))
        t = us_[len(us_) // 2]
        tot_us += t
        tot_fl += fl
        print(f"{i:3d} kind={kind:3d} GFLOP={fl/1e9:9.2f} {t:9.1f} us {fl/t/1e6:8.1f} TF {fl/t/1e6/peak:.3f}", flush=True)
    print(f"TOTAL {tot_fl/1e9:.1f} GFLOP {tot_us:.1f} us {tot_fl/tot_us/1e6:.1f} TF {tot_fl/tot_us/1e6/peak:.3f}")


if __name__ == "__main__":
    main()
