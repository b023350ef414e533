#!/usr/bin/env python
"""Per-layer timing of the GEMM-shaped layers (forward / data-gradient / weight-gradient) on one MI355X.

    python scripts/layer_bench.py [--net U|P|D|DeltaP|all] [--batch B] [--dtype bf16|f32] [--iters N] [--modes fwd,dgrad,wgrad]

Prints one line per (layer, mode): GEMM dims, microseconds per launch (HIP events on the launch stream,
median of `iters` back-to-back launches after warm-up), algorithmic TFLOP/s and the fraction of the MFMA peak.
Operands are uniform random (never zero-filled: zero operands raise the clock and read high).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_amd import ops  # noqa: E402
from cfd_amd.ops import CONV_K3S1, CONV_K4S1, CONV_K4S2, CONVT_K4S2, DENSE, CONV_K4S1_BWD  # noqa: E402

PEAK = {"bf16": 2500.0, "f32": 157.3}
TAPS = {CONV_K4S2: 16, CONV_K4S1: 16, CONV_K3S1: 9, CONVT_K4S2: 4, DENSE: 1, CONV_K4S1_BWD: 16}


def gen_layers(arch):
    from cfd_amd.nets import GEN_ARCH
    a = GEN_ARCH[arch]
    L, h, w, c = [], 32, 256, a["enc"][0]
    for i, f in enumerate(a["enc"][1:], 2):
        L.append((f"e{i}", CONV_K4S2, h, w, c, f)); h, w, c = h // 2, w // 2, f
    L.append(("b", CONV_K4S2, h, w, c, a["bott"])); h, w, c = h // 2, w // 2, a["bott"]
    for i, ((f, _), s) in enumerate(zip(a["dec"], a["enc"][::-1]), 1):
        L.append((f"d{i}", CONVT_K4S2, h, w, c, f)); h, w, c = h * 2, w * 2, f + s
    return L


def disc_layers():
    return [("c2", CONV_K4S2, 32, 256, 128, 256), ("c3", CONV_K4S2, 16, 128, 256, 512),
            ("c4", CONV_K4S2, 8, 64, 512, 1024), ("c5", CONV_K4S1, 4, 32, 1024, 1024)]


def cnn_layers():
    return [("g3", CONV_K3S1, 32, 256, 64, 128), ("g4", CONV_K3S1, 32, 256, 128, 256),
            ("g6", CONV_K3S1, 16, 128, 256, 512), ("fc1", DENSE, 1, 1, 262144, 256)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ts)
    return v[len(v) // 2] * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="all")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--modes", default="fwd,dgrad,wgrad")
    ap.add_argument("--igemm256", type=int, default=-1, help="-1 heuristic, 0 never, 1 whenever eligible")
    ap.add_argument("--pd", type=int, default=-1)
    ap.add_argument("--wgrad256", type=int, default=-1)
    ap.add_argument("--class-order", type=int, default=-1, help="-1 heuristic, 0 class-major, 1 interleaved")
    ap.add_argument("--staged-out", type=int, default=-1, help="-1/1 staged bf16 epilogue where legal, 0 off")
    ap.add_argument("--ntile", type=int, default=-1, help="-1 heuristic, 128 / 256: n tile of the ping-pong kernels")
    ap.add_argument("--only", default="", help="comma list of layer names, e.g. e2,d4")
    ap.add_argument("--bias", type=int, default=0, help="1: forward launches add a bias vector (as the nets do)")
    ap.add_argument("--onephase", default="", help="comma list of CFD_TUNE_ONEPHASE values timed side by side (e.g. 0,1)")
    ap.add_argument("--patch", default="-1", help="comma list of CFD_TUNE_PATCH values timed side by side per layer "
                                                   "(-1/1 patch kernels, 0 off, 2 co-resident for 128-wide, 3 co-resident everywhere)")
    args = ap.parse_args()
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    ops.set_tuning(ops.TUNE_IGEMM256, args.igemm256)
    ops.set_tuning(ops.TUNE_IGEMM256_PD, args.pd)
    ops.set_tuning(ops.TUNE_WGRAD256, args.wgrad256)
    ops.set_tuning(ops.TUNE_CLASS_ORDER, args.class_order)
    ops.set_tuning(ops.TUNE_STAGED_OUT, args.staged_out)
    ops.set_tuning(ops.TUNE_NTILE, args.ntile)
    B = args.batch
    nets = ["U", "P", "D", "DeltaP"] if args.net == "all" else [args.net]
    modes = args.modes.split(",")
    tot = {m: [0.0, 0.0] for m in modes}
    for net in nets:
        layers = gen_layers(net) if net in ("U", "P") else (disc_layers() if net == "D" else cnn_layers())
        for name, kind, h, w, cin, cout in layers:
            if args.only and name not in args.only.split(","):
                continue
            g = torch.Generator(device="cuda").manual_seed(1)
            if kind == DENSE:
                x = (torch.rand((B, cin), device="cuda", generator=g) - 0.5).to(tdt)
                oshape = (B, cout)
            else:
                x = (torch.rand((B, h, w, cin), device="cuda", generator=g) - 0.5).to(tdt)
                oshape = ops.conv_out_shape(kind, B, h, w, cout)
            ntw = {CONV_K3S1: 9, DENSE: 1}.get(kind, 16)
            wt = ((torch.rand((cout, ntw, cin), device="cuda", generator=g) - 0.5) * 0.1).to(tdt)
            wtt = ((torch.rand((cin, ntw, cout), device="cuda", generator=g) - 0.5) * 0.1).to(tdt)
            y = torch.empty(oshape, dtype=tdt, device="cuda")
            dy = (torch.rand(oshape, device="cuda", generator=g) - 0.5).to(tdt)
            dx = torch.empty_like(x)
            dw = torch.empty((cout, ntw, cin), dtype=torch.float32, device="cuda")
            opix = y.numel() // cout
            flops = 2.0 * opix * cout * TAPS[kind] * cin
            bk = {CONV_K4S2: CONVT_K4S2, CONVT_K4S2: CONV_K4S2, CONV_K3S1: CONV_K3S1, DENSE: DENSE,
                  CONV_K4S1: CONV_K4S1_BWD}[kind]
            bias = torch.zeros(cout, device="cuda") if args.bias else None
            fns = {"fwd": lambda: ops.conv_forward(x, wt, bias, kind, cout, out=y),
                   "dgrad": lambda: ops.conv_forward(dy, wtt, None, bk, cin, out=dx),
                   "wgrad": lambda: ops.conv_wgrad(x, dy, kind, cout, dw)}
            patches = [int(v) for v in args.patch.split(",")]
            if args.onephase:                       # A/B of the one-cluster schedule instead of the patch modes
                patches = [1000 + int(v) for v in args.onephase.split(",")]
            for m in modes:
                for pm in patches:
                    if pm >= 1000:
                        ops.set_tuning(ops.TUNE_ONEPHASE, pm - 1000)
                    else:
                        ops.set_tuning(ops.TUNE_PATCH, pm)
                    us = timeit(fns[m], args.iters)
                    tf = flops / us / 1e6
                    if pm == patches[0]:
                        tot[m][0] += flops; tot[m][1] += us
                    tag = f" patch={pm}" if len(patches) > 1 else ""
                    print(f"{net:6s} {name:4s} {m:5s} B={B} in=({h},{w},{cin}) cout={cout} GFLOP={flops/1e9:8.2f} "
                          f"{us:9.1f} us {tf:8.1f} TF {tf/PEAK[args.dtype]:.3f}{tag}", flush=True)
            ops.set_tuning(ops.TUNE_PATCH, -1)
            ops.set_tuning(ops.TUNE_ONEPHASE, -1)
    for m in modes:
        if tot[m][1] > 0:
            tf = tot[m][0] / tot[m][1] / 1e6
            print(f"TOTAL {m}: {tot[m][0]/1e9:.1f} GFLOP {tot[m][1]:.1f} us {tf:.1f} TF {tf/PEAK[args.dtype]:.3f}")


if __name__ == "__main__":
    main()
