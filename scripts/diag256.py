#!/usr/bin/env python
"""Diagnostic build of igemm256 (cycle stamps per phase segment) on one layer shape: prints the mean cycles per
K tile spent in each segment (memory cluster issue, vmcnt wait, barrier a, MFMA cluster, barrier b)."""
import ctypes as C, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_amd import ops, _lib
from cfd_amd.ops import CONV_K4S2, CONVT_K4S2
B, h, w, cin, cout, kind = 64, 8, 64, 1024, 512, CONVT_K4S2   # U d4
x = (torch.rand((B, h, w, cin), device="cuda") - 0.5).bfloat16()
wt = ((torch.rand((cout, 16, cin), device="cuda") - 0.5) * 0.1).bfloat16()
y = torch.empty(ops.conv_out_shape(kind, B, h, w, cout), dtype=torch.bfloat16, device="cuda")
lib = _lib.load()
ops.set_tuning(ops.TUNE_IGEMM256, 1)
ops.set_tuning(ops.TUNE_IGEMM256_PD, 106)
nblocks = (B * h * w // 256) * (cout // 256) * 4
ws = torch.zeros(nblocks * 8 * 8, dtype=torch.int64, device="cuda")
d = _lib.ConvDesc(kind=kind, dtype=_lib.BF16, out_dtype=_lib.BF16, B=B, Hi=h, Wi=w, Cin=cin, Cout=cout, ldx=cin,
                  ldy=cout, act=0, leaky_slope=0.2, split_k=1)
for _ in range(3):
    _lib.check(lib.cfd_conv_forward(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wt.data_ptr()), None,
                                    C.c_void_p(y.data_ptr()), C.c_void_p(ws.data_ptr()),
                                    torch.cuda.current_stream().cuda_stream), "fwd")
torch.cuda.synchronize()
t = ws.view(nblocks, 8, 8).double().cpu()
nkt = t[..., 5].mean().item()
names = ["mem-cluster issue(+lds return)", "vmcnt wait", "barrier a", "mfma cluster", "barrier b"]
tot = 0
for g, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    print(g)
    for i, n in enumerate(names):
        v = t[:, sl, i].mean().item() / nkt
        print(f"   {n:34s} {v:8.1f} cycles per K tile")
    print(f"   total {t[:, sl, :5].sum(-1).mean().item() / nkt:8.1f} (ideal 2048)")
