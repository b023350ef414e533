#!/usr/bin/env python
"""Where a tile's time goes in igemm256 (one tile per workgroup): diagnostic build with four s_memtime stamps per wave
(entry, pipeline primed, K loop done, stores issued) plus the CU it ran on.  Prints the mean length of each stage and
the gap between consecutive workgroups on the same CU (= launch + LDS hand-over), in microseconds and in K tiles."""
import ctypes as C, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_amd import ops, _lib

B, H, W, cin, cout = 128, 16, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 512
x = torch.randn((B, H, W, cin), device="cuda").to(torch.bfloat16)
w = (torch.randn((cout, 4, 4, cin), device="cuda") * 0.05).to(torch.bfloat16)
y = torch.empty((B, H // 2, W // 2, cout), device="cuda", dtype=torch.bfloat16)
nblocks = (B * (H // 2) * (W // 2) // 256) * (cout // 256)
ws = torch.zeros(nblocks * 8 * 8, dtype=torch.int64, device="cuda")
lib = _lib.load()
ops.set_tuning(ops.TUNE_IGEMM256, 1)
ops.set_tuning(ops.TUNE_IGEMM256_PD, 226)
d = _lib.ConvDesc(kind=ops.CONV_K4S2, dtype=_lib.BF16, out_dtype=_lib.BF16, B=B, Hi=H, Wi=W, Cin=cin, Cout=cout, ldx=cin,
                  ldy=cout, act=0, leaky_slope=0.2, split_k=1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    e0.record()
    _lib.check(lib.cfd_conv_forward(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), None,
                                    C.c_void_p(y.data_ptr()), C.c_void_p(ws.data_ptr()),
                                    torch.cuda.current_stream().cuda_stream), "fwd")
    e1.record()
torch.cuda.synchronize()
kernel_us = e0.elapsed_time(e1) * 1e3
t = ws.view(nblocks, 8, 8).cpu().numpy().astype(np.int64)
assert (t[..., 0] > 0).all(), "some workgroup did not write its stamps"
t0 = t[..., 0].min()
span = (t[..., 3].max() - t0)
tick_us = kernel_us / span            # s_memtime ticks -> us, from the kernel's own duration
t5 = t[..., 5] >> 8                     # stamp after the address set-up (packed above the XCC id)
low = t[..., 0] & ((1 << 55) - 1)
setup = ((t5 - low) & ((1 << 55) - 1)).mean()
st = (setup, (t[..., 1] - t[..., 0]).mean() - setup, (t[..., 2] - t[..., 1]).mean(), (t[..., 7] - t[..., 2]).mean(),
      (t[..., 3] - t[..., 7]).mean())
nkt = int(t[0, 0, 6])
print(f"cin {cin}: {nblocks} tiles, {nkt} K tiles each, kernel {kernel_us:.1f} us, {span} ticks ({1/tick_us:.0f} ticks/us)")
per_k = st[2] * tick_us / nkt
for name, v in zip(("entry -> addresses set up", "first loads issued -> pipeline primed", "K loop",
                    "K loop done -> tail loads drained (vmcnt 0)", "bias / activation / convert + stores issued"), st):
    print(f"   {name:48s} {v * tick_us:8.2f} us  = {v * tick_us / per_k:6.1f} K tiles")
# gaps on one CU: group workgroups by (xcc, hw_id CU/SE bits), sort by entry
hw, xcc = t[:, 0, 4], t[:, 0, 5] & 0xf
cu_key = (xcc << 32) | (hw & ~np.int64(0xff))   # drop wave / simd bits
gaps = []
for k in np.unique(cu_key):
    idx = np.where(cu_key == k)[0]
    idx = idx[np.argsort(t[idx, :, 0].min(axis=1))]
    for a, b in zip(idx[:-1], idx[1:]):
        gaps.append(t[b, :, 0].min() - t[a, :, 3].max())
gaps = np.array(gaps, dtype=np.float64)
print(f"   distinct CUs seen: {len(np.unique(cu_key))}")
if len(gaps):
    print(f"   gap: last wave's stores issued -> next workgroup's first wave enters   mean {gaps.mean() * tick_us:6.2f} us "
          f"(median {np.median(gaps) * tick_us:.2f}, max {gaps.max() * tick_us:.2f}) = {gaps.mean() * tick_us / per_k:.1f} K tiles")
first = (t[:, :, 0].min(axis=1) - t0) * tick_us
print(f"   first-wave entry of the first 256 workgroups: mean {np.sort(first)[:256].mean():.2f} us after kernel start")
