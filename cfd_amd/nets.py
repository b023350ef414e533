"""The three surrogate graphs of the reference, executed layer by layer on libcfd_amd.so kernels.

  Generator      U/models/generator.py:44-74 (U), P/models/generator.py:69-114 (P)
  Discriminator  U/models/discriminator.py:12-44
  CnnRegressor   DeltaP/models/generator.py:15-49

Data flow of a generator forward (all NHWC, activations in the compute dtype):

  x fp32 --conv_first--> e1 (LeakyReLU)  and  cat5[..., C_d5:] (ReLU copy = skip half of d5's concat)
  e_{i-1} --igemm k4s2--> raw_i --bn_act--> e_i (LeakyReLU)  and  cat_{6-i}[..., C:] (ReLU copy)
  e5 --igemm k4s2 + ReLU--> b
  prev --igemm convT (4 parity classes)--> raw --bn_act(+dropout, ReLU)--> cat_i[..., :C]
  cat5 --tap GEMM + gather + sigmoid--> y fp32

so the Concatenate of the decoder (generator.py:38) never exists as a copy: both halves are
written in place by the kernels that produce them.  BatchNorm statistics are per sample (the
reference only ever runs batch 1), which makes a batch of B equal to B reference predictions.
Activation buffers are cached per batch size; a forward allocates nothing after the first call.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops
from .ops import (ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, CONV_K3S1, CONV_K4S1, CONV_K4S2, CONVT_K4S2, DENSE)
from .params import ParamStore, Spec

U_GEN = dict(enc=[256, 512, 512, 1024, 1024], bott=1024,
             dec=[(1024, True), (1024, True), (512, True), (512, False), (256, False)])
P_GEN = dict(enc=[128, 256, 256, 512, 1024], bott=2048,
             dec=[(1024, True), (512, False), (256, False), (256, False), (128, False)])
GEN_ARCH = {"U": U_GEN, "P": P_GEN}
DISC = [(128, 2, False), (256, 2, True), (512, 2, True), (1024, 2, True), (1024, 1, True)]
BN_EPS = 1e-3
DROP_RATE = 0.5
# In bf16 mode the pre-BatchNorm conv output ("raw") is kept in fp32 for layers up to this many
# elements per sample: those tensors cost little bandwidth and rounding them BEFORE the mean is
# subtracted is the largest single error term of the bf16 path.  Larger (high-resolution) layers keep bf16.
RAW_F32_MAX_ELEMS = 524288


def _bn_specs(name: str, c: int) -> List[Spec]:
    return [Spec(name + ".bn.gamma", "bn", (c,)), Spec(name + ".bn.beta", "bn", (c,)),
            Spec(name + ".bn.moving_mean", "stat", (c,)), Spec(name + ".bn.moving_var", "stat", (c,))]


class _Net:
    def __init__(self, specs, device, dtype, seed):
        self.device, self.dtype = torch.device(device), dtype
        self.params = ParamStore(specs, device, dtype)
        self.params.init_random_normal(seed)
        self._plans: Dict[int, dict] = {}

    def buf(self, plan: dict, key: str, shape, dtype=None) -> torch.Tensor:
        t = plan.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype or self.dtype, device=self.device)
            plan[key] = t
        return t

    def plan(self, B: int) -> dict:
        return self._plans.setdefault(B, {})

    def raw_dtype(self, h: int, w: int, c: int) -> torch.dtype:
        return torch.float32 if (self.dtype == torch.float32 or h * w * c <= RAW_F32_MAX_ELEMS) else self.dtype


class Generator(_Net):
    def __init__(self, arch: str, image_shape=(64, 512, 1), device="cuda", dtype=torch.bfloat16, seed: int = 0):
        self.arch, self.a = arch, GEN_ARCH[arch]
        self.H, self.W, cin = image_shape
        if cin != 1 or self.H % 64 or self.W % 64:
            raise ValueError("generator needs a 1-channel image with H and W multiples of 64")
        a, specs = self.a, []
        c = cin
        for i, f in enumerate(a["enc"], 1):
            specs += [Spec(f"e{i}.kernel", "conv", (4, 4, c, f)), Spec(f"e{i}.bias", "vec", (f,))]
            if i > 1:
                specs += _bn_specs(f"e{i}", f)
            c = f
        specs += [Spec("b.kernel", "conv", (4, 4, c, a["bott"])), Spec("b.bias", "vec", (a["bott"],))]
        c = a["bott"]
        for i, ((f, _d), s) in enumerate(zip(a["dec"], a["enc"][::-1]), 1):
            specs += [Spec(f"d{i}.kernel", "convT", (4, 4, f, c)), Spec(f"d{i}.bias", "vec", (f,))]
            specs += _bn_specs(f"d{i}", f)
            c = f + s
        specs += [Spec("out.kernel", "convT", (4, 4, 1, c)), Spec("out.bias", "vec", (1,))]
        super().__init__(specs, device, dtype, seed)

    def dropout_shapes(self, B: int):
        h, w, out = self.H // 64, self.W // 64, []
        for f, drop in self.a["dec"]:
            h, w = h * 2, w * 2
            if drop:
                out.append((B, h, w, f))
        return out

    def forward(self, x: torch.Tensor, masks: Optional[List[torch.Tensor]] = None, seed: int = 0, sample0: int = 0,
                want_logits: bool = False, keep: bool = False):
        """x: fp32 (B,H,W,1) device tensor.  Dropout (always active, generator.py:36): explicit uint8
        `masks` (one per dropout layer, parity tests) or the kernel RNG keyed by (seed, sample0)."""
        P, a, B = self.params, self.a, x.shape[0]
        pl = self.plan(B)
        enc, dec = a["enc"], a["dec"]
        skips = enc[::-1]
        h, w = self.H, self.W
        # concat buffers of the decoder: cat_i = [d_i | skip e_{6-i}]
        cats = []
        hh, ww = self.H // 64, self.W // 64
        for i, ((f, _d), s) in enumerate(zip(dec, skips), 1):
            hh, ww = hh * 2, ww * 2
            cats.append(self.buf(pl, f"cat{i}", (B, hh, ww, f + s)))
        # ---- encoder
        h, w = h // 2, w // 2
        e = self.buf(pl, "e1", (B, h, w, enc[0]))
        ops.conv_first(x, None, P.m("e1.kernel"), P.m("e1.bias"), enc[0], 4, 2, ACT_LEAKY, self.dtype, out=e,
                       out2=cats[4][..., dec[4][0]:], act2=ACT_RELU)
        for i in range(2, 6):
            f = enc[i - 1]
            h, w = h // 2, w // 2
            raw = self.buf(pl, f"raw_e{i}", (B, h, w, f), self.raw_dtype(h, w, f))
            ops.conv_forward(e, P.c(f"e{i}.kernel"), P.m(f"e{i}.bias"), CONV_K4S2, f, out=raw)
            e = self.buf(pl, f"e{i}", (B, h, w, f))
            ops.bn_act(raw, P.m(f"e{i}.bn.gamma"), P.m(f"e{i}.bn.beta"), ACT_LEAKY, out=e,
                       out2=cats[5 - i][..., dec[5 - i][0]:], act2=ACT_RELU, eps=BN_EPS,
                       mean=self.buf(pl, f"mean_e{i}", (B, f), torch.float32),
                       var=self.buf(pl, f"var_e{i}", (B, f), torch.float32))
        # ---- bottleneck (no BN): conv + ReLU
        h, w = h // 2, w // 2
        g = self.buf(pl, "b", (B, h, w, a["bott"]))
        ops.conv_forward(e, P.c("b.kernel"), P.m("b.bias"), CONV_K4S2, a["bott"], act=ACT_RELU, out=g)
        # ---- decoder
        mi = 0
        for i, (f, drop) in enumerate(dec, 1):
            h, w = h * 2, w * 2
            raw = self.buf(pl, f"raw_d{i}", (B, h, w, f), self.raw_dtype(h, w, f))
            ops.conv_forward(g, P.c(f"d{i}.kernel"), P.m(f"d{i}.bias"), CONVT_K4S2, f, out=raw)
            mask, rate = None, 0.0
            if drop:
                rate = DROP_RATE
                if masks is not None:
                    mask = masks[mi]
                mi += 1
            ops.bn_act(raw, P.m(f"d{i}.bn.gamma"), P.m(f"d{i}.bn.beta"), ACT_RELU, out=cats[i - 1][..., :f],
                       drop_mask=mask, drop_rate=rate, drop_seed=seed * 16 + i, sample0=sample0, eps=BN_EPS,
                       mean=self.buf(pl, f"mean_d{i}", (B, f), torch.float32),
                       var=self.buf(pl, f"var_d{i}", (B, f), torch.float32))
            g = cats[i - 1]
        return ops.convT_out(g, P.c("out.kernel"), P.m("out.bias"), apply_sigmoid=True, want_logits=want_logits)


class Discriminator(_Net):
    def __init__(self, image_shape=(64, 512, 1), device="cuda", dtype=torch.bfloat16, seed: int = 1):
        self.H, self.W, cin = image_shape
        specs, c = [], 2 * cin
        for i, (f, _s, bn) in enumerate(DISC, 1):
            specs += [Spec(f"c{i}.kernel", "conv", (4, 4, c, f)), Spec(f"c{i}.bias", "vec", (f,))]
            if bn:
                specs += _bn_specs(f"c{i}", f)
            c = f
        specs += [Spec("c6.kernel", "conv", (4, 4, c, 1)), Spec("c6.bias", "vec", (1,))]
        super().__init__(specs, device, dtype, seed)

    @property
    def patch_shape(self):
        return (self.H // 16, self.W // 16, 1)

    def forward(self, src: torch.Tensor, tgt: torch.Tensor, training: bool, want_logits: bool = False):
        """src, tgt: fp32 (B,H,W,1).  training=True: batch statistics (per sample); False: moving statistics."""
        P, B = self.params, src.shape[0]
        pl = self.plan(B)
        h, w = self.H // 2, self.W // 2
        d = self.buf(pl, "c1", (B, h, w, DISC[0][0]))
        ops.conv_first(src, tgt, P.m("c1.kernel"), P.m("c1.bias"), DISC[0][0], 4, 2, ACT_LEAKY, self.dtype, out=d)
        for i in range(2, 6):
            f, s, _bn = DISC[i - 1]
            if s == 2:
                h, w = h // 2, w // 2
            raw = self.buf(pl, f"raw_c{i}", (B, h, w, f), self.raw_dtype(h, w, f))
            ops.conv_forward(d, P.c(f"c{i}.kernel"), P.m(f"c{i}.bias"), CONV_K4S2 if s == 2 else CONV_K4S1, f, out=raw)
            d = self.buf(pl, f"c{i}", (B, h, w, f))
            if training:
                ops.bn_act(raw, P.m(f"c{i}.bn.gamma"), P.m(f"c{i}.bn.beta"), ACT_LEAKY, out=d, eps=BN_EPS,
                           mean=self.buf(pl, f"mean_c{i}", (B, f), torch.float32),
                           var=self.buf(pl, f"var_c{i}", (B, f), torch.float32))
            else:
                ops.bn_act(raw, P.m(f"c{i}.bn.gamma"), P.m(f"c{i}.bn.beta"), ACT_LEAKY, out=d, eps=BN_EPS,
                           mean=P.m(f"c{i}.bn.moving_mean"), var=P.m(f"c{i}.bn.moving_var"), use_moving=True)
        logits = self.buf(pl, "logits", (B, h, w, 1), torch.float32)
        ops.conv_forward(d, P.c("c6.kernel"), P.m("c6.bias"), CONV_K4S1, 1, act=ACT_NONE, out=logits)
        if want_logits:
            return logits
        out = self.buf(pl, "patch", (B, h, w, 1), torch.float32)
        ops.conv_forward(d, P.c("c6.kernel"), P.m("c6.bias"), CONV_K4S1, 1, act=ACT_SIGMOID, out=out)
        return out


class CnnRegressor(_Net):
    def __init__(self, image_shape=(64, 512, 1), device="cuda", dtype=torch.bfloat16, seed: int = 2):
        self.H, self.W, cin = image_shape
        if cin != 1 or self.H % 8 or self.W % 8:
            raise ValueError("CNN needs a 1-channel image with H and W multiples of 8")
        flat = (self.H // 8) * (self.W // 8) * 512
        specs = []
        for name, ci, co in (("g1", cin, 64), ("g3", 64, 128), ("g4", 128, 256), ("g6", 256, 512)):
            specs += [Spec(name + ".kernel", "conv", (3, 3, ci, co)), Spec(name + ".bias", "vec", (co,))]
        for name, ci, co in (("fc1", flat, 256), ("fc2", 256, 128), ("fc3", 128, 1)):
            specs += [Spec(name + ".kernel", "dense", (ci, co)), Spec(name + ".bias", "vec", (co,))]
        super().__init__(specs, device, dtype, seed)

    def forward(self, x: torch.Tensor, drop_mask: Optional[torch.Tensor] = None):
        """x fp32 (B,H,W,1) -> fp32 (B,1).  Dropout(0.5) after fc1 only when a mask is given (training)."""
        P, B = self.params, x.shape[0]
        pl = self.plan(B)
        H, W = self.H, self.W
        g1 = self.buf(pl, "g1", (B, H, W, 64))
        ops.conv_first(x, None, P.m("g1.kernel"), P.m("g1.bias"), 64, 3, 1, ACT_LEAKY, self.dtype, out=g1)
        g2 = ops.maxpool2(g1)
        g3 = self.buf(pl, "g3", (B, H // 2, W // 2, 128))
        ops.conv_forward(g2, P.c("g3.kernel"), P.m("g3.bias"), CONV_K3S1, 128, act=ACT_RELU, out=g3)
        g4 = self.buf(pl, "g4", (B, H // 2, W // 2, 256))
        ops.conv_forward(g3, P.c("g4.kernel"), P.m("g4.bias"), CONV_K3S1, 256, act=ACT_RELU, out=g4)
        g5 = ops.maxpool2(g4)
        g6 = self.buf(pl, "g6", (B, H // 4, W // 4, 512))
        ops.conv_forward(g5, P.c("g6.kernel"), P.m("g6.bias"), CONV_K3S1, 512, act=ACT_RELU, out=g6)
        g7 = ops.maxpool2(g6)
        flat = g7.view(B, -1)  # Flatten in (H,W,C) order == NHWC memory order
        f1 = self.buf(pl, "fc1", (B, 256))
        ops.conv_forward(flat, P.c("fc1.kernel"), P.m("fc1.bias"), DENSE, 256, act=ACT_RELU, out=f1)
        if drop_mask is not None:
            raise NotImplementedError("training-time dropout is applied by the train step")
        f2 = self.buf(pl, "fc2", (B, 128))
        ops.conv_forward(f1, P.c("fc2.kernel"), P.m("fc2.bias"), DENSE, 128, act=ACT_RELU, out=f2)
        y = self.buf(pl, "y", (B, 1), torch.float32)
        ops.conv_forward(f2, P.c("fc3.kernel"), P.m("fc3.bias"), DENSE, 1, act=ACT_NONE, out=y)
        return y
