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

import os

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
FUSE_OUT_BN = os.environ.get("CFD_AMD_FUSE_OUT_BN", "1") != "0"   # inference: the last decoder block's BatchNorm + ReLU run inside the output layer's load
BN_SPLIT_MIN_ELEMS = 1 << 18      # per-sample tensor size from which BatchNorm statistics get their own chip-wide pass


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

    # ---- weight gradients beside the data-gradient chain ------------------------------------------------------------
    # A layer's weight gradient is needed only by the optimiser at the end of the step, while its data gradient heads the
    # chain of the next layer's backward.  With `side_wgrad` the weight-gradient GEMMs go to a second stream: they fill
    # the CUs that the short elementwise / split-K epilogue kernels of the chain leave idle.  Their operands (the layer's
    # input activation and `draw_*`, one buffer per layer) are not rewritten before the next forward.
    def wgrad_begin(self, side: bool) -> None:
        self._wg_side = bool(side)
        if side and getattr(self, "_wg_stream", None) is None:
            self._wg_stream = ops.side_stream("wgrad")

    def wgrad(self, fn) -> None:
        """run `fn` (weight-gradient launches) behind everything enqueued so far, on the side stream if enabled"""
        if not self._wg_side:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._wg_stream):
            self._wg_stream.wait_event(ev)
            ops._ws_lane = 1               # its own split-K / split-M scratch: these launches overlap the main stream's
            try:
                fn()
            finally:
                ops._ws_lane = 0

    def wgrad_join(self) -> None:
        """the caller's stream waits for every weight gradient launched so far"""
        if self._wg_side:
            torch.cuda.current_stream().wait_stream(self._wg_stream)

    def bn_stats(self, pl: dict, B: int) -> dict:
        """zeroed (sum, sumsq) fp32 (B, C) pairs, one per BatchNorm layer (slices of one flat buffer, one memset):
        the conv epilogue accumulates the per-sample statistics into them where the shape allows it"""
        buf = pl.get("bn_stats")
        if buf is None:
            tot = sum(2 * B * f for _n, f in self.bn_layers)
            buf = pl["bn_stats"] = torch.empty(tot, dtype=torch.float32, device=self.device)
            off, views, flat = 0, {}, {}
            for n, f in self.bn_layers:
                views[n] = (buf[off:off + B * f].view(B, f), buf[off + B * f:off + 2 * B * f].view(B, f))
                flat[n] = buf[off:off + 2 * B * f]
                off += 2 * B * f
            pl["bn_stats_views"], pl["bn_stats_flat"] = views, flat
        buf.zero_()
        return pl["bn_stats_views"]

    def bn_bwd_sums(self, pl: dict, B: int) -> dict:
        """the same flat buffer, zeroed again, as the (2, B, C) partial-sum scratch of the BatchNorm backward (the
        forward statistics are already condensed into mean / var by then)"""
        self.bn_stats(pl, B)
        return pl["bn_stats_flat"]

    def conv_bn(self, pl: dict, st: dict, name: str, x: torch.Tensor, kind: int, f: int, raw: torch.Tensor, act: int,
                out: torch.Tensor, **kw) -> None:
        """conv -> BatchNorm(per-sample batch statistics) -> [dropout] -> activation(s).  Statistics come from the
        conv epilogue (one read of `raw`) when the shape is fusable, else from bn_act's own first pass."""
        P, B = self.params, raw.shape[0]
        _y, fused = ops.conv_forward(x, P.c(name + ".kernel"), P.m(name + ".bias"), kind, f, out=raw, stats=st[name])
        mean = self.buf(pl, "mean_" + name, (B, f), torch.float32)
        var = self.buf(pl, "var_" + name, (B, f), torch.float32)
        g, b = P.m(name + ".bn.gamma"), P.m(name + ".bn.beta")
        _B, h, w, _c = raw.shape
        if not fused and f % 64 == 0 and 2048 % f == 0 and h * w * f >= BN_SPLIT_MIN_ELEMS and B * (f // 64) < 128:
            # small batch, large layer (the conv ran split-K, so its epilogue could not take the statistics): a
            # pixel-parallel statistics pass instead of bn_act's one block per (sample, 64 channels)
            ops.bn_stats(raw, st[name])
            fused = True
        if fused:
            ops.bn_apply(raw, st[name], g, b, act, out=out, mean=mean, var=var, eps=BN_EPS, **kw)
        else:
            ops.bn_act(raw, g, b, act, out=out, mean=mean, var=var, eps=BN_EPS, **kw)

    def check_image(self, *imgs: torch.Tensor) -> None:
        """every image operand must be the fp32 (B, H, W, 1) grid this net was built for: the layer buffers are sized
        from (H, W) and the kernels do not bounds-check"""
        self.params.join_updates()      # (every forward / backward starts here) parameter all-gathers still in flight
        B = imgs[0].shape[0]
        for t in imgs:
            if tuple(t.shape) != (B, self.H, self.W, 1) or t.dtype != torch.float32 or not t.is_cuda:
                raise ValueError(f"expected a float32 device tensor of shape ({B}, {self.H}, {self.W}, 1), got "
                                 f"{tuple(t.shape)} {t.dtype} on {t.device} (build the net with image_shape=...)")

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
        self.bn_layers = [(f"e{i}", f) for i, f in enumerate(a["enc"], 1) if i > 1] + \
                         [(f"d{i}", f) for i, (f, _d) in enumerate(a["dec"], 1)]
        super().__init__(specs, device, dtype, seed)
        self.params.no_transpose = frozenset({"e1.kernel", "out.kernel"})
        self.params.stored_grads = frozenset(n for n in self.params.specs
                                             if n.endswith(".kernel") and n not in ("e1.kernel", "out.kernel"))

    def dropout_shapes(self, B: int):
        h, w, out = self.H // 64, self.W // 64, []
        for f, drop in self.a["dec"]:
            h, w = h * 2, w * 2
            if drop:
                out.append((B, h, w, f))
        return out

    def forward(self, x: torch.Tensor, masks: Optional[List[torch.Tensor]] = None, seed: int = 0, sample0: int = 0,
                want_logits: bool = False, keep: bool = False, inference: bool = False):
        """x: fp32 (B,H,W,1) device tensor.  Dropout (always active, generator.py:36): explicit uint8
        `masks` (one per dropout layer, parity tests) or the kernel RNG keyed by (seed, sample0).
        inference=True (no backward will follow): the first encoder activation is written ONCE, as its LeakyReLU copy
        inside the last concat buffer; e2 reads it there and the output layer rectifies the buffer as it loads it
        (ReLU after Concatenate, generator.py:38-40) -- same values, one 0.5 GB write less per 128 samples (U)."""
        self.check_image(x)
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
        st = self.bn_stats(pl, B)
        # ---- encoder
        h, w = h // 2, w // 2
        if inference:
            e = cats[4][..., dec[4][0]:]
            ops.conv_first(x, None, P.m("e1.kernel"), P.m("e1.bias"), enc[0], 4, 2, ACT_LEAKY, self.dtype, out=e)
        else:
            e = self.buf(pl, "e1", (B, h, w, enc[0]))
            ops.conv_first(x, None, P.m("e1.kernel"), P.m("e1.bias"), enc[0], 4, 2, ACT_LEAKY, self.dtype, out=e,
                           out2=cats[4][..., dec[4][0]:], act2=ACT_RELU)
        for i in range(2, 6):
            f = enc[i - 1]
            h, w = h // 2, w // 2
            raw = self.buf(pl, f"raw_e{i}", (B, h, w, f), self.raw_dtype(h, w, f))
            e_out = self.buf(pl, f"e{i}", (B, h, w, f))
            self.conv_bn(pl, st, f"e{i}", e, CONV_K4S2, f, raw, ACT_LEAKY, e_out,
                         out2=cats[5 - i][..., dec[5 - i][0]:], act2=ACT_RELU)
            e = e_out
        # ---- bottleneck (no BN): conv + ReLU
        h, w = h // 2, w // 2
        g = self.buf(pl, "b", (B, h, w, a["bott"]))
        ops.conv_forward(e, P.c("b.kernel"), P.m("b.bias"), CONV_K4S2, a["bott"], act=ACT_RELU, out=g)
        # ---- decoder
        mi = 0
        for i, (f, drop) in enumerate(dec, 1):
            h, w = h * 2, w * 2
            raw = self.buf(pl, f"raw_d{i}", (B, h, w, f), self.raw_dtype(h, w, f))
            mask, rate = None, 0.0
            if drop:
                rate = DROP_RATE
                if masks is not None:
                    mask = masks[mi]
                mi += 1
            if FUSE_OUT_BN and inference and i == len(dec) and not drop and f == skips[i - 1] and f in (64, 128, 256) and (h * w) % 16 == 0:
                # the last block's BatchNorm + ReLU happen inside the output layer as it loads `raw`: the normalised
                # tensor (the largest activation of the net) is neither written nor read back
                name = f"d{i}"
                _y, fused = ops.conv_forward(g, P.c(name + ".kernel"), P.m(name + ".bias"), CONVT_K4S2, f, out=raw,
                                             stats=st[name])
                if not fused:
                    ops.bn_stats(raw, st[name])
                return ops.convT_out_bn(raw, st[name], P.m(name + ".bn.gamma"), P.m(name + ".bn.beta"),
                                        cats[i - 1][..., f:], P.c("out.kernel"), P.m("out.bias"), apply_sigmoid=True,
                                        want_logits=want_logits, eps=BN_EPS)
            self.conv_bn(pl, st, f"d{i}", g, CONVT_K4S2, f, raw, ACT_RELU, cats[i - 1][..., :f],
                         drop_mask=mask, drop_rate=rate, drop_seed=seed * 16 + i, sample0=sample0)
            g = cats[i - 1]
        return ops.convT_out(g, P.c("out.kernel"), P.m("out.bias"), apply_sigmoid=True, want_logits=want_logits,
                             relu_on_load=inference)

    def backward(self, dz: torch.Tensor, x: torch.Tensor, masks: Optional[List[torch.Tensor]] = None, seed: int = 0,
                 sample0: int = 0, ready=None, side_wgrad: bool = False) -> None:
        """Backward of the forward that JUST ran with the same (x, masks, seed, sample0): dz = dL/dlogits fp32
        (B,H,W,1).  Writes every trainable gradient of the generator into params.grad (bias gradients of
        convolutions followed by BatchNorm are identically zero and stay at the zero the step memset wrote).
        `ready(layer)` (optional) is called as soon as a layer's kernel and bias gradients are complete, last layer
        first: the data-parallel trainer starts the all-reduce of that part of the flat buffer there."""
        ready = ready or (lambda _name: None)
        self.check_image(x, dz)
        P, a, B = self.params, self.a, x.shape[0]
        P.wait_transposed()
        self.wgrad_begin(side_wgrad)
        pl = self.plan(B)
        sums = self.bn_bwd_sums(pl, B)
        enc, dec = a["enc"], a["dec"]
        skips = enc[::-1]
        dt = self.dtype
        # ---- output layer (Conv2DTranspose to one channel): taps family
        cat5 = pl["cat5"]
        _, h5, w5, c5 = cat5.shape
        dtaps = self.buf(pl, "dtaps", (B * h5 * w5, 16), torch.float32)
        ops.tap_scatter_grad(dz, dtaps, B, h5, w5, 0)
        ops.tap_dw(dtaps, cat5, P.g("out.kernel"), 16)
        ops.colsum(dz.view(-1, 1), P.g("out.bias"))
        ready("out")
        dcat = self.buf(pl, "dcat5", cat5.shape)
        ops.tap_dx(dtaps, P.c("out.kernel"), dcat, 16)
        # ---- decoder, last block first
        nd = len(dec)
        mi = sum(1 for _f, d in dec if d)
        skip_grads = {}
        for i in range(nd, 0, -1):
            f, drop = dec[i - 1]
            cat_i = pl[f"cat{i}"]
            raw = pl[f"raw_d{i}"]
            skip_grads[6 - i] = dcat[..., f:]          # gradient of the ReLU copy of e_{6-i}
            mask, rate = None, 0.0
            if drop:
                mi -= 1
                rate = DROP_RATE
                if masks is not None:
                    mask = masks[mi]
            draw = self.buf(pl, f"draw_d{i}", raw.shape)
            ops.bn_act_bwd(raw, pl[f"mean_d{i}"], pl[f"var_d{i}"], P.m(f"d{i}.bn.gamma"), P.m(f"d{i}.bn.beta"),
                           dcat[..., :f], ACT_RELU, draw, P.g(f"d{i}.bn.gamma"), P.g(f"d{i}.bn.beta"),
                           drop_mask=mask, drop_rate=rate, drop_seed=seed * 16 + i, sample0=sample0, eps=BN_EPS,
                           sums=sums[f"d{i}"])
            g_in = pl[f"cat{i - 1}"] if i > 1 else pl["b"]
            self.wgrad(lambda: ops.conv_wgrad(g_in, draw, CONVT_K4S2, f, P.g(f"d{i}.kernel")))
            ready(f"d{i}")
            dnext = self.buf(pl, f"dcat{i - 1}" if i > 1 else "db", g_in.shape)
            ops.conv_forward(draw, P.t(f"d{i}.kernel"), None, CONV_K4S2, g_in.shape[-1], out=dnext)
            dcat = dnext
        # ---- bottleneck: conv + ReLU (no BN)
        b = pl["b"]
        dpre = self.buf(pl, "dpre_b", b.shape)
        ops.act_bwd(b, dcat, ACT_RELU, dpre, dbias=P.g("b.bias"))       # (+ the bias gradient, in the same pass)
        e5 = pl["e5"]
        self.wgrad(lambda: ops.conv_wgrad(e5, dpre, CONV_K4S2, a["bott"], P.g("b.kernel")))
        ready("b")
        de = self.buf(pl, "de5", e5.shape)
        ops.conv_forward(dpre, P.t("b.kernel"), None, CONVT_K4S2, e5.shape[-1], out=de)
        # ---- encoder, deepest first: gradient arrives through the next conv (LeakyReLU) and the skip (ReLU)
        for i in range(5, 1, -1):
            f = enc[i - 1]
            raw = pl[f"raw_e{i}"]
            draw = self.buf(pl, f"draw_e{i}", raw.shape)
            ops.bn_act_bwd(raw, pl[f"mean_e{i}"], pl[f"var_e{i}"], P.m(f"e{i}.bn.gamma"), P.m(f"e{i}.bn.beta"),
                           de, ACT_LEAKY, draw, P.g(f"e{i}.bn.gamma"), P.g(f"e{i}.bn.beta"),
                           g2=skip_grads[i], act2=ACT_RELU, eps=BN_EPS, sums=sums[f"e{i}"])
            e_prev = pl[f"e{i - 1}"]
            self.wgrad(lambda: ops.conv_wgrad(e_prev, draw, CONV_K4S2, f, P.g(f"e{i}.kernel")))
            ready(f"e{i}")
            de = self.buf(pl, f"de{i - 1}", e_prev.shape)
            ops.conv_forward(draw, P.t(f"e{i}.kernel"), None, CONVT_K4S2, e_prev.shape[-1], out=de)
        e1 = pl["e1"]
        dpre1 = self.buf(pl, "dpre_e1", e1.shape)
        ops.act_bwd(e1, de, ACT_LEAKY, dpre1, g2=skip_grads[1], act2=ACT_RELU)
        ops.conv_first_wgrad(x, None, dpre1, P.g("e1.kernel"), P.g("e1.bias"), 4, 2)
        self.wgrad_join()
        ready("e1")


class Discriminator(_Net):
    def __init__(self, image_shape=(64, 512, 1), device="cuda", dtype=torch.bfloat16, seed: int = 1):
        self.H, self.W, cin = image_shape
        if cin != 1 or self.H % 16 or self.W % 16:
            raise ValueError("discriminator needs a 1-channel image with H and W multiples of 16")
        specs, c = [], 2 * cin
        for i, (f, _s, bn) in enumerate(DISC, 1):
            specs += [Spec(f"c{i}.kernel", "conv", (4, 4, c, f)), Spec(f"c{i}.bias", "vec", (f,))]
            if bn:
                specs += _bn_specs(f"c{i}", f)
            c = f
        specs += [Spec("c6.kernel", "conv", (4, 4, c, 1)), Spec("c6.bias", "vec", (1,))]
        self.bn_layers = [(f"c{i}", f) for i, (f, _s, bn) in enumerate(DISC, 1) if bn]
        super().__init__(specs, device, dtype, seed)
        self.params.no_transpose = frozenset({"c1.kernel", "c6.kernel"})
        self.params.flip_taps = frozenset({"c5.kernel"})
        self.params.stored_grads = frozenset(f"c{i}.kernel" for i in range(2, 6))

    @property
    def patch_shape(self):
        return (self.H // 16, self.W // 16, 1)

    def forward(self, src: torch.Tensor, tgt: torch.Tensor, training: bool, want_logits: bool = False):
        """src, tgt: fp32 (B,H,W,1).  training=True: batch statistics (per sample); False: moving statistics."""
        self.check_image(src, tgt)
        P, B = self.params, src.shape[0]
        pl = self.plan(B)
        h, w = self.H // 2, self.W // 2
        d = self.buf(pl, "c1", (B, h, w, DISC[0][0]))
        ops.conv_first(src, tgt, P.m("c1.kernel"), P.m("c1.bias"), DISC[0][0], 4, 2, ACT_LEAKY, self.dtype, out=d)
        st = self.bn_stats(pl, B) if training else None
        for i in range(2, 6):
            f, s, _bn = DISC[i - 1]
            if s == 2:
                h, w = h // 2, w // 2
            raw = self.buf(pl, f"raw_c{i}", (B, h, w, f), self.raw_dtype(h, w, f))
            kind = CONV_K4S2 if s == 2 else CONV_K4S1
            d_in, d = d, self.buf(pl, f"c{i}", (B, h, w, f))
            if training:
                self.conv_bn(pl, st, f"c{i}", d_in, kind, f, raw, ACT_LEAKY, d)
            else:
                ops.conv_forward(d_in, P.c(f"c{i}.kernel"), P.m(f"c{i}.bias"), kind, f, out=raw)
                ops.bn_act(raw, P.m(f"c{i}.bn.gamma"), P.m(f"c{i}.bn.beta"), ACT_LEAKY, out=d, eps=BN_EPS,
                           mean=P.m(f"c{i}.bn.moving_mean"), var=P.m(f"c{i}.bn.moving_var"), use_moving=True)
        out = self.buf(pl, "logits" if want_logits else "patch", (B, h, w, 1), torch.float32)
        ops.conv_forward(d, P.c("c6.kernel"), P.m("c6.bias"), CONV_K4S1, 1, act=ACT_NONE if want_logits else ACT_SIGMOID,
                         out=out)
        return out

    def update_moving(self, B: int) -> None:
        """moving statistics <- batch statistics of the forward that just ran (momentum 0.99)"""
        pl = self.plan(B)
        ops.bn_update_moving_multi([(self.params.m(f"c{i}.bn.moving_mean"), self.params.m(f"c{i}.bn.moving_var"),
                                     pl[f"mean_c{i}"], pl[f"var_c{i}"]) for i in range(2, 6)])

    def backward(self, dlogits: torch.Tensor, src: torch.Tensor, tgt: torch.Tensor, train_convs: bool = True,
                 want_input_grad: bool = False, side_wgrad: bool = False):
        """Backward of the training-mode forward that JUST ran on (src, tgt).  train_convs=False is the
        composite-GAN case: only BatchNorm gamma/beta receive gradients (U/models/generator.py:79-81) and the
        gradient w.r.t. the target image (the generator's output) is returned as fp32 (B,H,W,1)."""
        self.check_image(src, tgt)
        P, B = self.params, src.shape[0]
        if tuple(dlogits.shape) != (B,) + self.patch_shape:
            raise ValueError(f"dlogits must be {(B,) + self.patch_shape}, got {tuple(dlogits.shape)}")
        P.wait_transposed()
        self.wgrad_begin(side_wgrad and train_convs)
        pl = self.plan(B)
        sums = self.bn_bwd_sums(pl, B)
        c5 = pl["c5"]
        _, h, w, cc = c5.shape
        dtaps = self.buf(pl, "dtaps", (B * h * w, 16), torch.float32)
        ops.tap_scatter_grad(dlogits, dtaps, B, h, w, 1)
        if train_convs:
            ops.tap_dw(dtaps, c5, P.g("c6.kernel"), 16)
            ops.colsum(dlogits.view(-1, 1), P.g("c6.bias"))
        dc = self.buf(pl, "dc5", c5.shape)
        ops.tap_dx(dtaps, P.c("c6.kernel"), dc, 16)
        for i in range(5, 1, -1):
            f, s, _bn = DISC[i - 1]
            raw = pl[f"raw_c{i}"]
            draw = self.buf(pl, f"draw_c{i}", raw.shape)
            ops.bn_act_bwd(raw, pl[f"mean_c{i}"], pl[f"var_c{i}"], P.m(f"c{i}.bn.gamma"), P.m(f"c{i}.bn.beta"),
                           dc, ACT_LEAKY, draw, P.g(f"c{i}.bn.gamma"), P.g(f"c{i}.bn.beta"), eps=BN_EPS,
                           sums=sums[f"c{i}"])
            prev = pl[f"c{i - 1}"]
            kind = CONV_K4S2 if s == 2 else CONV_K4S1
            if train_convs:
                self.wgrad(lambda: ops.conv_wgrad(prev, draw, kind, f, P.g(f"c{i}.kernel")))
            dc = self.buf(pl, f"dc{i - 1}", prev.shape)
            ops.conv_forward(draw, P.t(f"c{i}.kernel"), None, CONVT_K4S2 if s == 2 else ops.CONV_K4S1_BWD,
                             prev.shape[-1], out=dc)
        c1 = pl["c1"]
        dpre1 = self.buf(pl, "dpre_c1", c1.shape)
        ops.act_bwd(c1, dc, ACT_LEAKY, dpre1)
        if train_convs:
            ops.conv_first_wgrad(src, tgt, dpre1, P.g("c1.kernel"), P.g("c1.bias"), 4, 2)
        self.wgrad_join()
        if not want_input_grad:
            return None
        # d(target image): a transposed conv of dpre1 to ONE channel with w[tap][co] = W[co][tap][ci=1]
        wt = self.buf(pl, "w_c1_tgt", (16, DISC[0][0]))
        ops.cast_copy(P.m("c1.kernel")[..., 1].permute(1, 2, 0).reshape(1, 1, 16, DISC[0][0]).contiguous(),
                      wt.view(1, 1, 16, DISC[0][0]))
        return ops.convT_out(dpre1, wt, None, apply_sigmoid=False)


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
        self.params.no_transpose = frozenset({"g1.kernel", "fc3.kernel"})
        self.params.stored_grads = frozenset(n + ".kernel" for n in ("g3", "g4", "g6", "fc1", "fc2"))

    def forward(self, x: torch.Tensor, training: bool = False, drop_mask: Optional[torch.Tensor] = None,
                seed: int = 0, sample0: int = 0):
        """x fp32 (B,H,W,1) -> fp32 (B,1).  Dropout(0.5) after fc1 only in training (explicit uint8 (B,256)
        mask, or the kernel RNG keyed by (seed, sample0))."""
        self.check_image(x)
        P, B = self.params, x.shape[0]
        pl = self.plan(B)
        H, W = self.H, self.W
        if training:
            g1 = self.buf(pl, "g1", (B, H, W, 64))
            ops.conv_first(x, None, P.m("g1.kernel"), P.m("g1.bias"), 64, 3, 1, ACT_LEAKY, self.dtype, out=g1)
            g2 = ops.maxpool2(g1, out=self.buf(pl, "g2", (B, H // 2, W // 2, 64)))
        else:          # inference: conv + LeakyReLU + pool in one pass, the full-resolution activation is never written
            g2 = self.buf(pl, "g2i", (B, H // 2, W // 2, 64))
            ops.conv3_pool2(x, P.m("g1.kernel"), P.m("g1.bias"), 64, ACT_LEAKY, self.dtype, out=g2)
        g3 = self.buf(pl, "g3", (B, H // 2, W // 2, 128))
        ops.conv_forward(g2, P.c("g3.kernel"), P.m("g3.bias"), CONV_K3S1, 128, act=ACT_RELU, out=g3)
        # inference at batch sizes that fill the chip without split-K: conv + ReLU + pool in one launch (the unpooled
        # g4 / g6 never reach HBM); training keeps them for the backward
        g5 = self.buf(pl, "g5", (B, H // 4, W // 4, 256))
        if not training and ops.conv_pool2_fusable(g3, CONV_K3S1, 256, ACT_RELU):
            ops.conv_forward_pool2(g3, P.c("g4.kernel"), P.m("g4.bias"), CONV_K3S1, 256, act=ACT_RELU, out=g5)
        else:
            g4 = self.buf(pl, "g4", (B, H // 2, W // 2, 256))
            if training and ops.conv_pool2_keep_fusable(g3, CONV_K3S1, 256, ACT_RELU):
                # training: the pooled tensor is written next to the unpooled one from the same output tile
                ops.conv_forward_pool2_keep(g3, P.c("g4.kernel"), P.m("g4.bias"), CONV_K3S1, 256, ACT_RELU, g4, g5)
            else:
                ops.conv_forward(g3, P.c("g4.kernel"), P.m("g4.bias"), CONV_K3S1, 256, act=ACT_RELU, out=g4)
                ops.maxpool2(g4, out=g5)
        g7 = self.buf(pl, "g7", (B, H // 8, W // 8, 512))
        if not training and ops.conv_pool2_fusable(g5, CONV_K3S1, 512, ACT_RELU):
            ops.conv_forward_pool2(g5, P.c("g6.kernel"), P.m("g6.bias"), CONV_K3S1, 512, act=ACT_RELU, out=g7)
        else:
            g6 = self.buf(pl, "g6", (B, H // 4, W // 4, 512))
            if training and ops.conv_pool2_keep_fusable(g5, CONV_K3S1, 512, ACT_RELU):
                ops.conv_forward_pool2_keep(g5, P.c("g6.kernel"), P.m("g6.bias"), CONV_K3S1, 512, ACT_RELU, g6, g7)
            else:
                ops.conv_forward(g5, P.c("g6.kernel"), P.m("g6.bias"), CONV_K3S1, 512, act=ACT_RELU, out=g6)
                ops.maxpool2(g6, out=g7)
        flat = g7.view(B, -1)  # Flatten in (H,W,C) order == NHWC memory order
        f1 = self.buf(pl, "fc1", (B, 256))
        ops.conv_forward(flat, P.c("fc1.kernel"), P.m("fc1.bias"), DENSE, 256, act=ACT_RELU, out=f1)
        f1d = f1
        if training:
            f1d = self.buf(pl, "fc1_drop", (B, 256))
            ops.dropout_apply(f1, f1d, DROP_RATE, mask=drop_mask, seed=seed, elem0=sample0 * 256)
        pl["fc1_in"] = f1d
        f2 = self.buf(pl, "fc2", (B, 128))
        ops.conv_forward(f1d, P.c("fc2.kernel"), P.m("fc2.bias"), DENSE, 128, act=ACT_RELU, out=f2)
        y = self.buf(pl, "y", (B, 1), torch.float32)
        ops.conv_forward(f2, P.c("fc3.kernel"), P.m("fc3.bias"), DENSE, 1, act=ACT_NONE, out=y)
        return y

    def backward(self, dy: torch.Tensor, x: torch.Tensor, drop_mask: Optional[torch.Tensor] = None, seed: int = 0,
                 sample0: int = 0, side_wgrad: bool = False) -> None:
        """Backward of the training forward that JUST ran; dy = dL/dy fp32 (B,1)."""
        self.check_image(x)
        self.wgrad_begin(side_wgrad)
        P, B = self.params, x.shape[0]
        if tuple(dy.shape) != (B, 1):
            raise ValueError(f"dy must be ({B}, 1), got {tuple(dy.shape)}")
        P.wait_transposed()
        pl = self.plan(B)
        dt = self.dtype
        H, W = self.H, self.W
        f2, f1d, f1 = pl["fc2"], pl["fc1_in"], pl["fc1"]
        # fc3 (one output): small-K kernels
        ops.tap_dw(dy, f2.view(B, 1, 1, 128), P.g("fc3.kernel"), 1)
        ops.colsum(dy, P.g("fc3.bias"))
        df2 = self.buf(pl, "dfc2", (B, 128))
        ops.tap_dx(dy, P.c("fc3.kernel"), df2.view(B, 1, 1, 128), 1)
        dpre2 = self.buf(pl, "dpre_fc2", (B, 128))
        ops.act_bwd(f2.view(B, 1, 1, 128), df2.view(B, 1, 1, 128), ACT_RELU, dpre2.view(B, 1, 1, 128), dbias=P.g("fc2.bias"))
        self.wgrad(lambda: ops.conv_wgrad(f1d, dpre2, DENSE, 128, P.g("fc2.kernel")))
        df1d = self.buf(pl, "dfc1d", (B, 256))
        ops.conv_forward(dpre2, P.t("fc2.kernel"), None, DENSE, 256, out=df1d)
        df1 = self.buf(pl, "dfc1", (B, 256))
        ops.dropout_apply(df1d, df1, DROP_RATE, mask=drop_mask, seed=seed, elem0=sample0 * 256)
        dpre1 = self.buf(pl, "dpre_fc1", (B, 256))
        ops.act_bwd(f1.view(B, 1, 1, 256), df1.view(B, 1, 1, 256), ACT_RELU, dpre1.view(B, 1, 1, 256), dbias=P.g("fc1.bias"))
        g7 = pl["g7"]
        flat = g7.view(B, -1)
        self.wgrad(lambda: ops.conv_wgrad(flat, dpre1, DENSE, 256, P.g("fc1.kernel")))
        dg7 = self.buf(pl, "dg7", g7.shape)
        ops.conv_forward(dpre1, P.t("fc1.kernel"), None, DENSE, flat.shape[1], out=dg7.view(B, -1))
        # g6 -> pool: pool backward fused with g6's ReLU backward
        g6, g5, g4, g3, g2, g1 = pl["g6"], pl["g5"], pl["g4"], pl["g3"], pl["g2"], pl["g1"]
        dpre6 = self.buf(pl, "dpre_g6", g6.shape)
        # (bias gradients ride with the pass that writes the pre-activation gradient: a separate column-sum pass re-read
        #  536 MB per B = 64 step at 1.9 TB/s, 6.9 % of the step)
        ops.maxpool2_act_bwd(g6, dg7, dpre6, ACT_RELU, dbias=P.g("g6.bias"))
        self.wgrad(lambda: ops.conv_wgrad(g5, dpre6, CONV_K3S1, 512, P.g("g6.kernel")))
        dg5 = self.buf(pl, "dg5", g5.shape)
        ops.conv_forward(dpre6, P.t("g6.kernel"), None, CONV_K3S1, 256, out=dg5)
        dpre4 = self.buf(pl, "dpre_g4", g4.shape)
        ops.maxpool2_act_bwd(g4, dg5, dpre4, ACT_RELU, dbias=P.g("g4.bias"))
        self.wgrad(lambda: ops.conv_wgrad(g3, dpre4, CONV_K3S1, 256, P.g("g4.kernel")))
        dg3 = self.buf(pl, "dg3", g3.shape)
        ops.conv_forward(dpre4, P.t("g4.kernel"), None, CONV_K3S1, 128, out=dg3)
        dpre3 = self.buf(pl, "dpre_g3", g3.shape)
        ops.act_bwd(g3, dg3, ACT_RELU, dpre3, dbias=P.g("g3.bias"))
        self.wgrad(lambda: ops.conv_wgrad(g2, dpre3, CONV_K3S1, 128, P.g("g3.kernel")))
        dg2 = self.buf(pl, "dg2", g2.shape)
        ops.conv_forward(dpre3, P.t("g3.kernel"), None, CONV_K3S1, 64, out=dg2)
        dpre1c = self.buf(pl, "dpre_g1", g1.shape)
        ops.maxpool2_act_bwd(g1, dg2, dpre1c, ACT_LEAKY)
        ops.conv_first_wgrad(x, None, dpre1c, P.g("g1.kernel"), P.g("g1.bias"), 3, 1)
        self.wgrad_join()
