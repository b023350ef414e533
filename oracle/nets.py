"""PyTorch-CPU fp32 restatement of the reference's Keras graphs (TEST INFRASTRUCTURE).

PARITY UNPINNED (no TensorFlow here, no reference fixtures) -- see oracle/__init__.py.

Follows, as text:
  U generator      /root/reference/U/models/generator.py:14-74
  P generator      /root/reference/P/models/generator.py:39-114
  discriminator    /root/reference/U/models/discriminator.py:12-49
  composite GAN    /root/reference/U/models/generator.py:77-94, P/models/generator.py:117-134
  DeltaP CNN       /root/reference/DeltaP/models/generator.py:15-55
  losses           /root/reference/U/models/loss.py:4-138
  train step order /root/reference/U/train.py:116-125, DeltaP/train.py:129-131

All tensors at the interface are NHWC float32 (numpy or torch), exactly what the
Keras models take and return.  Parameters are kept in KERAS layouts in a flat dict:
  '<layer>.kernel' Conv2D (kh,kw,Cin,Cout) | Conv2DTranspose (kh,kw,Cout,Cin) | Dense (in,out)
  '<layer>.bias'
  '<layer>.bn.gamma' '.bn.beta' '.bn.moving_mean' '.bn.moving_var'
Dropout masks are explicit inputs (0/1 float tensors, NHWC) so both sides of a
parity test draw the same ones; the kept values are scaled by 2.0 here.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import keras_semantics as KS

Params = Dict[str, torch.Tensor]

# --------------------------------------------------------------------------------------
# architecture tables (restated from the reference constructors)
# --------------------------------------------------------------------------------------
# U/models/generator.py:52-70
U_GEN = dict(enc=[256, 512, 512, 1024, 1024], bott=1024,
             dec=[(1024, True), (1024, True), (512, True), (512, False), (256, False)])
# P/models/generator.py:78-99
P_GEN = dict(enc=[128, 256, 256, 512, 1024], bott=2048,
             dec=[(1024, True), (512, False), (256, False), (256, False), (128, False)])
GEN_ARCH = {"U": U_GEN, "P": P_GEN}
# U/models/discriminator.py:22-41  (filters, stride, batchnorm)
DISC = [(128, 2, False), (256, 2, True), (512, 2, True), (1024, 2, True), (1024, 1, True)]


# --------------------------------------------------------------------------------------
# bf16 emulation (parity tests of the bf16 HIP path at STEP level)
# --------------------------------------------------------------------------------------
# The HIP path in bf16 mode keeps activations, inter-layer gradients and the MFMA weight operands in bf16 (fp32
# accumulation, fp32 master weights / statistics / losses).  Against the plain fp32 oracle a whole training step can
# only be compared loosely, and nothing separates "bf16 quantisation" from "kernel error".  Inside `bf16_emulation()`
# this oracle rounds to bf16 at the same points the HIP path stores bf16 -- after every activation, on the pre-BatchNorm
# conv output of the layers above RAW_F32_MAX_ELEMS per sample, on every weight that feeds an MFMA (not e1 / c1 / g1:
# the direct first-layer kernels read the fp32 master), and on the gradients flowing back through those points -- so
# what is left between the two is accumulation order and the single activations whose rounding flips.
# `noise` > 0 is a PROBE, not a model of anything: every stored activation is multiplied by (1 + noise * N(0,1)) before it
# is rounded, i.e. perturbed by a fraction of one bf16 ulp (2^-8 relative).  The change this makes to the oracle's OWN
# gradients measures how strongly the network amplifies single rounding flips (BatchNorm over 8-32 pixels in the
# deepest layers, ReLU / dropout kinks); the tests require the HIP path to sit inside that band.
_EMU = {"on": False, "raw_f32_max": 524288, "noise": 0.0}


class _RoundBoth(torch.autograd.Function):
    """value -> bf16 -> fp32 in the forward, the same on the gradient in the backward (a tensor stored in bf16 whose
    gradient is stored in bf16 too)"""

    @staticmethod
    def forward(ctx, x):
        if _EMU["noise"]:
            gen = torch.Generator().manual_seed(x.numel() % 1000003)
            x = x * (1.0 + _EMU["noise"] * torch.randn(x.shape, generator=gen))
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundFwd(torch.autograd.Function):
    """bf16 operand copy of an fp32 master weight: rounded in the forward, gradient passed through in fp32"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """fp32 value whose GRADIENT is stored in bf16 (the pre-BatchNorm output of a small layer: kept in fp32, but the
    BatchNorm backward writes d(raw) in the compute dtype, and the weight / data gradient GEMMs read it as bf16)"""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def R(x: torch.Tensor) -> torch.Tensor:
    return _RoundBoth.apply(x) if _EMU["on"] else x


def RW(w: torch.Tensor) -> torch.Tensor:
    return _RoundFwd.apply(w) if _EMU["on"] else w


def RPRE(x: torch.Tensor) -> torch.Tensor:
    """pre-activation of a layer without BatchNorm: never stored in the forward, its gradient is stored in bf16"""
    return _RoundBwd.apply(x) if _EMU["on"] else x


def RRAW(x: torch.Tensor) -> torch.Tensor:
    """pre-BatchNorm conv output: bf16 only above the HIP path's threshold (cfd_amd.nets.RAW_F32_MAX_ELEMS)"""
    if not _EMU["on"]:
        return x
    if x.shape[1] * x.shape[2] * x.shape[3] > _EMU["raw_f32_max"]:
        return _RoundBoth.apply(x)
    return _RoundBwd.apply(x)


class bf16_emulation:
    def __init__(self, raw_f32_max: int = 524288, noise: float = 0.0):
        self.raw, self.noise = raw_f32_max, noise

    def __enter__(self):
        self.prev = dict(_EMU)
        _EMU.update(on=True, raw_f32_max=self.raw, noise=self.noise)

    def __exit__(self, *a):
        _EMU.update(self.prev)


def _t(x) -> torch.Tensor:
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(torch.float32)


# --------------------------------------------------------------------------------------
# parameter construction
# --------------------------------------------------------------------------------------
def _normal(gen: torch.Generator, shape) -> torch.Tensor:
    return torch.randn(shape, generator=gen, dtype=torch.float32) * KS.INIT_STDDEV


def _bn_params(p: Params, name: str, c: int, gen: Optional[torch.Generator], jitter: bool):
    # Keras init is gamma=1, beta=0; `jitter` perturbs them so parity tests see non-trivial affine terms
    if jitter and gen is not None:
        p[name + ".bn.gamma"] = 1.0 + 0.1 * torch.randn(c, generator=gen)
        p[name + ".bn.beta"] = 0.1 * torch.randn(c, generator=gen)
    else:
        p[name + ".bn.gamma"] = torch.full((c,), KS.BN_GAMMA_INIT)
        p[name + ".bn.beta"] = torch.full((c,), KS.BN_BETA_INIT)
    p[name + ".bn.moving_mean"] = torch.full((c,), KS.BN_MOVING_MEAN_INIT)
    p[name + ".bn.moving_var"] = torch.full((c,), KS.BN_MOVING_VAR_INIT)


def _bias(gen, c, jitter):
    if jitter:
        return 0.02 * torch.randn(c, generator=gen)
    return torch.zeros(c)


def init_generator(arch: str, seed: int = 0, in_ch: int = 1, jitter: bool = False) -> Params:
    """RandomNormal(0.02) kernels, zero biases (U/models/generator.py:16,46)."""
    a = GEN_ARCH[arch]
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    cin = in_ch
    for i, f in enumerate(a["enc"], 1):
        p[f"e{i}.kernel"] = _normal(g, (4, 4, cin, f))
        p[f"e{i}.bias"] = _bias(g, f, jitter)
        if i > 1:
            _bn_params(p, f"e{i}", f, g, jitter)
        cin = f
    p["b.kernel"] = _normal(g, (4, 4, cin, a["bott"]))
    p["b.bias"] = _bias(g, a["bott"], jitter)
    cin = a["bott"]
    skips = a["enc"][::-1]
    for i, ((f, _drop), s) in enumerate(zip(a["dec"], skips), 1):
        p[f"d{i}.kernel"] = _normal(g, (4, 4, f, cin))
        p[f"d{i}.bias"] = _bias(g, f, jitter)
        _bn_params(p, f"d{i}", f, g, jitter)
        cin = f + s
    p["out.kernel"] = _normal(g, (4, 4, 1, cin))
    p["out.bias"] = _bias(g, 1, jitter)
    return p


def init_discriminator(seed: int = 1, in_ch: int = 2, jitter: bool = False) -> Params:
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    cin = in_ch
    for i, (f, _s, bn) in enumerate(DISC, 1):
        p[f"c{i}.kernel"] = _normal(g, (4, 4, cin, f))
        p[f"c{i}.bias"] = _bias(g, f, jitter)
        if bn:
            _bn_params(p, f"c{i}", f, g, jitter)
        cin = f
    p["c6.kernel"] = _normal(g, (4, 4, cin, 1))
    p["c6.bias"] = _bias(g, 1, jitter)
    return p


def init_cnn(seed: int = 2, image_shape=(64, 512, 1), jitter: bool = False) -> Params:
    """DeltaP/models/generator.py:23-46."""
    g = torch.Generator().manual_seed(seed)
    h, w, c = image_shape
    p: Params = {}
    for name, ci, co in (("g1", c, 64), ("g3", 64, 128), ("g4", 128, 256), ("g6", 256, 512)):
        p[name + ".kernel"] = _normal(g, (3, 3, ci, co))
        p[name + ".bias"] = _bias(g, co, jitter)
    flat = (h // 8) * (w // 8) * 512
    for name, ci, co in (("fc1", flat, 256), ("fc2", 256, 128), ("fc3", 128, 1)):
        p[name + ".kernel"] = _normal(g, (ci, co))
        p[name + ".bias"] = _bias(g, co, jitter)
    return p


# --------------------------------------------------------------------------------------
# layers (NHWC in / NHWC out)
# --------------------------------------------------------------------------------------
def _same_pad(n: int, k: int, s: int) -> Tuple[int, int]:
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv2d_same(x: torch.Tensor, kernel: torch.Tensor, bias: torch.Tensor, stride: int) -> torch.Tensor:
    """Keras Conv2D(padding='same'); kernel (kh,kw,Cin,Cout)."""
    kh, kw = kernel.shape[0], kernel.shape[1]
    pt, pb = _same_pad(x.shape[1], kh, stride)
    pl, pr = _same_pad(x.shape[2], kw, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, kernel.permute(3, 2, 0, 1), bias, stride=stride)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose_same(x: torch.Tensor, kernel: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """Keras Conv2DTranspose(4x4, strides 2, 'same'); kernel (kh,kw,Cout,Cin)."""
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), kernel.permute(3, 2, 0, 1), bias, stride=2, padding=1)
    return y.permute(0, 2, 3, 1)


def batchnorm(x: torch.Tensor, p: Params, name: str, training: bool, per_sample: bool,
              stats_out: Optional[dict] = None) -> torch.Tensor:
    """BatchNormalization(axis=-1).  training=True: batch statistics (biased variance).
    per_sample=True computes them over (H,W) of each sample separately -- identical to the
    reference, which only ever runs batch 1 (U/train.py:39, U/valid.py:29)."""
    gamma, beta = p[name + ".bn.gamma"], p[name + ".bn.beta"]
    if training:
        dims = (1, 2) if per_sample else (0, 1, 2)
        mean = x.mean(dim=dims, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=dims, keepdim=True)
        if stats_out is not None:
            stats_out[name] = (mean.detach().reshape(-1, x.shape[-1]), var.detach().reshape(-1, x.shape[-1]))
    else:
        mean, var = p[name + ".bn.moving_mean"], p[name + ".bn.moving_var"]
    return (x - mean) * torch.rsqrt(var + KS.BN_EPSILON) * gamma + beta


def update_moving(p: Params, stats: dict) -> None:
    """moving = moving*0.99 + batch*0.01, applied once per sample-group, in order."""
    for name, (mean, var) in stats.items():
        for i in range(mean.shape[0]):
            p[name + ".bn.moving_mean"] = p[name + ".bn.moving_mean"] * KS.BN_MOMENTUM + mean[i] * (1 - KS.BN_MOMENTUM)
            p[name + ".bn.moving_var"] = p[name + ".bn.moving_var"] * KS.BN_MOMENTUM + var[i] * (1 - KS.BN_MOMENTUM)


def leaky(x):
    return F.leaky_relu(x, KS.LEAKY_SLOPE)


# --------------------------------------------------------------------------------------
# the four graphs
# --------------------------------------------------------------------------------------
def generator_dropout_shapes(arch: str, batch: int, image_shape=(64, 512, 1)) -> List[Tuple[int, ...]]:
    """Shapes of the dropout masks generator_forward expects, in order d1, d2, ..."""
    a = GEN_ARCH[arch]
    h, w = image_shape[0] // 64, image_shape[1] // 64
    shapes = []
    for f, drop in a["dec"]:
        h, w = h * 2, w * 2
        if drop:
            shapes.append((batch, h, w, f))
    return shapes


def generator_forward(p: Params, x, arch: str, dropout_masks: Sequence, per_sample_bn: bool = True,
                      return_logits: bool = False, acts: Optional[dict] = None):
    """define_generator (U/models/generator.py:44-74, P/models/generator.py:69-114)."""
    a = GEN_ARCH[arch]
    x = _t(x)
    masks = [_t(m) for m in dropout_masks]
    skips = []
    g = x
    for i, _f in enumerate(a["enc"], 1):
        g = conv2d_same(g, p[f"e{i}.kernel"] if i == 1 else RW(p[f"e{i}.kernel"]), p[f"e{i}.bias"], 2)
        g = batchnorm(RRAW(g), p, f"e{i}", True, per_sample_bn) if i > 1 else RPRE(g)
        g = R(leaky(g))
        skips.append(g)
        if acts is not None:
            acts[f"e{i}"] = g
    g = R(torch.relu(RPRE(conv2d_same(g, RW(p["b.kernel"]), p["b.bias"], 2))))
    if acts is not None:
        acts["b"] = g
    mi = 0
    for i, (_f, drop) in enumerate(a["dec"], 1):
        g = conv2d_transpose_same(g, RW(p[f"d{i}.kernel"]), p[f"d{i}.bias"])
        g = batchnorm(RRAW(g), p, f"d{i}", True, per_sample_bn)
        if drop:
            g = g * masks[mi] * KS.DROPOUT_KEEP_SCALE
            mi += 1
        g = R(torch.relu(torch.cat([g, skips[-i]], dim=-1)))  # ReLU after concat: also rectifies the skip
        if acts is not None:
            acts[f"d{i}"] = g
    logits = conv2d_transpose_same(g, RW(p["out.kernel"]), p["out.bias"])
    return logits if return_logits else torch.sigmoid(logits)


def discriminator_forward(p: Params, src, tgt, training: bool, per_sample_bn: bool = True,
                          stats_out: Optional[dict] = None, return_logits: bool = False):
    """define_discriminator (U/models/discriminator.py:12-44).  BN uses the default training flag."""
    d = torch.cat([_t(src), _t(tgt)], dim=-1)
    for i, (_f, s, bn) in enumerate(DISC, 1):
        d = conv2d_same(d, p[f"c{i}.kernel"] if i == 1 else RW(p[f"c{i}.kernel"]), p[f"c{i}.bias"], s)
        d = batchnorm(RRAW(d), p, f"c{i}", training, per_sample_bn, stats_out) if bn else RPRE(d)
        d = R(leaky(d))
    logits = conv2d_same(d, RW(p["c6.kernel"]), p["c6.bias"], 1)
    return logits if return_logits else torch.sigmoid(logits)


def cnn_forward(p: Params, x, dropout_mask=None):
    """define_cnn_model (DeltaP/models/generator.py:15-49).  Dropout is a normal layer:
    active (mask given) in train_on_batch, identity in predict/test_on_batch."""
    g = _t(x)
    g = R(leaky(RPRE(conv2d_same(g, p["g1.kernel"], p["g1.bias"], 1))))
    g = F.max_pool2d(g.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    g = R(leaky(torch.relu(RPRE(conv2d_same(g, RW(p["g3.kernel"]), p["g3.bias"], 1)))))  # LeakyReLU after relu: no-op
    g = R(torch.relu(RPRE(conv2d_same(g, RW(p["g4.kernel"]), p["g4.bias"], 1))))
    g = F.max_pool2d(g.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    g = R(torch.relu(RPRE(conv2d_same(g, RW(p["g6.kernel"]), p["g6.bias"], 1))))
    g = F.max_pool2d(g.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    g = g.reshape(g.shape[0], -1)  # Flatten in (H,W,C) order
    g = R(torch.relu(RPRE(g @ RW(p["fc1.kernel"]) + p["fc1.bias"])))
    if dropout_mask is not None:
        g = R(g * _t(dropout_mask) * KS.DROPOUT_KEEP_SCALE)
    g = R(torch.relu(RPRE(g @ RW(p["fc2.kernel"]) + p["fc2.bias"])))
    return g @ RW(p["fc3.kernel"]) + p["fc3.bias"]


# --------------------------------------------------------------------------------------
# losses (Keras reduce = mean over every element)
# --------------------------------------------------------------------------------------
def bce_from_logits(logits: torch.Tensor, target) -> torch.Tensor:
    z = _t(target)
    if KS.BCE_FROM_LOGITS:
        return (torch.clamp(logits, min=0) - logits * z + torch.log1p(torch.exp(-logits.abs()))).mean()
    pr = torch.clamp(torch.sigmoid(logits), KS.BCE_CLIP, 1 - KS.BCE_CLIP)
    return (-(z * torch.log(pr) + (1 - z) * torch.log(1 - pr))).mean()


def mae(y_true, y_pred):
    """Keras 'mae' and MC_mae (U/models/loss.py:47-88): the mass-conservation term is computed
    from y_true on both sides (loss.py:59-60 vs 72-73) and is identically zero."""
    return (_t(y_pred) - _t(y_true)).abs().mean()


def mse(y_true, y_pred):
    """MC_mse (U/models/loss.py:4-45), mc term identically zero."""
    return ((_t(y_pred) - _t(y_true)) ** 2).mean()


def huber(y_true, y_pred, delta: float = 1.0):
    """Huber(delta=1.0) (DeltaP/models/generator.py:53) / MC_huber_loss (loss.py:90-138)."""
    e = _t(y_pred) - _t(y_true)
    a = e.abs()
    return torch.where(a <= delta, 0.5 * e * e, delta * (a - 0.5 * delta)).mean()


# --------------------------------------------------------------------------------------
# Adam, Keras form
# --------------------------------------------------------------------------------------
class KerasAdam:
    def __init__(self, lr: float, beta_1: float = 0.5):
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, KS.ADAM_BETA_2, KS.ADAM_EPSILON
        self.t = 0
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}

    def step(self, params: Params, grads: Dict[str, torch.Tensor]) -> None:
        self.t += 1
        alpha = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k, g in grads.items():
            m = self.m.setdefault(k, torch.zeros_like(g))
            v = self.v.setdefault(k, torch.zeros_like(g))
            m += (g - m) * (1 - self.b1)
            v += (g * g - v) * (1 - self.b2)
            params[k] = params[k] - (m * alpha) / (v.sqrt() + self.eps)


def trainable_keys(p: Params, bn_only: bool = False) -> List[str]:
    ks = []
    for k in p:
        if k.endswith("moving_mean") or k.endswith("moving_var"):
            continue
        if bn_only and ".bn." not in k:
            continue
        ks.append(k)
    return ks


def _grads(loss: torch.Tensor, p: Params, keys: List[str]) -> Dict[str, torch.Tensor]:
    gs = torch.autograd.grad(loss, [p[k] for k in keys], allow_unused=True)
    return {k: (g if g is not None else torch.zeros_like(p[k])) for k, g in zip(keys, gs)}


def _with_grad(p: Params, keys: List[str]) -> Params:
    q = dict(p)
    for k in keys:
        q[k] = p[k].detach().clone().requires_grad_(True)
    return q


# --------------------------------------------------------------------------------------
# train / test steps in the reference's order
# --------------------------------------------------------------------------------------
def d_train_on_batch(pd: Params, opt: KerasAdam, src, tgt, label, per_sample_bn=True,
                     return_grads: bool = False):
    """d_model.train_on_batch([A, B], y) (U/train.py:121,123): BCE, Adam(2e-3, .5)."""
    keys = trainable_keys(pd, bn_only=not KS.D_CONVS_TRAIN_IN_D_STEP)
    q = _with_grad(pd, keys)
    stats: dict = {}
    logits = discriminator_forward(q, src, tgt, True, per_sample_bn, stats, return_logits=True)
    loss = bce_from_logits(logits, label)
    grads = _grads(loss, q, keys)
    opt.step(pd, grads)
    update_moving(pd, stats)
    return (float(loss), grads) if return_grads else float(loss)


def gan_train_on_batch(pg: Params, pd: Params, opt: KerasAdam, arch: str, src, label, tgt, masks,
                       per_sample_bn=True, return_grads: bool = False):
    """gan_model.train_on_batch([A], [ones, B]) (U/train.py:125): loss = 1*BCE + 100*MAE;
    trainable = all of G plus D's BatchNorm gamma/beta (U/models/generator.py:79-81)."""
    gk = trainable_keys(pg)
    dk = trainable_keys(pd, bn_only=True)
    qg, qd = _with_grad(pg, gk), _with_grad(pd, dk)
    stats: dict = {}
    fake = generator_forward(qg, src, arch, masks, per_sample_bn)
    logits = discriminator_forward(qd, src, fake, True, per_sample_bn, stats, return_logits=True)
    l1 = bce_from_logits(logits, label)
    l2 = mae(tgt, fake)
    total = 1.0 * l1 + 100.0 * l2
    gs = torch.autograd.grad(total, [qg[k] for k in gk] + [qd[k] for k in dk])
    grads = {("g." + k): g for k, g in zip(gk, gs[: len(gk)])}
    grads.update({("d." + k): g for k, g in zip(dk, gs[len(gk):])})
    merged = {("g." + k): pg[k] for k in gk}
    merged.update({("d." + k): pd[k] for k in dk})
    opt.step(merged, grads)
    for k in gk:
        pg[k] = merged["g." + k]
    for k in dk:
        pd[k] = merged["d." + k]
    update_moving(pd, stats)
    out = [float(total), float(l1), float(l2)]
    return (out, grads) if return_grads else out


def gan_test_on_batch(pg: Params, pd: Params, arch: str, src, label, tgt, masks, per_sample_bn=True):
    """gan_model.test_on_batch (U/train.py:148): G's BN/dropout stay in training mode (they are
    called with training=True), D's BN uses its moving statistics."""
    with torch.no_grad():
        fake = generator_forward(pg, src, arch, masks, per_sample_bn)
        logits = discriminator_forward(pd, src, fake, False, per_sample_bn, return_logits=True)
        l1, l2 = bce_from_logits(logits, label), mae(tgt, fake)
        return [float(l1 + 100.0 * l2), float(l1), float(l2)]


def cnn_train_on_batch(p: Params, opt: KerasAdam, x, y, dropout_mask, return_grads: bool = False):
    """g_model.train_on_batch([X],[y]) (DeltaP/train.py:131): 100*Huber(1.0), Adam(1e-3,.5)."""
    keys = trainable_keys(p)
    q = _with_grad(p, keys)
    loss = 100.0 * huber(y, cnn_forward(q, x, dropout_mask))
    grads = _grads(loss, q, keys)
    opt.step(p, grads)
    return (float(loss), grads) if return_grads else float(loss)


def to_numpy(p: Params) -> Dict[str, np.ndarray]:
    return {k: v.detach().numpy().copy() for k, v in p.items()}
