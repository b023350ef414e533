"""Training steps in the reference's order, on the HIP kernels.

  DiscriminatorTrainer.train_on_batch  = d_model.train_on_batch([A, B], y)          U/train.py:121,123
  GanTrainer.train_on_batch            = gan_model.train_on_batch([A], [ones, B])    U/train.py:125
  GanTrainer.test_on_batch             = gan_model.test_on_batch                     U/train.py:148
  CnnTrainer.train_on_batch/test_on_batch                                           DeltaP/train.py:131,148

Optimisers are Keras-form Adam over the flat parameter buffers (one fused launch per buffer range).
Data parallelism: every rank runs the same step on its shard of the global batch; the flat gradient
buffers are summed with RCCL (`torch.distributed` all_reduce) in a few large buckets launched on a side
stream as soon as the backward has produced them, and Adam divides by the world size.  BatchNorm
statistics stay per sample, so the DP gradient is the mean of single-sample gradients exactly as in the
reference's batch-1 loop, independent of the world size.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from . import nets, ops

ADAM_BETA_2 = 0.999
ADAM_EPS = 1e-7


class Adam:
    """keras.optimizers.Adam(learning_rate, beta_1) on a range [lo, hi) of a ParamStore's flat buffers."""

    def __init__(self, params, lo: int, hi: int, lr: float, beta_1: float = 0.5):
        self.p, self.lo, self.hi, self.lr, self.b1 = params, lo, hi, lr, beta_1
        self.t = 0
        n = hi - lo
        self.m = torch.zeros(n, dtype=torch.float32, device=params.device)
        self.v = torch.zeros(n, dtype=torch.float32, device=params.device)

    def step(self, grad_scale: float = 1.0) -> None:
        self.t += 1
        alpha = self.lr * math.sqrt(1.0 - ADAM_BETA_2 ** self.t) / (1.0 - self.b1 ** self.t)
        P = self.p
        wc = None if P.compute is P.master else P.compute[self.lo:self.hi]
        ops.adam_step(P.master[self.lo:self.hi], P.grad[self.lo:self.hi], self.m, self.v, wc, alpha, self.b1,
                      ADAM_BETA_2, ADAM_EPS, grad_scale)


class _DP:
    """gradient all-reduce (SUM) over RCCL in large buckets; no-op for world size 1"""

    def __init__(self, bucket_elems: int = 32 * 1024 * 1024):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.bucket = bucket_elems
        self.handles: List = []
        self.stream = torch.cuda.Stream() if (self.world > 1 and torch.cuda.is_available()) else None

    def reduce_async(self, flat: torch.Tensor) -> None:
        """launch the all-reduce of `flat` (already fully written on the current stream)"""
        if self.world == 1:
            return
        if self.stream is None:      # gloo / CPU tests
            for s in range(0, flat.numel(), self.bucket):
                self.handles.append(self.dist.all_reduce(flat[s:s + self.bucket], async_op=True))
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            for s in range(0, flat.numel(), self.bucket):
                self.handles.append(self.dist.all_reduce(flat[s:s + self.bucket], async_op=True))

    def wait(self) -> None:
        for h in self.handles:
            h.wait()
        self.handles.clear()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def average_(self, t: torch.Tensor) -> None:
        """t <- mean over ranks (small state that must stay identical on every rank, e.g. BatchNorm moving
        statistics, whose per-rank updates see different samples)"""
        if self.world > 1:
            self.dist.all_reduce(t)
            t.mul_(1.0 / self.world)


class BackwardBuckets:
    """Overlap of the gradient all-reduce with the backward pass.  The main range [0, n_main) of the flat
    gradient buffer holds kernels and biases in layer order (e1 .. out) and the backward runs last layer first, so
    when layer L reports ready everything in [offset(L.kernel), hi) is final: that tail is handed to RCCL on the side
    stream (once it holds at least `min_elems`, to keep collectives large) while the earlier layers' gradient
    kernels keep running.  `finish()` sends what is left plus the BatchNorm range."""

    def __init__(self, dp: _DP, params, min_elems: int = 8 * 1024 * 1024):
        self.dp, self.p, self.min = dp, params, min_elems
        self.hi = params.n_main
        self.sent: List[tuple] = []          # (lo, hi) ranges handed to the collective, for tests

    def ready(self, layer: str) -> None:
        if self.dp.world == 1:
            return
        lo = self.p.specs[layer + ".kernel"].offset
        if self.hi - lo >= self.min or lo == 0:
            self._send(lo, self.hi)
            self.hi = lo

    def _send(self, lo: int, hi: int) -> None:
        if hi > lo:
            self.dp.reduce_async(self.p.grad[lo:hi])
            self.sent.append((lo, hi))

    def finish(self) -> None:
        if self.dp.world == 1:
            return
        self._send(0, self.hi)
        self.hi = 0
        self._send(self.p.n_main, self.p.n_trainable)


class DiscriminatorTrainer:
    def __init__(self, d: nets.Discriminator, lr: float = 2e-3, beta_1: float = 0.5, train_convs: bool = True):
        self.d = d
        d.params.enable_training()
        self.train_convs = train_convs
        lo = 0 if train_convs else d.params.n_main
        self.opt = Adam(d.params, lo, d.params.n_trainable, lr, beta_1)
        self.dp = _DP()
        self.loss = torch.zeros(1, dtype=torch.float32, device=d.device)
        self._pending = None

    def train_on_batch(self, src: torch.Tensor, tgt: torch.Tensor, label: float) -> float:
        self.begin(src, tgt, label)
        return self.finish()

    def begin(self, src: torch.Tensor, tgt: torch.Tensor, label: float) -> None:
        """forward + backward + launch of the gradient all-reduce.  Work that does not read D's weights (the
        generator's forward) may be enqueued between begin() and finish(): it then overlaps the collective, which
        D's short backward cannot hide by itself."""
        d, P = self.d, self.d.params
        B = src.shape[0]
        P.grad.zero_()
        self.loss.zero_()
        logits = d.forward(src, tgt, training=True, want_logits=True)
        dz = d.buf(d.plan(B), "dlogits", logits.shape, torch.float32)
        ops.bce_logits(logits, self.loss, dz, target_const=label)
        d.backward(dz, src, tgt, train_convs=self.train_convs)
        self.dp.reduce_async(P.grad[self.opt.lo:self.opt.hi])
        self._pending = B

    def finish(self) -> float:
        """wait for the all-reduce, Adam, moving statistics; returns the loss (one host sync)"""
        B, self._pending = self._pending, None
        if B is None:
            raise RuntimeError("finish() without begin()")
        self.dp.wait()
        self.opt.step(self.dp.grad_scale)
        if self.train_convs:
            self.d.params.refresh_transposed()
        self.d.update_moving(B)
        P = self.d.params
        self.dp.average_(P.master[P.n_trainable:P.n_total])
        return float(self.loss.item())


class GanTrainer:
    """Composite model of define_gan (U/models/generator.py:77-94): loss = 1*BCE(D(A,G(A)), y) + 100*MAE(G(A), B);
    trainable = all of G + D's BatchNorm gamma/beta; optimiser Adam(4e-4, 0.5)."""

    def __init__(self, g: nets.Generator, d: nets.Discriminator, lr: float = 4e-4, beta_1: float = 0.5,
                 field_loss: int = ops.MAE, field_weight: float = 100.0):
        self.g, self.d = g, d
        g.params.enable_training()
        d.params.enable_training()
        self.opt_g = Adam(g.params, 0, g.params.n_trainable, lr, beta_1)
        self.opt_dbn = Adam(d.params, d.params.n_main, d.params.n_trainable, lr, beta_1)
        self.dp = _DP()
        self.field_loss, self.field_weight = field_loss, field_weight
        self.losses = torch.zeros(2, dtype=torch.float32, device=g.device)
        self.step_count = 0

    def _forward_losses(self, src, label, tgt, masks, seed, sample0, training: bool, want_grads: bool, before_d=None):
        g, d = self.g, self.d
        B = src.shape[0]
        self.losses.zero_()
        fake = g.forward(src, masks=masks, seed=seed, sample0=sample0)
        if before_d is not None:
            before_d()
        logits = d.forward(src, fake, training=training, want_logits=True)
        dz_d = d.buf(d.plan(B), "dlogits", logits.shape, torch.float32) if want_grads else None
        ops.bce_logits(logits, self.losses[0:1], dz_d, target_const=label)
        return fake, dz_d

    def train_on_batch(self, src: torch.Tensor, label: float, tgt: torch.Tensor, masks=None, seed: Optional[int] = None,
                       sample0: int = 0, before_d=None) -> List[float]:
        """`before_d` (optional callable) runs after the generator's forward and before D is first read: a pending
        DiscriminatorTrainer.finish goes there, so that D's all-reduce overlaps the generator's forward."""
        g, d = self.g, self.d
        B = src.shape[0]
        self.step_count += 1
        seed = self.step_count if seed is None else seed
        g.params.grad.zero_()
        fake, dz_d = self._forward_losses(src, label, tgt, masks, seed, sample0, True, True, before_d)
        d.params.grad.zero_()      # after a pending D step (before_d) has consumed its gradients
        dfake = d.backward(dz_d, src, fake, train_convs=False, want_input_grad=True)
        # D's BN gradients are complete: overlap their all-reduce (and G's, bucket by bucket) with G's backward
        self.dp.reduce_async(d.params.grad[self.opt_dbn.lo:self.opt_dbn.hi])
        dz = g.buf(g.plan(B), "dz", fake.shape, torch.float32)
        ops.field_loss(fake, tgt, self.field_loss, self.losses[1:2], dz, weight=self.field_weight, dy_extra=dfake,
                       sigmoid_grad=True)
        buckets = BackwardBuckets(self.dp, g.params)
        g.backward(dz, src, masks=masks, seed=seed, sample0=sample0, ready=buckets.ready)
        buckets.finish()
        self.dp.wait()
        self.opt_g.step(self.dp.grad_scale)
        self.opt_dbn.step(self.dp.grad_scale)
        g.params.refresh_transposed()
        d.update_moving(B)
        self.dp.average_(d.params.master[d.params.n_trainable:d.params.n_total])
        l = self.losses.tolist()
        return [l[0] + self.field_weight * l[1], l[0], l[1]]

    def test_on_batch(self, src: torch.Tensor, label: float, tgt: torch.Tensor, masks=None, seed: int = 0,
                      sample0: int = 0) -> List[float]:
        """G keeps batch statistics and dropout (its layers are called with training=True); D uses its
        moving statistics."""
        fake, _ = self._forward_losses(src, label, tgt, masks, seed, sample0, False, False)
        ops.field_loss(fake, tgt, self.field_loss, self.losses[1:2], None, weight=self.field_weight)
        l = self.losses.tolist()
        return [l[0] + self.field_weight * l[1], l[0], l[1]]


class CnnTrainer:
    """define_cnn_model's compile: Adam(1e-3, 0.5), 100 * Huber(1.0) (DeltaP/models/generator.py:51-53)."""

    def __init__(self, c: nets.CnnRegressor, lr: float = 1e-3, beta_1: float = 0.5, weight: float = 100.0,
                 delta: float = 1.0):
        self.c = c
        c.params.enable_training()
        self.opt = Adam(c.params, 0, c.params.n_trainable, lr, beta_1)
        self.dp = _DP()
        self.weight, self.delta = weight, delta
        self.loss = torch.zeros(1, dtype=torch.float32, device=c.device)
        self.step_count = 0

    def train_on_batch(self, x: torch.Tensor, y: torch.Tensor, drop_mask=None, seed: Optional[int] = None,
                       sample0: int = 0) -> float:
        c, P = self.c, self.c.params
        B = x.shape[0]
        self.step_count += 1
        seed = self.step_count if seed is None else seed
        P.grad.zero_()
        self.loss.zero_()
        pred = c.forward(x, training=True, drop_mask=drop_mask, seed=seed, sample0=sample0)
        dy = c.buf(c.plan(B), "dy", pred.shape, torch.float32)
        ops.field_loss(pred, y, ops.HUBER, self.loss, dy, weight=self.weight, delta=self.delta)
        c.backward(dy, x, drop_mask=drop_mask, seed=seed, sample0=sample0)
        self.dp.reduce_async(P.grad)
        self.dp.wait()
        self.opt.step(self.dp.grad_scale)
        P.refresh_transposed()
        return self.weight * float(self.loss.item())

    def test_on_batch(self, x: torch.Tensor, y: torch.Tensor) -> float:
        self.loss.zero_()
        pred = self.c.forward(x, training=False)
        ops.field_loss(pred, y, ops.HUBER, self.loss, None, weight=self.weight, delta=self.delta)
        return self.weight * float(self.loss.item())
