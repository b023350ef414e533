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

    def step(self, grad_scale: float = 1.0, segments=None) -> None:
        """One update of [lo, hi).  `segments` = None: the whole range, gradient read from params.grad.  Otherwise a
        list of (a, b, g) -- absolute ranges [a, b) inside [lo, hi) with their (summed) gradient in the compact tensor
        `g` (or None = params.grad[a:b]): only those ranges are touched.  The sharded data-parallel step hands in this
        rank's chunks (`_DP.take`), so Adam's 7 x 4 B + 2 B per parameter shrink by the world size."""
        self.t += 1
        alpha = self.lr * math.sqrt(1.0 - ADAM_BETA_2 ** self.t) / (1.0 - self.b1 ** self.t)
        P = self.p
        if segments is None:
            segments = [(self.lo, self.hi, None)]
        for a, b, g in segments:
            if not (self.lo <= a <= b <= self.hi):
                raise ValueError(f"Adam segment [{a}, {b}) outside the optimiser's range [{self.lo}, {self.hi})")
            if b == a:
                continue
            wc = None if P.compute is P.master else P.compute[a:b]
            ops.adam_step(P.master[a:b], P.grad[a:b] if g is None else g, self.m[a - self.lo:b - self.lo],
                          self.v[a - self.lo:b - self.lo], wc, alpha, self.b1, ADAM_BETA_2, ADAM_EPS, grad_scale)


def _cast(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst <- src across dtypes (bf16 gradient exchange): the library's cast kernel on the GPU, torch on CPU tensors
    (the gloo tests)"""
    if src.is_cuda:
        n = src.numel()
        ops.cast_copy(src.view(1, 1, 1, n), dst.view(1, 1, 1, n))
    else:
        dst.copy_(src)


class _DP:
    """Gradient exchange of the data-parallel step over RCCL (`torch.distributed`); a no-op for world size 1.

    Two forms, chosen per range of the flat gradient buffer:
      * sharded (default for world in {2, 4, 8} and ranges of at least `shard_min` elements): `reduce_scatter_tensor`
        leaves rank r with the sum of chunk r of the range, Adam updates only that chunk (optimizer traffic / world),
        and `all_gather_into_tensor` redistributes the updated COMPUTE copy (bf16 in bf16 mode: 2 B per parameter on
        the wire instead of 4).  On the fully connected xGMI node every rank sends a chunk to every peer at once, so
        all seven links carry data in both phases.  The fp32 master (and Adam's m, v) of the chunks a rank does not own
        go stale until `sync_state()` (a collective: every rank calls it) -- checkpoints call it.
      * replicated: bucketed `all_reduce` (SUM), every rank applies the same Adam step (small ranges, other world sizes,
        CFD_AMD_DP_SHARD=0).
    CFD_AMD_DP_GRAD_BF16=1 exchanges the gradients in bf16 (half the bytes; the sum of `world` bf16 terms then carries
    bf16 rounding -- off by default, covered by a tolerance test).
    ParamStore aligns every tensor to 512 elements, so every range between layer boundaries divides by world * 64."""

    def __init__(self, bucket_elems: int = 32 * 1024 * 1024, shard: Optional[bool] = None, shard_min: int = 1 << 20,
                 grad_bf16: Optional[bool] = None):
        import os
        import torch.distributed as dist
        self.dist = dist
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if on else 1
        self.rank = dist.get_rank() if on else 0
        self.bucket = bucket_elems
        self.shard = (os.environ.get("CFD_AMD_DP_SHARD", "1") != "0") if shard is None else bool(shard)
        self.shard = self.shard and self.world in (2, 4, 8)
        self.shard_min = shard_min
        self.grad_bf16 = (os.environ.get("CFD_AMD_DP_GRAD_BF16", "0") == "1") if grad_bf16 is None else bool(grad_bf16)
        self.handles: List = []
        self.stream = torch.cuda.Stream() if (self.world > 1 and torch.cuda.is_available()) else None
        self._segs = {}        # id(params) -> ranges handed over this step: (a, b, shard tensor | None, 16-bit staging | None)
        self._pool = {}        # id(params) -> dict of grow-only scratch tensors
        self.partition = {}    # id(params) -> {(a, b)}: every range that has ever been updated sharded (for sync_state)

    # ---- plumbing ------------------------------------------------------------------------------------------------
    def _scratch(self, P, key, n: int, dtype) -> torch.Tensor:
        pool = self._pool.setdefault(id(P), {})
        t = pool.get(key)
        if t is None or t.numel() < n:
            t = pool[key] = torch.empty(n, dtype=dtype, device=P.grad.device if hasattr(P, "grad") else P.master.device)
        return t[:n]

    def _launch(self, fn) -> None:
        """run the collective launches of `fn` behind everything already enqueued on the current stream, on the side
        stream (GPU), so that they overlap what the caller enqueues next"""
        if self.stream is None:          # gloo / CPU tests
            fn()
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            fn()

    def reduce_async(self, flat: torch.Tensor) -> None:
        """launch the bucketed all-reduce of `flat` (already fully written on the current stream)"""
        if self.world == 1:
            return

        def go():
            for s in range(0, flat.numel(), self.bucket):
                self.handles.append(self.dist.all_reduce(flat[s:s + self.bucket], async_op=True))
        self._launch(go)

    def reduce_range_async(self, P, a: int, b: int) -> None:
        """hand params.grad[a:b) (final on the current stream) to the collective: reduce-scatter when the range is
        eligible for the sharded update, all-reduce otherwise.  `take(P, ...)` returns the matching Adam segments."""
        if self.world == 1 or b <= a:
            return
        n = b - a
        segs = self._segs.setdefault(id(P), [])
        off = sum(x[1] - x[0] for x in segs)              # bump pointer into this step's scratch
        if self.shard and n >= self.shard_min and n % (self.world * 64) == 0:
            c = n // self.world
            if self.grad_bf16:
                src = self._scratch(P, "g16", P.n_trainable, torch.bfloat16)[a:b]
                _cast(P.grad[a:b], src)
                dst = self._scratch(P, "s16", P.n_trainable // self.world + 64, torch.bfloat16)[off // self.world:off // self.world + c]
            else:
                src = P.grad[a:b]
                dst = None
            gs = self._scratch(P, "s32", P.n_trainable // self.world + 64, torch.float32)[off // self.world:off // self.world + c]
            out = gs if dst is None else dst
            self._launch(lambda: self.handles.append(self.dist.reduce_scatter_tensor(out, src, async_op=True)))
            segs.append((a, b, gs, dst))
            self.partition.setdefault(id(P), set()).add((a, b))
        else:
            if self.grad_bf16:
                g16 = self._scratch(P, "g16", P.n_trainable, torch.bfloat16)[a:b]
                _cast(P.grad[a:b], g16)
                self.reduce_async(g16)
                segs.append((a, b, None, g16))
            else:
                self.reduce_async(P.grad[a:b])
                segs.append((a, b, None, None))

    def wait(self) -> None:
        for h in self.handles:
            h.wait()
        self.handles.clear()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def chunk(self, a: int, b: int):
        """this rank's chunk of a sharded range"""
        c = (b - a) // self.world
        return a + self.rank * c, a + (self.rank + 1) * c

    def take(self, P, lo: int, hi: int):
        """After wait(): the Adam segments of the ranges handed over this step that lie in [lo, hi) -- this rank's
        chunk with its compact summed gradient for a sharded range, the whole range for a replicated one -- or None for
        world size 1 (Adam then runs over its whole range)."""
        if self.world == 1:
            return None
        segs = self._segs.get(id(P), [])
        mine = [x for x in segs if lo <= x[0] and x[1] <= hi]
        self._segs[id(P)] = [x for x in segs if not (lo <= x[0] and x[1] <= hi)]
        covered = sum(x[1] - x[0] for x in mine)
        if covered != hi - lo:
            raise RuntimeError(f"data-parallel step: gradient ranges handed over cover {covered} of the {hi - lo} "
                               f"elements of the optimiser's range [{lo}, {hi})")
        out = []
        for a, b, gs, g16 in mine:
            if gs is not None:
                if g16 is not None:
                    _cast(g16, gs)
                ca, cb = self.chunk(a, b)
                out.append((ca, cb, gs))
            else:
                if g16 is not None:
                    _cast(g16, P.grad[a:b])
                out.append((a, b, None))
        return out

    def gather_params(self, P, segments) -> None:
        """after Adam: every rank's updated chunk of the compute copy -> all ranks (sharded segments only)"""
        if self.world == 1 or not segments:
            return
        c = self.world
        for ca, cb, gs in segments:
            if gs is None:
                continue
            n = (cb - ca) * c
            a = ca - self.rank * (cb - ca)
            self._all_gather(P.compute, a, a + n, P, "ag")

    def _all_gather(self, T: torch.Tensor, a: int, b: int, P, key: str) -> None:
        ca, cb = self.chunk(a, b)
        stage = self._scratch(P, key + str(T.dtype), cb - ca, T.dtype)
        stage.copy_(T[ca:cb])             # the send buffer must not alias the receive buffer on every backend
        self.dist.all_gather_into_tensor(T[a:b], stage)

    def sync_state(self, P, opts=()) -> None:
        """COLLECTIVE (every rank calls it): bring the fp32 master weights and the Adam moments of `opts` up to date on
        every rank for all ranges that are updated sharded.  Needed before anything reads the whole master buffer:
        checkpoints, get_weights, refresh_compute."""
        if self.world == 1:
            return
        join_averages()
        for a, b in sorted(self.partition.get(id(P), ())):
            if P.compute is not P.master:
                self._all_gather(P.master, a, b, P, "ag")
            for o in opts:
                if o.p is P and o.lo <= a and b <= o.hi:
                    for T in (o.m, o.v):
                        ca, cb = self.chunk(a, b)
                        stage = self._scratch(P, "agm", cb - ca, T.dtype)
                        stage.copy_(T[ca - o.lo:cb - o.lo])
                        self.dist.all_gather_into_tensor(T[a - o.lo:b - o.lo], stage)

    def broadcast_(self, tensors, src: int = 0) -> None:
        """every tensor <- rank `src`'s (initial parameters, optimiser state: replicas must START equal, gradients
        alone keep them equal)"""
        if self.world > 1:
            for t in tensors:
                self.dist.broadcast(t, src=src)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def average_async(self, t: torch.Tensor) -> None:
        """t <- mean over ranks (small state that must stay identical on every rank, e.g. BatchNorm moving
        statistics, whose per-rank updates see different samples), on the side stream.  Whoever next reads or writes
        `t` on the compute stream calls join_averages() first (the next step's update of the moving statistics: a
        whole step later, so the collective is hidden)."""
        if self.world == 1:
            return

        def go():
            h = self.dist.all_reduce(t, async_op=True)
            h.wait()                       # orders the scaling behind the collective on THIS (side) stream
            t.mul_(1.0 / self.world)
            if self.stream is not None:
                ev = torch.cuda.Event()
                ev.record()
                _pending_averages.append(ev)
        self._launch(go)


_pending_averages: List = []     # events of average_async launches not yet joined (any trainer's side stream)


def join_averages() -> None:
    """the compute stream waits for every pending average_async (process-wide: D's moving statistics are averaged by
    both the discriminator's and the composite's trainer)"""
    while _pending_averages:
        torch.cuda.current_stream().wait_event(_pending_averages.pop())


def sync_from_rank0(dp: _DP, params, opts=()) -> None:
    """Data-parallel replicas must start from the same parameters and optimiser state: broadcast rank 0's (models built
    through the Keras surface seed themselves from os.urandom per process).  COLLECTIVE; a no-op for world size 1."""
    if dp.world == 1:
        return
    dp.broadcast_([params.master] + [t for o in opts for t in (o.m, o.v)])
    ts = torch.tensor([o.t for o in opts] or [0], dtype=torch.int64, device=params.master.device)
    dp.broadcast_([ts])
    for o, t in zip(opts, ts.tolist()):
        o.t = int(t)
    params.refresh_compute()
    params.refresh_transposed()


class BackwardBuckets:
    """Overlap of the gradient exchange with the backward pass.  The main range [0, n_main) of the flat
    gradient buffer holds kernels and biases in layer order (e1 .. out) and the backward runs last layer first, so
    when layer L reports ready everything in [offset(L.kernel), hi) is final: that tail is handed to RCCL on the side
    stream (once it holds at least `min_elems`, to keep collectives large) while the earlier layers' gradient
    kernels keep running.  `finish()` sends what is left plus the BatchNorm range.  The cut points depend only on the
    layer sizes, so the sharded update owns the same chunks every step."""

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
            self.dp.reduce_range_async(self.p, lo, hi)
            self.sent.append((lo, hi))

    def finish(self) -> None:
        if self.dp.world == 1:
            return
        self._send(0, self.hi)
        self.hi = 0
        self._send(self.p.n_main, self.p.n_trainable)


def _side_wgrad(dp) -> bool:
    """weight-gradient GEMMs on a second stream beside the data-gradient chain (nets._Net.wgrad): single-GPU steps only
    -- under data parallelism `ready(layer)` hands gradient ranges to RCCL as the backward produces them, which needs
    them complete on the launching stream.  Opt-in (CFD_AMD_WGRAD_STREAM=1): measured 8.65 -> 8.49 ms per B=16 step on
    one box (-1.9 %), i.e. inside the box-to-box spread, while every GEMM's own duration grows under the sharing (the
    per-kernel roofline brackets overlap), so the default keeps one stream."""
    import os
    return dp.world == 1 and os.environ.get("CFD_AMD_WGRAD_STREAM", "0") == "1"


class DiscriminatorTrainer:
    def __init__(self, d: nets.Discriminator, lr: float = 2e-3, beta_1: float = 0.5, train_convs: bool = True):
        self.d = d
        d.params.enable_training()
        self.train_convs = train_convs
        lo = 0 if train_convs else d.params.n_main
        self.opt = Adam(d.params, lo, d.params.n_trainable, lr, beta_1)
        self.dp = _DP()
        sync_from_rank0(self.dp, d.params, [self.opt])
        self.loss = torch.zeros(1, dtype=torch.float32, device=d.device)
        self._loss_out = self.loss
        self._pending = None

    def sync_state(self) -> None:
        """COLLECTIVE: full master weights + Adam moments on every rank (before a checkpoint under data parallelism)"""
        self.dp.sync_state(self.d.params, [self.opt])

    def train_on_batch(self, src: torch.Tensor, tgt: torch.Tensor, label: float) -> float:
        self.begin(src, tgt, label)
        return self.finish()

    def begin(self, src: torch.Tensor, tgt: torch.Tensor, label: float, loss: Optional[torch.Tensor] = None) -> None:
        """forward + backward + launch of the gradient all-reduce.  Work that does not read D's weights (the
        generator's forward) may be enqueued between begin() and finish(): it then overlaps the collective, which
        D's short backward cannot hide by itself.  `loss`: a 1-element fp32 device tensor that receives this update's loss
        instead of the trainer's own (a training loop that reads its three losses once per step, see finish(sync=False))."""
        d, P = self.d, self.d.params
        B = src.shape[0]
        P.zero_grad()
        self._loss_out = self.loss if loss is None else loss
        self._loss_out.zero_()
        logits = d.forward(src, tgt, training=True, want_logits=True)
        dz = d.buf(d.plan(B), "dlogits", logits.shape, torch.float32)
        ops.bce_logits(logits, self._loss_out, dz, target_const=label)
        d.backward(dz, src, tgt, train_convs=self.train_convs, side_wgrad=_side_wgrad(self.dp))
        if self.opt.lo < P.n_main:
            self.dp.reduce_range_async(P, self.opt.lo, P.n_main)
        self.dp.reduce_range_async(P, max(self.opt.lo, P.n_main), self.opt.hi)
        self._pending = B

    def finish(self, sync: bool = True):
        """wait for the all-reduce, Adam, moving statistics; returns the loss (one host sync), or -- sync=False -- the
        device tensor that will hold it, without waiting: U/train.py:121-125 uses the three losses of a step only after
        its third call, and every host sync drains the GPU's queue (three per step cost 1-2 % at batch 16)"""
        B, self._pending = self._pending, None
        if B is None:
            raise RuntimeError("finish() without begin()")
        self.dp.wait()
        P = self.d.params
        segs = self.dp.take(P, self.opt.lo, self.opt.hi)
        self.opt.step(self.dp.grad_scale, segs)
        self.dp.gather_params(P, segs)
        if self.train_convs:
            P.refresh_transposed(side_stream=True)
        join_averages()
        self.d.update_moving(B)
        self.dp.average_async(P.master[P.n_trainable:P.n_total])
        return float(self._loss_out.item()) if sync else self._loss_out


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
        sync_from_rank0(self.dp, g.params, [self.opt_g])
        sync_from_rank0(self.dp, d.params, [self.opt_dbn])
        self.field_loss, self.field_weight = field_loss, field_weight
        self.losses = torch.zeros(2, dtype=torch.float32, device=g.device)
        self.step_count = 0

    def sync_state(self) -> None:
        """COLLECTIVE: full master weights + Adam moments on every rank (before a checkpoint under data parallelism)"""
        self.dp.sync_state(self.g.params, [self.opt_g])
        self.dp.sync_state(self.d.params, [self.opt_dbn])

    def _forward_losses(self, src, label, tgt, masks, seed, sample0, training: bool, want_grads: bool, before_d=None):
        g, d = self.g, self.d
        B = src.shape[0]
        if not training:
            join_averages()               # D reads its moving statistics
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
        g.params.zero_grad()
        fake, dz_d = self._forward_losses(src, label, tgt, masks, seed, sample0, True, True, before_d)
        d.params.zero_grad()       # after a pending D step (before_d) has consumed its gradients
        dfake = d.backward(dz_d, src, fake, train_convs=False, want_input_grad=True)
        # D's BN gradients are complete: overlap their all-reduce (and G's, bucket by bucket) with G's backward
        self.dp.reduce_range_async(d.params, self.opt_dbn.lo, self.opt_dbn.hi)
        dz = g.buf(g.plan(B), "dz", fake.shape, torch.float32)
        ops.field_loss(fake, tgt, self.field_loss, self.losses[1:2], dz, weight=self.field_weight, dy_extra=dfake,
                       sigmoid_grad=True)
        buckets = BackwardBuckets(self.dp, g.params)
        g.backward(dz, src, masks=masks, seed=seed, sample0=sample0, ready=buckets.ready,
                   side_wgrad=_side_wgrad(self.dp))
        buckets.finish()
        self.dp.wait()
        segs = self.dp.take(g.params, self.opt_g.lo, self.opt_g.hi)
        self.opt_g.step(self.dp.grad_scale, segs)
        self.dp.gather_params(g.params, segs)
        self.opt_dbn.step(self.dp.grad_scale, self.dp.take(d.params, self.opt_dbn.lo, self.opt_dbn.hi))
        g.params.refresh_transposed(side_stream=True)
        join_averages()
        d.update_moving(B)
        self.dp.average_async(d.params.master[d.params.n_trainable:d.params.n_total])
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
        sync_from_rank0(self.dp, c.params, [self.opt])
        self.weight, self.delta = weight, delta
        self.loss = torch.zeros(1, dtype=torch.float32, device=c.device)
        self.step_count = 0

    def train_on_batch(self, x: torch.Tensor, y: torch.Tensor, drop_mask=None, seed: Optional[int] = None,
                       sample0: int = 0) -> float:
        c, P = self.c, self.c.params
        B = x.shape[0]
        self.step_count += 1
        seed = self.step_count if seed is None else seed
        P.zero_grad()
        self.loss.zero_()
        pred = c.forward(x, training=True, drop_mask=drop_mask, seed=seed, sample0=sample0)
        dy = c.buf(c.plan(B), "dy", pred.shape, torch.float32)
        ops.field_loss(pred, y, ops.HUBER, self.loss, dy, weight=self.weight, delta=self.delta)
        c.backward(dy, x, drop_mask=drop_mask, seed=seed, sample0=sample0)
        self.dp.reduce_range_async(P, 0, P.n_trainable)
        self.dp.wait()
        segs = self.dp.take(P, 0, P.n_trainable)
        self.opt.step(self.dp.grad_scale, segs)
        self.dp.gather_params(P, segs)
        P.refresh_transposed(side_stream=True)
        return self.weight * float(self.loss.item())

    def sync_state(self) -> None:
        """COLLECTIVE: full master weights + Adam moments on every rank (before a checkpoint under data parallelism)"""
        self.dp.sync_state(self.c.params, [self.opt])

    def test_on_batch(self, x: torch.Tensor, y: torch.Tensor) -> float:
        self.loss.zero_()
        pred = self.c.forward(x, training=False)
        ops.field_loss(pred, y, ops.HUBER, self.loss, None, weight=self.weight, delta=self.delta)
        return self.weight * float(self.loss.item())
