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

    def begin(self) -> float:
        """start update t + 1: returns its step size alpha_t (Keras: lr * sqrt(1 - b2^t) / (1 - b1^t))"""
        self.t += 1
        return self.lr * math.sqrt(1.0 - ADAM_BETA_2 ** self.t) / (1.0 - self.b1 ** self.t)

    def apply(self, a: int, b: int, g, alpha: float, grad_scale: float = 1.0) -> None:
        """update the absolute range [a, b) inside [lo, hi) with gradient `g` (compact tensor of b - a elements, or None
        = params.grad[a:b]).  The sharded data-parallel step hands in this rank's chunks, so Adam's 7 x 4 B + 2 B per
        parameter shrink by the world size."""
        if not (self.lo <= a <= b <= self.hi):
            raise ValueError(f"Adam segment [{a}, {b}) outside the optimiser's range [{self.lo}, {self.hi})")
        if b == a:
            return
        P = self.p
        wc = None if P.compute is P.master else P.compute[a:b]
        ops.adam_step(P.master[a:b], P.grad[a:b] if g is None else g, self.m[a - self.lo:b - self.lo],
                      self.v[a - self.lo:b - self.lo], wc, alpha, self.b1, ADAM_BETA_2, ADAM_EPS, grad_scale)

    def step(self, grad_scale: float = 1.0, segments=None) -> None:
        """One update of [lo, hi).  `segments` = None: the whole range, gradient read from params.grad.  Otherwise a
        list of (a, b, g) as for apply(): only those ranges are touched."""
        alpha = self.begin()
        for a, b, g in ([(self.lo, self.hi, None)] if segments is None else segments):
            self.apply(a, b, g, alpha, grad_scale)


def _cast(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst <- src across dtypes (bf16 gradient exchange): the library's cast kernel on the GPU, torch on CPU tensors
    (the gloo tests)"""
    if src.is_cuda:
        n = src.numel()
        ops.cast_copy(src.view(1, 1, 1, n), dst.view(1, 1, 1, n))
    else:
        dst.copy_(src)


_side_stream = None


def _dp_stream():
    """ONE side stream per process for every collective launch: the discriminator's and the composite's trainers each
    own a _DP, and their collectives must reach RCCL in one program order on every rank"""
    global _side_stream
    if _side_stream is None:
        _side_stream = torch.cuda.Stream()
    return _side_stream


class _Seg:
    """one range [a, b) of the flat gradient buffer handed to a collective this step"""
    __slots__ = ("a", "b", "gs", "g16", "handles")

    def __init__(self, a, b, gs, g16):
        self.a, self.b, self.gs, self.g16, self.handles = a, b, gs, g16, []


class _DP:
    """Gradient exchange of the data-parallel step over RCCL (`torch.distributed`); a no-op for world size 1.

    Two forms, chosen per range of the flat gradient buffer:
      * sharded (default for world in {2, 4, 8}; ranges of at least `shard_min` elements INSIDE [0, P.n_kernels), the
        GEMM kernels): `reduce_scatter_tensor` leaves rank r with the sum of chunk r of the range, Adam updates only
        that chunk (optimizer traffic / world), and `all_gather_into_tensor` redistributes the updated COMPUTE copy
        (bf16 in bf16 mode: 2 B per parameter on the wire instead of 4).  On the fully connected xGMI node every rank
        sends a chunk to every peer at once, so all seven links carry data in both phases.  The fp32 master (and
        Adam's m, v) of the chunks a rank does not own go stale until `sync_state()` (a collective: every rank calls
        it) -- checkpoints call it.  Nothing in a forward or backward reads the master of that range: every tensor the
        kernels read in fp32 (biases, the first layers' kernels, BatchNorm gamma / beta) lies in
        [P.n_kernels, P.n_trainable), which is ALWAYS updated replicated, so the replicas stay coherent.
      * replicated: bucketed `all_reduce` (SUM), every rank applies the same Adam step (the fp32-read range above, small
        ranges, other world sizes, CFD_AMD_DP_SHARD=0).
    Per bucket: `apply()` makes the compute stream wait for ONE range's collective, runs Adam on it and hands the
    updated chunk straight to an asynchronous all-gather on the side stream -- the first Adam segment starts when the
    first reduce-scatter (the last layers' gradients, launched from inside the backward) is done, not when all are, and
    the all-gathers overlap the remaining Adam segments and whatever the caller enqueues next; the next reader of the
    parameters waits for them (`ParamStore.join_updates`, called by every forward).
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
        self.stream = _dp_stream() if (self.world > 1 and torch.cuda.is_available()) else None
        self._segs = {}        # id(params) -> [_Seg]: ranges handed over this step
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
        if self.stream is None:          # gloo / CPU tests, virtual ranks
            fn()
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            fn()

    def reduce_async(self, flat: torch.Tensor, handles: Optional[List] = None) -> List:
        """launch the bucketed all-reduce of `flat` (already fully written on the current stream); returns the handles"""
        handles = [] if handles is None else handles
        if self.world == 1:
            return handles

        def go():
            for s in range(0, flat.numel(), self.bucket):
                handles.append(self.dist.all_reduce(flat[s:s + self.bucket], async_op=True))
        self._launch(go)
        return handles

    def reduce_range_async(self, P, a: int, b: int) -> None:
        """hand params.grad[a:b) (final on the current stream) to the collective: reduce-scatter when the range is
        eligible for the sharded update, all-reduce otherwise.  `apply(P, opt)` runs the matching Adam segments."""
        if self.world == 1 or b <= a:
            return
        nk = getattr(P, "n_kernels", b)
        if a < nk < b:                                    # never shard across the fp32-read range
            self.reduce_range_async(P, a, nk)
            self.reduce_range_async(P, nk, b)
            return
        n = b - a
        segs = self._segs.setdefault(id(P), [])
        off = sum(x.b - x.a for x in segs)                # bump pointer into this step's scratch
        if self.shard and b <= nk and n >= self.shard_min and n % (self.world * 64) == 0:
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
            seg = _Seg(a, b, gs, dst)
            self._launch(lambda: seg.handles.append(self.dist.reduce_scatter_tensor(out, src, async_op=True)))
            self.partition.setdefault(id(P), set()).add((a, b))
        else:
            seg = _Seg(a, b, None, None)
            if self.grad_bf16:
                seg.g16 = self._scratch(P, "g16", P.n_trainable, torch.bfloat16)[a:b]
                _cast(P.grad[a:b], seg.g16)
                self.reduce_async(seg.g16, seg.handles)
            else:
                self.reduce_async(P.grad[a:b], seg.handles)
        segs.append(seg)

    def wait(self) -> None:
        """the current stream waits for every collective launched through reduce_range_async so far (apply() waits
        range by range instead; this is for callers that only want the summed gradients)"""
        for segs in self._segs.values():
            for sg in segs:
                for h in sg.handles:
                    h.wait()
                sg.handles = []

    def chunk(self, a: int, b: int):
        """this rank's chunk of a sharded range"""
        c = (b - a) // self.world
        return a + self.rank * c, a + (self.rank + 1) * c

    def apply(self, P, opt: "Adam") -> None:
        """One optimiser step of `opt` over the ranges of P handed over this step (they must cover [opt.lo, opt.hi)
        exactly), bucket by bucket in the order they were launched: wait for that range's collective only, Adam on this
        rank's chunk (sharded) or on the whole range (replicated), then -- sharded -- start the asynchronous all-gather
        of the updated compute copy.  World size 1: plain Adam over the optimiser's range."""
        if self.world == 1:
            opt.step()
            return
        segs = self._segs.get(id(P), [])
        mine = [x for x in segs if opt.lo <= x.a and x.b <= opt.hi]
        self._segs[id(P)] = [x for x in segs if not (opt.lo <= x.a and x.b <= opt.hi)]
        covered = sum(x.b - x.a for x in mine)
        if covered != opt.hi - opt.lo:
            raise RuntimeError(f"data-parallel step: gradient ranges handed over cover {covered} of the {opt.hi - opt.lo} "
                               f"elements of the optimiser's range [{opt.lo}, {opt.hi})")
        alpha = opt.begin()
        P._pending_updates = []        # (the forward of the step that produced these gradients joined the previous ones)
        for sg in mine:
            for h in sg.handles:
                h.wait()               # the CURRENT stream waits for this range's collective (no host block on RCCL)
            sg.handles = []
            if sg.gs is not None:
                if sg.g16 is not None:
                    _cast(sg.g16, sg.gs)
                ca, cb = self.chunk(sg.a, sg.b)
                opt.apply(ca, cb, sg.gs, alpha, self.grad_scale)
                self.gather_async(P, P.compute, sg.a, sg.b)
            else:
                if sg.g16 is not None:
                    _cast(sg.g16, P.grad[sg.a:sg.b])
                opt.apply(sg.a, sg.b, None, alpha, self.grad_scale)

    def gather_async(self, P, T: torch.Tensor, a: int, b: int) -> None:
        """every rank's chunk of T[a:b) -> all ranks, on the side stream; readers wait through P.join_updates()"""
        ca, cb = self.chunk(a, b)
        stage = self._scratch(P, ("ag", a, T.dtype), cb - ca, T.dtype)   # one per range: several are in flight at once
        stage.copy_(T[ca:cb])             # the send buffer must not alias the receive buffer on every backend

        def go():
            h = self.dist.all_gather_into_tensor(T[a:b], stage, async_op=True)
            h.wait()                      # the side stream (not the compute stream) waits for the collective
            if self.stream is not None:
                ev = torch.cuda.Event()
                ev.record()
                P._pending_updates.append(ev)
        self._launch(go)

    def _all_gather(self, T: torch.Tensor, a: int, b: int, P, key: str) -> None:
        ca, cb = self.chunk(a, b)
        stage = self._scratch(P, key + str(T.dtype), cb - ca, T.dtype)
        stage.copy_(T[ca:cb])
        self.dist.all_gather_into_tensor(T[a:b], stage)

    def sync_state(self, P, opts=()) -> None:
        """COLLECTIVE (every rank calls it): bring the fp32 master weights and the Adam moments of `opts` up to date on
        every rank for all ranges that are updated sharded.  Needed before anything reads the whole master buffer:
        checkpoints, get_weights, refresh_compute."""
        if self.world == 1:
            return
        join_averages()
        if hasattr(P, "join_updates"):
            P.join_updates()
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
    """Overlap of the gradient exchange with the backward pass.  The range [0, n_kernels) of the flat
    gradient buffer holds the GEMM kernels in layer order (e2 .. out) and the backward runs last layer first, so
    when layer L reports ready everything in [offset(L.kernel), hi) is final: that tail is handed to RCCL on the side
    stream (once it holds at least `min_elems`, to keep collectives large) while the earlier layers' gradient
    kernels keep running.  `finish()` sends what is left plus the small replicated range (biases, first-layer kernel,
    BatchNorm gamma / beta).  The cut points depend only on the
    layer sizes, so the sharded update owns the same chunks every step."""

    def __init__(self, dp: _DP, params, min_elems: int = 8 * 1024 * 1024):
        self.dp, self.p, self.min = dp, params, min_elems
        self.hi = getattr(params, "n_kernels", params.n_main)
        self.sent: List[tuple] = []          # (lo, hi) ranges handed to the collective, for tests

    def ready(self, layer: str) -> None:
        if self.dp.world == 1:
            return
        lo = self.p.specs[layer + ".kernel"].offset
        if lo >= self.hi:               # a first-layer kernel: it lives in the fp32-read range that finish() sends
            return
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
        self._send(getattr(self.p, "n_kernels", self.p.n_main), self.p.n_trainable)


def _side_wgrad(dp) -> bool:
    """weight-gradient GEMMs on a second stream beside the data-gradient chain (nets._Net.wgrad): single-GPU steps only
    -- under data parallelism `ready(layer)` hands gradient ranges to RCCL as the backward produces them, which needs
    them complete on the launching stream.  On by default since round 3 (CFD_AMD_WGRAD_STREAM=0 turns it off): the
    B = 16 U step 7.92 -> 7.77 ms (-1.9 %, three interleaved pairs on one box; the same -1.9 % round 2 measured).  The
    small-M layers' GEMMs leave most CUs idle for part of their run; a second GEMM fills them.  Every GEMM's OWN
    duration grows under the sharing, so bench.py's per-launch roofline brackets are taken with the switch off (the
    brackets of two streams overlap and would count the shared time twice); the step time is measured with it on.
    The side streams are process-wide (ops.side_stream): with one per net the step had more streams than hardware queues
    and a slow mode (9.4 ms) that came and went with the queue assignment."""
    import os
    return dp.world == 1 and os.environ.get("CFD_AMD_WGRAD_STREAM", "1") != "0"


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
        self.dp.reduce_range_async(P, self.opt.lo, self.opt.hi)      # (split at n_kernels: sharded | replicated)
        self._pending = B

    def finish(self, sync: bool = True):
        """wait for the all-reduce, Adam, moving statistics; returns the loss (one host sync), or -- sync=False -- the
        device tensor that will hold it, without waiting: U/train.py:121-125 uses the three losses of a step only after
        its third call, and every host sync drains the GPU's queue (three per step cost 1-2 % at batch 16)"""
        B, self._pending = self._pending, None
        if B is None:
            raise RuntimeError("finish() without begin()")
        P = self.d.params
        self.dp.apply(P, self.opt)
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
        self.dp.apply(g.params, self.opt_g)          # per bucket: wait, Adam, asynchronous all-gather
        self.dp.apply(d.params, self.opt_dbn)
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
        c.backward(dy, x, drop_mask=drop_mask, seed=seed, sample0=sample0, side_wgrad=_side_wgrad(self.dp))
        self.dp.reduce_range_async(P, 0, P.n_trainable)
        self.dp.apply(P, self.opt)
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
