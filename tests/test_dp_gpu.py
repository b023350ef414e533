"""Data-parallel training on the GPU, world size 2, as two VIRTUAL ranks inside one process.

The test box has one GPU.  Two processes on it do work with torch.distributed (gloo), but the driver time-slices
their queues so coarsely that one training step with a dozen collectives took 31 s when tried; RCCL refuses two
ranks on one device.  So each rank here is a thread that runs the unmodified trainer code (pipelined step of
bench.py, bucketed all-reduce launched from inside the backward, 1/world scaling in Adam, rank-averaged moving
statistics) against a stand-in for torch.distributed whose all_reduce is a rendezvous of the two threads; only one
thread runs at a time, the hand-over happens inside the collectives.  The transport itself (torch.distributed) is
covered by tests/test_dp_gloo.py.  Checked: one sample per rank leaves the same weights as one process stepping on
both samples, and both ranks end bit-identical."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class VirtualWorld:
    def __init__(self, world):
        self.world = world
        self.run_lock = threading.Lock()         # held by the one thread that is executing trainer code
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.local = threading.local()
        self.calls = [[] for _ in range(world)]  # per rank: sizes of the collectives, in order

    # --- the part of torch.distributed the trainer uses
    def get_world_size(self):
        return self.world

    def all_reduce(self, t, async_op=False):
        rank = self.local.rank
        torch.cuda.synchronize()
        self.slots[rank] = t
        self.calls[rank].append(t.numel())
        self.run_lock.release()
        if self.barrier.wait(timeout=120) == 0:
            assert len({s.numel() for s in self.slots}) == 1, "ranks disagree on a collective's size"
            total = torch.stack([s for s in self.slots]).sum(0)
            for s in self.slots:
                s.copy_(total)
            torch.cuda.synchronize()
        self.barrier.wait(timeout=120)
        self.run_lock.acquire()

        class _Done:
            def wait(self_inner):
                return True
        return _Done()


def _make(world=None):
    from cfd_amd import nets, trainer
    g = nets.Generator("P", dtype=torch.float32, seed=5)
    d = nets.Discriminator(dtype=torch.float32, seed=6)
    dtr, gtr = trainer.DiscriminatorTrainer(d), trainer.GanTrainer(g, d)
    if world is not None:
        for tr in (dtr, gtr):
            tr.dp.dist, tr.dp.world, tr.dp.stream = world, world.world, None
    return g, d, dtr, gtr


def _step(g, d, dtr, gtr, x, u, sample0):
    dtr.begin(x, u, 1.0)
    fake = g.forward(x, seed=4, sample0=sample0)
    dtr.finish()
    dtr.begin(x, fake, 0.0)
    gtr.train_on_batch(x, 1.0, u, seed=5, sample0=sample0, before_d=dtr.finish)
    torch.cuda.synchronize()


def test_dp_world2_equals_one_process_on_both_samples():
    from cfd_amd import synthetic
    xs, us, _ = synthetic.model_batch([0, 1])
    xs, us = torch.from_numpy(xs).cuda(), torch.from_numpy(us).cuda()
    vw = VirtualWorld(2)
    ranks = [_make(vw) for _ in range(2)]
    errors = []

    def run(rank):
        vw.local.rank = rank
        vw.run_lock.acquire()
        try:
            _step(*ranks[rank], xs[rank:rank + 1], us[rank:rank + 1], sample0=rank)
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            errors.append((rank, repr(e)))
            vw.barrier.abort()
        finally:
            if vw.run_lock.locked():
                vw.run_lock.release()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors and not any(t.is_alive() for t in threads), errors
    assert vw.calls[0] == vw.calls[1] and len(vw.calls[0]) >= 6      # D x2, D-BN, >= 2 buckets of G, moving stats
    ref = _make()
    _step(*ref, xs, us, sample0=0)
    # Adam's first step moves a weight by ~lr * sign(g): an element whose near-zero gradient changes sign between two
    # fp32 summation orders (the ranks run batch 1, the reference batch 2: different split-K and atomics order) lands
    # 2 * lr away.  So: nothing further than that, and almost no element further than lr / 2.
    for name, k, lr in (("G", 0, 4e-4), ("D", 1, 2e-3)):
        w_ref = ref[k].params
        for rank in range(2):
            P = ranks[rank][k].params
            n = P.n_trainable      # kernels, biases, BatchNorm gamma / beta
            diff = (P.master[:n] - w_ref.master[:n]).abs()
            assert float(diff.max()) <= 2.5 * lr, (name, rank, float(diff.max()))
            assert float((diff > 0.5 * lr).double().mean()) <= 1e-3, (name, rank)
        # both ranks hold the same state bit for bit, D's rank-averaged moving statistics included
        assert torch.equal(ranks[0][k].params.master, ranks[1][k].params.master), name
