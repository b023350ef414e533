"""world_size-2 checks of the data-parallel plumbing on CPU (gloo): bucketed gradient all-reduce, the
1/world scaling that Adam applies, and contiguous case sharding for inference.  The kernels themselves
need a GPU; here the flat gradient buffers are plain CPU tensors."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cfd_amd import trainer, sharding
    dp = trainer._DP(bucket_elems=1000)            # several buckets
    g = torch.arange(4096, dtype=torch.float32) * (rank + 1)
    dp.reduce_async(g)
    dp.wait()
    expect = torch.arange(4096, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(g, expect) and abs(dp.grad_scale - 1.0 / world) < 1e-12
    # backward-overlap buckets: ranges handed over tail first, disjoint, covering the trainable prefix exactly once
    class _Spec:
        def __init__(self, offset):
            self.offset = offset

    class _P:
        n_main, n_trainable = 6000, 6400
        specs = {"e1.kernel": _Spec(0), "e2.kernel": _Spec(1024), "d1.kernel": _Spec(3000), "out.kernel": _Spec(5500)}
        grad = torch.ones(6400, dtype=torch.float32) * (rank + 1)
    bk = trainer.BackwardBuckets(dp, _P, min_elems=2000)
    for layer in ("out", "d1", "e2", "e1"):
        bk.ready(layer)
    bk.finish()
    dp.wait()
    tot = float(sum(r + 1 for r in range(world)))
    spans = sorted(bk.sent)
    ok = ok and spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans[:-2], spans[1:-1])) \
        and spans[-2][1] == 6000 and spans[-1] == (6000, 6400) and bk.sent[0] == (3000, 6000) \
        and bool((_P.grad == tot).all())
    lo, hi = sharding.case_range(10001, rank, world)
    out.put((rank, ok, lo, hi))
    dist.destroy_process_group()


def test_dp_allreduce_and_sharding_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res)
    # contiguous, disjoint, covering
    assert res[0][2] == 0 and res[0][3] == res[1][2] and res[1][3] == 10001
    assert abs((res[0][3] - res[0][2]) - (res[1][3] - res[1][2])) <= 1


def test_case_range_properties():
    from cfd_amd import sharding
    for n in (0, 1, 7, 10000):
        for w in (1, 2, 4, 8):
            spans = [sharding.case_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
