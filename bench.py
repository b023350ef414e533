#!/usr/bin/env python
"""bench.py -- throughput of the CFD-surrogate hot path on MI355X (contract: task statement ④).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--batch B] [--no-also]

One "step" = one pass of the hot path over one batch of synthetic geometries resident in HBM.
Workloads (BASELINE.json configs):
  infer_pd_fp32   (headline, configs[1]) P generator + DeltaP CNN batched inference, fp32 MFMA
  infer_all_bf16  (configs[2])           U + P generators + DeltaP CNN batched inference, bf16 MFMA
  train_u_bf16    (configs[3])           U GAN training step, bf16, data parallel (B = 16 and 32 per GPU)
  resample        (north_star item 1)    scattered points -> 64x512 grids, fp64, queries/s against the HBM roofline
Rank 0 prints ONE JSON line: `value` etc. are the headline workload's; with no --workload the same line carries an
`also` object with the other workloads, each with its own ms_per_step and live HIP-event roofline (and, for training,
cpu_baseline).  Inference and the resample shard by case with no data-path collective ("scaling": "weak"); training
exchanges gradients over RCCL every step.
`--gpus N` with N > 1: under `python -m torch.distributed.run` (WORLD_SIZE set) this process is one rank; started
plainly, it starts the N ranks itself as child processes (before anything touches a GPU) and relays rank 0's line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X peaks (MI355X_MICROARCH.md): dense MFMA TFLOP/s by arithmetic type, HBM GB/s
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
PEAK_HBM_GBS = 8000.0
# forward GFLOP per sample of the GEMM-shaped layers only (SURVEY.md §2.1 minus the direct first/last layers)
WORKLOADS = {
    "infer_pd_fp32": dict(nets=["P", "DeltaP"], dtype="f32", batch=128),
    "infer_all_bf16": dict(nets=["U", "P", "DeltaP"], dtype="bf16", batch=128),
    # configs[3]: U surrogate training, bf16, data parallel; one step = the reference's four calls
    # (U/train.py:119-125): G predict, D on real, D on fake, composite GAN update -- incl. Adam and all-reduce
    "train_u_bf16": dict(nets=["U", "D"], dtype="bf16", batch=16, train=True),
    "train_u_fp32": dict(nets=["U", "D"], dtype="f32", batch=4, train=True),
    "train_p_bf16": dict(nets=["P", "D"], dtype="bf16", batch=16, train=True),      # P/train.py: the same step on the P generator
    # the DeltaP CNN's training step (DeltaP/train.py:131: one train_on_batch = forward, Huber loss, backward, Adam)
    "train_deltap_bf16": dict(nets=["DeltaP"], dtype="bf16", batch=64, train=True),
}


# Control-flow test hook (tests/test_host_logic_cpu.py): no GPU, gloo ranks, every workload replaced by a canned result
# marked "dry" -- exercises the spawner, the watchdog and the exit codes, measures nothing.
DRY = os.environ.get("CFD_BENCH_DRY") == "1"


def spawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` started plainly: start N ranks as fresh child processes (nothing in this process has
    touched a GPU: torch.cuda.device_count() does not initialise one) and relay their output; rank 0 prints the line."""
    import socket
    import subprocess
    have = n if DRY else torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    # poll all ranks: when one dies the others would sit in their next collective until the RCCL timeout (or for ever),
    # and this parent with them -- end them and return the first failure's code
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    q.terminate()
                t_end = time.time() + 10.0
                for q in live:
                    try:
                        q.wait(max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
    return rc


def dist_setup(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if DRY:
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        return rank, world, local
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return rank, world, local


def dry_workload(name, rank, world):
    """CFD_BENCH_DRY: one collective per workload (so a rank that never arrives hangs the others, as on RCCL)"""
    if os.environ.get("CFD_BENCH_HANG_EXTRA") == name and rank == world - 1:
        time.sleep(3600)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(torch.ones(1))
    return {"value": 0.0, "unit": "samples/s", "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": 0.0, "dtype": "none",
            "config": {"workload": name, "dry": True}, "roofline": None}


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if not DRY:
        torch.cuda.synchronize()


def build_nets(names, tdt):
    from cfd_amd import nets
    out = {}
    for n in names:
        if n in ("U", "P"):
            out[n] = nets.Generator(n, dtype=tdt, seed={"U": 11, "P": 13}[n])   # fixed: str hashes differ per process
        else:
            out[n] = nets.CnnRegressor(dtype=tdt, seed=5)
    return out


SYNC3 = os.environ.get("CFD_BENCH_SYNC3", "0") == "1"     # A/B: read every loss at its own call (three host syncs per step)


class TrainJob:
    def __init__(self, tdt, arch="U"):
        from cfd_amd import nets, trainer
        self.g = nets.Generator(arch, dtype=tdt, seed=7)
        self.d = nets.Discriminator(dtype=tdt, seed=8)
        self.dtr = trainer.DiscriminatorTrainer(self.d)
        self.gtr = trainer.GanTrainer(self.g, self.d)
        self.dloss = torch.zeros(2, dtype=torch.float32, device="cuda")

    def step(self, x, y, step, sample0):
        """The reference's four calls (U/train.py:119-125).  The generator's forwards do not read D's weights, so
        they are enqueued between a D update's all-reduce launch and its Adam step: same arithmetic as the
        reference's order, with D's gradient exchange hidden behind them when world > 1.  The three losses are read
        once, after the third call (the reference only logs them): one host sync per step instead of three."""
        self.dtr.begin(x, y, 1.0, loss=self.dloss[0:1])                   # d_model.train_on_batch([A,B], ones) ...
        fake = self.g.forward(x, seed=2 * step, sample0=sample0, inference=True)   # generate_fake_samples (utils.py:52-62)
        self.dtr.finish(sync=SYNC3)                                       # ... its all-reduce, Adam
        self.dtr.begin(x, fake, 0.0, loss=self.dloss[1:2])                # d_model.train_on_batch([A,fake], zeros)
        out = self.gtr.train_on_batch(x, 1.0, y, seed=2 * step + 1, sample0=sample0,
                                      before_d=lambda: self.dtr.finish(sync=SYNC3))        # gan_model.train_on_batch
        return self.dloss.tolist() + out


class CnnJob:
    """DeltaP/train.py:131: model.train_on_batch(X, y) on the CNN regressor (Huber, Adam(1e-3, 0.5))"""

    def __init__(self, tdt):
        from cfd_amd import nets, trainer
        self.c = nets.CnnRegressor(dtype=tdt, seed=5)
        self.tr = trainer.CnnTrainer(self.c)

    def step(self, x, y, step, sample0):
        return self.tr.train_on_batch(x, y, seed=step, sample0=sample0)


def run_step(models, x, step, sample0):
    for name, m in models.items():
        if name == "DeltaP":
            m.forward(x)
        else:
            m.forward(x, seed=step, sample0=sample0, inference=True)      # predict: no backward follows


def cpu_baseline(names, seconds_budget=20.0):
    """The oracle (PyTorch-CPU fp32 restatement of the Keras graphs) on the host cores, batch 1 per
    predict call like the reference's valid.py loop.  Reported only."""
    from oracle import nets as O
    from cfd_amd import synthetic
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    params = {}
    for n in names:
        params[n] = O.init_cnn(seed=5) if n == "DeltaP" else O.init_generator(n, seed=3)
    x, _, _ = synthetic.model_batch([0])
    masks = {n: [torch.ones(s) for s in O.generator_dropout_shapes(n, 1)] for n in names if n != "DeltaP"}

    def one():
        with torch.no_grad():
            for n in names:
                if n == "DeltaP":
                    O.cnn_forward(params[n], x)
                else:
                    O.generator_forward(params[n], x, n, masks[n])
    one()  # warm-up
    t0 = time.perf_counter()
    n_done = 0
    while True:
        one()
        n_done += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n_done >= 64:
            break
    return dict(value=n_done / el, unit="samples/s", cores=cores, kind="port",
                sample=f"{n_done} synthetic 64x512 cases, batch 1 per call, nets {'+'.join(names)}, oracle (PyTorch-CPU fp32)")


_cpu_train_cache = {}


def cpu_baseline_train(seconds_budget=25.0):
    """The oracle's U training step (batch 1, reference order U/train.py:116-125) on the host cores (timed once per
    process: every U training workload of the line reports the same figure)."""
    if "r" not in _cpu_train_cache:
        _cpu_train_cache["r"] = _cpu_baseline_train(seconds_budget)
    return dict(_cpu_train_cache["r"])


def _cpu_baseline_train(seconds_budget):
    from oracle import nets as O
    from cfd_amd import synthetic
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    pg, pd = O.init_generator("U", seed=3), O.init_discriminator(seed=4)
    od, og = O.KerasAdam(2e-3, 0.5), O.KerasAdam(4e-4, 0.5)
    x, u, _ = synthetic.model_batch([0])
    ones, zeros = np.ones((1, 4, 32, 1), np.float32), np.zeros((1, 4, 32, 1), np.float32)
    masks = [torch.ones(s) for s in O.generator_dropout_shapes("U", 1)]

    def one():
        with torch.no_grad():
            fake = O.generator_forward(pg, x, "U", masks).numpy()
        O.d_train_on_batch(pd, od, x, u, ones)
        O.d_train_on_batch(pd, od, x, fake, zeros)
        O.gan_train_on_batch(pg, pd, og, "U", x, ones, u, masks)
    one()
    t0, n = time.perf_counter(), 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 16:
            break
    return dict(value=n / el, unit="samples/s", cores=cores, kind="port",
                sample=f"{n} U training steps at batch 1 (G predict + 2 D updates + GAN update), oracle (PyTorch-CPU fp32)")


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources and the host modules that choose and launch them
    (cfd_amd/csrc/*, cfd_amd/*.py): identifies the code a profile was taken on (the GPU box has no .git;
    scripts/collect_profiles.sh stamps the same digest into every profile it collects).  bench.py itself is left out:
    the workloads are named in the profile's own bench line, and an edit to the reporting code changes no kernel."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "cfd_amd", "csrc", "*"))) + sorted(glob.glob(os.path.join(ROOT, "cfd_amd", "*.py"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


_mfma_probe_memo = {}


def mfma_register_only():
    """What the matrix cores sustain in THIS process on THIS chip with nothing else in the loop (cfd_mfma_probe: bf16
    16x16x32 MFMAs on register operands, 8 waves per CU), on random operands and on zeros -- measured live, HIP events,
    median of 5 launches of ~2 ms after 3 warm-up launches.  Context for the bf16 rooflines: `peak` stays the 2.5 PFLOP/s
    of the guide; the clock the chip holds under MFMA load depends on how many operand bits toggle."""
    if _mfma_probe_memo:
        return _mfma_probe_memo
    from cfd_amd import ops
    out = torch.empty(256 * 512, dtype=torch.float32, device="cuda")
    res = {}
    for name in ("random", "zero"):
        src = (torch.randn(65536 * 8, device="cuda") if name == "random" else torch.zeros(65536 * 8, device="cuda")).to(torch.bfloat16)
        for _ in range(3):
            flop = ops.mfma_probe(src, out, 4096)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            a.record()
            ops.mfma_probe(src, out, 4096)
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)[2]
        res[name + "_operands"] = round(flop / (ms * 1e-3) / 1e12, 1)
    res.update(unit="TFLOP/s", kernel="cfd_mfma_probe: v_mfma_f32_16x16x32_bf16 on register operands, 8 waves per CU, no LDS / memory in the loop",
               note="measured in this run; the ceiling of any schedule on operands that toggle, beside the guide's 2500 (peak)")
    _mfma_probe_memo.update(res)
    return _mfma_probe_memo


def profile_source(path, summary):
    """where a PMC-derived figure in the line comes from: NOT this run -- a committed profile of the same command"""
    return {"file": os.path.relpath(path, ROOT), "git_head": summary.get("git_head"), "csrc_sha16": summary.get("csrc_sha16"),
            "same_sources_as_this_run": (summary.get("csrc_sha16") == csrc_sha16()) if summary.get("csrc_sha16") else None,
            "note": "rocprofv3 --pmc passes of this bench command, collected separately (scripts/collect_profiles.sh); "
                    "not measured in this run"}


def pmc_traffic(workload):
    """HBM bytes per launch of the MFMA conv kernels for this workload, from the committed PMC passes
    (profiles/*_<workload>_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same
    command, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM), the time-weighted MFMA-busy fraction of the same
    profile, and the profile's identity.  (None, None, None) when no profile of the workload is committed."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_summary.json")))
    if not fs:
        return None, None, None
    try:
        summary = json.load(open(fs[-1]))
        ks = summary["kernels"]
    except Exception:
        return None, None, None
    n = b = 0
    busy = []
    for name, k in ks.items():
        if name.startswith(("igemm", "wgrad")) and "hbm_read_bytes_per_launch" in k:
            n += k["launches"]
            b += k["launches"] * (k["hbm_read_bytes_per_launch"] + k.get("hbm_write_bytes_per_launch", 0))
            if "mfma_busy_frac" in k:
                busy.append((k["total_us"], k["mfma_busy_frac"]))
    if not n:
        return None, None, None
    mb = sum(t * f for t, f in busy) / sum(t for t, _ in busy) if busy else None
    return round(b / n), (round(mb, 4) if mb is not None else None), profile_source(fs[-1], summary)


def pmc_traffic_resample():
    """HBM bytes of ONE cfd_resample_nearest call (all of its launches together), from the committed PMC passes of
    `bench.py --workload resample` (profiles/r*_resample_summary.json); None when none is committed."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_resample_summary.json")))
    if not fs:
        return None, None
    try:
        summary = json.load(open(fs[-1]))
        ks = summary["kernels"]
        calls = ks["nearest_kernel"]["launches"]
        tot = sum(k["launches"] * (k.get("hbm_read_bytes_per_launch", 0) + k.get("hbm_write_bytes_per_launch", 0))
                  for n, k in ks.items() if n != "nearest_kernel_counted")
        return round(tot / calls), profile_source(fs[-1], summary)
    except Exception:
        return None, None


def timed_steps(step_fn, steps, warmup, world):
    """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks; -> seconds"""
    for s in range(warmup):
        step_fn(s)
    barrier(world)
    t0 = time.perf_counter()
    for s in range(steps):
        step_fn(warmup + s)
    barrier(world)
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def run_model_workload(name, B, steps, warmup, rank, world, cpu_base):
    """one of the model workloads -> the fields of its JSON object (every rank runs it; rank 0 reports)"""
    from cfd_amd import ops, synthetic
    wl = WORKLOADS[name]
    B = B or wl["batch"]
    tdt = torch.float32 if wl["dtype"] == "f32" else torch.bfloat16
    # synthetic geometries of the reference shape, resident in HBM before the timed region
    xs, us, _ = synthetic.model_batch([rank * B + i for i in range(B)])
    x = torch.from_numpy(xs).cuda()
    train = wl.get("train", False)
    if train and wl["nets"] == ["DeltaP"]:
        job = CnnJob(tdt)
        y = torch.from_numpy(np.asarray([[0.1 + 0.01 * (i % 7)] for i in range(B)], dtype=np.float32)).cuda()   # DeltaP targets
    elif train:
        job = TrainJob(tdt, wl["nets"][0])
        y = torch.from_numpy(us).cuda()
    if train:
        def step_fn(step):
            job.step(x, y, step, rank * B)
    else:
        models = build_nets(wl["nets"], tdt)

        def step_fn(step):
            run_step(models, x, step, rank * B)

    dt = timed_steps(step_fn, steps, warmup, world)

    # ---- roofline of the dominant kernels (the MFMA implicit GEMMs): HIP events around every launch, same workload
    # (a training step holds collectives, so every rank runs this pass when training; rank 0 reports it)
    roof = None
    if rank == 0 or (train and world > 1):
        # The brackets must see kernel time only: if the GPU ran dry, the host time between an event record and the
        # launch behind it would count.  ONE calibrated on-stream delay ahead of the pass lets the host run a few steps
        # ahead of the GPU, and the first step behind it is not counted.  (A delay ahead of EVERY step, as round 1 had it,
        # idles the chip for 4 ms out of every 13: the clock governor answers with a lower clock for the layers that
        # follow -- scripts/net_layer_times.py --sleep: 8.11 ms of GEMM time per bf16 inference step against 7.55 without.)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        torch.cuda._sleep(1_000_000)
        c1.record()
        torch.cuda.synchronize()
        step_ms = dt / steps * 1e3
        head_start = int(1_000_000 * max(4.0, 1.5 * step_ms) / max(c0.elapsed_time(c1), 1e-3))
        n_roof = min(steps, 5)
        kt = ops.KernelTimer(sync=False)   # (creates its events here, ahead of the delay)
        torch.cuda.synchronize()
        torch.cuda._sleep(head_start)
        # (the training steps put their weight-gradient GEMMs on a second stream by default; brackets on two streams
        # overlap and would count the shared time twice, so THIS pass runs them on the launch stream: the roofline is of
        # the kernels, `ms_per_step` above is of the step as shipped)
        wstream = os.environ.get("CFD_AMD_WGRAD_STREAM")
        os.environ["CFD_AMD_WGRAD_STREAM"] = "0"
        try:
            step_fn(999)                   # not counted: runs while the host queues the counted steps
            with kt:
                for s in range(n_roof):
                    step_fn(1000 + s)
        finally:
            if wstream is None:
                del os.environ["CFD_AMD_WGRAD_STREAM"]
            else:
                os.environ["CFD_AMD_WGRAD_STREAM"] = wstream
        sm = kt.summary()
        achieved = sm["total_flops"] / (sm["total_ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[wl["dtype"]]
        # the committed PMC passes were taken at the workload's default batch; they say nothing about another one
        traffic, busy, src = pmc_traffic(name) if B == wl["batch"] else (None, None, None)
        roof = dict(bound="mfma", kernel="igemm256_kernel+igemm_kernel" + ("+wgrad_kernel" if train else ""),
                    achieved=round(achieved, 2), peak=peak, unit="TFLOP/s",
                    frac=round(achieved / peak, 4), traffic=traffic, mfma_busy_frac_pmc=busy, traffic_source=src,
                    launches_per_step=sm["launches"] // n_roof,
                    avg_launch_us=round(sm["total_ms"] * 1e3 / sm["launches"], 2))
        if wl["dtype"] == "bf16":
            roof["mfma_register_only"] = mfma_register_only()
    out = {"value": round(world * B * steps / dt, 2), "unit": "samples/s", "n_gpus": world, "steps": steps,
           "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "dtype": wl["dtype"],
           "config": {"workload": name, "nets": wl["nets"], "per_gpu_batch": B, "global_batch": B * world,
                      "grid": "64x512", "parallelism": (f"dp{world}, RCCL reduce-scatter + sharded Adam + all-gather"
                                                        if train else f"case-sharded x{world}, no collectives")},
           "roofline": roof}
    if train:
        out["steps_per_s"] = round(steps / dt, 2)      # configs[0] is quoted per step (the reference's loop is batch 1)
    if cpu_base and rank == 0 and world == 1:
        if train and wl["nets"] == ["DeltaP"]:
            pass                                        # (no oracle timing of the CNN's step: reported without a baseline)
        else:
            out["cpu_baseline"] = cpu_baseline_train() if train else cpu_baseline(wl["nets"])
    return out


def run_resample_workload(steps, warmup, rank, world, cpu_base, ncase=256):
    """north_star item 1 / SURVEY.md §8(d): `ncase` synthetic scattered cases (101 slices x 21 points, jittered) ->
    64x512 grids per step, inputs and outputs resident in HBM; queries/s = pixels resampled per second."""
    from cfd_amd import ops, synthetic
    H, W = 64, 512
    cases = [synthetic.scattered_case(rank * ncase + i, jitter=0.2) for i in range(ncase)]
    job = ops.ResampleJob(cases, H, W)
    dt = timed_steps(lambda _s: job.run(), steps, warmup, world)
    roof = None
    if rank == 0:
        n_roof = min(steps, 5)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_roof)]
        torch.cuda.synchronize()
        for a, b in ev:
            a.record()
            job.run()
            b.record()
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in ev) / n_roof
        gbs = job.algorithmic_bytes() / (ms * 1e-3) / 1e9
        tr_bytes, tr_src = pmc_traffic_resample()
        roof = dict(bound="hbm", kernel="one cfd_resample_nearest call = case_prepare_kernel (a block per case: slice walls, cell "
                                        "grid, counting sort) + nearest_kernel",
                    achieved=round(gbs, 2), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 5),
                    traffic=tr_bytes, traffic_source=tr_src, algorithmic_bytes_per_launch=job.algorithmic_bytes(), avg_launch_us=round(ms * 1e3, 2))
    roof2 = None
    if rank == 0:
        # The bound DESIGN section 5 names for the search kernel: not HBM but the FP64 vector rate (and, beside it, a chain of
        # dependent loads per block that no roofline prices).  Work = MEASURED distance evaluations of one call
        # (cfd_resample_nearest_counted) x 5 flops (two subtractions, two products, one sum; -ffp-contract=off, so no FMA)
        # against the non-packed FP64 vector peak, 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 Tflop/s without FMA
        # (78.6 with); time = the whole call.
        evals = job.count_evaluations()
        gf = evals * 5.0 / (ms * 1e-3) / 1e12
        peak64 = 256 * 4 * 16 * 2.4e9 / 1e12
        roof2 = dict(bound="valu_fp64", kernel="nearest_kernel (window scan + ring search)", achieved=round(gf, 3), peak=round(peak64, 1),
                     unit="TFLOP/s", frac=round(gf / peak64, 4), evaluations_per_call=evals,
                     evaluations_per_pixel=round(evals / (ncase * H * W), 2), flops_per_evaluation=5,
                     note="fp64 add/mul without FMA (bit-exactness forbids contraction); the call's time, launches included")
    out = {"value": round(world * ncase * H * W * steps / dt, 1), "unit": "queries/s", "n_gpus": world, "steps": steps,
           "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "dtype": "f64",
           "cases_per_s": round(world * ncase * steps / dt, 1),
           "config": {"workload": "resample", "cases_per_step": ncase, "points_per_case": int(job.n_points // ncase),
                      "grid": "64x512", "parallelism": f"case-sharded x{world}, no collectives"},
           "roofline": roof, "roofline_fp64": roof2}
    if cpu_base and rank == 0 and world == 1:
        from oracle import resample as OR
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 10.0 and n < len(cases):
            OR.resample(cases[n], H, W)
            n += 1
        el = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=round(n * H * W / el, 1), unit="queries/s", cores=1, kind="port",
                                   sample=f"{n} of the same cases through oracle/resample.py (numpy + scipy cKDTree, one thread)")
    return out


def free_device_memory():
    import gc
    from cfd_amd import ops
    gc.collect()
    ops._ws_cache.clear()
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS) + ["resample"])
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="only the headline workload")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank, world, local = dist_setup(args.gpus)
    cpu_base = not args.no_cpu_baseline
    name = args.workload or "infer_pd_fp32"
    out = {"metric": "surrogate samples/sec (infer + train) at 1/2/4/8 GPUs; field rel-err vs CPU ref"}
    # With several ranks a hang in a collective would otherwise sit until the RCCL timeout (or for ever): a watchdog per
    # phase prints what HAS been measured (rank 0) and leaves with a NON-ZERO code, naming the workload in flight -- a
    # hang must never read as a green run.
    in_flight = {"workload": name, "phase": "headline"}
    watchdog = [None]

    def arm(seconds):
        if world == 1:
            return
        import threading
        if watchdog[0] is not None:
            watchdog[0].cancel()

        def bail():
            out["watchdog"] = (f"rank {rank}: workload {in_flight['workload']!r} ({in_flight['phase']}) did not finish within "
                               f"{seconds:.0f} s -- most likely a collective that not every rank reached; line printed by "
                               f"the watchdog, exit code 3")
            if rank == 0:
                print(json.dumps(out), flush=True)
            print(out["watchdog"], file=sys.stderr, flush=True)
            os._exit(3)
        watchdog[0] = threading.Timer(seconds + 5.0 * rank, bail)
        watchdog[0].daemon = True
        watchdog[0].start()

    wd_s = float(os.environ.get("CFD_BENCH_WATCHDOG_S", "0") or 0)
    arm(wd_s or 900.0)
    if DRY:
        res = dry_workload(name, rank, world)
    elif name == "resample":
        res = run_resample_workload(args.steps, args.warmup, rank, world, cpu_base)
    else:
        res = run_model_workload(name, args.batch, args.steps, args.warmup, rank, world, cpu_base)
    out.update({k: res[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")})
    out.update({"higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": res["dtype"],
                "data": "synthetic", "config": res["config"], "roofline": res["roofline"]})
    for k2 in ("cpu_baseline", "roofline_fp64", "steps_per_s", "cases_per_s"):
        if k2 in res:
            out[k2] = res[k2]
    failed = None
    if args.workload is None and not args.no_also:
        # the other BASELINE configs under the same clock: fewer steps each, same contract (warm-up, barrier, max over ranks)
        also = {}
        out["also"] = also
        k, w = max(3, min(args.steps, 10)), min(args.warmup, 2)
        # The headline above is measured and is printed whatever happens below.  A workload that raises is recorded as an
        # error in the line (and, with several ranks, ends the extras: the ranks may no longer agree on the next
        # collective) -- and the process then exits NON-ZERO (4) after printing the line.
        for key, wname, B, base in EXTRAS:
            in_flight.update(workload=key, phase="also")
            arm(wd_s or 600.0)
            try:
                if os.environ.get("CFD_BENCH_FAIL_EXTRA") == key:      # test hook: tests/test_host_logic_cpu.py
                    raise RuntimeError("forced by CFD_BENCH_FAIL_EXTRA")
                if DRY:
                    also[key] = dry_workload(key, rank, world)
                    continue
                free_device_memory()
                if wname is None:
                    also[key] = run_resample_workload(k, w, rank, world, base and cpu_base)
                else:
                    also[key] = run_model_workload(wname, B, k, w, rank, world, base and cpu_base)
            except Exception as e:   # noqa: BLE001 -- reported in the line, never silently dropped
                also[key] = {"error": f"{type(e).__name__}: {e}"[:400]}
                failed = key
                if world > 1:
                    break
    if watchdog[0] is not None:
        watchdog[0].cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if failed is not None:
        print(f"bench.py: extra workload {failed!r} failed (see its 'error' in the line); exit code 4", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(4)      # (with several ranks they may disagree on the next collective: leave without one)
    if world > 1:
        import torch.distributed as dist
        barrier(world)
        dist.destroy_process_group()


# the other workloads of the default line: (key in `also`, workload, per-GPU batch (0 = its default), cpu_baseline?)
EXTRAS = (("infer_all_bf16", "infer_all_bf16", 0, False),
          ("train_u_bf16", "train_u_bf16", 16, True),
          ("train_u_bf16_b32", "train_u_bf16", 32, False),
          ("train_u_fp32_b1", "train_u_fp32", 1, True),       # BASELINE configs[0]: the reference's own loop shape (batch 1, fp32)
          ("train_deltap_bf16", "train_deltap_bf16", 0, False),
          ("resample", None, 0, True))


if __name__ == "__main__":
    main()
