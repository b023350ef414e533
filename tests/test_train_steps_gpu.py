"""GPU parity of whole training steps (forward, backward, Adam, moving statistics) against the oracle's
autograd steps on identical weights, inputs and dropout masks."""
import numpy as np
import pytest
import torch

from oracle import nets as O

pytestmark = pytest.mark.gpu


def relerr(y, ref):
    y, ref = torch.as_tensor(y).double().cpu(), torch.as_tensor(ref).double().cpu()
    return float((y - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def rel_l2(y, ref):
    y, ref = torch.as_tensor(y).double().cpu(), torch.as_tensor(ref).double().cpu()
    return float((y - ref).norm() / ref.norm().clamp_min(1e-30))


def inputs(B, seed0=0):
    from cfd_amd import synthetic
    return synthetic.model_batch(range(seed0, seed0 + B))


def grads_keras(net):
    """the flat gradient buffer exported in Keras layouts (same transform as the weights)"""
    P = net.params
    saved = P.master
    tmp = torch.zeros_like(P.master)
    tmp[:P.n_trainable] = P.grad
    P.master = tmp
    try:
        return P.to_keras()
    finally:
        P.master = saved


def compare_grads(gk, ref_grads, prefix, tol, skip_bias_of_bn=True, report=None):
    worst = 0.0
    for k, g in ref_grads.items():
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        mine = torch.from_numpy(gk[name])
        gr = g.detach()
        if gr.abs().max() < 1e-6 * max(1.0, float(mine.abs().max())) or \
                (name.endswith(".bias") and (name[:-5] + ".bn.gamma") in gk):
            # bias before BatchNorm: the true gradient is 0; the oracle holds rounding noise, we hold exact zeros
            assert float(mine.abs().max()) <= 1e-4, name
            continue
        e = rel_l2(mine, gr)
        worst = max(worst, e)
        if report is not None:
            report.append((name, e))
        assert e <= tol, f"{name}: rel L2 {e:.3e} > {tol}"
    return worst


def compare_grads_bf16(gk, ref, probe, prefix, floor=2e-2):
    """bf16 step: per-tensor rel-L2 of the HIP gradients against the bf16-EMULATED oracle (oracle.nets.bf16_emulation:
    rounds where the HIP path stores bf16) must stay inside the band by which the emulated oracle's own gradients move
    when every stored activation is perturbed by an eighth of a bf16 ulp (`probe`), plus `floor` (the north-star
    bf16 tolerance).  That separates quantisation -- amplified by BatchNorm over 8-32 pixels and by ReLU / dropout kinks in
    the deepest layers -- from kernel error: a wrong kernel would leave the band, rounding noise cannot."""
    worst = (0.0, 0.0, "")
    for k, g in ref.items():
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        mine, gr, gp = torch.from_numpy(gk[name]), g.detach(), probe[k].detach()
        if gr.abs().max() < 1e-6 * max(1.0, float(mine.abs().max())) or \
                (name.endswith(".bias") and (name[:-5] + ".bn.gamma") in gk):
            assert float(mine.abs().max()) <= 1e-4, name
            continue
        e, band = rel_l2(mine, gr), rel_l2(gp, gr)
        if e > worst[0]:
            worst = (e, band, name)
        assert e <= band + floor, f"{name}: rel L2 {e:.3e} outside the perturbation band {band:.3e} + {floor}"
    return worst


def check_adam_delta(mine, ref_after, ref_before, lr, keys, dtype):
    """the first Adam step is ~lr*sign(g) per element; compare the applied updates, tolerating the few
    elements whose near-zero gradient changes sign with the summation order"""
    for k in keys:
        dm = torch.from_numpy(mine[k]).double() - ref_before[k].double()
        dr = ref_after[k].double() - ref_before[k].double()
        bad = ((dm - dr).abs() > 0.5 * lr).double().mean().item()
        assert bad <= (0.01 if dtype == torch.float32 else 0.08), (k, bad)
        assert float((dm.abs().mean() / lr)) > 0.1, k   # the step really happened


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 1.5e-1)])
def test_d_train_on_batch(dtype, tol):
    from cfd_amd import nets, trainer
    B = 2
    po = O.init_discriminator(seed=21, jitter=True)
    d = nets.Discriminator(dtype=dtype)
    d.params.load_keras(O.to_numpy(po))
    tr = trainer.DiscriminatorTrainer(d)
    x, u, _ = inputs(B)
    opt = O.KerasAdam(2e-3, 0.5)
    w0 = {k: v.clone() for k, v in po.items()}
    for step, label in enumerate((1.0, 0.0)):
        lab = np.full((B, 4, 32, 1), label, dtype=np.float32)
        loss_ref, gref = O.d_train_on_batch(po, opt, x, u, lab, return_grads=True)
        loss = tr.train_on_batch(torch.from_numpy(x).cuda(), torch.from_numpy(u).cuda(), label)
        if step == 0:
            assert abs(loss - loss_ref) <= (1e-4 if dtype == torch.float32 else 5e-3) * max(1.0, abs(loss_ref))
            if dtype == torch.bfloat16:
                refs = []
                for noise in (0.0, 5e-4):
                    with O.bf16_emulation(noise=noise):
                        refs.append(O.d_train_on_batch(dict(w0), O.KerasAdam(2e-3, 0.5), x, u, lab, return_grads=True))
                assert abs(loss - refs[0][0]) <= 2e-3 * max(1.0, abs(refs[0][0])), (loss, refs[0][0])
                e, band, name = compare_grads_bf16(grads_keras(d), refs[0][1], refs[1][1], "")
                print(f"D bf16 vs emulated oracle: loss {loss:.6f} vs {refs[0][0]:.6f}; worst grad {name} {e:.2e} (band {band:.2e})")
            rep = []
            w = compare_grads(grads_keras(d), gref, "", tol, report=rep)
            print(f"D step {step} {dtype}: loss {loss:.6f} vs {loss_ref:.6f}, worst grad rel-L2 {w:.3e}",
                  ", ".join(f"{n}={e:.1e}" for n, e in rep))
            wk = d.params.to_keras()
            check_adam_delta(wk, po, w0, 2e-3, ("c2.kernel", "c5.kernel", "c3.bn.gamma", "c6.kernel"), dtype)
            for k in ("c4.bn.moving_mean", "c4.bn.moving_var", "c2.bn.moving_mean"):   # momentum-0.99 update
                dm = torch.from_numpy(wk[k]) - w0[k]
                assert rel_l2(dm, po[k] - w0[k]) <= (1e-3 if dtype == torch.float32 else 5e-2), k
        else:
            # Adam's first step moves every weight by ~lr*sign(g): elements whose tiny gradient differs in sign
            # between two fp32 evaluation orders land 2*lr apart, so later steps agree only loosely
            assert abs(loss - loss_ref) <= 5e-2 * max(1.0, abs(loss_ref)), (loss, loss_ref)


# Gradient tolerances are rel-L2 per tensor.  They are looser than the forward tolerance for a structural
# reason: every ReLU/LeakyReLU derivative is a step function of the pre-activation, and an element whose
# pre-activation rounds to the other side of zero (fp32: ~1e-6 of them, bf16: ~3e-3 of them per layer)
# changes that element's gradient by O(1).  The error therefore grows with depth (printed per tensor).
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-2), (torch.bfloat16, 3.5e-1)])
@pytest.mark.parametrize("arch", ["U", "P"])
def test_gan_train_on_batch(arch, dtype, tol):
    from cfd_amd import nets, trainer
    B = 2
    pg = O.init_generator(arch, seed=22, jitter=True)
    pd = O.init_discriminator(seed=23, jitter=True)
    g = nets.Generator(arch, dtype=dtype)
    d = nets.Discriminator(dtype=dtype)
    g.params.load_keras(O.to_numpy(pg))
    d.params.load_keras(O.to_numpy(pd))
    tr = trainer.GanTrainer(g, d)
    x, u, _ = inputs(B)
    gen = torch.Generator().manual_seed(3)
    masks = [(torch.rand(s, generator=gen) < 0.5) for s in O.generator_dropout_shapes(arch, B)]
    opt = O.KerasAdam(4e-4, 0.5)
    pg0 = {k: v.clone() for k, v in pg.items()}
    pd0 = {k: v.clone() for k, v in pd.items()}
    lab = np.ones((B, 4, 32, 1), dtype=np.float32)
    ref, gref = O.gan_train_on_batch(pg, pd, opt, arch, x, lab, u, [m.float() for m in masks], return_grads=True)
    out = tr.train_on_batch(torch.from_numpy(x).cuda(), 1.0, torch.from_numpy(u).cuda(),
                            masks=[m.to(torch.uint8).cuda() for m in masks])
    ltol = 1e-4 if dtype == torch.float32 else 5e-3       # north_star's bf16 bar is 2e-2; measured 1.6e-3
    for a, b in zip(out, ref):
        assert abs(a - b) <= ltol * max(1.0, abs(b)), (out, ref)
    if dtype == torch.bfloat16:
        refs = []
        for noise in (0.0, 5e-4):
            with O.bf16_emulation(noise=noise):
                refs.append(O.gan_train_on_batch(dict(pg0), dict(pd0), O.KerasAdam(4e-4, 0.5), arch, x, lab, u,
                                                 [m.float() for m in masks], return_grads=True))
        for a, b in zip(out, refs[0][0]):
            assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (out, refs[0][0])
        e, band, name = compare_grads_bf16(grads_keras(g), refs[0][1], refs[1][1], "g.")
        print(f"GAN {arch} bf16 vs emulated oracle: losses {out} vs {refs[0][0]}; worst grad {name} {e:.2e} (band {band:.2e})")
    rep = []
    wg = compare_grads(grads_keras(g), gref, "g.", tol, report=rep)
    wd = compare_grads(grads_keras(d), {k: v for k, v in gref.items() if k.startswith("d.")}, "d.", tol)
    print(f"GAN {arch} {dtype}: losses {out} vs {ref}; worst grad rel-L2 G {wg:.3e} D-bn {wd:.3e}")
    print("   per-tensor:", ", ".join(f"{n}={e:.1e}" for n, e in rep))
    # after the Adam step: G weights moved like the oracle's; D conv weights untouched
    check_adam_delta(g.params.to_keras(), pg, pg0, 4e-4, ("e2.kernel", "d3.kernel", "d5.bn.gamma", "out.kernel"), dtype)
    dk = d.params.to_keras()
    check_adam_delta(dk, pd, pd0, 4e-4, ("c3.bn.gamma",), dtype)
    assert np.array_equal(dk["c3.kernel"], pd0["c3.kernel"].numpy())   # D's convs are frozen inside the GAN
    # test_on_batch: D on moving statistics
    t_ref = O.gan_test_on_batch(pg, pd, arch, x, lab, u, [m.float() for m in masks])
    t_out = tr.test_on_batch(torch.from_numpy(x).cuda(), 1.0, torch.from_numpy(u).cuda(),
                             masks=[m.to(torch.uint8).cuda() for m in masks])
    for a, b in zip(t_out, t_ref):
        assert abs(a - b) <= 5e-2 * max(1.0, abs(b)), (t_out, t_ref)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 3.5e-1)])
def test_cnn_train_on_batch(dtype, tol):
    from cfd_amd import nets, trainer
    B = 3
    po = O.init_cnn(seed=24, jitter=True)
    c = nets.CnnRegressor(dtype=dtype)
    c.params.load_keras(O.to_numpy(po))
    tr = trainer.CnnTrainer(c)
    x, _, _ = inputs(B)
    y = np.array([[0.3], [0.9], [0.1]], dtype=np.float32)
    mask = torch.rand((B, 256), generator=torch.Generator().manual_seed(1)) < 0.5
    opt = O.KerasAdam(1e-3, 0.5)
    po0 = {k: v.clone() for k, v in po.items()}
    loss_ref, gref = O.cnn_train_on_batch(po, opt, x, y, mask.float(), return_grads=True)
    loss = tr.train_on_batch(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(),
                             drop_mask=mask.to(torch.uint8).cuda())
    assert abs(loss - loss_ref) <= (1e-4 if dtype == torch.float32 else 5e-3) * max(1.0, abs(loss_ref)), (loss, loss_ref)
    if dtype == torch.bfloat16:
        with O.bf16_emulation():
            l_emu, g_emu = O.cnn_train_on_batch(dict(po0), O.KerasAdam(1e-3, 0.5), x, y, mask.float(), return_grads=True)
        assert abs(loss - l_emu) <= 1e-3 * max(1.0, abs(l_emu)), (loss, l_emu)
        w_emu = compare_grads(grads_keras(c), g_emu, "", 2e-2)          # no BatchNorm here: the plain bar holds
        print(f"CNN bf16 vs emulated oracle: loss {loss:.6f} vs {l_emu:.6f}; worst grad rel-L2 {w_emu:.2e}")
    rep = []
    w = compare_grads(grads_keras(c), gref, "", tol, report=rep)
    print(f"CNN {dtype}: loss {loss:.5f} vs {loss_ref:.5f}; worst grad rel-L2 {w:.3e}",
          ", ".join(f"{n}={e:.1e}" for n, e in rep))
    check_adam_delta(c.params.to_keras(), po, po0, 1e-3, ("g3.kernel", "g6.kernel", "fc2.kernel"), dtype)
    v_ref = 100.0 * float(O.huber(y, O.cnn_forward(po, x)))
    v = tr.test_on_batch(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    assert abs(v - v_ref) <= 5e-2 * max(1.0, abs(v_ref))


def test_upscaled_4x_preprocess_and_train_step():
    """BASELINE.json configs[4]: the 4x-pixel grid (128x1024) end to end -- scattered table -> GPU resample (bit-exact
    vs the oracle) -> GPU sign mask / DeltaP / signed distance (bit-exact) -> one U GAN training step in fp32 against
    the oracle's autograd step on the same 128x1024 input."""
    from cfd_amd import nets, ops, synthetic, trainer
    from oracle import dataset as OD, resample as OR
    H, W = 128, 1024
    arr = synthetic.scattered_case(5, jitter=0.1)
    rows = ops.resample_nearest([arr], H, W)[0]
    ref_rows, ties = OR.resample(arr, H, W, return_ties=True)
    assert not ties.any() and np.array_equal(rows, ref_rows)
    sdf, dp = ops.dataset_grids(rows[:, 4].reshape(1, H, W), cell=4.0 / H)
    sdf_ref, _p, u_ref, dp_ref = OD.grids_from_table(rows[:, 4], rows[:, 3], H, W)
    assert np.array_equal(sdf[0], sdf_ref) and dp[0] == dp_ref
    x = sdf[0][None, :, :, None].astype(np.float32)
    u = (u_ref / u_ref.max())[None, :, :, None].astype(np.float32)

    shape = (H, W, 1)
    pg = O.init_generator("U", seed=31, jitter=True)
    pd = O.init_discriminator(seed=32, jitter=True)
    g = nets.Generator("U", image_shape=shape, dtype=torch.float32)
    d = nets.Discriminator(image_shape=shape, dtype=torch.float32)
    g.params.load_keras(O.to_numpy(pg))
    d.params.load_keras(O.to_numpy(pd))
    tr = trainer.GanTrainer(g, d)
    gen = torch.Generator().manual_seed(4)
    masks = [(torch.rand(s, generator=gen) < 0.5) for s in O.generator_dropout_shapes("U", 1, shape)]
    lab = np.ones((1, H // 16, W // 16, 1), dtype=np.float32)
    ref, gref = O.gan_train_on_batch(pg, pd, O.KerasAdam(4e-4, 0.5), "U", x, lab, u, [m.float() for m in masks],
                                     return_grads=True)
    out = tr.train_on_batch(torch.from_numpy(x).cuda(), 1.0, torch.from_numpy(u).cuda(),
                            masks=[m.to(torch.uint8).cuda() for m in masks])
    for a, b in zip(out, ref):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (out, ref)
    wg = compare_grads(grads_keras(g), gref, "g.", 1e-2)
    print(f"4x grid U fp32: losses {out} vs {ref}; worst grad rel-L2 {wg:.3e}")


def test_side_stream_weight_gradients_equal_single_stream(monkeypatch):
    """CFD_AMD_WGRAD_STREAM=1 moves the weight-gradient GEMMs of G's and D's backward to a second stream; the gradients
    of one step from identical weights must equal the single-stream ones up to the atomics' commit order (a missing
    stream dependency would show as garbage or stale gradients)."""
    from cfd_amd import nets, trainer
    B = 2
    x, u, _ = inputs(B)
    x, u = torch.from_numpy(x).cuda(), torch.from_numpy(u).cuda()
    grads = []
    for flag in ("0", "1", "1"):
        monkeypatch.setenv("CFD_AMD_WGRAD_STREAM", flag)
        g = nets.Generator("P", dtype=torch.float32, seed=5)
        d = nets.Discriminator(dtype=torch.float32, seed=6)
        dtr, gtr = trainer.DiscriminatorTrainer(d), trainer.GanTrainer(g, d)
        assert trainer._side_wgrad(gtr.dp) == (flag == "1")
        la = dtr.train_on_batch(x, u, 1.0)
        gd = d.params.grad[:d.params.n_trainable].clone()
        lg = gtr.train_on_batch(x, 1.0, u, seed=3)
        torch.cuda.synchronize()
        grads.append((la, lg, gd, g.params.grad[:g.params.n_trainable].clone()))
    for la, lg, gd, gg in grads[1:]:
        assert np.allclose(la, grads[0][0], rtol=1e-5) and np.allclose(lg, grads[0][1], rtol=1e-5, atol=1e-6)
        assert rel_l2(gd, grads[0][2]) <= 2e-3 and rel_l2(gg, grads[0][3]) <= 2e-3     # run-to-run noise ~1e-5; stale = O(1)


def test_pipelined_step_equals_reference_order():
    """bench.py's training step enqueues the generator's forwards between a D update's all-reduce launch and its
    Adam step (DiscriminatorTrainer.begin / finish, GanTrainer before_d).  Same arithmetic as the reference's order
    (U/train.py:119-125): two steps of each from identical weights must give the same losses and weights."""
    from cfd_amd import nets, trainer
    B = 2
    x, u, _ = inputs(B)
    x, u = torch.from_numpy(x).cuda(), torch.from_numpy(u).cuda()

    def make():
        g = nets.Generator("P", dtype=torch.float32, seed=5)
        d = nets.Discriminator(dtype=torch.float32, seed=6)
        return g, d, trainer.DiscriminatorTrainer(d), trainer.GanTrainer(g, d)

    g1, d1, dtr1, gtr1 = make()
    g2, d2, dtr2, gtr2 = make()
    out1, out2 = [], []
    for step in range(2):
        fake = g1.forward(x, seed=2 * step)
        a = dtr1.train_on_batch(x, u, 1.0)
        b = dtr1.train_on_batch(x, fake, 0.0)
        out1.append([a, b] + gtr1.train_on_batch(x, 1.0, u, seed=2 * step + 1))

        dtr2.begin(x, u, 1.0)
        fake = g2.forward(x, seed=2 * step)
        a = dtr2.finish()
        dtr2.begin(x, fake, 0.0)
        res = {}
        c = gtr2.train_on_batch(x, 1.0, u, seed=2 * step + 1, before_d=lambda: res.setdefault("b", dtr2.finish()))
        out2.append([a, res["b"]] + c)
    # first step: same arithmetic up to atomics order; second step: downstream of Adam's sign-sensitive first update
    assert np.allclose(out1[0], out2[0], rtol=1e-5, atol=1e-6), (out1, out2)
    assert np.allclose(out1[1], out2[1], rtol=2e-3, atol=1e-5), (out1, out2)
    # Weights: the two orders run the same kernels on the same data, but the fp32 atomics inside them commit in a
    # different order from run to run, and Adam's first steps move every weight by ~lr * sign(g): an element whose
    # near-zero gradient changes sign lands 2 * lr away.  So: no element further than two such flips, and few flips.
    for n1, n2, lr in ((g1, g2, 4e-4), (d1, d2, 2e-3)):
        n = n1.params.n_trainable
        diff = (n1.params.master[:n] - n2.params.master[:n]).abs()
        assert float(diff.max()) <= 4.5 * lr, float(diff.max())
        assert float((diff > 0.5 * lr).double().mean()) <= 3e-3      # run-to-run noise of the atomics: seen 0.7e-3 .. 1.01e-3
        # moving statistics (D): relative, they scale with the activations
        assert rel_l2(n1.params.master[n:], n2.params.master[n:]) <= 1e-3 or n1.params.master[n:].numel() == 0
    with pytest.raises(RuntimeError):
        dtr2.finish()
