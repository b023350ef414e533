"""GPU parity of the backward / loss / optimiser kernels against torch autograd on the oracle's layers."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets as O
from oracle import keras_semantics as KS

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def relerr(y, ref):
    y, ref = y.double().cpu(), ref.double().cpu()
    return float((y - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def q(t, dtype):
    return t.to(dtype).float()


CONVS = [
    ("k4s2", 2, 8, 16, 64, 128), ("k4s2", 3, 4, 12, 128, 64), ("k4s1", 2, 4, 32, 128, 128),
    ("k3s1", 2, 8, 16, 64, 128), ("convT", 2, 4, 8, 128, 128), ("convT", 1, 1, 8, 64, 256),
    ("dense", 5, 1, 1, 512, 256), ("dense", 3, 1, 1, 128, 8), ("k4s1", 2, 4, 32, 64, 192), ("k3s1", 1, 6, 10, 64, 64),
]


def _fwd_ref(kind, x, wk, bias):
    if kind == "dense":
        return x.reshape(x.shape[0], -1) @ wk + bias
    if kind == "convT":
        return O.conv2d_transpose_same(x, wk, bias)
    s = 2 if kind == "k4s2" else 1
    return O.conv2d_same(x, wk, bias, s)


def _kernel_shape(kind, cin, cout):
    return {"dense": (cin, cout), "convT": (4, 4, cout, cin), "k3s1": (3, 3, cin, cout)}.get(kind, (4, 4, cin, cout))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,B,H,W,cin,cout", CONVS)
def test_conv_backward(kind, B, H, W, cin, cout, dtype):
    """dX (forward kernel on transposed weights) and dW (MFMA wgrad) vs autograd"""
    from cfd_amd import ops
    dev = "cuda"
    code = {"k4s2": ops.CONV_K4S2, "k4s1": ops.CONV_K4S1, "k3s1": ops.CONV_K3S1, "convT": ops.CONVT_K4S2,
            "dense": ops.DENSE}[kind]
    x = q(rnd((B, H, W, cin), 1), dtype).requires_grad_(True)
    wk = q(rnd(_kernel_shape(kind, cin, cout), 2, 0.05), dtype).requires_grad_(True)
    bias = rnd((cout,), 3, 0.1)
    y = _fwd_ref(kind, x, wk, bias)
    dy = q(rnd(y.shape, 4), dtype)
    y.backward(dy)
    # product path
    pack = {"dense": ops.pack_dense_weight, "convT": ops.pack_convT_weight}.get(kind, ops.pack_conv_weight)
    wp = pack(wk.detach().to(dev), dtype)                      # [Cout][taps][Cin]
    xd = x.detach().to(dev).to(dtype)
    dyd = dy.to(dev).to(dtype)
    if kind == "dense":
        xd = xd.reshape(B, cin)
    # dW
    dw = torch.zeros(wp.shape, dtype=torch.float32, device=dev)
    ops.conv_wgrad(xd, dyd, code, cout, dw)
    dw_ref = pack(wk.grad.to(dev), torch.float32)
    assert relerr(dw, dw_ref) <= 1e-4, "dW"
    dw2 = torch.zeros_like(dw)
    ops.conv_wgrad(xd, dyd, code, cout, dw2, split_m=3)
    assert relerr(dw2, dw_ref) <= 1e-4, "dW split"
    # dX
    flip = kind in ("k3s1", "k4s1")
    wt = ops.transpose_weight(wp, flip)
    bkind = {"k4s2": ops.CONVT_K4S2, "convT": ops.CONV_K4S2, "k3s1": ops.CONV_K3S1, "k4s1": ops.CONV_K4S1_BWD,
             "dense": ops.DENSE}[kind]
    if cout % 64 == 0:
        dx = ops.conv_forward(dyd, wt, None, bkind, cin, out_dtype=torch.float32)
        assert relerr(dx.reshape(x.shape), x.grad) <= 1e-4, "dX"
    else:   # Dense with a handful of outputs (fc3): dX through the small-K kernel
        dx = torch.empty((B, cin), dtype=dtype, device=dev)
        ops.tap_dx(dyd.float().contiguous(), wp, dx, cout)
        assert relerr(dx.float().reshape(x.shape), x.grad) <= TOL[dtype], "dX small"
    # dbias
    db = torch.zeros(cout, device=dev)
    ops.colsum(dyd, db)
    assert relerr(db, dy.reshape(-1, cout).sum(0)) <= 1e-4


WG256 = [
    ("k4s2", 2, 16, 32, 256, 256, 1),    # reduction grid 8x16, gathered X with padding on every side
    ("k4s2", 3, 8, 128, 256, 512, 3),    # Wm = 64: a K tile is one grid row; ragged split
    ("k4s1", 2, 4, 32, 256, 256, 2),     # asymmetric 'same' padding
    ("k3s1", 2, 8, 16, 256, 256, 1),     # 9 taps
    ("convT", 2, 8, 16, 256, 256, 2),    # gathered operand = dY (co side)
    ("convT", 1, 1, 8, 512, 256, 1),     # M = 8 < one K tile: ragged tail rows must read as zero
    ("dense", 70, 1, 1, 256, 256, 1),    # 1x1 grid, M = 70 (ragged)
    ("k4s2", 2, 16, 32, 128, 256, 1),    # Cin = 128: a tile's ci columns are two adjacent taps (the discriminator's c2)
    ("k4s2", 3, 8, 128, 128, 512, 3),    # the same with a ragged split and one grid row per K tile
    ("k4s1", 2, 4, 32, 128, 256, 2),     # stride 1, asymmetric padding: the second tap of a pair leaves the image on the right
    ("k3s1", 2, 8, 16, 128, 256, 1),     # nine taps: pairs cross kernel rows, the ninth tap has no partner (DeltaP g4)
    ("k3s1", 3, 4, 32, 128, 512, 2),
]


@pytest.mark.parametrize("kind,B,H,W,cin,cout,split", WG256)
def test_conv_wgrad256(kind, B, H, W, cin, cout, split):
    """the 256x256 ping-pong bf16 weight-gradient kernel (wgrad256.hip), forced on, vs autograd"""
    from cfd_amd import ops
    dev, dtype = "cuda", torch.bfloat16
    code = {"k4s2": ops.CONV_K4S2, "k4s1": ops.CONV_K4S1, "k3s1": ops.CONV_K3S1, "convT": ops.CONVT_K4S2,
            "dense": ops.DENSE}[kind]
    x = q(rnd((B, H, W, cin), 1), dtype).requires_grad_(True)
    wk = q(rnd(_kernel_shape(kind, cin, cout), 2, 0.05), dtype).requires_grad_(True)
    y = _fwd_ref(kind, x, wk, torch.zeros(cout))
    dy = q(rnd(y.shape, 4), dtype)
    y.backward(dy)
    pack = {"dense": ops.pack_dense_weight, "convT": ops.pack_convT_weight}.get(kind, ops.pack_conv_weight)
    dw_ref = pack(wk.grad.to(dev), torch.float32)
    xd = x.detach().to(dev).to(dtype)
    dyd = dy.to(dev).to(dtype)
    if kind == "dense":
        xd = xd.reshape(B, cin)
    # operands as channel slices of wider buffers (the concat layout the nets use)
    xw = torch.zeros(xd.shape[:-1] + (cin + 64,), dtype=dtype, device=dev)
    xw[..., 64:] = xd
    dw = torch.zeros(dw_ref.shape, dtype=torch.float32, device=dev)
    ops.set_tuning(ops.TUNE_WGRAD256, 1)
    try:
        ops.conv_wgrad(xw[..., 64:], dyd, code, cout, dw, split_m=split)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning(ops.TUNE_WGRAD256, -1)
    assert relerr(dw, dw_ref) <= 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C,drop,two", [(2, 4, 32, 128, False, True), (3, 2, 16, 64, True, False),
                                              (1, 16, 64, 64, False, False),
                                              (2, 64, 128, 256, True, False),     # large: two-pass pixel-parallel path
                                              (2, 32, 128, 512, False, True)])
def test_bn_act_bwd(B, H, W, C, drop, two, dtype):
    from cfd_amd import ops
    dev = "cuda"
    x = (q(rnd((B, H, W, C), 1) * 0.7 + 0.3, dtype)).requires_grad_(True)
    gamma = (1 + 0.1 * rnd((C,), 2)).requires_grad_(True)
    beta = (0.1 * rnd((C,), 3)).requires_grad_(True)
    p = {"l.bn.gamma": gamma, "l.bn.beta": beta}
    st = {}
    bn = O.batchnorm(x, p, "l", True, True, st)
    mask = None
    z = bn
    if drop:
        mask = torch.rand((B, H, W, C), generator=torch.Generator().manual_seed(5)) < 0.5
        z = bn * mask.float() * 2.0
    g1 = q(rnd((B, H, W, C), 6), dtype)
    g2 = q(rnd((B, H, W, C), 7), dtype) if two else None
    if two:   # encoder: y1 = leaky(bn), y2 = relu(bn)
        loss = (O.leaky(z) * g1).sum() + (torch.relu(bn) * g2).sum()
        act1 = ops.ACT_LEAKY
    else:     # decoder: y1 = relu(drop(bn))
        loss = (torch.relu(z) * g1).sum()
        act1 = ops.ACT_RELU
    loss.backward()
    mean, var = st["l"]
    dx = torch.empty((B, H, W, C), dtype=dtype, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.bn_act_bwd(x.detach().to(dev).to(dtype), mean.to(dev).contiguous(), var.to(dev).contiguous(),
                   gamma.detach().to(dev), beta.detach().to(dev), g1.to(dev).to(dtype), act1, dx, dg, db,
                   g2=None if g2 is None else g2.to(dev).to(dtype), act2=ops.ACT_RELU,
                   drop_mask=None if mask is None else mask.to(torch.uint8).to(dev),
                   sums=torch.zeros(2 * B * C, device=dev))
    assert relerr(dx.float(), x.grad) <= TOL[dtype]
    assert relerr(dg, gamma.grad) <= 1e-4
    assert relerr(db, beta.grad) <= 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_act_pool_bwd(dtype):
    from cfd_amd import ops
    dev = "cuda"
    pre = q(rnd((2, 8, 12, 64), 1), dtype).requires_grad_(True)
    y = O.leaky(pre)
    g1, g2 = q(rnd(y.shape, 2), dtype), q(rnd(y.shape, 3), dtype)
    ((y * g1).sum() + (torch.relu(pre) * g2).sum()).backward()
    dx = torch.empty(y.shape, dtype=dtype, device=dev)
    ops.act_bwd(y.detach().to(dev).to(dtype), g1.to(dev).to(dtype), ops.ACT_LEAKY, dx, g2=g2.to(dev).to(dtype),
                act2=ops.ACT_RELU)
    assert relerr(dx.float(), pre.grad) <= TOL[dtype]
    # maxpool(leaky(pre)) backward
    # (the activation is the leaf, already representable in the compute dtype, so that both sides see
    #  the same window maxima; first maximum in row-major window order wins ties on both sides)
    a = q(O.leaky(rnd((2, 8, 12, 64), 4)), dtype).requires_grad_(True)
    pooled = F.max_pool2d(a.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    gp = q(rnd(pooled.shape, 5), dtype)
    (pooled * gp).sum().backward()
    expect = a.grad * torch.where(a.detach() > 0, 1.0, 0.2)
    dx2 = torch.empty(a.shape, dtype=dtype, device=dev)
    ops.maxpool2_act_bwd(a.detach().to(dev).to(dtype), gp.to(dev).to(dtype), dx2, ops.ACT_LEAKY)
    assert relerr(dx2.float(), expect) <= TOL[dtype]


def test_losses():
    from cfd_amd import ops
    dev = "cuda"
    z = rnd((3, 4, 32, 1), 1, 2.0).requires_grad_(True)
    t = (torch.rand(z.shape, generator=torch.Generator().manual_seed(2)) < 0.5).float()
    for target, tc in ((None, 1.0), (None, 0.0), (t, 0.0)):
        z.grad = None
        tt = torch.full_like(z, tc) if target is None else target
        l = O.bce_from_logits(z, tt)
        l.backward()
        loss = torch.zeros(1, device=dev)
        dz = torch.empty(z.shape, device=dev)
        ops.bce_logits(z.detach().to(dev), loss, dz, target=None if target is None else target.to(dev),
                       target_const=tc)
        assert abs(float(loss) - float(l)) <= 1e-5 * max(1.0, abs(float(l)))
        assert relerr(dz, z.grad) <= 1e-5
    # field losses on a sigmoid output, with an extra upstream gradient
    logit = rnd((2, 16, 32, 1), 3).requires_grad_(True)
    tgt = torch.rand(logit.shape, generator=torch.Generator().manual_seed(4))
    extra = rnd(logit.shape, 5, 1e-3)
    for kind, fn, w in ((ops.MAE, O.mae, 100.0), (ops.MSE, O.mse, 1.0), (ops.HUBER, lambda a, b: O.huber(a, b, 0.3), 100.0)):
        logit.grad = None
        y = torch.sigmoid(logit)
        l = fn(tgt, y)
        (w * l + (y * extra).sum()).backward()
        loss = torch.zeros(1, device=dev)
        dz = torch.empty(logit.shape, device=dev)
        ops.field_loss(y.detach().to(dev), tgt.to(dev), kind, loss, dz, weight=w, delta=0.3,
                       dy_extra=extra.to(dev), sigmoid_grad=True)
        assert abs(float(loss) - float(l)) <= 1e-5 * max(1.0, abs(float(l)))
        assert relerr(dz, logit.grad) <= 1e-4


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_adam_matches_keras_form(cdt):
    from cfd_amd import ops
    dev = "cuda"
    n = 4096 + 64
    w0 = rnd((n,), 1)
    p = {"w": w0.clone()}
    opt = O.KerasAdam(4e-4, 0.5)
    w = w0.clone().to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    wc = torch.zeros(n, dtype=cdt, device=dev) if cdt == torch.bfloat16 else None
    for t in range(1, 4):
        g = rnd((n,), 10 + t, 0.01)
        opt.step(p, {"w": g.clone()})
        alpha = 4e-4 * math.sqrt(1 - KS.ADAM_BETA_2 ** t) / (1 - 0.5 ** t)
        ops.adam_step(w, (2.0 * g).to(dev), m, v, wc, alpha, 0.5, grad_scale=0.5)
    assert relerr(w, p["w"]) <= 1e-6
    if wc is not None:
        assert torch.equal(wc, w.to(torch.bfloat16))


def test_bn_update_moving():
    from cfd_amd import ops
    dev = "cuda"
    C, B = 96, 3
    mm, mv = rnd((C,), 1), 1 + rnd((C,), 2).abs()
    mean, var = rnd((B, C), 3), rnd((B, C), 4).abs()
    p = {"l.bn.moving_mean": mm.clone(), "l.bn.moving_var": mv.clone()}
    O.update_moving(p, {"l": (mean, var)})
    a, b = mm.clone().to(dev), mv.clone().to(dev)
    ops.bn_update_moving(a, b, mean.to(dev), var.to(dev))
    assert relerr(a, p["l.bn.moving_mean"]) <= 1e-6 and relerr(b, p["l.bn.moving_var"]) <= 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [0, 1])
def test_tap_family_backward(geom, dtype):
    """one-output-channel layers: generator `out` (Conv2DTranspose) and discriminator c6 (Conv2D k4 s1)"""
    from cfd_amd import ops
    dev = "cuda"
    B, H, W, Cc = 2, 4, 16, 128
    x = q(rnd((B, H, W, Cc), 1), dtype).requires_grad_(True)
    if geom == 0:
        wk = q(rnd((4, 4, 1, Cc), 2, 0.05), dtype).requires_grad_(True)
        y = O.conv2d_transpose_same(x, wk, torch.zeros(1))
        wp = ops.pack_convT_weight(wk.detach().to(dev), dtype)      # [1][4][4][C]
        wg = lambda: ops.pack_convT_weight(wk.grad.to(dev), torch.float32)
    else:
        wk = q(rnd((4, 4, Cc, 1), 2, 0.05), dtype).requires_grad_(True)
        y = O.conv2d_same(x, wk, torch.zeros(1), 1)
        wp = ops.pack_conv_weight(wk.detach().to(dev), dtype)
        wg = lambda: ops.pack_conv_weight(wk.grad.to(dev), torch.float32)
    dz = rnd(y.shape, 3)
    y.backward(dz)
    dtaps = torch.empty((B * H * W, 16), device=dev)
    ops.tap_scatter_grad(dz.to(dev), dtaps, B, H, W, geom)
    dx = torch.empty((B, H, W, Cc), dtype=dtype, device=dev)
    ops.tap_dx(dtaps, wp, dx, 16)
    assert relerr(dx.float(), x.grad) <= TOL[dtype]
    dw = torch.zeros((1, 4, 4, Cc), device=dev)
    ops.tap_dw(dtaps, x.detach().to(dev).to(dtype), dw, 16)
    assert relerr(dw, wg()) <= 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W", [(2, 16, 48), (3, 6, 22)])        # the second: a pixel count that is no multiple of a step
@pytest.mark.parametrize("k,s,c1,c2,cout", [(4, 2, 1, 0, 256), (4, 2, 1, 1, 128), (3, 1, 1, 0, 64), (3, 1, 1, 0, 96),
                                            (4, 2, 1, 0, 40)])     # 96 / 40: channels that do not fill their thread group
def test_conv_first_wgrad(k, s, c1, c2, cout, B, H, W, dtype):
    from cfd_amd import ops
    dev = "cuda"
    x = rnd((B, H, W, c1), 1)
    x2 = rnd((B, H, W, c2), 4) if c2 else None
    wk = rnd((k, k, c1 + c2, cout), 2, 0.1).requires_grad_(True)
    bias = rnd((cout,), 3, 0.1).requires_grad_(True)
    xin = x if x2 is None else torch.cat([x, x2], -1)
    pre = O.conv2d_same(xin, wk, bias, s)
    dpre = q(rnd(pre.shape, 5), dtype)
    pre.backward(dpre)
    dw = torch.zeros((cout, k, k, c1 + c2), device=dev)
    db = torch.zeros(cout, device=dev)
    ops.conv_first_wgrad(x.to(dev), None if x2 is None else x2.to(dev), dpre.to(dev).to(dtype), dw, db, k, s)
    assert relerr(dw, ops.pack_conv_weight(wk.grad.to(dev), torch.float32)) <= 1e-4
    assert relerr(db, bias.grad) <= 1e-4


def test_dropout_apply_matches_mask():
    from cfd_amd import ops
    dev = "cuda"
    x = rnd((4, 256), 1).to(dev)
    y = torch.empty_like(x)
    ops.dropout_apply(x, y, 0.5, seed=9, elem0=0)
    m = ops.dropout_mask((4, 1, 1, 256), 0.5, 9).view(4, 256)
    assert torch.equal(y, x * m.float() * 2.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cout,kh,cin,flip", [(256, 4, 128, False), (72, 3, 40, True), (130, 4, 6, False), (8, 1, 1024, False)])
def test_transpose_weight_exact(cout, kh, cin, flip, dtype):
    """[Cout][kh][kw][Cin] -> [Cin][taps'][Cout], taps reversed when flip: a pure permutation, so bit-exact (the bf16
    16-byte path for channel counts that are multiples of 8, the scalar path otherwise)"""
    from cfd_amd import ops
    w = torch.randn((cout, kh, kh, cin), device="cuda").to(dtype)
    wt = ops.transpose_weight(w, flip)
    ref = w.reshape(cout, kh * kh, cin)
    if flip:
        ref = ref.flip(1)
    ref = ref.permute(2, 1, 0).contiguous()
    assert torch.equal(wt.reshape(cin, kh * kh, cout), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 8, 12, 64), (8, 128, 128, 128), (3, 16, 24, 512), (5, 2, 2, 256)])
def test_act_pool_bwd_with_bias_gradient(shape, dtype):
    """cfd_act_bwd_dbias / cfd_maxpool2_act_bwd_dbias: the same dx as the plain kernels bit for bit, and the bias
    gradient = column sums of dx AS STORED (what the separate cfd_colsum pass gave), accumulated into dbias"""
    from cfd_amd import ops
    dev = "cuda"
    C = shape[-1]
    y = q(O.leaky(rnd(shape, 1)), dtype).to(dev).to(dtype)
    g1, g2 = q(rnd(shape, 2), dtype).to(dev).to(dtype), q(rnd(shape, 3), dtype).to(dev).to(dtype)
    dx0, dx1 = torch.empty(shape, dtype=dtype, device=dev), torch.empty(shape, dtype=dtype, device=dev)
    ops.act_bwd(y, g1, ops.ACT_LEAKY, dx0, g2=g2, act2=ops.ACT_RELU)
    db = torch.full((C,), 0.5, dtype=torch.float32, device=dev)              # accumulated INTO
    ops.act_bwd(y, g1, ops.ACT_LEAKY, dx1, g2=g2, act2=ops.ACT_RELU, dbias=db)
    assert torch.equal(dx0, dx1)
    ref = dx0.double().sum(dim=(0, 1, 2))
    # (fp32 partial sums + atomics: error relative to the sum of magnitudes)
    assert float(((db.double() - 0.5) - ref).abs().max()) <= 2e-6 * float(dx0.double().abs().sum(dim=(0, 1, 2)).max()) + 1e-5
    B, H, W, _ = shape
    gp = q(rnd((B, H // 2, W // 2, C), 5), dtype).to(dev).to(dtype)
    p0, p1 = torch.empty(shape, dtype=dtype, device=dev), torch.empty(shape, dtype=dtype, device=dev)
    ops.maxpool2_act_bwd(y, gp, p0, ops.ACT_LEAKY)
    db2 = torch.zeros((C,), dtype=torch.float32, device=dev)
    ops.maxpool2_act_bwd(y, gp, p1, ops.ACT_LEAKY, dbias=db2)
    assert torch.equal(p0, p1)
    ref2 = p0.double().sum(dim=(0, 1, 2))
    assert float((db2.double() - ref2).abs().max()) <= 2e-6 * float(p0.double().abs().sum(dim=(0, 1, 2)).max()) + 1e-5
