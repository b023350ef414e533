"""The drop-in boundary end to end on the GPU: preprocess CLI -> train script (+ resume) -> evaluation,
through the same names, flags and files the reference's scripts use."""
import os
import sys

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_processed_folder(tmp, n_cases):
    """raw OpenFOAM-like tables -> processed 64x512 CSVs via the preprocess module (GPU resample)"""
    sys.path.insert(0, ROOT)
    import pixels_coords_testing as pct
    from cfd_amd import synthetic
    folder = tmp / "cases" / "processed"
    folder.mkdir(parents=True)
    for s in range(n_cases):
        arr = synthetic.scattered_case(s, jitter=0.2)
        pd.DataFrame(arr, columns=["x", "y", "z", "p", "Umean"]).to_csv(folder / f"case_{s:03d}.csv", index=False)
    pct.height, pct.width = 64, 512
    pct.process_all_csvs_with_pixel_coordinates(folder)
    df = pd.read_csv(folder / "case_000.csv")
    assert list(df.columns) == pct.OUT_COLUMNS and len(df) == 64 * 512
    return str(folder) + "/"


def test_models_namespace_matches_reference():
    sys.path.insert(0, os.path.join(ROOT, "U"))
    import importlib
    for pkg, names in (("U", ["define_generator", "define_discriminator", "define_gan", "save_model", "load_model",
                              "MC_mse", "MC_mae", "MC_huber_loss"]),):
        sys.modules.pop("models", None)
        sys.path.insert(0, os.path.join(ROOT, pkg))
        m = importlib.import_module("models")
        for n in names:
            assert hasattr(m, n), n
        sys.path.pop(0)
        sys.modules.pop("models", None)


@pytest.mark.parametrize("field", ["U", "P", "DeltaP"])
def test_train_resume_evaluate(field, tmp_path, monkeypatch):
    """<field>/train.py -n 1, resume, valid.py, test.py end to end.  P: the targets are divided by the training
    set's maximum (P/train.py:79-82) and the GAN step is the U script's (P/train.py:117-146)."""
    from cfd_amd import cli, keras_api as K
    folder = make_processed_folder(tmp_path, 20)
    monkeypatch.chdir(tmp_path)
    hist = cli.train(field, ["-n", "1", "-t", folder])
    out = tmp_path / field / os.path.basename(os.path.normpath(cli.FIELDS[field]["default_dir"]))
    assert (out / "steps.txt").read_text() == "18"            # 20 cases -> 18 train / 2 valid, batch 1
    assert len(np.loadtxt(out / "train_indices.txt")) == 18 and len(np.loadtxt(out / "valid_indices.txt")) == 2
    assert (out / "model_ckpt" / "model_G.keras").exists()
    if field in ("U", "P"):
        assert (out / "model_ckpt" / "model_D.keras").exists() and (out / "model_ckpt" / "model_GAN.keras").exists()
        # U/train.py:155-157 saves the best-validation generator; P/train.py:166-168 has that save commented out
        assert (out / "model_eval.keras").exists() == (field == "U")
    assert len(hist["g"]) == 1 and np.isfinite(hist["g"][0]) and np.isfinite(hist["val"][0])
    # resume from the checkpoint: continues at the epoch boundary, optimizer state restored
    hist2 = cli.train(field, ["-c", "true", "-n", "2", "-t", folder])
    assert (out / "steps.txt").read_text() == "36" and len(hist2["g"]) == 1
    # weights survive save -> load bit for bit
    g = K.load_model(str(out / "model_ckpt" / "model_G.keras"), compile=False)
    w = g.get_weights_dict()
    K.save_model(g, str(tmp_path / "again.keras"))
    w2 = K.load_model(str(tmp_path / "again.keras")).get_weights_dict()
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    # <field>/valid.py: U scores the last 10 % of the folder, DeltaP the valid_indices.txt cases the training saved
    rows, mean_err = cli.evaluate(field, folder, model=g, out_dir=str(tmp_path / "eval"), mode="valid")
    assert len(rows) == (20 - int(0.9 * 20 + 1) if field == "U" else 2) and np.isfinite(mean_err)
    assert (tmp_path / "eval" / cli.EVAL_CSV).exists()
    # <field>/test.py: every case of the folder, scale from the training folder + train_indices.txt
    rows, mean_err = cli.evaluate(field, folder, model=g, out_dir=str(tmp_path / "eval_t"), mode="test", train_folder=folder)
    assert len(rows) == 20 and np.isfinite(mean_err)


def test_keras_surface_shapes():
    from cfd_amd import keras_api as K
    d = K.define_discriminator((64, 512, 1))
    g = K.define_generator_P((64, 512, 1))
    gan = K.define_gan(g, d, (64, 512, 1))
    assert d.output_shape[1] == 4                                     # U/train.py:41
    assert all(isinstance(l, K.BatchNormalization) or not l.trainable for l in d.layers)
    x = np.random.default_rng(0).standard_normal((2, 64, 512, 1)).astype("float32")
    y = g.predict([x])
    assert y.shape == (2, 64, 512, 1) and y.dtype == np.float32
    out = gan.train_on_batch([x], [np.ones((2, 4, 32, 1)), np.clip(y, 0, 1)])
    assert len(out) == 3 and abs(out[0] - (out[1] + 100 * out[2])) < 1e-3 * max(1, abs(out[0]))
    assert g.count_params() == 103_314_177 and d.count_params() == 27_821_953


def test_predict_hip_graph_and_empty_batch(monkeypatch):
    """CFD_AMD_GRAPHS=1: small-batch predict replays one captured HIP graph.  The DeltaP CNN (no dropout at inference)
    must give the eager result bit for bit; the generators draw fresh dropout on every replay (device-resident epoch,
    cfd_dropout_epoch) and stay within the spread of eager draws.  An empty batch returns an empty array."""
    from cfd_amd import keras_api as K, synthetic
    x, _, _ = synthetic.model_batch([0, 1])
    c = K.define_cnn_model((64, 512, 1))
    g = K.define_generator_P((64, 512, 1))
    monkeypatch.setenv("CFD_AMD_GRAPHS", "0")
    yc = c.predict(x)
    ye = np.stack([g.predict(x) for _ in range(4)])
    monkeypatch.setenv("CFD_AMD_GRAPHS", "1")
    for _ in range(2):
        assert np.array_equal(c.predict(x), yc)
    a, b = g.predict(x), g.predict(x)
    assert a.shape == (2, 64, 512, 1) and np.isfinite(a).all() and (a != b).any()
    spread = np.abs(ye - ye.mean(0)).max()
    assert np.abs(a - ye.mean(0)).max() < 4 * spread + 1e-3 and np.abs(b - ye.mean(0)).max() < 4 * spread + 1e-3
    assert g.predict(x[:0]).shape == (0, 64, 512, 1) and c.predict(x[:0]).shape == (0, 1)
    # the epoch the replays advanced touches only bit-63 seeds: training-range seeds still give one mask in the
    # forward (elementwise.hip) and backward (train.hip) kernels, and the backward refuses an epoch-keyed seed
    import torch
    from cfd_amd import _lib, ops
    t = torch.randn((4, 256), device="cuda")
    y = torch.empty_like(t)
    ops.dropout_apply(t, y, 0.5, seed=9, elem0=0)
    assert torch.equal(y, t * ops.dropout_mask((4, 1, 1, 256), 0.5, 9).view(4, 256).float() * 2.0)
    m1 = ops.dropout_mask((1, 2, 2, 64), 0.5, (1 << 63) | 5)
    ops.dropout_epoch(1)
    assert not torch.equal(m1, ops.dropout_mask((1, 2, 2, 64), 0.5, (1 << 63) | 5))
    r = torch.zeros((1, 2, 2, 64), device="cuda")
    v = torch.ones((1, 64), device="cuda")
    with pytest.raises(_lib.CfdError):
        ops.bn_act_bwd(r, v, v, v[0], v[0], r, ops.ACT_RELU, torch.empty_like(r), torch.zeros(64, device="cuda"),
                       torch.zeros(64, device="cuda"), drop_rate=0.5, drop_seed=(1 << 63) | 5)
    ops.dropout_epoch(0, set_=True)


def test_predict_in_chunks_equals_one_pass(monkeypatch):
    """A predict over more cases than CFD_AMD_PREDICT_CHUNK runs chunk by chunk (a ragged last chunk included) and
    returns what one pass over the whole batch returns: per-sample BatchNorm statistics, dropout keyed by the sample
    index.  Only the split-K factor (fp32 summation order) may differ with the batch size, as in
    test_generator_batch_equals_single.  The DeltaP CNN has no dropout at inference; U has three dropout layers."""
    import torch
    from cfd_amd import keras_api as K, synthetic
    monkeypatch.setenv("CFD_AMD_DTYPE", "fp32")
    monkeypatch.setenv("CFD_AMD_GRAPHS", "0")       # batches <= 8 would replay a graph with an epoch-keyed draw
    x, _, _ = synthetic.model_batch(range(5))
    xd = torch.from_numpy(x).cuda()
    g = K.define_generator_U((64, 512, 1))
    c = K.define_cnn_model((64, 512, 1))
    g._seed = 77
    seed = g._next_seed()
    whole = g.net.forward(xd, seed=seed).cpu().numpy().copy()
    whole_c = c.net.forward(xd, training=False).cpu().numpy().copy()
    monkeypatch.setenv("CFD_AMD_PREDICT_CHUNK", "2")
    g._seed = 77
    y = g.predict(x)
    g._seed = 77
    yd = g.predict_on_device(x).cpu().numpy()
    yc = c.predict(x)
    assert y.shape == whole.shape and yc.shape == (5, 1)
    assert np.abs(y - whole).max() <= 1e-5 * np.abs(whole).max()
    assert np.abs(y - yd).max() <= 1e-5 * np.abs(whole).max()    # two runs differ by the commit order of the BN atomics
    assert np.abs(yc - whole_c).max() <= 1e-5 * max(np.abs(whole_c).max(), 1e-30)
    # a different chunking of the same call: the same draw again (sample-indexed dropout), chunk 3 + 2
    monkeypatch.setenv("CFD_AMD_PREDICT_CHUNK", "3")
    g._seed = 77
    assert np.abs(g.predict(x) - whole).max() <= 1e-5 * np.abs(whole).max()


@pytest.mark.parametrize("mode", ["compile", "call"])
def test_discriminator_own_step_freeze_switch(mode, monkeypatch):
    """define_gan sets trainable=False on D's conv layers AFTER D was compiled (U/models/generator.py:79-81).
    CFD_AMD_D_FREEZE=compile (default, Keras-2 snapshot): D's own train_on_batch still updates its convolutions;
    =call (Keras-3 reading): only BatchNorm gamma/beta move.  The composite never touches D's convolutions."""
    from cfd_amd import keras_api as K, synthetic
    monkeypatch.setenv("CFD_AMD_D_FREEZE", mode)
    x, u, _ = synthetic.model_batch([0])
    d = K.define_discriminator((64, 512, 1))
    g = K.define_generator_P((64, 512, 1))
    gan = K.define_gan(g, d, (64, 512, 1))
    P = d.net.params
    k0, g0 = P.m("c3.kernel").clone(), P.m("c3.bn.gamma").clone()
    d.train_on_batch([x, u], np.ones((1, 4, 32, 1)))
    assert d.d_step_trains_convs() == (mode == "compile")
    assert (not bool((P.m("c3.kernel") == k0).all())) == (mode == "compile")
    assert not bool((P.m("c3.bn.gamma") == g0).all())
    k1 = P.m("c3.kernel").clone()
    gan.train_on_batch([x], [np.ones((1, 4, 32, 1)), u])
    assert bool((P.m("c3.kernel") == k1).all())


def _keras_style_archive(path, datasets, image_shape, empty=()):
    """a `.keras` archive laid out the way Keras 3 writes it (config.json = a Functional model config, metadata.json,
    model.weights.h5) -- built by hand: Keras is not installed, so this is the published layout, not Keras' own output"""
    import json
    import zipfile
    from cfd_amd import h5
    tmp = str(path) + ".h5"
    h5.write(tmp, datasets, empty_groups=("vars",) + tuple(empty))
    cfg = {"module": "keras", "class_name": "Functional", "registered_name": "Functional",
           "config": {"name": "functional", "trainable": True, "layers": [
               {"module": "keras.layers", "class_name": "InputLayer", "registered_name": None,
                "config": {"batch_shape": [None] + list(image_shape), "dtype": "float32", "sparse": False, "name": "input_layer"}}]}}
    with zipfile.ZipFile(str(path), "w") as z:
        z.writestr("metadata.json", json.dumps({"keras_version": "3.3.3", "date_saved": "2024-06-01@00:00:00"}))
        z.writestr("config.json", json.dumps(cfg))
        z.write(tmp, "model.weights.h5")
    os.remove(tmp)


def _golden_fill(var_id, shape, positive=False):
    """the content formula of tests/golden/make_keras_archive.py, restated (data, not layout)"""
    n = int(np.prod(shape)) if len(shape) else 1
    v = ((np.arange(n, dtype=np.int64) * 7919 + var_id * 104729) % 2003).astype(np.float32) / 2003.0 - 0.5
    v = v * 0.05
    if positive:
        v = np.abs(v) + 0.01
    return v.reshape(shape)


def test_load_model_reads_hand_laid_out_archive(tmp_path, monkeypatch):
    """(f3) without testing the reader with the writer: tests/golden/make_keras_archive.py lays out `.keras` archives of
    the DeltaP CNN and the discriminator with every HDF5 path and the optimizer's variable order TYPED OUT from Keras 3's
    published saving rule, imports nothing from this repository and writes with h5py under the image's conda
    interpreter.  load_model must put dataset k into the tensor this test names -- also literally -- for it.  Still
    unpinned against real Keras: both sides are readings of the same published rule."""
    import shutil
    import subprocess
    from cfd_amd import h5, keras_api as K
    py = next((p for p in ("/opt/conda/bin/python3.9", shutil.which("python3.9") or "") if p and os.path.exists(p)), None)
    if not h5.available() or not py or subprocess.run([py, "-c", "import h5py, numpy"], capture_output=True).returncode:
        pytest.skip("no HDF5 C library / no interpreter with h5py in this image")
    monkeypatch.setenv("CFD_AMD_DTYPE", "fp32")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_keras_archive.py")
    H, W = 16, 64
    # ---- DeltaP CNN: variable id -> our tensor, in the order DeltaP/models/generator.py:23-46 creates the layers
    r = subprocess.run([py, script, "cnn", str(tmp_path / "cnn.keras"), str(H), str(W)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cnn_vars = ["g1.kernel", "g1.bias", "g3.kernel", "g3.bias", "g4.kernel", "g4.bias", "g6.kernel", "g6.bias",
                "fc1.kernel", "fc1.bias", "fc2.kernel", "fc2.bias", "fc3.kernel", "fc3.bias"]
    m = K.load_model(str(tmp_path / "cnn.keras"))
    assert isinstance(m, K.CnnModel) and m.image_shape == (H, W, 1)
    w = m.get_weights_dict()
    for vid, name in enumerate(cnn_vars):
        assert np.array_equal(w[name], _golden_fill(vid, w[name].shape)), name
    opt = m._tr().opt
    assert opt.t == 5
    names, mom, vel = K._opt_keras(m.net.params, opt)
    assert names == cnn_vars                                     # Adam's slots follow trainable_variables
    for vid, name in enumerate(cnn_vars):
        assert np.array_equal(mom[name], _golden_fill(1000 + vid, mom[name].shape)), name
        assert np.array_equal(vel[name], _golden_fill(2000 + vid, vel[name].shape, True)), name
    x = np.random.default_rng(0).standard_normal((2, H, W, 1)).astype("float32")
    assert np.isfinite(m.predict(x)).all() and np.isfinite(m.train_on_batch(x, np.zeros((2, 1), "float32")))
    # ---- discriminator, compiled before define_gan froze anything: Adam over kernels, biases, gamma, beta
    r = subprocess.run([py, script, "disc", str(tmp_path / "d.keras"), str(H), str(W)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    disc_vars = ["c1.kernel", "c1.bias",
                 "c2.kernel", "c2.bias", "c2.bn.gamma", "c2.bn.beta", "c2.bn.moving_mean", "c2.bn.moving_var",
                 "c3.kernel", "c3.bias", "c3.bn.gamma", "c3.bn.beta", "c3.bn.moving_mean", "c3.bn.moving_var",
                 "c4.kernel", "c4.bias", "c4.bn.gamma", "c4.bn.beta", "c4.bn.moving_mean", "c4.bn.moving_var",
                 "c5.kernel", "c5.bias", "c5.bn.gamma", "c5.bn.beta", "c5.bn.moving_mean", "c5.bn.moving_var",
                 "c6.kernel", "c6.bias"]
    d = K.load_model(str(tmp_path / "d.keras"))
    assert isinstance(d, K.DiscriminatorModel) and d._d_freeze == "compile"
    wd = d.get_weights_dict()
    assert set(wd) == set(disc_vars)
    for vid, name in enumerate(disc_vars):
        assert np.array_equal(wd[name], _golden_fill(vid, wd[name].shape, name.endswith("moving_var"))), name
    optd = d._tr().opt
    assert optd.t == 3
    names, mom, vel = K._opt_keras(d.net.params, optd)
    trainable = [n for n in disc_vars if "moving" not in n]
    assert names == trainable
    for name in trainable:
        vid = disc_vars.index(name)
        assert np.array_equal(mom[name], _golden_fill(1000 + vid, mom[name].shape)), name
        assert np.array_equal(vel[name], _golden_fill(2000 + vid, vel[name].shape, True)), name


def test_load_model_reads_keras3_archives(tmp_path, monkeypatch):
    """`.keras` archives as Keras 3 lays them out (zip + model.weights.h5 with
    layers/<class>_<k>/vars/<i> and optimizer/vars/<i>) load into the HIP nets: graph recognised from the shapes, image
    shape from the InputLayer, weights and Adam state placed by Keras' variable order.  And save_model writes that same
    member, so its weights file is what `model.load_weights` of real Keras expects."""
    import zipfile
    from cfd_amd import h5, keras_api as K, keras_h5
    if not h5.available():
        pytest.skip("no HDF5 C library")
    monkeypatch.setenv("CFD_AMD_DTYPE", "fp32")
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 64, 512, 1)).astype("float32")
    # ---- DeltaP CNN with optimizer state: train two steps, hand the state over in Keras' layout, continue on both
    a = K.define_cnn_model((64, 512, 1))
    y = rng.standard_normal((2, 1)).astype("float32")
    for _ in range(2):
        a.train_on_batch(x, y)
    specs = a.net.params.specs
    w = a.get_weights_dict()
    ds = {path: w[n] for n, path in keras_h5.weight_paths(specs).items()}
    names, m, v = K._opt_keras(a.net.params, a._tr().opt)
    assert names == keras_h5.trainable_order(specs)
    ds.update(keras_h5.optimizer_datasets([(names, m, v)], a._tr().opt.t, 1e-3))
    _keras_style_archive(tmp_path / "cnn.keras", ds, (64, 512, 1),
                         empty=("layers/input_layer/vars", "layers/leaky_re_lu/vars", "layers/max_pooling2d/vars",
                                "layers/flatten/vars", "layers/dropout/vars"))
    b = K.load_model(str(tmp_path / "cnn.keras"))
    assert isinstance(b, K.CnnModel) and b.image_shape == (64, 512, 1)
    wb = b.get_weights_dict()
    assert all(np.array_equal(w[k], wb[k]) for k in w)
    assert b._tr().opt.t == 2
    b._seed = a._seed                                                   # same dropout draw in the next step
    la, lb = a.train_on_batch(x, y), b.train_on_batch(x, y)
    assert np.allclose(la, lb, rtol=1e-5, atol=1e-6), (la, lb)
    # same weights AND same Adam moments -> the same third update
    wa, wb = a.get_weights_dict(), b.get_weights_dict()
    assert all(np.allclose(wa[k], wb[k], rtol=1e-5, atol=1e-7) for k in wa)
    # ---- P generator (no optimizer), 128 x 1024 image: shape comes from Keras' config
    g = K.define_generator_P((128, 1024, 1))
    wg = g.get_weights_dict()
    _keras_style_archive(tmp_path / "g.keras", {p: wg[n] for n, p in keras_h5.weight_paths(g.net.params.specs).items()},
                         (128, 1024, 1))
    g2 = K.load_model(str(tmp_path / "g.keras"), compile=False)
    assert isinstance(g2, K.GeneratorModel) and g2.arch == "P" and g2.image_shape == (128, 1024, 1)
    wg2 = g2.get_weights_dict()
    assert all(np.array_equal(wg[k], wg2[k]) for k in wg)
    # ---- discriminator whose optimizer holds BatchNorm variables only: written by a Keras that read `trainable` at call time
    d = K.define_discriminator((64, 512, 1))
    sd = d.net.params.specs
    wd = d.get_weights_dict()
    ds = {p: wd[n] for n, p in keras_h5.weight_paths(sd).items()}
    bn = keras_h5.trainable_order(sd, convs=False)
    ds.update(keras_h5.optimizer_datasets([(bn, {n: 0.25 * np.ones(sd[n].keras_shape, "float32") for n in bn},
                                            {n: 0.5 * np.ones(sd[n].keras_shape, "float32") for n in bn})], 7, 2e-3))
    _keras_style_archive(tmp_path / "d.keras", ds, (64, 512, 1))
    d2 = K.load_model(str(tmp_path / "d.keras"))
    assert d2._d_freeze == "call" and all(np.array_equal(wd[k], d2.get_weights_dict()[k]) for k in wd)
    # ---- what save_model writes is that same member
    K.save_model(a, str(tmp_path / "ours.keras"))
    with zipfile.ZipFile(str(tmp_path / "ours.keras")) as z:
        assert "model.weights.h5" in z.namelist() and "model.weights.npz" not in z.namelist()
        z.extract("model.weights.h5", str(tmp_path))
    data, groups = h5.read(str(tmp_path / "model.weights.h5"))
    assert np.array_equal(data["layers/dense/vars/0"], wa["fc1.kernel"]) and data["layers/dense/vars/0"].shape == (8 * 64 * 512, 256)
    assert int(data["optimizer/vars/0"]) == 3 and "vars" in groups
    c = K.load_model(str(tmp_path / "ours.keras"))
    assert c._tr().opt.t == 3 and all(np.array_equal(wa[k], c.get_weights_dict()[k]) for k in wa)
    # the npz member of round 1 still loads
    monkeypatch.setenv("CFD_AMD_KERAS_NPZ", "1")
    K.save_model(a, str(tmp_path / "npz.keras"))
    e = K.load_model(str(tmp_path / "npz.keras"))
    assert e._tr().opt.t == 3 and all(np.array_equal(wa[k], e.get_weights_dict()[k]) for k in wa)
