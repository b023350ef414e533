"""CPU tests of the `.keras` HDF5 interchange (U/train.py:198-203, 214-216; U/valid.py:39):
the ctypes HDF5 binding against h5py where an interpreter with h5py exists in the image, and the Keras 3 weight-file
layout of the four reference graphs.  No GPU: models are not built here (tests/test_boundary_gpu.py loads archives)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from cfd_amd import h5, keras_h5

pytestmark = pytest.mark.skipif(not h5.available(), reason="no HDF5 C library in this image")

H5PY_PYTHON = next((p for p in ("/opt/conda/bin/python3.9", shutil.which("python3.9") or "") if p and os.path.exists(p)), None)


def _has_h5py() -> bool:
    if not H5PY_PYTHON:
        return False
    return subprocess.run([H5PY_PYTHON, "-c", "import h5py, numpy"], capture_output=True).returncode == 0


def test_h5_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    d = {"layers/conv2d/vars/0": rng.standard_normal((4, 4, 1, 8)).astype(np.float32),
         "layers/conv2d/vars/1": np.zeros(8, np.float32),
         "layers/dense_2/vars/0": rng.standard_normal((5, 3)).astype(np.float32).T,        # non-contiguous input
         "optimizer/vars/0": np.array(7, dtype=np.int64),                                   # 0-d stays 0-d
         "optimizer/vars/1": np.array(4e-4, dtype=np.float32),
         "x/f64": rng.standard_normal(3), "x/i32": np.arange(4, dtype=np.int32), "x/empty": np.zeros((0, 2), np.float32)}
    path = str(tmp_path / "t.h5")
    h5.write(path, d, empty_groups=("vars", "layers/leaky_re_lu/vars"))
    got, groups = h5.read(path)
    assert set(got) == set(d)
    for k, v in d.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    assert {"vars", "layers/leaky_re_lu/vars", "layers/conv2d/vars", "optimizer/vars"} <= set(groups)
    with pytest.raises(h5.H5Error):
        h5.read(str(tmp_path / "missing.h5"))
    with pytest.raises(h5.H5Error):
        h5.write(path, {"a": np.zeros(2, np.complex64)})


@pytest.mark.skipif(not _has_h5py(), reason="no interpreter with h5py in this image")
def test_h5_against_h5py(tmp_path):
    """the independent implementation: h5py reads what h5.write wrote, and h5.read reads what h5py wrote the way Keras'
    H5IOStore does (`group.create_group('vars')`, `vars['0'] = array`)"""
    rng = np.random.default_rng(1)
    ours, theirs = str(tmp_path / "ours.h5"), str(tmp_path / "theirs.h5")
    a = rng.standard_normal((4, 4, 2, 6)).astype(np.float32)
    h5.write(ours, {"layers/conv2d/vars/0": a, "optimizer/vars/0": np.array(12, np.int64)}, empty_groups=("vars",))
    np.save(str(tmp_path / "a.npy"), a)
    script = f"""
import h5py, numpy as np, json
a = np.load({str(tmp_path / 'a.npy')!r})
f = h5py.File({ours!r}, 'r')
ok = bool(np.array_equal(f['layers/conv2d/vars/0'][()], a)) and f['layers/conv2d/vars/0'].dtype == np.float32
ok = ok and int(f['optimizer/vars/0'][()]) == 12 and f['optimizer/vars/0'].shape == () and 'vars' in f
g = h5py.File({theirs!r}, 'w')
g.create_group('vars')
v = g.create_group('layers/batch_normalization_3').create_group('vars')
for i in range(4):
    v[str(i)] = (a.reshape(-1)[:32] * (i + 1)).astype(np.float32)
o = g.create_group('optimizer').create_group('vars')
o['0'] = np.int64(5); o['1'] = np.float32(2e-3); o['2'] = a.astype(np.float16)
g.close()
print(json.dumps({{"ok": ok}}))
"""
    r = subprocess.run([H5PY_PYTHON, "-c", script], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["ok"]
    got, groups = h5.read(theirs)
    for i in range(4):
        assert np.array_equal(got[f"layers/batch_normalization_3/vars/{i}"], (a.reshape(-1)[:32] * (i + 1)).astype(np.float32))
    assert got["optimizer/vars/0"].dtype == np.int64 and int(got["optimizer/vars/0"]) == 5
    assert got["optimizer/vars/1"] == np.float32(2e-3)
    assert np.array_equal(got["optimizer/vars/2"], a.astype(np.float16).astype(np.float32))      # f16 reads as f32
    assert "vars" in groups


def _specs(kind, arch=None):
    """the ParamStore specs of a net without building it on a GPU"""
    from cfd_amd import nets, params

    captured = {}

    def fake_init(self, specs, device, dtype, seed):
        self.params = params.ParamStore(specs, "cpu", torch.float32)
        captured["specs"] = self.params.specs

    saved = nets._Net.__init__
    nets._Net.__init__ = fake_init
    try:
        if kind == "generator":
            nets.Generator(arch, device="cpu", dtype=torch.float32)
        elif kind == "discriminator":
            nets.Discriminator(device="cpu", dtype=torch.float32)
        else:
            nets.CnnRegressor(device="cpu", dtype=torch.float32)
    finally:
        nets._Net.__init__ = saved
    return captured["specs"]


def test_keras3_weight_paths_of_the_reference_graphs():
    """`layers/<snake class>[_k]/vars/<i>` in model.layers order (U/models/generator.py:44-74, discriminator.py:12-49,
    DeltaP/models/generator.py:15-55)"""
    g = keras_h5.weight_paths(_specs("generator", "U"))
    assert g["e1.kernel"] == "layers/conv2d/vars/0" and g["e1.bias"] == "layers/conv2d/vars/1"
    assert g["e2.kernel"] == "layers/conv2d_1/vars/0"
    assert g["e2.bn.gamma"] == "layers/batch_normalization/vars/0"           # e1 has no BatchNorm (generator.py:47)
    assert g["e5.bn.moving_var"] == "layers/batch_normalization_3/vars/3"
    assert g["b.kernel"] == "layers/conv2d_5/vars/0"
    assert g["d1.kernel"] == "layers/conv2d_transpose/vars/0" and g["d1.bn.beta"] == "layers/batch_normalization_4/vars/1"
    assert g["d5.bn.moving_mean"] == "layers/batch_normalization_8/vars/2"
    assert g["out.kernel"] == "layers/conv2d_transpose_5/vars/0" and g["out.bias"] == "layers/conv2d_transpose_5/vars/1"
    assert len(set(g.values())) == len(g)
    d = keras_h5.weight_paths(_specs("discriminator"), "layers/functional_1/")
    assert d["c1.kernel"] == "layers/functional_1/layers/conv2d/vars/0"
    assert d["c2.bn.gamma"] == "layers/functional_1/layers/batch_normalization/vars/0"
    assert d["c6.bias"] == "layers/functional_1/layers/conv2d_5/vars/1"
    c = keras_h5.weight_paths(_specs("cnn"))
    assert c["g1.kernel"] == "layers/conv2d/vars/0" and c["g6.kernel"] == "layers/conv2d_3/vars/0"
    assert c["fc1.kernel"] == "layers/dense/vars/0" and c["fc3.bias"] == "layers/dense_2/vars/1"


def test_trainable_order_and_optimizer_vars():
    specs = _specs("discriminator")
    all_ = keras_h5.trainable_order(specs)
    assert all_[:6] == ["c1.kernel", "c1.bias", "c2.kernel", "c2.bias", "c2.bn.gamma", "c2.bn.beta"]
    assert all_[-2:] == ["c6.kernel", "c6.bias"] and len(all_) == 6 * 2 + 4 * 2
    bn = keras_h5.trainable_order(specs, convs=False)
    assert bn == [f"c{i}.bn.{v}" for i in range(2, 6) for v in ("gamma", "beta")]
    m = {n: np.full(specs[n].keras_shape, i, np.float32) for i, n in enumerate(bn)}
    v = {n: np.full(specs[n].keras_shape, 100 + i, np.float32) for i, n in enumerate(bn)}
    ds = keras_h5.optimizer_datasets([(bn, m, v)], t=9, lr=2e-3)
    assert ds["optimizer/vars/0"].dtype == np.int64 and int(ds["optimizer/vars/0"]) == 9
    assert ds["optimizer/vars/2"][0] == 0 and ds["optimizer/vars/3"][0] == 100 and ds["optimizer/vars/4"][0] == 1
    vars_ = keras_h5.optimizer_vars(ds)
    assert len(vars_) == 2 + 2 * len(bn)
    t, m2, v2 = keras_h5.split_optimizer(vars_, bn)
    assert t == 9 and all(np.array_equal(m2[n], m[n]) and np.array_equal(v2[n], v[n]) for n in bn)
    assert keras_h5.split_optimizer(vars_, all_) is None                    # count mismatch: not this variable list


def test_classify_and_config_shape():
    firsts = {256: "U", 128: "P"}
    z = lambda *s: np.zeros(s, np.float32)
    assert keras_h5.classify({"layers/conv2d/vars/0": z(4, 4, 1, 256)}, firsts) == ("generator", "U", "")
    assert keras_h5.classify({"layers/conv2d/vars/0": z(4, 4, 1, 128)}, firsts) == ("generator", "P", "")
    assert keras_h5.classify({"layers/conv2d/vars/0": z(4, 4, 2, 128)}, firsts)[0] == "discriminator"
    assert keras_h5.classify({"layers/conv2d/vars/0": z(3, 3, 1, 64)}, firsts)[0] == "cnn"
    assert keras_h5.classify({"layers/functional/layers/conv2d/vars/0": z(4, 4, 1, 128),
                              "layers/functional_1/layers/conv2d/vars/0": z(4, 4, 2, 128)}, firsts) == \
        ("gan", "P", "layers/functional/")
    with pytest.raises(ValueError):
        keras_h5.classify({"layers/dense/vars/0": z(3, 3)}, firsts)
    cfg = {"module": "keras", "class_name": "Functional",
           "config": {"name": "functional", "layers": [
               {"module": "keras.layers", "class_name": "InputLayer",
                "config": {"batch_shape": [None, 128, 1024, 1], "dtype": "float32", "name": "input_layer"}},
               {"module": "keras.layers", "class_name": "Conv2D", "config": {"filters": 64, "trainable": True}}]}}
    assert keras_h5.input_shape_from_config(cfg) == (128, 1024, 1)
    cfg2 = {"class_name": "Functional", "config": {"layers": [{"class_name": "InputLayer",
                                                                "config": {"batch_input_shape": [None, 64, 512, 1]}}]}}
    assert keras_h5.input_shape_from_config(cfg2) == (64, 512, 1)
    assert keras_h5.input_shape_from_config({"format": "x"}) is None
