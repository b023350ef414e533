"""Keras 3 `model.weights.h5` layout for the four reference graphs (the HDF5 member of a `.keras` archive written by
`save_model(...)`, U/train.py:198-203, and read by `load_model(...)`, U/train.py:214-216, U/valid.py:39).

Restated from Keras 3's published saving library (`keras/src/saving/saving_lib.py`, `_save_state` /
`_save_container_state`, H5IOStore), which TensorFlow 2.16-2.18 -- the versions the reference's comments name -- ship:

  * a model's layers are stored under `layers/<snake_case(class name)>[_<k>]/vars/<i>`, `<k>` counting earlier layers of
    the SAME class in `model.layers` order (first one has no suffix), `<i>` the position in the layer's
    `trainable_variables + non_trainable_variables`: Conv2D / Conv2DTranspose / Dense = [kernel, bias],
    BatchNormalization = [gamma, beta, moving_mean, moving_variance];
  * a Functional model nested as a layer (the composite GAN: [InputLayer, generator, discriminator]) is `functional`,
    `functional_1`, with its own `layers/...` below;
  * a compiled model's optimizer is `optimizer/vars/<i>` = [iterations, learning_rate, m_0, v_0, m_1, v_1, ...] over
    `model.trainable_variables` (layer order; kernel, bias / gamma, beta per layer) -- Adam.build creates momentum and
    velocity per variable in turn;
  * layers without variables get an empty `vars` group, and so does the model itself (`vars` at the root).

Every weighted layer of the four graphs lies on ONE chain (each consumes the previous one's output), so `model.layers`
order restricted to a class is the chain order -- the order of the ParamStore's kernels.

Parity unpinned: neither Keras nor a Keras-written archive exists in this image or in the reference; the HDF5 container
itself is cross-checked against h5py (tests/test_keras_h5_cpu.py).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np

_CLS = {"conv": "conv2d", "convT": "conv2d_transpose", "dense": "dense"}
_BN_VARS = ("gamma", "beta", "moving_mean", "moving_var")


def _layer_bases(specs) -> List[str]:
    """layer prefixes in graph order (ParamStore.specs iterates in declaration order, whatever the offsets)"""
    return [n[:-len(".kernel")] for n in specs if n.endswith(".kernel")]


def weight_paths(specs, prefix: str = "") -> "OrderedDict[str, str]":
    """spec name -> HDF5 dataset path of that variable in a Keras 3 weights file"""
    out: "OrderedDict[str, str]" = OrderedDict()
    count: Dict[str, int] = {}

    def layer_dir(cls: str) -> str:
        k = count.get(cls, -1) + 1
        count[cls] = k
        return f"{prefix}layers/{cls}" + (f"_{k}" if k else "")

    for base in _layer_bases(specs):
        d = layer_dir(_CLS[specs[base + ".kernel"].kind])
        out[base + ".kernel"] = d + "/vars/0"
        out[base + ".bias"] = d + "/vars/1"
        if base + ".bn.gamma" in specs:
            d = layer_dir("batch_normalization")
            for i, v in enumerate(_BN_VARS):
                out[f"{base}.bn.{v}"] = f"{d}/vars/{i}"
    missing = set(specs) - set(out)
    if missing:
        raise KeyError(f"no Keras path for {sorted(missing)[:4]}")
    return out


def trainable_order(specs, convs: bool = True) -> List[str]:
    """`model.trainable_variables` order: per layer kernel, bias, then its BatchNorm's gamma, beta.
    convs=False: the convolution kernels / biases are frozen (D inside the composite; D's own step in 'call' mode)."""
    out: List[str] = []
    for base in _layer_bases(specs):
        if convs:
            out += [base + ".kernel", base + ".bias"]
        if base + ".bn.gamma" in specs:
            out += [base + ".bn.gamma", base + ".bn.beta"]
    return out


def optimizer_datasets(entries, t: int, lr: float) -> Dict[str, np.ndarray]:
    """entries: [(names, m_by_name, v_by_name)] in `trainable_variables` order (Keras layouts) -> optimizer/vars/*"""
    out = {"optimizer/vars/0": np.array(int(t), dtype=np.int64), "optimizer/vars/1": np.array(lr, dtype=np.float32)}
    i = 2
    for names, m, v in entries:
        for n in names:
            out[f"optimizer/vars/{i}"] = np.asarray(m[n], dtype=np.float32)
            out[f"optimizer/vars/{i + 1}"] = np.asarray(v[n], dtype=np.float32)
            i += 2
    return out


def optimizer_vars(data: Dict[str, np.ndarray]) -> List[np.ndarray]:
    out = []
    while f"optimizer/vars/{len(out)}" in data:
        out.append(data[f"optimizer/vars/{len(out)}"])
    return out


def split_optimizer(vars_: List[np.ndarray], names: List[str]) -> Optional[Tuple[int, Dict[str, np.ndarray], Dict[str, np.ndarray]]]:
    """[iterations, learning_rate, m_0, v_0, ...] against `names` -> (t, m, v), or None when the count does not fit"""
    if len(vars_) != 2 + 2 * len(names):
        return None
    m = {n: vars_[2 + 2 * i] for i, n in enumerate(names)}
    v = {n: vars_[3 + 2 * i] for i, n in enumerate(names)}
    return int(np.asarray(vars_[0]).reshape(-1)[0]), m, v


def pick(data: Dict[str, np.ndarray], paths: "OrderedDict[str, str]", specs) -> Dict[str, np.ndarray]:
    """spec name -> array (Keras layout) out of a weights file, shapes checked"""
    out = {}
    for name, path in paths.items():
        if path not in data:
            raise KeyError(f"{path} ({name}) is not in the weights file")
        a = data[path]
        if tuple(a.shape) != tuple(specs[name].keras_shape):
            raise ValueError(f"{path} ({name}): shape {tuple(a.shape)}, expected {tuple(specs[name].keras_shape)}")
        out[name] = a
    return out


def classify(data: Dict[str, np.ndarray], gen_first_filters: Dict[int, str]) -> Tuple[str, Optional[str], str]:
    """-> (kind, arch, prefix of the first model) from the shapes in a weights file.
    gen_first_filters: {filters of the first encoder convolution: arch name}"""
    for prefix, nested in (("layers/functional/", True), ("", False)):
        k = data.get(prefix + "layers/conv2d/vars/0")
        if k is None:
            continue
        kh, kw, ci, co = k.shape
        if nested:
            if (kh, ci) == (4, 1) and co in gen_first_filters:
                return "gan", gen_first_filters[co], prefix
            break
        if (kh, ci) == (4, 1) and co in gen_first_filters:
            return "generator", gen_first_filters[co], prefix
        if (kh, ci) == (4, 2):
            return "discriminator", None, prefix
        if (kh, ci) == (3, 1):
            return "cnn", None, prefix
        break
    raise ValueError("the weights file matches none of the four reference graphs (U / P generator, discriminator, "
                     "DeltaP CNN, composite GAN)")


def input_shape_from_config(cfg) -> Optional[Tuple[int, ...]]:
    """first InputLayer's shape in a Keras model config (Keras 3 `batch_shape`, Keras 2 `batch_input_shape`)"""
    if isinstance(cfg, dict):
        if cfg.get("class_name") == "InputLayer":
            c = cfg.get("config", {})
            shp = c.get("batch_shape") or c.get("batch_input_shape")
            if shp:
                return tuple(int(s) for s in shp[1:])
        for v in cfg.values():
            r = input_shape_from_config(v)
            if r:
                return r
    elif isinstance(cfg, list):
        for v in cfg:
            r = input_shape_from_config(v)
            if r:
                return r
    return None
