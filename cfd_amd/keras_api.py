"""Host-side mirror of the reference's model boundary.

The reference's entry scripts touch their Keras models through a handful of names only:

  constructors  define_generator(image_shape)            U/models/generator.py:44, P/models/generator.py:69
                define_discriminator(image_shape)        U/models/discriminator.py:12
                define_gan(g_model, d_model, image_shape) U/models/generator.py:77, P/models/generator.py:117
                define_cnn_model(image_shape)            DeltaP/models/generator.py:15
  methods       .predict(x) .train_on_batch(x, y) .test_on_batch(x, y) .output_shape .layers .summary()
                (U/train.py:41,121-125,148; utils.py:56; U/valid.py:29)
  persistence   save_model(model, path, overwrite=True, include_optimizer=True), load_model(path, compile=...)
                (U/train.py:156,199-203,214-216; U/valid.py:39): `.keras` zip archives with Keras 3's
                `model.weights.h5` member (cfd_amd/h5.py, cfd_amd/keras_h5.py)
  losses        MC_mse, MC_mae, MC_huber_loss            U/models/loss.py

Same names, same argument meaning, numpy NHWC fp32 in / numpy out, Python exceptions on error.
Everything below the method calls runs on the HIP kernels (cfd_amd.nets / cfd_amd.trainer).

Environment switches:
  CFD_AMD_DTYPE = fp32 | bf16   arithmetic type of the MFMA kernels.  Default fp32: the reference's own arithmetic (Keras
                                float32; exact-f32 MFMA, within 1e-4 of the oracle) -- a drop-in gives the reference's
                                numbers unless asked otherwise.  bf16 (north_star's target arithmetic: bf16 operands, fp32
                                accumulation, outputs within 2e-2 of the fp32 oracle) is ~2.7x faster at batch 128.
"""
from __future__ import annotations

import io
import json
import os
import zipfile
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import h5, keras_h5, nets, ops, trainer


def _dtype() -> torch.dtype:
    return {"bf16": torch.bfloat16, "fp32": torch.float32, "f32": torch.float32}[
        os.environ.get("CFD_AMD_DTYPE", "fp32").lower()]


GRAPH_MAX_BATCH = 8     # largest batch whose predict is replayed as one captured HIP graph (CFD_AMD_GRAPHS=0 turns it off)


class _GraphedForward:
    """One captured HIP graph of `net.forward` for a fixed batch size: the per-case predict loop of the reference
    (utils.py:55-58, U/valid.py:25-35) issues ~80 kernels per case, and at batch 1 the host is the slower side: U
    0.64 ms eager, 0.48 ms replayed (P 0.59 -> 0.39).  The graph starts with the dropout-epoch kernel so that every
    replay draws fresh dropout.  Weights are read in place (loading and training update the same buffers), and the
    graph keeps the shared workspace it was captured with alive."""

    def __init__(self, net, x: torch.Tensor, **fwd):
        self.x = x.clone()
        net.forward(self.x, **fwd)                     # warm-up: buffers, function attributes, workspace
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.keep = list(ops._ws_cache.values())       # the grow-only workspace may be replaced later: hold this one
        with torch.cuda.graph(self.graph):
            ops.dropout_epoch(1)
            self.y = net.forward(self.x, **fwd)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        self.x.copy_(x)
        self.graph.replay()
        return self.y


PREDICT_CHUNK = 128    # cases per forward pass of a large predict (CFD_AMD_PREDICT_CHUNK): the bench's batch


def _predict_chunk() -> int:
    return max(1, int(os.environ.get("CFD_AMD_PREDICT_CHUNK", PREDICT_CHUNK)))


def _forward_chunked(forward, xd: torch.Tensor, with_sample0: bool, **kw) -> torch.Tensor:
    """`forward` over xd in chunks of PREDICT_CHUNK cases, results gathered in one device tensor.  Keras' predict
    batches the same way (batch_size=32).  Every case is independent of its batch here (per-sample BatchNorm statistics,
    dropout keyed by the sample index `sample0 + i`), so a chunked call computes the same function of each case as
    one pass -- equal WITHIN THE STATED TOLERANCE (1e-4 fp32 / 2e-2 bf16), not bit for bit: which BatchNorm-statistics
    path a layer takes (sums fused into the conv epilogue through fp32 atomics, variance as E[x^2] - mean^2; or the
    shifted sums of `bn_act`) and the split-K factor depend on the batch size, so the fp32 summation order differs
    between a batch-1 predict, a full chunk and a ragged tail.  Chunking bounds the activation
    workspace (U at fp32: ~45 MB a case) and keeps every kernel's element count far below 2^31 however many cases
    (configs[2]: 1250 per GPU) one call hands over."""
    B, C = xd.shape[0], _predict_chunk()
    if B <= C:
        return forward(xd, **kw)
    out = None
    for lo in range(0, B, C):
        if with_sample0:
            kw["sample0"] = lo
        y = forward(xd[lo:lo + C], **kw)           # a slice along the batch axis is contiguous
        if out is None:
            out = torch.empty((B,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
        out[lo:lo + C].copy_(y)                     # y lives in the net's reused buffers
    return out


def _use_graphs() -> bool:
    return os.environ.get("CFD_AMD_GRAPHS", "1") != "0"


def _dev(a) -> torch.Tensor:
    """numpy / list-of-one-numpy (Keras accepts both, utils.py:56) -> fp32 device tensor"""
    if isinstance(a, (list, tuple)):
        if len(a) != 1:
            raise ValueError(f"expected one input array, got {len(a)}")
        a = a[0]
    if isinstance(a, torch.Tensor):
        return a.to("cuda", torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


class Layer:
    """what `define_gan` iterates over: `for layer in d_model.layers: if not isinstance(layer, BatchNormalization):
    layer.trainable = False` (U/models/generator.py:79-81)"""

    def __init__(self, name: str):
        self.name, self.trainable = name, True


class BatchNormalization(Layer):
    pass


class Conv2D(Layer):
    pass


class Conv2DTranspose(Layer):
    pass


class Dense(Layer):
    pass


class Model:
    kind = "model"

    def __init__(self, net, image_shape):
        self.net = net
        self.image_shape = tuple(image_shape)
        self.layers: List[Layer] = []
        self._seed = int.from_bytes(os.urandom(4), "little")   # dropout is random run to run, as in Keras
        self._graphs = {}                                      # batch size -> _GraphedForward (small-batch predict)
        self.compiled = True

    # -- Keras surface ---------------------------------------------------------------------
    def count_params(self) -> int:
        return sum(s.numel for s in self.net.params.specs.values())

    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.kind}"  input {(None,) + self.image_shape}  output {self.output_shape}')
        for name, s in self.net.params.specs.items():
            print_fn(f"  {name:28s} {str(s.keras_shape):24s} {s.numel:>12,d}")
        print_fn(f"Total params: {self.count_params():,d}")

    def sync_state(self) -> None:
        """Under data parallelism the sharded optimizer step (trainer._DP) leaves the fp32 master weights and Adam
        moments of the chunks other ranks own stale: bring them up to date.  COLLECTIVE -- every rank calls it (every
        rank runs the same script, so every rank saves).  A no-op for one process."""
        for tr in (getattr(self, "_trainer", None), getattr(self, "_gan_trainer", None)):
            if tr is not None:
                tr.sync_state()

    def get_weights_dict(self):
        self.sync_state()
        return self.net.params.to_keras()

    def set_weights_dict(self, w):
        self.net.params.load_keras(w)
        self.net.params.refresh_transposed()

    def _next_seed(self) -> int:
        self._seed = (self._seed * 6364136223846793005 + 1442695040888963407) % (1 << 62)
        return self._seed >> 8


class GeneratorModel(Model):
    kind = "generator"

    def __init__(self, arch: str, image_shape):
        super().__init__(nets.Generator(arch, image_shape, dtype=_dtype(), seed=int.from_bytes(os.urandom(4), "little")),
                         image_shape)
        self.arch = arch
        self.layers = [Layer(n) for n in self.net.params.specs if n.endswith(".kernel")]

    @property
    def output_shape(self):
        return (None,) + self.image_shape

    def predict(self, x, verbose=0, batch_size=None):
        """BatchNorm and Dropout run in training mode here too (they are built with training=True,
        generator.py:21,33,36): per-sample statistics, fresh dropout draw per call."""
        xd = _dev(x)
        B = xd.shape[0]
        if B == 0:
            return np.zeros((0,) + tuple(self.image_shape), dtype=np.float32)
        if B <= GRAPH_MAX_BATCH and _use_graphs():
            g = self._graphs.get(B)
            if g is None:
                # bit 59 of the seed -> bit 63 of every layer's drop_seed (seed * 16 + layer): epoch-keyed
                g = self._graphs[B] = _GraphedForward(self.net, xd, inference=True,
                                                      seed=(1 << 59) | (self._next_seed() & ((1 << 59) - 1)))
            return g(xd).cpu().numpy()
        return _forward_chunked(self.net.forward, xd, True, seed=self._next_seed(), inference=True).cpu().numpy()

    def predict_on_device(self, x) -> torch.Tensor:
        """predict without the copy back to the host (not in Keras): the evaluation drivers feed the result to
        ops.eval_metrics and fetch a few numbers per case instead of the fields"""
        return _forward_chunked(self.net.forward, _dev(x), True, seed=self._next_seed(), inference=True)

    __call__ = predict


class DiscriminatorModel(Model):
    kind = "discriminator"

    def __init__(self, image_shape):
        super().__init__(nets.Discriminator(image_shape, dtype=_dtype(), seed=int.from_bytes(os.urandom(4), "little")),
                         image_shape)
        self.layers = []
        for i in range(1, 7):
            self.layers.append(Conv2D(f"c{i}"))
            if 2 <= i <= 5:
                self.layers.append(BatchNormalization(f"c{i}.bn"))
        self._trainer: Optional[trainer.DiscriminatorTrainer] = None

    @property
    def output_shape(self):
        return (None,) + self.net.patch_shape

    def _tr(self):
        if self._trainer is None:
            self._trainer = trainer.DiscriminatorTrainer(self.net, lr=2e-3, beta_1=0.5, train_convs=self.d_step_trains_convs())
        return self._trainer

    def d_step_trains_convs(self) -> bool:
        """Does d_model.train_on_batch update D's convolutions?  define_discriminator compiles D BEFORE define_gan
        sets `layer.trainable = False` on its conv layers (U/models/generator.py:79-81).
          CFD_AMD_D_FREEZE=compile (default): trainability is what it was at compile() -- the Keras-2 snapshot and the
              pix2pix intent: D trains fully in its own step, frozen inside the composite.
          CFD_AMD_D_FREEZE=call: `layer.trainable` is read when the first train_on_batch builds the step -- the
              Keras-3 reading of the same script (the reference's comments name tensorflow 2.16-2.18): D's own step then
              updates only its BatchNorm gamma/beta once define_gan has run.
        Which of the two real Keras does cannot be checked here (no TensorFlow: parity unpinned); the
        choice is stored in the checkpoint's config.json."""
        mode = getattr(self, "_d_freeze", None) or os.environ.get("CFD_AMD_D_FREEZE", "compile").lower()
        if mode not in ("compile", "call"):
            raise ValueError("CFD_AMD_D_FREEZE must be 'compile' or 'call'")
        self._d_freeze = mode
        if mode == "compile":
            return True
        return all(l.trainable for l in self.layers if isinstance(l, Conv2D))

    @staticmethod
    def _label(y) -> float:
        y = np.asarray(y)
        v = float(y.flat[0])
        if not np.all(y == v):
            raise NotImplementedError("patch labels must be constant (the reference uses all-ones / all-zeros)")
        return v

    def train_on_batch(self, x, y):
        a, b = x
        return self._tr().train_on_batch(_dev(a), _dev(b), self._label(y))

    def predict(self, x, verbose=0):
        a, b = x
        return self.net.forward(_dev(a), _dev(b), training=False).cpu().numpy()


class GanModel(Model):
    kind = "gan"

    def __init__(self, g_model: GeneratorModel, d_model: DiscriminatorModel, image_shape):
        self.g_model, self.d_model = g_model, d_model
        self.image_shape = tuple(image_shape)
        self.layers = [g_model, d_model]
        self.net = g_model.net
        self._seed = g_model._seed
        self.compiled = True
        self._trainer = trainer.GanTrainer(g_model.net, d_model.net, lr=4e-4, beta_1=0.5, field_loss=ops.MAE,
                                           field_weight=100.0)
        g_model._gan_trainer = d_model._gan_trainer = self._trainer      # who else updates their weights (sync_state)

    @property
    def output_shape(self):
        return [self.d_model.output_shape, self.g_model.output_shape]

    def count_params(self):
        return self.g_model.count_params() + self.d_model.count_params()

    def train_on_batch(self, x, y):
        """x = [A]; y = [patch labels, B] -> [total, bce, mae] (U/train.py:125)"""
        y_patch, y_field = y
        return self._trainer.train_on_batch(_dev(x), DiscriminatorModel._label(y_patch), _dev(y_field),
                                            seed=self.g_model._next_seed())

    def test_on_batch(self, x, y):
        y_patch, y_field = y
        return self._trainer.test_on_batch(_dev(x), DiscriminatorModel._label(y_patch), _dev(y_field),
                                           seed=self.g_model._next_seed())

    def predict(self, x, verbose=0):
        fake = self.g_model.predict(x)
        return [self.d_model.predict([x[0] if isinstance(x, (list, tuple)) else x, fake]), fake]


class CnnModel(Model):
    kind = "cnn"

    def __init__(self, image_shape):
        super().__init__(nets.CnnRegressor(image_shape, dtype=_dtype(), seed=int.from_bytes(os.urandom(4), "little")),
                         image_shape)
        self.layers = [Layer(n) for n in self.net.params.specs if n.endswith(".kernel")]
        self._trainer: Optional[trainer.CnnTrainer] = None

    @property
    def output_shape(self):
        return (None, 1)

    def _tr(self):
        if self._trainer is None:
            self._trainer = trainer.CnnTrainer(self.net, lr=1e-3, beta_1=0.5, weight=100.0, delta=1.0)
        return self._trainer

    def predict(self, x, verbose=0):
        xd = _dev(x)
        B = xd.shape[0]
        if B == 0:
            return np.zeros((0, 1), dtype=np.float32)
        if B <= GRAPH_MAX_BATCH and _use_graphs():
            g = self._graphs.get(B)
            if g is None:
                g = self._graphs[B] = _GraphedForward(self.net, xd, training=False)
            return g(xd).cpu().numpy()
        return _forward_chunked(self.net.forward, xd, False, training=False).cpu().numpy()

    def train_on_batch(self, x, y):
        return self._tr().train_on_batch(_dev(x), _dev(y).reshape(-1, 1), seed=self._next_seed())

    def test_on_batch(self, x, y):
        return self._tr().test_on_batch(_dev(x), _dev(y).reshape(-1, 1))


# ---- constructors with the reference's signatures -------------------------------------------
def define_generator_U(image_shape=(64, 512, 1)):
    return GeneratorModel("U", image_shape)


def define_generator_P(image_shape=(64, 512, 1)):
    return GeneratorModel("P", image_shape)


def define_discriminator(image_shape=(64, 512, 1)):
    return DiscriminatorModel(image_shape)


def define_gan(g_model, d_model, image_shape):
    for layer in d_model.layers:                       # the reference's loop, kept for its side effect on .trainable
        if not isinstance(layer, BatchNormalization):
            layer.trainable = False
    return GanModel(g_model, d_model, image_shape)


def define_cnn_model(image_shape=(64, 512, 1)):
    return CnnModel(image_shape)


# ---- losses exported by `from models import *` (U/models/loss.py) -----------------------------
def _loss(kind, y_true, y_pred, delta=1.0) -> float:
    yt, yp = _dev(y_true).reshape(-1), _dev(y_pred).reshape(-1)
    acc = torch.zeros(1, dtype=torch.float32, device="cuda")
    ops.field_loss(yp, yt, kind, acc, None, delta=delta)
    return float(acc.item())


def MC_mse(y_true, y_pred):
    """MSE; the reference's mass-conservation term is computed from y_true on both sides and is 0 (loss.py:13-37)"""
    return _loss(ops.MSE, y_true, y_pred)


def MC_mae(y_true, y_pred):
    return _loss(ops.MAE, y_true, y_pred)


def MC_huber_loss(y_true, y_pred, delta=1.0):
    return _loss(ops.HUBER, y_true, y_pred, delta)


# ---- persistence ------------------------------------------------------------------------------
FORMAT = "cfd_amd.keras-archive.v1"


def _npz_bytes(arrays: dict) -> bytes:
    bio = io.BytesIO()
    np.savez(bio, **arrays)
    return bio.getvalue()


def _opt_state(opt: Optional[trainer.Adam], prefix: str) -> dict:
    if opt is None:
        return {}
    return {prefix + "m": opt.m.cpu().numpy(), prefix + "v": opt.v.cpu().numpy(),
            prefix + "t": np.array([opt.t], dtype=np.int64)}


def _load_opt(opt: trainer.Adam, z, prefix: str) -> None:
    if prefix + "m" in z:
        opt.m.copy_(torch.from_numpy(z[prefix + "m"]))
        opt.v.copy_(torch.from_numpy(z[prefix + "v"]))
        opt.t = int(z[prefix + "t"][0])


def _use_h5() -> bool:
    """write `model.weights.h5` (Keras 3's member) when an HDF5 C library is present; CFD_AMD_KERAS_NPZ=1 keeps the
    round-1 npz member instead"""
    return os.environ.get("CFD_AMD_KERAS_NPZ", "0") != "1" and h5.available()


def _opt_keras(params, opt: Optional[trainer.Adam]):
    """(names, m, v) of one Adam range in `trainable_variables` order and Keras layouts"""
    if opt is None:
        return None
    names = [n for n in keras_h5.trainable_order(params.specs) if opt.lo <= params.specs[n].offset < opt.hi]
    return names, params.to_keras(opt.m, opt.lo, names), params.to_keras(opt.v, opt.lo, names)


def _opt_from_keras(params, opt: trainer.Adam, t: int, m: dict, v: dict) -> None:
    params.pack_keras(m, opt.m, opt.lo)
    params.pack_keras(v, opt.v, opt.lo)
    opt.t = int(t)


def save_model(model, filepath, overwrite=True, include_optimizer=True):
    """Same call as keras.models.save_model; same file names (`*.keras`, a zip archive as in Keras 3).
    Members: config.json, metadata.json and -- when the HDF5 C library is available (cfd_amd/h5.py) -- `model.weights.h5`
    in Keras 3's own layout (cfd_amd/keras_h5.py: `layers/conv2d_3/vars/0`, ..., `optimizer/vars/<i>`; weights in KERAS
    layouts: Conv2D (kh,kw,Cin,Cout), Conv2DTranspose (kh,kw,Cout,Cin), Dense (in,out)), so a Keras user can
    `define_generator(...).load_weights(<that member>)`.  config.json is this library's own (the Keras model config is
    not reproduced), so keras.models.load_model on these archives is not supported; load_model below reads both these
    and archives Keras wrote.  Without HDF5 (or with CFD_AMD_KERAS_NPZ=1) the weights go to `model.weights.npz` +
    `optimizer.npz` as in round 1."""
    filepath = str(filepath)
    if os.path.exists(filepath) and not overwrite:
        raise FileExistsError(filepath)
    os.makedirs(os.path.dirname(os.path.abspath(filepath)), exist_ok=True)
    cfg = {"format": FORMAT, "class_name": model.kind, "image_shape": list(model.image_shape),
           "compute_dtype": str(_dtype()).replace("torch.", "")}
    weights, opt = {}, {}
    use_h5 = _use_h5()
    h5sets, h5empty = {}, ["vars"]
    model.sync_state()
    if isinstance(model, DiscriminatorModel):
        cfg["d_freeze"] = getattr(model, "_d_freeze", None) or os.environ.get("CFD_AMD_D_FREEZE", "compile").lower()
        cfg["d_convs_trainable"] = all(l.trainable for l in model.layers if isinstance(l, Conv2D))
    if isinstance(model, GanModel):
        model.g_model.sync_state()
        model.d_model.sync_state()
        cfg["arch"] = model.g_model.arch
        tr = model._trainer
        if use_h5:
            for sub, pre in ((model.g_model, "layers/functional/"), (model.d_model, "layers/functional_1/")):
                w = sub.get_weights_dict()
                for n, path in keras_h5.weight_paths(sub.net.params.specs, pre).items():
                    h5sets[path] = w[n]
                h5empty.append(pre + "vars")
            if include_optimizer:
                h5sets.update(keras_h5.optimizer_datasets(
                    [_opt_keras(model.g_model.net.params, tr.opt_g), _opt_keras(model.d_model.net.params, tr.opt_dbn)],
                    tr.opt_g.t, tr.opt_g.lr))
        else:
            weights.update({"g/" + k: v for k, v in model.g_model.get_weights_dict().items()})
            weights.update({"d/" + k: v for k, v in model.d_model.get_weights_dict().items()})
            if include_optimizer:
                opt.update(_opt_state(tr.opt_g, "gan_g/"))
                opt.update(_opt_state(tr.opt_dbn, "gan_dbn/"))
    else:
        if isinstance(model, GeneratorModel):
            cfg["arch"] = model.arch
        tr = getattr(model, "_trainer", None)
        if use_h5:
            w = model.get_weights_dict()
            for n, path in keras_h5.weight_paths(model.net.params.specs).items():
                h5sets[path] = w[n]
            if include_optimizer and tr is not None:
                h5sets.update(keras_h5.optimizer_datasets([_opt_keras(model.net.params, tr.opt)], tr.opt.t, tr.opt.lr))
        else:
            weights.update(model.get_weights_dict())
            if include_optimizer and tr is not None:
                opt.update(_opt_state(tr.opt, "opt/"))
    tmp = filepath + ".tmp"
    tmp_h5 = filepath + ".weights.tmp.h5"
    try:
        with zipfile.ZipFile(tmp, "w", zipfile.ZIP_STORED) as z:
            z.writestr("config.json", json.dumps(cfg))
            z.writestr("metadata.json", json.dumps({"writer": "cfd_amd", "format": FORMAT}))
            if use_h5:
                h5.write(tmp_h5, h5sets, empty_groups=h5empty)
                # (ZipFile.write copies in 8 KiB pieces: 150 000 Python-level read / crc32 / write rounds for the 1.2 GB
                #  of a GAN archive; 32 MiB pieces take a third of the time)
                with z.open("model.weights.h5", "w", force_zip64=True) as dst, open(tmp_h5, "rb") as src:
                    while True:
                        chunk = src.read(1 << 25)
                        if not chunk:
                            break
                        dst.write(chunk)
            else:
                z.writestr("model.weights.npz", _npz_bytes(weights))
                if opt:
                    z.writestr("optimizer.npz", _npz_bytes(opt))
    finally:
        if os.path.exists(tmp_h5):
            os.remove(tmp_h5)
    os.replace(tmp, filepath)


def _keras_conv_flags(cfg) -> List[bool]:
    """`trainable` of every Conv2D layer in a Keras model config"""
    out: List[bool] = []
    if isinstance(cfg, dict):
        if cfg.get("class_name") == "Conv2D" and isinstance(cfg.get("config"), dict):
            out.append(bool(cfg["config"].get("trainable", True)))
        for v in cfg.values():
            out += _keras_conv_flags(v)
    elif isinstance(cfg, list):
        for v in cfg:
            out += _keras_conv_flags(v)
    return out


def _load_h5_archive(z: zipfile.ZipFile, cfg: dict, ours: bool, compile: bool, filepath: str):
    """a `.keras` archive whose weights are Keras 3's `model.weights.h5`: ours, or one Keras itself wrote"""
    if not h5.available():
        raise NotImplementedError("this archive stores its weights as HDF5 (`model.weights.h5`) and no HDF5 C library "
                                  "was found: set CFD_AMD_HDF5 to a libhdf5.so (cfd_amd/h5.py)")
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".h5", dir=os.path.dirname(os.path.abspath(filepath))) as tf:
        with z.open("model.weights.h5") as src:
            while True:
                blk = src.read(1 << 24)
                if not blk:
                    break
                tf.write(blk)
        tf.flush()
        data, _groups = h5.read(tf.name)
    firsts = {a["enc"][0]: name for name, a in nets.GEN_ARCH.items()}
    if ours:
        kind, arch, shape = cfg["class_name"], cfg.get("arch"), tuple(cfg["image_shape"])
    else:
        kind, arch, _ = keras_h5.classify(data, firsts)
        shape = keras_h5.input_shape_from_config(cfg) or (64, 512, 1)
    ovars = keras_h5.optimizer_vars(data) if compile else []

    def fill(m, prefix=""):
        specs = m.net.params.specs
        m.set_weights_dict(keras_h5.pick(data, keras_h5.weight_paths(specs, prefix), specs))

    if kind == "generator":
        m = GeneratorModel(arch, shape)
        fill(m)
    elif kind == "cnn":
        m = CnnModel(shape)
        fill(m)
        st = keras_h5.split_optimizer(ovars, keras_h5.trainable_order(m.net.params.specs)) if ovars else None
        if st:
            _opt_from_keras(m.net.params, m._tr().opt, *st)
    elif kind == "discriminator":
        m = DiscriminatorModel(shape)
        fill(m)
        specs = m.net.params.specs
        if ours:
            frozen = not cfg.get("d_convs_trainable", True)
            mode = cfg.get("d_freeze")
        else:
            flags = _keras_conv_flags(cfg)
            frozen = bool(flags) and not any(flags)
            # an optimizer that holds BatchNorm variables only was built AFTER define_gan froze the convolutions: the
            # archive's writer read `trainable` at call time (keras_api.DiscriminatorModel.d_step_trains_convs)
            n_all, n_bn = len(keras_h5.trainable_order(specs)), len(keras_h5.trainable_order(specs, convs=False))
            mode = {2 + 2 * n_all: "compile", 2 + 2 * n_bn: "call"}.get(len(ovars)) if ovars else None
        if mode:
            m._d_freeze = mode
        if frozen:
            for l in m.layers:
                if not isinstance(l, BatchNormalization):
                    l.trainable = False
        if ovars:
            opt = m._tr().opt
            names = [n for n in keras_h5.trainable_order(specs) if opt.lo <= specs[n].offset < opt.hi]
            st = keras_h5.split_optimizer(ovars, names)
            if st:
                _opt_from_keras(m.net.params, opt, *st)
    elif kind == "gan":
        g, d = GeneratorModel(arch, shape), DiscriminatorModel(shape)
        fill(g, "layers/functional/")
        fill(d, "layers/functional_1/")
        m = define_gan(g, d, shape)
        if ovars:
            ng = keras_h5.trainable_order(g.net.params.specs)
            nd = keras_h5.trainable_order(d.net.params.specs, convs=False)
            st = keras_h5.split_optimizer(ovars, ng + nd)
            if st:
                t, mm, vv = st
                # (the names of the two nets overlap -- c2.bn.gamma vs e2.bn.gamma do not, but keep them apart anyway)
                k = 2 + 2 * len(ng)
                sg = keras_h5.split_optimizer(ovars[:k], ng)
                sd = keras_h5.split_optimizer(ovars[:2] + ovars[k:], nd)
                _opt_from_keras(g.net.params, m._trainer.opt_g, *sg)
                _opt_from_keras(d.net.params, m._trainer.opt_dbn, *sd)
    else:
        raise ValueError(f"unknown model class {kind}")
    return m


def load_model(filepath, compile=True, custom_objects=None):
    """keras.models.load_model counterpart: reads the archives save_model above writes (HDF5 or npz member) AND `.keras`
    archives written by Keras 3 itself for the four reference graphs (U/train.py:198-203): the graph is recognised from
    the shapes in `model.weights.h5`, the image shape from the InputLayer in Keras' config.json, the optimizer state
    from `optimizer/vars/*` (compile=True).  custom_objects (the MC_* losses) is accepted and not needed.
    PARITY UNPINNED for Keras-written archives: the variable order assumed below -- [iterations, learning_rate, then
    (momentum, velocity) per trainable variable; for the composite GAN all of G's variables before D's BatchNorm
    gamma / beta] -- is Keras 3's published rule, never checked against a file Keras wrote (none exists here).  Shapes
    are verified per name; two same-shaped tensors listed in another order (e.g. D before G) would be assigned silently."""
    filepath = str(filepath)
    with zipfile.ZipFile(filepath) as z:
        names = set(z.namelist())
        cfg = json.loads(z.read("config.json"))
        ours = isinstance(cfg, dict) and cfg.get("format") == FORMAT
        if "model.weights.h5" in names:
            return _load_h5_archive(z, cfg, ours, compile, filepath)
        if not ours:
            raise ValueError(f"unknown archive format in {filepath}")
        w = dict(np.load(io.BytesIO(z.read("model.weights.npz")), allow_pickle=False))
        o = dict(np.load(io.BytesIO(z.read("optimizer.npz")), allow_pickle=False)) if "optimizer.npz" in names else {}
    shape = tuple(cfg["image_shape"])
    kind = cfg["class_name"]
    if kind == "generator":
        m = GeneratorModel(cfg["arch"], shape)
        m.set_weights_dict(w)
    elif kind == "discriminator":
        m = DiscriminatorModel(shape)
        m.set_weights_dict(w)
        if cfg.get("d_freeze"):
            m._d_freeze = cfg["d_freeze"]
            if not cfg.get("d_convs_trainable", True):
                for l in m.layers:                      # saved after define_gan froze the convolutions: restore that
                    if not isinstance(l, BatchNormalization):
                        l.trainable = False
        if compile and o:
            _load_opt(m._tr().opt, o, "opt/")
    elif kind == "cnn":
        m = CnnModel(shape)
        m.set_weights_dict(w)
        if compile and o:
            _load_opt(m._tr().opt, o, "opt/")
    elif kind == "gan":
        g, d = GeneratorModel(cfg["arch"], shape), DiscriminatorModel(shape)
        g.set_weights_dict({k[2:]: v for k, v in w.items() if k.startswith("g/")})
        d.set_weights_dict({k[2:]: v for k, v in w.items() if k.startswith("d/")})
        m = define_gan(g, d, shape)
        if compile and o:
            _load_opt(m._trainer.opt_g, o, "gan_g/")
            _load_opt(m._trainer.opt_dbn, o, "gan_dbn/")
    else:
        raise ValueError(f"unknown model class {kind}")
    return m


def link_gan(gan_model: GanModel, g_model: GeneratorModel, d_model: DiscriminatorModel) -> GanModel:
    """After loading three separate archives (U/train.py:214-216) the composite must train the SAME G and D
    objects the script keeps using; Keras restores that sharing only inside one process graph, so the entry
    scripts call this to re-tie them (the composite's optimizer state is kept)."""
    tr_old = gan_model._trainer
    new = GanModel(g_model, d_model, gan_model.image_shape)
    for a, b in ((new._trainer.opt_g, tr_old.opt_g), (new._trainer.opt_dbn, tr_old.opt_dbn)):
        a.m.copy_(b.m)
        a.v.copy_(b.v)
        a.t = b.t
    return new
