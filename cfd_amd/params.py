"""Parameter storage for the surrogate nets: one flat fp32 master buffer in KERNEL layouts
(conv weights packed [Cout][kh][kw][Cin], dense [out][in]) plus a flat shadow copy in the compute
dtype for the MFMA operands.  Keras layouts appear only at the interchange boundary
(`load_keras` / `to_keras`, used by checkpoints, get/set_weights and the parity tests).

Order inside the flat buffer: trainable non-BN tensors, then BN gamma/beta (one contiguous range,
because `define_gan` leaves exactly those of D trainable, U/models/generator.py:79-81), then the
non-trainable moving statistics.  Every tensor starts on a 512-element boundary: 16-B vector loads (bias,
gamma) and a single fused Adam launch over the trainable prefix stay aligned, and every range between
two layer boundaries divides into 2, 4 or 8 chunks of whole 64-element groups -- the reduce-scatter /
sharded-Adam / all-gather data-parallel step (trainer._DP) splits such ranges by rank.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import ops

ALIGN = 512


class Spec:
    __slots__ = ("name", "kind", "keras_shape", "packed_shape", "offset", "numel")

    def __init__(self, name: str, kind: str, keras_shape: Tuple[int, ...]):
        self.name, self.kind, self.keras_shape = name, kind, tuple(keras_shape)
        if kind == "conv":       # Keras (kh,kw,Cin,Cout) -> [Cout][kh][kw][Cin]
            kh, kw, ci, co = keras_shape
            self.packed_shape = (co, kh, kw, ci)
        elif kind == "convT":    # Keras (kh,kw,Cout,Cin) -> [Cout][kh][kw][Cin]
            kh, kw, co, ci = keras_shape
            self.packed_shape = (co, kh, kw, ci)
        elif kind == "dense":    # Keras (in,out) -> [out][in]
            ci, co = keras_shape
            self.packed_shape = (co, ci)
        else:                    # vec / bn / stat
            self.packed_shape = tuple(keras_shape)
        self.numel = int(np.prod(self.packed_shape))
        self.offset = -1


def _group(kind: str, name: str) -> int:
    if kind == "stat":
        return 2
    if kind == "bn":
        return 1
    return 0


class ParamStore:
    def __init__(self, specs: List[Spec], device, compute_dtype: torch.dtype):
        self.device, self.compute_dtype = torch.device(device), compute_dtype
        ordered = sorted(range(len(specs)), key=lambda i: (_group(specs[i].kind, specs[i].name), i))
        self.specs: "OrderedDict[str, Spec]" = OrderedDict()
        off = 0
        bounds = [0, 0, 0]
        for i in ordered:
            s = specs[i]
            s.offset = off
            off += (s.numel + ALIGN - 1) // ALIGN * ALIGN
            self.specs[s.name] = s
            bounds[_group(s.kind, s.name)] = off
        self.n_main = bounds[0]                       # [0, n_main): conv/dense kernels + biases
        self.n_trainable = max(bounds[0], bounds[1])  # [n_main, n_trainable): BN gamma/beta
        self.n_total = off
        self.master = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.compute = self.master if compute_dtype == torch.float32 else \
            torch.zeros(self.n_trainable, dtype=compute_dtype, device=self.device)

    # ---- views -------------------------------------------------------------------------
    def m(self, name: str) -> torch.Tensor:
        s = self.specs[name]
        return self.master[s.offset:s.offset + s.numel].view(s.packed_shape)

    def c(self, name: str) -> torch.Tensor:
        s = self.specs[name]
        return self.compute[s.offset:s.offset + s.numel].view(s.packed_shape)

    # ---- training state -----------------------------------------------------------------
    def enable_training(self) -> None:
        """allocate the flat gradient buffer and the transposed compute-dtype weights used by the
        data-gradient GEMMs (dX of a conv = forward kernel on Wt[ci][taps][co])"""
        if getattr(self, "grad", None) is None:
            self.grad = torch.zeros(self.n_trainable, dtype=torch.float32, device=self.device)
            self.compute_t = torch.zeros(self.n_main, dtype=self.compute_dtype, device=self.device)
            self.refresh_transposed()

    def g(self, name: str) -> torch.Tensor:
        s = self.specs[name]
        return self.grad[s.offset:s.offset + s.numel].view(s.packed_shape)

    def t(self, name: str) -> torch.Tensor:
        """transposed weights [Cin][taps][Cout] (taps flipped for the stride-1 convs) in the compute dtype"""
        s = self.specs[name]
        shp = (s.packed_shape[-1],) + tuple(s.packed_shape[1:-1]) + (s.packed_shape[0],)
        return self.compute_t[s.offset:s.offset + s.numel].view(shp)

    def refresh_transposed(self) -> None:
        if getattr(self, "compute_t", None) is None:
            return
        for name, s in self.specs.items():
            if s.kind in ("conv", "convT", "dense") and name not in self.no_transpose:
                flip = s.kind == "conv" and s.packed_shape[1] * s.packed_shape[2] in (9,) or name in self.flip_taps
                ops.transpose_weight(self.c(name), bool(flip), out=self.t(name))

    no_transpose: frozenset = frozenset()
    flip_taps: frozenset = frozenset()

    def refresh_compute(self) -> None:
        """compute copy <- master (after loading weights; the fused Adam kernel keeps it in sync later)"""
        if self.compute is not self.master:
            n = self.n_trainable
            ops.cast_copy(self.master[:n].view(1, 1, 1, n), self.compute.view(1, 1, 1, n))

    # ---- Keras interchange ----------------------------------------------------------------
    def load_keras(self, weights: Dict[str, np.ndarray]) -> None:
        missing = set(self.specs) - set(weights)
        if missing:
            raise KeyError(f"missing weights: {sorted(missing)[:5]} ...")
        for name, s in self.specs.items():
            w = np.ascontiguousarray(np.asarray(weights[name], dtype=np.float32))
            if tuple(w.shape) != s.keras_shape:
                raise ValueError(f"{name}: expected Keras shape {s.keras_shape}, got {w.shape}")
            src = torch.from_numpy(w).to(self.device)
            if s.kind == "conv":
                self.m(name).copy_(ops.pack_conv_weight(src, torch.float32))
            elif s.kind == "convT":
                self.m(name).copy_(ops.pack_convT_weight(src, torch.float32))
            elif s.kind == "dense":
                self.m(name).copy_(ops.pack_dense_weight(src, torch.float32))
            else:
                self.m(name).copy_(src)
        self.refresh_compute()

    def to_keras(self) -> Dict[str, np.ndarray]:
        """checkpoint / test export in Keras layouts.  The layout change runs on the device (a strided host transpose
        of the 128 M-parameter generator took 2 s per save: the training script's wall time was its checkpoints)."""
        out = {}
        for name, s in self.specs.items():
            a = self.master[s.offset:s.offset + s.numel].view(s.packed_shape)
            if s.kind == "conv":       # [co][kh][kw][ci] -> (kh,kw,ci,co)
                a = a.permute(1, 2, 3, 0)
            elif s.kind == "convT":    # [co][kh][kw][ci] -> (kh,kw,co,ci)
                a = a.permute(1, 2, 0, 3)
            elif s.kind == "dense":
                a = a.permute(1, 0)
            out[name] = a.contiguous().cpu().numpy()
        return out

    def init_random_normal(self, seed: int, stddev: float = 0.02) -> None:
        """RandomNormal(stddev=0.02) kernels, zero biases, BN gamma=1 beta=0, moving mean 0 var 1
        (U/models/generator.py:16).  Initialisation only -- drawn on the host, uploaded once."""
        g = torch.Generator().manual_seed(seed)
        for name, s in self.specs.items():
            v = self.m(name)
            if s.kind in ("conv", "convT", "dense"):
                v.copy_((torch.randn(s.packed_shape, generator=g) * stddev).to(self.device))
            elif name.endswith("gamma") or name.endswith("moving_var"):
                v.fill_(1.0)
            else:
                v.zero_()
        self.refresh_compute()
