"""Parameter storage for the surrogate nets: one flat fp32 master buffer in KERNEL layouts
(conv weights packed [Cout][kh][kw][Cin], dense [out][in]) plus a flat shadow copy in the compute
dtype for the MFMA operands.  Keras layouts appear only at the interchange boundary
(`load_keras` / `to_keras`, used by checkpoints, get/set_weights and the parity tests).

Order inside the flat buffer: the GEMM kernels (read by the MFMA kernels through the compute-dtype shadow), then
the small tensors every kernel reads in fp32 straight from the master -- biases and the first layers' kernels
(Cin <= 2: conv_first) --, then BN gamma/beta (one contiguous range, because `define_gan` leaves exactly those
of D trainable, U/models/generator.py:79-81), then the non-trainable moving statistics.  [0, n_kernels) is the only
range the data-parallel step shards: a rank's fp32 master of chunks it does not own goes stale there, which nothing
in a forward or backward reads; [n_kernels, n_trainable) is updated replicated on every rank (trainer._DP).  Every tensor starts on a 512-element boundary: 16-B vector loads (bias,
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


def _group(s: "Spec") -> int:
    if s.kind == "stat":
        return 3
    if s.kind == "bn":
        return 2
    if s.kind == "vec" or (s.kind == "conv" and s.keras_shape[2] <= 2):
        return 1          # read in fp32 from the master by the kernels (biases; conv_first's weights)
    return 0


class ParamStore:
    def __init__(self, specs: List[Spec], device, compute_dtype: torch.dtype):
        self.device, self.compute_dtype = torch.device(device), compute_dtype
        ordered = sorted(range(len(specs)), key=lambda i: (_group(specs[i]), i))
        # `specs` keeps the DECLARATION order (= graph order: keras_h5 and the Keras surface's `.layers` walk it);
        # offsets follow the group order
        self.specs: "OrderedDict[str, Spec]" = OrderedDict((s.name, s) for s in specs)
        off = 0
        bounds = [0, 0, 0, 0]
        for i in ordered:
            s = specs[i]
            s.offset = off
            off += (s.numel + ALIGN - 1) // ALIGN * ALIGN
            for k in range(_group(s), 4):
                bounds[k] = off
        self.n_kernels = bounds[0]                    # [0, n_kernels): GEMM kernels (layer order)
        self.n_main = bounds[1]                       # [n_kernels, n_main): biases + first-layer kernels
        self.n_trainable = bounds[2]                  # [n_main, n_trainable): BN gamma/beta
        self.n_total = off
        self._pending_updates: List = []              # events of parameter all-gathers in flight (trainer._DP)
        self.master = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.compute = self.master if compute_dtype == torch.float32 else \
            torch.zeros(self.n_trainable, dtype=compute_dtype, device=self.device)

    # ---- asynchronous parameter updates (data-parallel step) --------------------------------
    def join_updates(self) -> None:
        """the current stream waits for every parameter all-gather of the last optimiser step still in flight
        (trainer._DP.gather_async); every forward, the transposed-weight refresh and sync_state call it"""
        for ev in self._pending_updates:
            torch.cuda.current_stream().wait_event(ev)

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

    def refresh_transposed(self, side_stream: bool = False) -> None:
        """compute_t <- transposed compute copy (after the weights changed).  side_stream=True (the train steps): the
        transposes are HBM-bound and only the NEXT backward reads them, so they run on a side stream behind everything
        enqueued so far and overlap the MFMA-bound kernels the caller enqueues next; `wait_transposed()` (called by
        every backward) joins them."""
        if getattr(self, "compute_t", None) is None:
            return
        if side_stream and self.master.is_cuda:
            if self._t_stream is None:
                self._t_stream = ops.side_stream("transpose")
            self.wait_transposed()                     # never two refreshes in flight
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._t_stream):
                self._t_stream.wait_event(ev)
                self._transpose_all()
                self._t_done = torch.cuda.Event()
                self._t_done.record()
        else:
            self.wait_transposed()
            self._transpose_all()

    def _transpose_all(self) -> None:
        self.join_updates()
        for name, s in self.specs.items():
            if s.kind in ("conv", "convT", "dense") and name not in self.no_transpose:
                flip = s.kind == "conv" and s.packed_shape[1] * s.packed_shape[2] in (9,) or name in self.flip_taps
                ops.transpose_weight(self.c(name), bool(flip), out=self.t(name))

    def wait_transposed(self) -> None:
        """the current stream waits for a refresh_transposed(side_stream=True) still in flight"""
        if self._t_done is not None:
            torch.cuda.current_stream().wait_event(self._t_done)
            self._t_done = None

    def zero_grad(self) -> None:
        """Clear what the backward ACCUMULATES into: every trainable tensor except the GEMM-shaped weight gradients
        named in `stored_grads`, which ops.conv_wgrad overwrites (or zeroes itself on its atomic path).  One launch
        over a device-resident range table instead of a memset of the whole flat buffer."""
        if self._zero_table is None:
            ranges = []
            for name, s in sorted(self.specs.items(), key=lambda kv: kv[1].offset):
                if s.offset >= self.n_trainable or name in self.stored_grads:
                    continue
                span = (s.numel + ALIGN - 1) // ALIGN * ALIGN            # the padding behind a tensor may be cleared too
                if ranges and ranges[-1][0] + ranges[-1][1] == s.offset:
                    ranges[-1][1] += span
                else:
                    ranges.append([s.offset, span])
            for off, cnt in ranges:
                if off < 0 or off + cnt > self.n_trainable:
                    raise RuntimeError("zero_grad: range outside the gradient buffer")
            self._zero_max = max((c for _o, c in ranges), default=0)
            self._zero_table = torch.tensor(ranges or [[0, 0]], dtype=torch.int64, device=self.device).reshape(-1, 2)
        if self._zero_max > 0:
            ops.zero_ranges(self.grad, self._zero_table, self._zero_max)

    no_transpose: frozenset = frozenset()
    flip_taps: frozenset = frozenset()
    stored_grads: frozenset = frozenset()      # kernels whose gradient ops.conv_wgrad writes (set by the nets)
    _t_stream = None
    _t_done = None
    _zero_table = None
    _zero_max = 0

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

    def to_keras(self, flat: torch.Tensor = None, base: int = 0, names=None) -> Dict[str, np.ndarray]:
        """checkpoint / test export in Keras layouts.  The layout change runs on the device (a strided host transpose
        of the 128 M-parameter generator took 2 s per save: the training script's wall time was its checkpoints).
        flat / base: any flat buffer laid out like the master from offset `base` on (Adam's m and v); names: a subset."""
        out = {}
        src = self.master if flat is None else flat
        for name in (self.specs if names is None else names):
            s = self.specs[name]
            a = src[s.offset - base:s.offset - base + s.numel].view(s.packed_shape)
            if s.kind == "conv":       # [co][kh][kw][ci] -> (kh,kw,ci,co)
                a = a.permute(1, 2, 3, 0)
            elif s.kind == "convT":    # [co][kh][kw][ci] -> (kh,kw,co,ci)
                a = a.permute(1, 2, 0, 3)
            elif s.kind == "dense":
                a = a.permute(1, 0)
            out[name] = a.contiguous().cpu().numpy()
        return out

    def pack_keras(self, weights: Dict[str, np.ndarray], flat: torch.Tensor, base: int = 0) -> None:
        """Keras-layout arrays -> their packed places in `flat` (laid out like the master from offset `base` on):
        Adam moments of an imported checkpoint.  Only the names given are written."""
        for name, w in weights.items():
            s = self.specs[name]
            w = np.ascontiguousarray(np.asarray(w, dtype=np.float32))
            if tuple(w.shape) != s.keras_shape:
                raise ValueError(f"{name}: expected Keras shape {s.keras_shape}, got {w.shape}")
            src = torch.from_numpy(w).to(self.device)
            dst = flat[s.offset - base:s.offset - base + s.numel].view(s.packed_shape)
            if s.kind == "conv":
                dst.copy_(ops.pack_conv_weight(src, torch.float32))
            elif s.kind == "convT":
                dst.copy_(ops.pack_convT_weight(src, torch.float32))
            elif s.kind == "dense":
                dst.copy_(ops.pack_dense_weight(src, torch.float32))
            else:
                dst.copy_(src)

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
