"""Thin host wrappers: torch tensors (device memory + stream plumbing) -> C ABI of libcfd_amd.so.

Every function here launches hand-written gfx950 kernels; none computes with torch ops.
Activations are NHWC.  A tensor may be a channel-slice view `buf[..., a:b]` of a dense NHWC
buffer (that is how concat buffers are written in place); `_nhwc` turns it into (pointer, ld).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import (ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, BF16, CONV_K3S1, CONV_K4S1, CONV_K4S2, CONVT_K4S2,
                   DENSE, F32, TUNE_CLASS_ORDER, TUNE_IGEMM256, TUNE_STAGED_OUT, TUNE_IGEMM256_PD, TUNE_WGRAD256, TUNE_NTILE, TUNE_PATCH, TUNE_RESAMPLE_FUSED, TUNE_ONEPHASE, TUNE_SKINNY, ConvDesc, check)

LEAKY_SLOPE = 0.2
_DT = {torch.float32: F32, torch.bfloat16: BF16}
_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.CfdError("cfd_amd ops need device tensors (there is no CPU path)")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(_dev(t).data_ptr())


def _nhwc(t: torch.Tensor) -> Tuple[C.c_void_p, int]:
    """(pointer, channel stride) of a dense NHWC tensor or a channel-slice view of one."""
    _dev(t)
    if t.dim() == 2:  # (B, F): dense features
        if t.stride(1) != 1:
            raise _lib.CfdError("feature tensor must be contiguous in its last dim")
        return C.c_void_p(t.data_ptr()), t.stride(0)
    b, h, w, c = t.shape
    sb, sh, sw, sc = t.stride()
    ld = sw if w > 1 else (sh if h > 1 else (sb if b > 1 else c))
    if sc != 1 or (w > 1 and sh != w * ld) or (h > 1 and b > 1 and sb != h * w * ld) or ld < c:
        raise _lib.CfdError(f"tensor is not a channel slice of a dense NHWC buffer: shape {tuple(t.shape)} "
                            f"strides {tuple(t.stride())}")
    return C.c_void_p(t.data_ptr()), ld


def _need(cond: bool, what: str) -> None:
    """Host-side operand check: the kernels index by the sizes they are given and never bounds-check, so every
    wrapper verifies that each tensor really has the extent its kernel will touch before launching."""
    if not cond:
        raise _lib.CfdError("shape check failed: " + what)


def _numel(t: Optional[torch.Tensor], n: int, what: str, dtype: Optional[torch.dtype] = None) -> None:
    if t is None:
        return
    _need(t.numel() >= n and t.is_contiguous(), f"{what} needs {n} contiguous elements, has {tuple(t.shape)}")
    if dtype is not None:
        _need(t.dtype == dtype, f"{what} must be {dtype}, is {t.dtype}")


def conv_out_shape(kind: int, B: int, Hi: int, Wi: int, cout: int):
    if kind == CONV_K4S2:
        return (B, Hi // 2, Wi // 2, cout)
    if kind == CONVT_K4S2:
        return (B, Hi * 2, Wi * 2, cout)
    return (B, Hi, Wi, cout)


def set_tuning(key: int, value: int) -> None:
    """cfd_set_tuning: A/B switches of the library (see include/cfd_amd.h); -1 restores the heuristic."""
    check(_lib.load().cfd_set_tuning(int(key), int(value)), "cfd_set_tuning")


_ws_cache = {}
_ws_lane = 0          # which scratch `_workspace` hands out: 0 = the launch stream's, 1 = the weight-gradient side stream's


_side_streams: dict = {}


def side_stream(name: str) -> "torch.cuda.Stream":
    """ONE stream per purpose and process ('wgrad', 'transpose'), shared by every net and parameter store.  The box gives a
    process four hardware queues (GPU_MAX_HW_QUEUES); with a stream per object the U training step had five (compute,
    a weight-gradient stream per net, a transpose stream per parameter store), two of them folded onto one queue, and
    WHICH two differed from process to process: the same step ran 7.54 ms or 9.4 ms (round 3, three processes in a row on
    one box).  Work on streams of one purpose never overlaps in time anyway (G's and D's backward alternate)."""
    st = _side_streams.get(name)
    if st is None:
        st = _side_streams[name] = torch.cuda.Stream()
    return st


class KernelTimer:
    """Brackets every cfd_conv_forward launch with HIP events on the launch stream (torch's current
    stream IS the stream the kernels are launched on) and records their algorithmic FLOPs.
    Used by bench.py for the `roofline` object; off by default (zero overhead)."""

    def __init__(self, max_launches: int = 1024, sync: bool = True):
        self.records = []   # (tag, flops, start_event, end_event)
        self.sync = sync    # False: the caller keeps work queued ahead of the first bracket (bench.py's head start)
        # torch creates the HIP event at its first record(): do that here, outside every bracket, so that a bracket
        # costs two hipEventRecord calls and nothing else
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * max_launches)]
        for e in self.pool:
            e.record()

    def pair(self):
        if len(self.pool) >= 2:
            return self.pool.pop(), self.pool.pop()
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        global _timer
        if self.sync:
            torch.cuda.synchronize()
        _timer = self
        return self

    def __exit__(self, *a):
        global _timer
        _timer = None

    def summary(self):
        torch.cuda.synchronize()
        tot_ms = sum(s.elapsed_time(e) for _, _, s, e in self.records)
        tot_fl = sum(f for _, f, _, _ in self.records)
        return dict(launches=len(self.records), total_ms=tot_ms, total_flops=tot_fl)


_timer: Optional[KernelTimer] = None


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only fp32 scratch per device and lane (split-K slabs, tap buffers)."""
    # one scratch per (device, lane): launches that may overlap (nets._Net.wgrad's side stream sets lane 1) must not
    # share slabs
    key = (device.index if device.index is not None else torch.cuda.current_device(), _ws_lane)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


def conv_forward(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], kind: int, cout: int,
                 act: int = ACT_NONE, out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None,
                 split_k: Optional[int] = None, slope: float = LEAKY_SLOPE,
                 stats: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """y = act(conv(x, w) + bias) on MFMA.  x NHWC (or (B,F) for DENSE); w packed [Cout][kh][kw][Cin].
    stats = (sum, sumsq) fp32 (B, cout) buffers, zeroed by the caller: if the shape allows it the kernel also
    accumulates the per-sample BatchNorm statistics into them and (y, True) is returned, else (y, False)."""
    lib = _lib.load()
    if kind == DENSE:
        B, cin = x.shape
        Hi = Wi = 1
    else:
        B, Hi, Wi, cin = x.shape
    xp, ldx = _nhwc(x)
    odt = out_dtype or (out.dtype if out is not None else x.dtype)
    if out is None:
        shp = (B, cout) if kind == DENSE else conv_out_shape(kind, B, Hi, Wi, cout)
        out = torch.empty(shp, dtype=odt, device=x.device)
    yp, ldy = _nhwc(out)
    want = (B, cout) if kind == DENSE else conv_out_shape(kind, B, Hi, Wi, cout)
    _need(tuple(out.shape) == tuple(want), f"conv_forward out is {tuple(out.shape)}, kind {kind} on {tuple(x.shape)} gives {want}")
    _numel(w, cout * _TAPS[kind] * cin, "conv weight")
    _numel(bias, cout, "conv bias", torch.float32)
    if stats is not None:
        _numel(stats[0], B * cout, "stats sum", torch.float32)
        _numel(stats[1], B * cout, "stats sumsq", torch.float32)
    d = ConvDesc(kind=kind, dtype=_DT[x.dtype], out_dtype=_DT[out.dtype], B=B, Hi=Hi, Wi=Wi, Cin=cin, Cout=cout,
                 ldx=ldx, ldy=ldy, act=act, leaky_slope=slope, split_k=1)
    if w.dtype != x.dtype:
        raise _lib.CfdError(f"weight dtype {w.dtype} != activation dtype {x.dtype}")
    d.split_k = int(split_k) if split_k else lib.cfd_conv_suggest_split_k(C.byref(d))
    fused = stats is not None and act == ACT_NONE and bool(lib.cfd_conv_stats_fusable(C.byref(d)))
    ws = None
    if d.split_k > 1:
        nbytes = lib.cfd_conv_workspace_bytes(C.byref(d))
        if nbytes < 0:
            check(-1, "cfd_conv_workspace_bytes")
        ws = _workspace(nbytes, x.device)
    if _timer is not None:
        e0, e1 = _timer.pair()
        e0.record()
    if fused:
        check(lib.cfd_conv_forward_stats(C.byref(d), xp, _ptr(w), _ptr(bias), yp, _ptr(ws), _ptr(stats[0]),
                                         _ptr(stats[1]), _stream()), "cfd_conv_forward_stats")
    else:
        check(lib.cfd_conv_forward(C.byref(d), xp, _ptr(w), _ptr(bias), yp, _ptr(ws), _stream()), "cfd_conv_forward")
    if _timer is not None:
        e1.record()
        taps = {CONV_K4S2: 16, CONV_K4S1: 16, CONV_K3S1: 9, CONVT_K4S2: 4, DENSE: 1, 5: 16}[kind]
        opix = out.numel() // cout
        _timer.records.append((kind, 2.0 * opix * cout * taps * cin, e0, e1))
    return (out, fused) if stats is not None else out


def conv_pool2_fusable(x: torch.Tensor, kind: int, cout: int, act: int = ACT_LEAKY, slope: float = LEAKY_SLOPE) -> bool:
    B, Hi, Wi, cin = x.shape
    d = ConvDesc(kind=kind, dtype=_DT[x.dtype], out_dtype=_DT[x.dtype], B=B, Hi=Hi, Wi=Wi, Cin=cin, Cout=cout,
                 ldx=_nhwc(x)[1], ldy=cout, act=act, leaky_slope=slope, split_k=1)
    lib = _lib.load()
    # fusable, and worth it: the pooled epilogue cannot split K, so only where the plain launch would not either
    return bool(lib.cfd_conv_pool2_fusable(C.byref(d))) and lib.cfd_conv_suggest_split_k(C.byref(d)) == 1


def conv_forward_pool2(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], kind: int, cout: int,
                       act: int = ACT_LEAKY, out: Optional[torch.Tensor] = None, slope: float = LEAKY_SLOPE) -> torch.Tensor:
    """MaxPooling2D(2x2)(act(conv(x, w) + bias)) in ONE launch (inference: the unpooled activation is never written).
    out: (B, Hi/2, Wi/2, cout), same dtype as x.  Shapes: see conv_pool2_fusable."""
    lib = _lib.load()
    B, Hi, Wi, cin = x.shape
    xp, ldx = _nhwc(x)
    if out is None:
        out = torch.empty((B, Hi // 2, Wi // 2, cout), dtype=x.dtype, device=x.device)
    _need(tuple(out.shape) == (B, Hi // 2, Wi // 2, cout) and out.dtype == x.dtype,
          f"conv_forward_pool2 out is {tuple(out.shape)} {out.dtype}, wanted {(B, Hi // 2, Wi // 2, cout)} {x.dtype}")
    yp, ldy = _nhwc(out)
    _numel(w, cout * _TAPS[kind] * cin, "conv weight")
    _numel(bias, cout, "conv bias", torch.float32)
    if w.dtype != x.dtype:
        raise _lib.CfdError(f"weight dtype {w.dtype} != activation dtype {x.dtype}")
    d = ConvDesc(kind=kind, dtype=_DT[x.dtype], out_dtype=_DT[out.dtype], B=B, Hi=Hi, Wi=Wi, Cin=cin, Cout=cout,
                 ldx=ldx, ldy=ldy, act=act, leaky_slope=slope, split_k=1)
    if _timer is not None:
        e0, e1 = _timer.pair()
        e0.record()
    check(lib.cfd_conv_forward_pool2(C.byref(d), xp, _ptr(w), _ptr(bias), yp, _stream()), "cfd_conv_forward_pool2")
    if _timer is not None:
        e1.record()
        _timer.records.append((kind, 2.0 * B * Hi * Wi * cout * _TAPS[kind] * cin, e0, e1))
    return out


def conv_pool2_keep_fusable(x: torch.Tensor, kind: int, cout: int, act: int = ACT_LEAKY, slope: float = LEAKY_SLOPE) -> bool:
    """can conv + activation (kept) + 2x2 max-pool run as one launch (training forward)?"""
    B, Hi, Wi, cin = x.shape
    if x.dtype != torch.bfloat16:
        return False
    d = ConvDesc(kind=kind, dtype=_DT[x.dtype], out_dtype=_DT[x.dtype], B=B, Hi=Hi, Wi=Wi, Cin=cin, Cout=cout,
                 ldx=_nhwc(x)[1], ldy=cout, act=act, leaky_slope=slope, split_k=1)
    lib = _lib.load()
    return bool(lib.cfd_conv_pool2_keep_fusable(C.byref(d))) and lib.cfd_conv_suggest_split_k(C.byref(d)) == 1


def conv_forward_pool2_keep(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], kind: int, cout: int,
                            act: int, out: torch.Tensor, pooled: torch.Tensor, slope: float = LEAKY_SLOPE) -> torch.Tensor:
    """out = act(conv(x, w) + bias) (B, Hi, Wi, cout) AND pooled = MaxPooling2D(2x2)(out), one launch (training forward:
    the pool's backward needs `out`).  Shapes: see conv_pool2_keep_fusable.  Returns `pooled`."""
    lib = _lib.load()
    B, Hi, Wi, cin = x.shape
    xp, ldx = _nhwc(x)
    _need(tuple(out.shape) == (B, Hi, Wi, cout) and out.dtype == x.dtype and
          tuple(pooled.shape) == (B, Hi // 2, Wi // 2, cout) and pooled.dtype == x.dtype,
          f"conv_forward_pool2_keep: out {tuple(out.shape)} / pooled {tuple(pooled.shape)} do not match {(B, Hi, Wi, cout)}")
    yp, ldy = _nhwc(out)
    pp, ldp = _nhwc(pooled)
    _numel(w, cout * _TAPS[kind] * cin, "conv weight")
    _numel(bias, cout, "conv bias", torch.float32)
    if w.dtype != x.dtype:
        raise _lib.CfdError(f"weight dtype {w.dtype} != activation dtype {x.dtype}")
    d = ConvDesc(kind=kind, dtype=_DT[x.dtype], out_dtype=_DT[out.dtype], B=B, Hi=Hi, Wi=Wi, Cin=cin, Cout=cout,
                 ldx=ldx, ldy=ldy, act=act, leaky_slope=slope, split_k=1)
    if _timer is not None:
        e0, e1 = _timer.pair()
        e0.record()
    check(lib.cfd_conv_forward_pool2_keep(C.byref(d), xp, _ptr(w), _ptr(bias), yp, pp, ldp, _stream()),
          "cfd_conv_forward_pool2_keep")
    if _timer is not None:
        e1.record()
        _timer.records.append((kind, 2.0 * B * Hi * Wi * cout * _TAPS[kind] * cin, e0, e1))
    return pooled


def pack_conv_weight(keras_hwio: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    kh, kw, cin, cout = keras_hwio.shape
    src = _dev(keras_hwio).contiguous().float()
    dst = torch.empty((cout, kh, kw, cin), dtype=dtype, device=src.device)
    check(_lib.load().cfd_pack_conv_weight(_ptr(src), _ptr(dst), kh, kw, cin, cout, _DT[dtype], _stream()),
          "cfd_pack_conv_weight")
    return dst


def pack_convT_weight(keras_hwoi: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    kh, kw, cout, cin = keras_hwoi.shape
    src = _dev(keras_hwoi).contiguous().float()
    dst = torch.empty((cout, kh, kw, cin), dtype=dtype, device=src.device)
    check(_lib.load().cfd_pack_convT_weight(_ptr(src), _ptr(dst), kh, kw, cin, cout, _DT[dtype], _stream()),
          "cfd_pack_convT_weight")
    return dst


def pack_dense_weight(keras_io: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    cin, cout = keras_io.shape
    src = _dev(keras_io).contiguous().float()
    dst = torch.empty((cout, cin), dtype=dtype, device=src.device)
    check(_lib.load().cfd_pack_dense_weight(_ptr(src), _ptr(dst), cin, cout, _DT[dtype], _stream()),
          "cfd_pack_dense_weight")
    return dst


def conv_first(x: torch.Tensor, x2: Optional[torch.Tensor], w: torch.Tensor, bias: Optional[torch.Tensor],
               cout: int, ksize: int, stride: int, act: int, out_dtype: torch.dtype,
               out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None, act2: int = ACT_RELU,
               slope: float = LEAKY_SLOPE) -> torch.Tensor:
    """Direct conv for Cin<=4 inputs (fp32 image(s) in, `out_dtype` out).  w fp32 packed [Cout][k][k][Cin]."""
    B, Hi, Wi, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()) or x.dtype != torch.float32:
        raise _lib.CfdError("conv_first inputs must be dense fp32 NHWC")
    if out is None:
        out = torch.empty((B, Hi // stride, Wi // stride, cout), dtype=out_dtype, device=x.device)
    yp, ldy = _nhwc(out)
    y2p, ldy2 = (None, 0) if out2 is None else _nhwc(out2)
    want = (B, Hi // stride, Wi // stride, cout)
    _need(Hi % stride == 0 and Wi % stride == 0, "conv_first: H, W must be multiples of the stride")
    _need(x2 is None or (tuple(x2.shape[:3]) == (B, Hi, Wi) and x2.dtype == torch.float32),
          f"conv_first x2 {None if x2 is None else tuple(x2.shape)} does not match x {tuple(x.shape)}")
    _need(tuple(out.shape) == want, f"conv_first out is {tuple(out.shape)}, expected {want}")
    _need(out2 is None or (tuple(out2.shape) == want and out2.dtype == out.dtype),
          f"conv_first out2 {None if out2 is None else tuple(out2.shape)} != {want}")
    _numel(w, cout * ksize * ksize * (c1 + c2), "conv_first weight", torch.float32)
    _numel(bias, cout, "conv_first bias", torch.float32)
    check(_lib.load().cfd_conv_first(_ptr(x), _ptr(x2), c1, c2, _ptr(w), _ptr(bias), yp, ldy, y2p, ldy2, B, Hi, Wi,
                                     cout, ksize, stride, act, act2, slope, _DT[out.dtype], _stream()),
          "cfd_conv_first")
    return out


def conv3_pool2(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], cout: int, act: int,
                out_dtype: torch.dtype, out: Optional[torch.Tensor] = None, slope: float = LEAKY_SLOPE) -> torch.Tensor:
    """Conv2D 3x3 'same' on a 1-channel fp32 image + activation + MaxPooling2D(2,2), fused (inference)."""
    B, Hi, Wi, c1 = x.shape
    _need(c1 == 1 and x.is_contiguous() and x.dtype == torch.float32, "conv3_pool2: dense fp32 (B,H,W,1) input")
    _need(Hi % 2 == 0 and Wi % 2 == 0, "conv3_pool2 needs even H and W")
    want = (B, Hi // 2, Wi // 2, cout)
    if out is None:
        out = torch.empty(want, dtype=out_dtype, device=x.device)
    _need(tuple(out.shape) == want and out.is_contiguous(), f"conv3_pool2 out is {tuple(out.shape)}, expected dense {want}")
    _numel(w, cout * 9, "conv3_pool2 weight", torch.float32)
    _numel(bias, cout, "conv3_pool2 bias", torch.float32)
    check(_lib.load().cfd_conv3_pool2(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), B, Hi, Wi, cout, act, slope,
                                      _DT[out.dtype], _stream()), "cfd_conv3_pool2")
    return out


def convT_out(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], apply_sigmoid: bool = True,
              want_logits: bool = False, relu_on_load: bool = False):
    """Conv2DTranspose(1,(4,4),s2,'same') [+ sigmoid]; returns fp32 (B,2H,2W,1) (and the logits).
    relu_on_load: ReLU applied to x while it is read (the inference path's single copy of the e1 skip)."""
    B, Hi, Wi, cin = x.shape
    xp, ldx = _nhwc(x)
    taps = _workspace(B * Hi * Wi * 16 * 4, x.device)
    y = torch.empty((B, 2 * Hi, 2 * Wi, 1), dtype=torch.float32, device=x.device)
    logits = torch.empty_like(y) if want_logits else None
    _numel(w, 16 * cin, "convT_out weight", x.dtype)
    _numel(bias, 1, "convT_out bias", torch.float32)
    check(_lib.load().cfd_convT_out_relu(xp, ldx, _ptr(w), _ptr(bias), _ptr(taps), _ptr(y), _ptr(logits), B, Hi, Wi, cin,
                                         _DT[x.dtype], int(apply_sigmoid), int(relu_on_load), _stream()), "cfd_convT_out")
    return (y, logits) if want_logits else y


def convT_out_bn(raw: torch.Tensor, stats: Tuple[torch.Tensor, torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                 skip: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], apply_sigmoid: bool = True,
                 want_logits: bool = False, eps: float = 1e-3):
    """The output layer reading [ReLU(BN(raw)) | ReLU(skip)] without that tensor ever being written (inference).
    raw: pre-BatchNorm output of the last decoder convolution (B,H,W,C1) with its per-sample sums `stats`; skip: the
    first encoder activation (B,H,W,C2), possibly a channel slice of a wider buffer.  Same values as bn_apply (ReLU)
    + convT_out(relu_on_load=True) on the concat buffer, bit for bit."""
    B, Hi, Wi, c1 = raw.shape
    c2 = skip.shape[-1]
    _need(tuple(skip.shape) == (B, Hi, Wi, c2), f"skip is {tuple(skip.shape)}, raw {tuple(raw.shape)}")
    _need(raw.dtype in (skip.dtype, torch.float32), f"raw {raw.dtype} with compute dtype {skip.dtype}")
    rp, ldr = _nhwc(raw)
    sp, lds = _nhwc(skip)
    _numel(stats[0], B * c1, "stats sum", torch.float32)
    _numel(stats[1], B * c1, "stats sumsq", torch.float32)
    _numel(gamma, c1, "bn gamma", torch.float32)
    _numel(beta, c1, "bn beta", torch.float32)
    _numel(w, 16 * (c1 + c2), "convT_out weight", skip.dtype)
    _numel(bias, 1, "convT_out bias", torch.float32)
    taps = _workspace(B * Hi * Wi * 16 * 4, raw.device)
    y = torch.empty((B, 2 * Hi, 2 * Wi, 1), dtype=torch.float32, device=raw.device)
    logits = torch.empty_like(y) if want_logits else None
    check(_lib.load().cfd_convT_out_bn(rp, ldr, _DT[raw.dtype], _ptr(stats[0]), _ptr(stats[1]), _ptr(gamma), _ptr(beta),
                                       eps, sp, lds, _ptr(w), _ptr(bias), _ptr(taps), _ptr(y), _ptr(logits), B, Hi, Wi,
                                       c1, c2, _DT[skip.dtype], int(apply_sigmoid), _stream()), "cfd_convT_out_bn")
    return (y, logits) if want_logits else y


def _bn_checks(x, out, out2, gamma, beta, drop_mask) -> None:
    shp, Cc = tuple(x.shape), x.shape[-1]
    _need(tuple(out.shape) == shp, f"bn out {tuple(out.shape)} != x {shp}")
    _need(out2 is None or tuple(out2.shape) == shp, f"bn out2 {None if out2 is None else tuple(out2.shape)} != x {shp}")
    _numel(gamma, Cc, "bn gamma", torch.float32)
    _numel(beta, Cc, "bn beta", torch.float32)
    _need(drop_mask is None or tuple(drop_mask.shape) == shp, f"drop_mask shape != x {shp}")


def bn_act(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, act: int, out: Optional[torch.Tensor] = None,
           out2: Optional[torch.Tensor] = None, act2: int = ACT_RELU, drop_mask: Optional[torch.Tensor] = None,
           drop_rate: float = 0.0, drop_seed: int = 0, sample0: int = 0,
           mean: Optional[torch.Tensor] = None, var: Optional[torch.Tensor] = None,
           use_moving: bool = False, eps: float = 1e-3, slope: float = LEAKY_SLOPE) -> torch.Tensor:
    """BatchNorm (per-sample batch statistics, or given moving statistics) + dropout + activation."""
    B, H, W, Cc = x.shape
    xp, ldx = _nhwc(x)
    if out is None:
        out = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device)
    yp, ldy = _nhwc(out)
    y2p, ldy2 = (None, 0) if out2 is None else _nhwc(out2)
    if out2 is not None and out2.dtype != out.dtype:
        raise _lib.CfdError("out and out2 must share a dtype")
    if drop_mask is not None and (drop_mask.dtype != torch.uint8 or not drop_mask.is_contiguous()):
        raise _lib.CfdError("drop_mask must be dense uint8 NHWC")
    if drop_mask is not None and drop_rate == 0.0:
        drop_rate = 0.5
    _bn_checks(x, out, out2, gamma, beta, drop_mask)
    nstat = Cc if use_moving else B * Cc
    _numel(mean, nstat, "bn mean", torch.float32)
    _numel(var, nstat, "bn var", torch.float32)
    check(_lib.load().cfd_bn_act(xp, ldx, _DT[x.dtype], _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var),
                                 int(use_moving), _ptr(drop_mask), drop_rate, drop_seed, sample0, yp, ldy, y2p, ldy2,
                                 B, H * W, Cc, eps, act, act2, slope, _DT[out.dtype], _stream()), "cfd_bn_act")
    return out


def bn_apply(x: torch.Tensor, stats: Tuple[torch.Tensor, torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
             act: int, out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None, act2: int = ACT_RELU,
             drop_mask: Optional[torch.Tensor] = None, drop_rate: float = 0.0, drop_seed: int = 0, sample0: int = 0,
             mean: Optional[torch.Tensor] = None, var: Optional[torch.Tensor] = None, eps: float = 1e-3,
             slope: float = LEAKY_SLOPE) -> torch.Tensor:
    """bn_act for a conv output whose (sum, sumsq) statistics were accumulated by conv_forward(stats=...)."""
    B, H, W, Cc = x.shape
    xp, ldx = _nhwc(x)
    if out is None:
        out = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device)
    yp, ldy = _nhwc(out)
    y2p, ldy2 = (None, 0) if out2 is None else _nhwc(out2)
    if out2 is not None and out2.dtype != out.dtype:
        raise _lib.CfdError("out and out2 must share a dtype")
    if drop_mask is not None and (drop_mask.dtype != torch.uint8 or not drop_mask.is_contiguous()):
        raise _lib.CfdError("drop_mask must be dense uint8 NHWC")
    if drop_mask is not None and drop_rate == 0.0:
        drop_rate = 0.5
    _bn_checks(x, out, out2, gamma, beta, drop_mask)
    for t, what in ((stats[0], "stats sum"), (stats[1], "stats sumsq"), (mean, "bn mean"), (var, "bn var")):
        _numel(t, B * Cc, what, torch.float32)
    check(_lib.load().cfd_bn_apply(xp, ldx, _DT[x.dtype], _ptr(stats[0]), _ptr(stats[1]), _ptr(gamma), _ptr(beta),
                                   _ptr(mean), _ptr(var), _ptr(drop_mask), drop_rate, drop_seed, sample0, yp, ldy,
                                   y2p, ldy2, B, H * W, Cc, eps, act, act2, slope, _DT[out.dtype], _stream()),
          "cfd_bn_apply")
    return out


def bn_stats(x: torch.Tensor, stats: Tuple[torch.Tensor, torch.Tensor]) -> None:
    """accumulate per-sample (sum, sumsq) of x [B,H,W,C] into the zeroed fp32 (B, C) buffers `stats` (for bn_apply)"""
    B, H, W, Cc = x.shape
    xp, ldx = _nhwc(x)
    _numel(stats[0], B * Cc, "stats sum", torch.float32)
    _numel(stats[1], B * Cc, "stats sumsq", torch.float32)
    check(_lib.load().cfd_bn_stats(xp, ldx, _DT[x.dtype], _ptr(stats[0]), _ptr(stats[1]), B, H * W, Cc, _stream()),
          "cfd_bn_stats")


def dropout_epoch(value: int, set_: bool = False) -> None:
    """advance (or set) the device-resident dropout epoch on the current stream (see cfd_dropout_epoch)"""
    check(_lib.load().cfd_dropout_epoch(int(value), int(set_), _stream()), "cfd_dropout_epoch")


def dropout_mask(shape, rate: float, seed: int, sample0: int = 0, device="cuda") -> torch.Tensor:
    """The uint8 keep-mask bn_act's RNG mode applies for (seed, sample0) on a [B,H,W,C] tensor."""
    B, H, W, Cc = shape
    m = torch.empty(shape, dtype=torch.uint8, device=device)
    check(_lib.load().cfd_dropout_mask(_ptr(m), sample0, B, H * W, Cc, rate, seed, _stream()), "cfd_dropout_mask")
    return m


def maxpool2(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, H, W, Cc = x.shape
    if not x.is_contiguous():
        raise _lib.CfdError("maxpool2 input must be dense NHWC")
    _need(H % 2 == 0 and W % 2 == 0, "maxpool2 needs even H and W")
    y = out if out is not None else torch.empty((B, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
    _need(tuple(y.shape) == (B, H // 2, W // 2, Cc) and y.is_contiguous() and y.dtype == x.dtype,
          f"maxpool2 out must be dense {(B, H // 2, W // 2, Cc)} of x's dtype")
    check(_lib.load().cfd_maxpool2(_ptr(x), _ptr(y), B, H, W, Cc, _DT[x.dtype], _stream()), "cfd_maxpool2")
    return y


def cast_copy(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    xp, ldx = _nhwc(x)
    yp, ldy = _nhwc(out)
    pixels = x.numel() // x.shape[-1]
    _need(tuple(out.shape) == tuple(x.shape), f"cast_copy out {tuple(out.shape)} != x {tuple(x.shape)}")
    check(_lib.load().cfd_cast_copy(xp, ldx, _DT[x.dtype], yp, ldy, _DT[out.dtype], pixels, x.shape[-1], _stream()),
          "cfd_cast_copy")
    return out


# ----------------------------------------------------------------------------------------
# resample (pixels_coords_testing.py:151-227)
# ----------------------------------------------------------------------------------------
Y_RANGE, Z_RANGE, N_SLICES, TOL_1, TOL_2, P_CONVERT = 4e-3, 50e-3, 101, 1e-5, 1e-4, 1072.033


class ResampleJob:
    """Inputs of cfd_resample_nearest resident in HBM: the concatenated point tables of `cases` (each (N_i,5) fp64 rows
    x,y,z,p,Umean), their offsets, the pixel-centre / slice vectors, the workspace and the output.  `run()` launches
    the kernels on the current stream (no host transfer); bench.py times exactly that."""

    def __init__(self, cases, height: int = 64, width: int = 512, device="cuda"):
        lib = _lib.load()
        # grid vectors exactly as the reference builds them with numpy (:164-168, :188); set-up, not hot
        y_final = -0.002 + (np.arange(height) + 0.5) * (Y_RANGE / height)
        z_final = (np.arange(width) + 0.5) * (Z_RANGE / width)
        z_samples = np.linspace(0, Z_RANGE, N_SLICES)
        offs = np.zeros(len(cases) + 1, dtype=np.int64)
        for i, c in enumerate(cases):
            if c.ndim != 2 or c.shape[1] != 5:
                raise ValueError("each case must be (N,5): x,y,z,p,Umean")
            offs[i + 1] = offs[i] + c.shape[0]
        self.H, self.W, self.ncase, self.n_points = height, width, len(cases), int(offs[-1])
        self.pts = torch.from_numpy(np.ascontiguousarray(np.concatenate(cases, axis=0), dtype=np.float64)).to(device)
        d = lambda a: torch.from_numpy(a).to(device)
        self.offs, self.yf, self.zf, self.zs = d(offs), d(y_final), d(z_final), d(z_samples)
        self.out = torch.empty((self.ncase, height * width, 5), dtype=torch.float64, device=device)
        wsb = lib.cfd_resample_workspace_bytes(self.ncase, width, N_SLICES, self.n_points)
        if wsb < 0:
            check(-1, "cfd_resample_workspace_bytes")
        self.ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=device)

    def run(self) -> torch.Tensor:
        check(_lib.load().cfd_resample_nearest(_ptr(self.pts), _ptr(self.offs), self.ncase, self.n_points, _ptr(self.yf),
                                               self.H, _ptr(self.zf), self.W, _ptr(self.zs), N_SLICES, TOL_1, TOL_2,
                                               P_CONVERT, _ptr(self.out), _ptr(self.ws), _stream()),
              "cfd_resample_nearest")
        return self.out

    def count_evaluations(self) -> int:
        """fp64 distance evaluations of one run's search (cfd_resample_nearest_counted; same output as run())"""
        n = torch.zeros(1, dtype=torch.int64, device=self.out.device)
        check(_lib.load().cfd_resample_nearest_counted(_ptr(self.pts), _ptr(self.offs), self.ncase, self.n_points,
                                                       _ptr(self.yf), self.H, _ptr(self.zf), self.W, _ptr(self.zs),
                                                       N_SLICES, TOL_1, TOL_2, P_CONVERT, _ptr(self.out), _ptr(self.ws),
                                                       _ptr(n), _stream()), "cfd_resample_nearest_counted")
        return int(n.item())

    def run_noalloc(self, valid_slices: Optional[torch.Tensor] = None) -> torch.Tensor:
        """the same call through cfd_resample_nearest_noalloc: the library's built-in scratch instead of self.ws, no
        allocation anywhere (valid_slices: optional int32 (ncase,) device tensor the caller owns)"""
        if valid_slices is not None:
            _numel(valid_slices, self.ncase, "valid_slices", torch.int32)
        check(_lib.load().cfd_resample_nearest_noalloc(_ptr(self.pts), _ptr(self.offs), self.ncase, self.n_points,
                                                       _ptr(self.yf), self.H, _ptr(self.zf), self.W, _ptr(self.zs),
                                                       N_SLICES, TOL_1, TOL_2, P_CONVERT, _ptr(self.out),
                                                       _ptr(valid_slices), _stream()), "cfd_resample_nearest_noalloc")
        return self.out

    def algorithmic_bytes(self) -> int:
        """HBM bytes = input points + 5 output columns = 40 B x (N_pts + H*W) per case"""
        return 40 * (self.n_points + self.ncase * self.H * self.W)

    def n_valid_slices(self) -> np.ndarray:
        stride = 2 * N_SLICES + 2 * self.W + 2
        col = self.ws[:self.ncase * stride].view(self.ncase, stride)[:, 2 * N_SLICES + 2 * self.W]
        return np.ascontiguousarray(col.cpu().numpy()).view(np.int32)[::2]


def resample_nearest(cases, height: int = 64, width: int = 512, device="cuda", return_device: bool = False):
    """cases: list of (N_i,5) fp64 arrays with rows x,y,z,p,Umean (the reference's input CSV as
    parsed by pandas).  Returns (ncase, H*W, 5) fp64 rows y_center,z_center,p_center,Umean_center,
    p_convert -- the reference's output table -- computed on the GPU, bit-exact."""
    job = ResampleJob(cases, height, width, device)
    out = job.run()
    if (job.n_valid_slices() == 0).any():
        # numpy: np.interp raises "array of sample points is empty" (:210)
        raise ValueError("array of sample points is empty")
    return out if return_device else out.cpu().numpy()


def csv_format_rows(rows: np.ndarray) -> bytes:
    """(n_rows, n_cols) fp64 HOST table -> the body of the CSV pandas' `to_csv(index=False)` writes for it, byte for byte
    (pixels_coords_testing.py:223-226), formatted by the library's host-side writer (cfd_csv_format_rows)."""
    lib = _lib.load()
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    if rows.ndim != 2 or rows.shape[1] < 1:
        raise ValueError("rows must be (n_rows, n_cols) with n_cols >= 1")
    cap = rows.shape[0] * rows.shape[1] * 26
    buf = C.create_string_buffer(max(cap, 1))
    n = lib.cfd_csv_format_rows(rows.ctypes.data, rows.shape[0], rows.shape[1], C.addressof(buf), cap)
    if n < 0:
        check(int(n), "cfd_csv_format_rows")
    return buf.raw[:n]


def dataset_grids(p: np.ndarray, cell: float, device="cuda", return_device: bool = False):
    """p: (ncase, H, W) fp64 `p_convert` grids.  Returns (sdf (ncase,H,W) fp64, delta_p (ncase,) fp64): the model
    input and DeltaP of utils.load_data (utils.py:89-103), computed on the GPU; the signed distance is scikit-fmm's
    fast-marching scheme (order 2), one thread per case.  Raises ValueError for a case without a zero contour (all
    fluid or all solid), as skfmm.distance does."""
    lib = _lib.load()
    p = np.ascontiguousarray(p, dtype=np.float64)
    if p.ndim != 3:
        raise ValueError("p must be (ncase, H, W)")
    n, H, W = p.shape
    pd_ = torch.from_numpy(p).to(device)
    sdf = torch.empty((n, H, W), dtype=torch.float64, device=device)
    dp = torch.empty(n, dtype=torch.float64, device=device)
    flags = torch.empty(n, dtype=torch.int32, device=device)
    wsb = lib.cfd_dataset_workspace_bytes(n, H, W)
    if wsb < 0:
        check(-1, "cfd_dataset_workspace_bytes")
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    check(lib.cfd_dataset_grids(_ptr(pd_), n, H, W, float(cell), _ptr(sdf), _ptr(dp), _ptr(flags), _ptr(ws), _stream()),
          "cfd_dataset_grids")
    fl = flags.cpu().numpy()
    if ((fl & 3) != 3).any():
        raise ValueError("the array phi contains no zero contour (no zero level set)")      # skfmm/pfmm.py
    if return_device:
        return sdf, dp
    return sdf.cpu().numpy(), dp.cpu().numpy()


def eval_metrics(pred: torch.Tensor, target: torch.Tensor, scale: float = 1.0, want_colmax: bool = True):
    """pred, target: fp32 device tensors (n, H, W) (a trailing channel axis of 1 is accepted).  Returns
    (metrics (n, 4) float64 numpy: max |Y-y|, % error at the arg-max pixel, mean |Y-y|, mean row cosine similarity;
     colmax (n, W) float32 numpy or None) -- U/valid.py:86-111 on the device, one launch for all cases."""
    if pred.dim() == 4 and pred.shape[-1] == 1:
        pred = pred[..., 0]
    if target.dim() == 4 and target.shape[-1] == 1:
        target = target[..., 0]
    _need(pred.dim() == 3 and tuple(pred.shape) == tuple(target.shape), f"eval_metrics: pred {tuple(pred.shape)} vs target {tuple(target.shape)}")
    _need(pred.dtype == torch.float32 and target.dtype == torch.float32 and pred.is_contiguous() and target.is_contiguous(),
          "eval_metrics: dense fp32 tensors")
    n, H, W = pred.shape
    out = torch.empty((n, 4), dtype=torch.float64, device=pred.device)
    col = torch.empty((n, W), dtype=torch.float32, device=pred.device) if want_colmax else None
    check(_lib.load().cfd_eval_metrics(_ptr(pred), float(scale), _ptr(target), n, H, W, _ptr(out), _ptr(col), _stream()),
          "cfd_eval_metrics")
    return out.cpu().numpy(), (col.cpu().numpy() if want_colmax else None)


# ----------------------------------------------------------------------------------------
# training-step kernels (backward, losses, optimiser)
# ----------------------------------------------------------------------------------------
from ._lib import CONV_K4S1_BWD  # noqa: E402

_TAPS = {CONV_K4S2: 16, CONV_K4S1: 16, CONV_K3S1: 9, CONVT_K4S2: 16, DENSE: 1, CONV_K4S1_BWD: 16}


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, kind: int, cout: int, dw: torch.Tensor,
               split_m: Optional[int] = None) -> torch.Tensor:
    """dw[Cout][taps][Cin] (fp32 view of the flat gradient buffer) for the FORWARD layer (kind, x -> dy)."""
    lib = _lib.load()
    if kind == DENSE:
        B, cin = x.shape
        Hi = Wi = 1
    else:
        B, Hi, Wi, cin = x.shape
    xp, ldx = _nhwc(x)
    yp, ldy = _nhwc(dy)
    if x.dtype != dy.dtype:
        raise _lib.CfdError("wgrad: x and dy must share the compute dtype")
    want = (B, cout) if kind == DENSE else conv_out_shape(kind, B, Hi, Wi, cout)
    _need(tuple(dy.shape) == tuple(want), f"wgrad dy is {tuple(dy.shape)}, kind {kind} on {tuple(x.shape)} gives {want}")
    _numel(dw, cout * _TAPS[kind] * cin, "wgrad dw", torch.float32)
    d = ConvDesc(kind=kind, dtype=_DT[x.dtype], out_dtype=_DT[x.dtype], B=B, Hi=Hi, Wi=Wi, Cin=cin, Cout=cout,
                 ldx=ldx, ldy=ldy, act=0, leaky_slope=0.0, split_k=1)
    s = int(split_m) if split_m else lib.cfd_conv_wgrad_suggest_split(C.byref(d))
    wsb = lib.cfd_conv_wgrad_workspace_bytes(C.byref(d), s) if s > 1 else 0
    ws = _workspace(wsb, x.device) if wsb > 0 else None
    if s > 1 and ws is None:
        dw.zero_()  # memset node on the stream; the kernel accumulates with atomics
    if _timer is not None:
        e0, e1 = _timer.pair()
        e0.record()
    check(lib.cfd_conv_wgrad_ws(C.byref(d), xp, yp, _ptr(dw), s, _ptr(ws), _stream()), "cfd_conv_wgrad_ws")
    if _timer is not None:
        e1.record()
        opix = (dy.numel() // cout) if kind != CONVT_K4S2 else (x.numel() // cin)
        _timer.records.append((100 + kind, 2.0 * opix * cout * _TAPS[kind] * cin, e0, e1))
    return dw


def transpose_weight(w: torch.Tensor, flip: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout][taps][Cin] -> [Cin][taps'][Cout] (taps reversed if flip): operand of the data-gradient GEMM."""
    cout, cin = w.shape[0], w.shape[-1]
    taps = w.numel() // (cout * cin)
    if out is None:
        out = torch.empty((cin,) + tuple(w.shape[1:-1]) + (cout,), dtype=w.dtype, device=w.device)
    _numel(w, cout * taps * cin, "transpose_weight in")
    _numel(out, cout * taps * cin, "transpose_weight out", w.dtype)
    check(_lib.load().cfd_transpose_weight(_ptr(w), _ptr(out), cout, taps, cin, int(flip), _DT[w.dtype], _stream()),
          "cfd_transpose_weight")
    return out


def colsum(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    xp, ldx = _nhwc(x)
    Cc = x.shape[-1]
    _numel(out, Cc, "colsum out", torch.float32)
    check(_lib.load().cfd_colsum(xp, ldx, _ptr(out), x.numel() // Cc, Cc, _DT[x.dtype], _stream()), "cfd_colsum")
    return out


def bn_act_bwd(x, mean, var, gamma, beta, g1, act1, dx, dgamma, dbeta, g2=None, act2=ACT_RELU, drop_mask=None,
               drop_rate=0.0, drop_seed=0, sample0=0, eps=1e-3, slope=LEAKY_SLOPE, sums=None):
    """`sums`: optional zeroed fp32 (2, B, C) scratch; large tensors then take the two-pass pixel-parallel path"""
    B, H, W, Cc = x.shape
    xp, ldx = _nhwc(x)
    g1p, ldg1 = _nhwc(g1)
    g2p, ldg2 = (None, 0) if g2 is None else _nhwc(g2)
    dxp, lddx = (None, 0) if dx is None else _nhwc(dx)
    if drop_mask is not None and drop_rate == 0.0:
        drop_rate = 0.5
    shp = tuple(x.shape)
    _need(tuple(g1.shape) == shp and (g2 is None or tuple(g2.shape) == shp) and (dx is None or tuple(dx.shape) == shp),
          f"bn_act_bwd: g1/g2/dx must have x's shape {shp}")
    _need(g2 is None or g2.dtype == g1.dtype, "bn_act_bwd: g1 and g2 must share a dtype")
    _need(dx is None or dx.dtype == g1.dtype, "bn_act_bwd: dx must have g1's dtype")
    _need(drop_mask is None or (tuple(drop_mask.shape) == shp and drop_mask.dtype == torch.uint8 and drop_mask.is_contiguous()),
          "bn_act_bwd: drop_mask must be dense uint8 of x's shape")
    for t, n, what in ((mean, B * Cc, "mean"), (var, B * Cc, "var"), (gamma, Cc, "gamma"), (beta, Cc, "beta"),
                       (dgamma, Cc, "dgamma"), (dbeta, Cc, "dbeta")):
        _numel(t, n, "bn_act_bwd " + what, torch.float32)
    if sums is not None and (sums.numel() < 2 * B * Cc or sums.dtype != torch.float32 or not sums.is_contiguous()):
        raise _lib.CfdError("sums must be contiguous fp32 with at least 2*B*C elements")
    check(_lib.load().cfd_bn_act_bwd_ws(xp, ldx, _DT[x.dtype], _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), g1p, ldg1,
                                        g2p, ldg2, _ptr(drop_mask), drop_rate, drop_seed, sample0, dxp, lddx,
                                        _ptr(dgamma), _ptr(dbeta), B, H * W, Cc, eps, act1, act2, slope, _DT[g1.dtype],
                                        _ptr(sums), _stream()), "cfd_bn_act_bwd_ws")
    return dx


def _dbias_ok(Cc: int) -> bool:
    """can the backward kernels carry the bias-gradient reduction for this channel count? (C / 8 divides 256)"""
    return Cc % 8 == 0 and Cc <= 2048 and 256 % (Cc // 8) == 0


def act_bwd(y, g1, act1, dx, g2=None, act2=ACT_RELU, slope=LEAKY_SLOPE, dbias=None):
    """dbias (fp32 (C,), accumulated into): the bias gradient of the layer whose pre-activation gradient is written,
    taken in the same pass (else a colsum pass over dx follows for channel counts the fused form does not take)"""
    yp, ldy = _nhwc(y)
    g1p, ldg1 = _nhwc(g1)
    g2p, ldg2 = (None, 0) if g2 is None else _nhwc(g2)
    dxp, lddx = _nhwc(dx)
    Cc = y.shape[-1]
    shp = tuple(y.shape)
    _need(tuple(g1.shape) == shp and tuple(dx.shape) == shp and (g2 is None or tuple(g2.shape) == shp),
          f"act_bwd: g1/g2/dx must have y's shape {shp}")
    _need(g1.dtype == y.dtype and dx.dtype == y.dtype and (g2 is None or g2.dtype == y.dtype), "act_bwd: one dtype")
    if dbias is not None and _dbias_ok(Cc):
        _numel(dbias, Cc, "act_bwd dbias", torch.float32)
        check(_lib.load().cfd_act_bwd_dbias(yp, ldy, g1p, ldg1, g2p, ldg2, dxp, lddx, y.numel() // Cc, Cc, act1, act2, slope,
                                            _DT[y.dtype], _ptr(dbias), _stream()), "cfd_act_bwd_dbias")
        return dx
    check(_lib.load().cfd_act_bwd(yp, ldy, g1p, ldg1, g2p, ldg2, dxp, lddx, y.numel() // Cc, Cc, act1, act2, slope,
                                  _DT[y.dtype], _stream()), "cfd_act_bwd")
    if dbias is not None:
        colsum(dx, dbias)
    return dx


def maxpool2_act_bwd(x, dy, dx, act, slope=LEAKY_SLOPE, dbias=None):
    B, H, W, Cc = x.shape
    _need(H % 2 == 0 and W % 2 == 0 and tuple(dy.shape) == (B, H // 2, W // 2, Cc) and tuple(dx.shape) == (B, H, W, Cc)
          and x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous() and dy.dtype == x.dtype
          and dx.dtype == x.dtype, "maxpool2_act_bwd: dy must be (B,H/2,W/2,C), dx (B,H,W,C), dense, x's dtype")
    if dbias is not None and _dbias_ok(Cc):
        _numel(dbias, Cc, "maxpool2_act_bwd dbias", torch.float32)
        check(_lib.load().cfd_maxpool2_act_bwd_dbias(_ptr(x), _ptr(dy), _ptr(dx), B, H, W, Cc, act, slope, _DT[x.dtype],
                                                     _ptr(dbias), _stream()), "cfd_maxpool2_act_bwd_dbias")
        return dx
    check(_lib.load().cfd_maxpool2_act_bwd(_ptr(x), _ptr(dy), _ptr(dx), B, H, W, Cc, act, slope, _DT[x.dtype],
                                           _stream()), "cfd_maxpool2_act_bwd")
    if dbias is not None:
        colsum(dx, dbias)
    return dx


def dropout_apply(x, out, rate=0.5, mask=None, seed=0, elem0=0):
    _need(x.is_contiguous() and out.is_contiguous() and out.numel() == x.numel() and out.dtype == x.dtype,
          "dropout_apply: out must match x")
    _numel(mask, x.numel(), "dropout mask", torch.uint8)
    check(_lib.load().cfd_dropout_apply(_ptr(x), _ptr(out), _ptr(mask), x.numel(), rate, seed, elem0, _DT[x.dtype],
                                        _stream()), "cfd_dropout_apply")
    return out


def bce_logits(z, loss, dz=None, target=None, target_const=1.0, grad_scale=1.0):
    _need(z.is_contiguous() and z.dtype == torch.float32, "bce_logits: z must be dense fp32")
    _numel(target, z.numel(), "bce target", torch.float32)
    _numel(dz, z.numel(), "bce dz", torch.float32)
    _numel(loss, 1, "bce loss", torch.float32)
    check(_lib.load().cfd_bce_logits(_ptr(z), _ptr(target), float(target_const), z.numel(), _ptr(loss), _ptr(dz),
                                     float(grad_scale), _stream()), "cfd_bce_logits")


MAE, MSE, HUBER = 0, 1, 2


def field_loss(y, target, kind, loss, dout=None, weight=1.0, delta=1.0, dy_extra=None, sigmoid_grad=False):
    _need(y.is_contiguous() and y.dtype == torch.float32, "field_loss: y must be dense fp32")
    _need(target.numel() == y.numel(), f"field_loss: target {tuple(target.shape)} != prediction {tuple(y.shape)}")
    _numel(target, y.numel(), "field_loss target", torch.float32)
    _numel(dout, y.numel(), "field_loss dout", torch.float32)
    _numel(dy_extra, y.numel(), "field_loss dy_extra", torch.float32)
    _numel(loss, 1, "field_loss loss", torch.float32)
    check(_lib.load().cfd_field_loss(_ptr(y), _ptr(target), y.numel(), kind, float(delta), float(weight), _ptr(loss),
                                     _ptr(dy_extra), _ptr(dout), int(sigmoid_grad), _stream()), "cfd_field_loss")


def mfma_probe(operands: torch.Tensor, out: torch.Tensor, iters: int) -> float:
    """one launch of the register-only bf16 MFMA loop (cfd_mfma_probe); returns the FLOP it performs"""
    _need(operands.dtype == torch.bfloat16 and operands.numel() >= 65536 * 8 and operands.is_contiguous(),
          "mfma_probe: operands must be >= 524288 contiguous bf16 values")
    _numel(out, 256 * 512, "mfma_probe out", torch.float32)
    check(_lib.load().cfd_mfma_probe(_ptr(operands), _ptr(out), int(iters), _stream()), "cfd_mfma_probe")
    return 2.0 * 16 * 16 * 32 * 32.0 * iters * 8 * 256


def adam_step(w, g, m, v, w_compute, alpha, beta1, beta2=0.999, eps=1e-7, grad_scale=1.0):
    n = w.numel()
    for t, what in ((w, "w"), (g, "g"), (m, "m"), (v, "v")):
        _numel(t, n, "adam " + what, torch.float32)
    _numel(w_compute, n, "adam compute copy")
    cdt = BF16 if (w_compute is not None and w_compute.dtype == torch.bfloat16) else F32
    wc = w_compute if cdt == BF16 else None
    check(_lib.load().cfd_adam_step(_ptr(w), _ptr(g), _ptr(m), _ptr(v), _ptr(wc), cdt, n, float(alpha), float(beta1),
                                    float(beta2), float(eps), float(grad_scale), _stream()), "cfd_adam_step")


def bn_update_moving(mm, mv, mean, var, momentum=0.99):
    B, Cc = mean.shape
    for t, nn in ((mm, Cc), (mv, Cc), (mean, B * Cc), (var, B * Cc)):
        _numel(t, nn, "bn_update_moving operand", torch.float32)
    check(_lib.load().cfd_bn_update_moving(_ptr(mm), _ptr(mv), _ptr(mean), _ptr(var), B, Cc, momentum, _stream()),
          "cfd_bn_update_moving")


def bn_update_moving_multi(layers, momentum=0.99):
    """layers: list of (moving_mean, moving_var, mean (B,C), var (B,C)) -- one launch for all of them (<= 16)"""
    n = len(layers)
    B = layers[0][2].shape[0]
    arr = lambda k: (C.c_void_p * n)(*[_dev(l[k]).data_ptr() for l in layers])
    cs = (C.c_int * n)(*[l[2].shape[1] for l in layers])
    for mm, mv, mean, var in layers:
        Cc = mean.shape[1]
        _need(mean.shape[0] == B and tuple(var.shape) == (B, Cc), "bn_update_moving_multi: one batch size, var like mean")
        for t, nn in ((mm, Cc), (mv, Cc), (mean, B * Cc), (var, B * Cc)):
            _numel(t, nn, "bn_update_moving operand", torch.float32)
    check(_lib.load().cfd_bn_update_moving_multi(n, arr(0), arr(1), arr(2), arr(3), cs, B, momentum, _stream()),
          "cfd_bn_update_moving_multi")


def zero_ranges(base: torch.Tensor, table: torch.Tensor, max_count: int) -> None:
    """base[off:off+cnt] = 0 for every (off, cnt) row of the int64 device `table` (validated when it was built)"""
    _need(base.dtype == torch.float32 and base.is_contiguous() and table.dtype == torch.int64 and table.dim() == 2
          and table.shape[1] == 2 and table.is_contiguous(), "zero_ranges: fp32 base, int64 (n, 2) table")
    check(_lib.load().cfd_zero_ranges(_ptr(base), _ptr(table), table.shape[0], int(max_count), _stream()), "cfd_zero_ranges")


def tap_scatter_grad(dz, dtaps, B, Hi, Wi, geom):
    # geom 0: dz is the (B,2Hi,2Wi) image gradient of a stride-2 transposed conv; geom 1: (B,Hi,Wi) of a stride-1 conv
    _numel(dz, B * Hi * Wi * (4 if geom == 0 else 1), "tap_scatter_grad dz", torch.float32)
    _numel(dtaps, B * Hi * Wi * 16, "tap_scatter_grad dtaps", torch.float32)
    check(_lib.load().cfd_tap_scatter_grad(_ptr(dz), _ptr(dtaps), B, Hi, Wi, geom, _stream()), "cfd_tap_scatter_grad")


def tap_dx(d, w, dx, ntaps):
    dxp, lddx = _nhwc(dx)
    Cc = dx.shape[-1]
    _numel(d, (dx.numel() // Cc) * ntaps, "tap_dx d", torch.float32)
    _numel(w, ntaps * Cc, "tap_dx w", dx.dtype)
    check(_lib.load().cfd_tap_dx(_ptr(d), _ptr(w), dxp, lddx, dx.numel() // Cc, Cc, ntaps, _DT[dx.dtype], _stream()),
          "cfd_tap_dx")
    return dx


def tap_dw(d, x, dw, ntaps):
    xp, ldx = _nhwc(x)
    Cc = x.shape[-1]
    _numel(d, (x.numel() // Cc) * ntaps, "tap_dw d", torch.float32)
    _numel(dw, ntaps * Cc, "tap_dw dw", torch.float32)
    check(_lib.load().cfd_tap_dw(_ptr(d), xp, ldx, _ptr(dw), x.numel() // Cc, Cc, ntaps, _DT[x.dtype], _stream()),
          "cfd_tap_dw")
    return dw


def conv_first_wgrad(x, x2, dpre, dw, db, ksize, stride):
    B, Hi, Wi, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    dp, ldd = _nhwc(dpre)
    cout = dpre.shape[-1]
    _need(x.is_contiguous() and x.dtype == torch.float32 and
          (x2 is None or (tuple(x2.shape[:3]) == (B, Hi, Wi) and x2.is_contiguous() and x2.dtype == torch.float32)),
          "conv_first_wgrad: x / x2 must be dense fp32 images of one shape")
    _need(tuple(dpre.shape) == (B, Hi // stride, Wi // stride, cout),
          f"conv_first_wgrad: dpre {tuple(dpre.shape)} does not match x {tuple(x.shape)} at stride {stride}")
    _numel(dw, cout * ksize * ksize * (c1 + c2), "conv_first_wgrad dw", torch.float32)
    _numel(db, cout, "conv_first_wgrad db", torch.float32)
    check(_lib.load().cfd_conv_first_wgrad(_ptr(x), _ptr(x2), c1, c2, dp, ldd, _ptr(dw), _ptr(db), B, Hi, Wi,
                                           dpre.shape[-1], ksize, stride, _DT[dpre.dtype], _stream()),
          "cfd_conv_first_wgrad")
