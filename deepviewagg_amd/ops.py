"""Autograd-aware Python entry points of the HIP kernels (through the C ABI, see ``_lib.py``).

Each op mirrors a primitive the reference obtains from ``torch_scatter`` or composes from torch
ops; names and argument meaning follow the reference call sites:

=========================  ==========================================================
``segment_csr``            torch_scatter.segment_csr  (pooling.py:63,289,295,628,787,807)
``gather_csr``             pooling.py:813-841
``segment_gather_csr``     pooling.py:844-856
``segment_softmax_csr``    pooling.py:758-810
``view_attention``         pooling.py:284-300 / :514-530 (softmax + weighted sum + gating)
``gather_nearest``         core/multimodal/image.py:1285 (``x[feature_map_indexing]``)
``gather_bilinear``        core/multimodal/image.py:105-170 (``sparse_interpolation``)
=========================  ==========================================================

All ops require tensors on a HIP device and raise otherwise (no CPU fallback).
"""
import os

import threading

import torch

from . import _lib
from ._lib import check, dtype_code, ptr, require_device, stream_of

# 0 = auto, 1 = generic kernels, 2 = fused wavefront-team kernels (tests flip this)
ROWS_GRAD_ALGO = 0    # 0: segmented reduction over the row plan; 1: fp32 atomics
ATTENTION_ALGO = 0


class KernelTimer:
    """Optional per-kernel timing with HIP events on the launch stream (bench.py sets
    ``ops.TIMER = KernelTimer()``): records an event pair around every C-ABI launch together with the
    algorithmic bytes of that launch; ``summary()`` synchronises once and aggregates."""

    def __init__(self, only=None):
        # ``only``: names to time; every other launch goes out without events.  An event pair costs the stream ~6 us of
        # bubble per launch on this part (rocprofv3 trace, round 5: 43 pairs = 0.27 ms of an 11.2 ms step), so a region
        # whose WALL time matters times only the kernel(s) it reports on
        self.records = []
        self.only = None if only is None else set(only)

    def launch(self, name, nbytes):
        if self.only is not None and name not in self.only:
            return _NO_TIMER
        return _TimedLaunch(self, name, nbytes)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, nbytes, e0, e1 in self.records:
            a = agg.setdefault(name, dict(launches=0, ms=0.0, bytes=0))
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["bytes"] += nbytes
        return agg


class _TimedLaunch:
    def __init__(self, timer, name, nbytes):
        self.timer, self.name, self.nbytes = timer, name, nbytes

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *exc):
        self.e1.record()
        self.timer.records.append((self.name, self.nbytes, self.e0, self.e1))
        return False


class _NoTimer:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_TIMER = _NoTimer()
TIMER = None

# ---------------------------------------------------------------------------------------------
# small zero-filled tensors (statistics accumulators, gradient arenas) without a fill launch each
# ---------------------------------------------------------------------------------------------
_ZERO_POOLS = {}
_ZERO_LOCK = threading.Lock()
_ZERO_POOL_BYTES = 32 << 20
_ZERO_SMALL_BYTES = 1 << 20


def zeros_small(shape, dtype, device):
    """``torch.zeros(shape, dtype=dtype, device=device)`` for the small accumulators of a step (fp64 statistics of a
    BatchNorm layer, the fp32 arena of a backward): a piece of a 32 MiB zero-filled pool per (device, stream) that only
    moves forward -- a piece is handed out once, so it is zero when the kernel that accumulates into it runs, and the
    pool is replaced (one fill) when it is used up.  A step of the pooling path asks for ~10 such tensors; each used to
    be its own 4-5 us fill kernel (round 5).  Large requests and requests during a HIP-graph capture (a captured fill is
    replayed, the pool's fill is not) take the plain ``torch.zeros``."""
    device = torch.device(device)
    n = 1
    for d in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)):
        n *= int(d)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    if (device.type != "cuda" or nbytes == 0 or nbytes > _ZERO_SMALL_BYTES
            or torch.cuda.is_current_stream_capturing()):
        return torch.zeros(shape, dtype=dtype, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    step = (nbytes + 255) & ~255
    with _ZERO_LOCK:        # the autograd thread and the caller's thread may both ask (never the same piece twice)
        return _zero_piece(key, step, nbytes, shape, dtype, device)


def _zero_piece(key, step, nbytes, shape, dtype, device):
    pool = _ZERO_POOLS.get(key)
    if pool is None or pool[1] + step > _ZERO_POOL_BYTES:
        # the pieces of a used-up pool are still alive when its successor is allocated: the first pool reserves the
        # second block as well (held while the first is allocated, then handed back to the caching allocator), so that
        # no hipMalloc lands in a later step (a pool lasts ~200 steps of the headline workload)
        spare = torch.empty(_ZERO_POOL_BYTES, dtype=torch.uint8, device=device) if pool is None else None
        pool = [torch.zeros(_ZERO_POOL_BYTES, dtype=torch.uint8, device=device), 0]
        del spare
        _ZERO_POOLS[key] = pool
    off = pool[1]
    pool[1] = off + step
    # NOT a view of the pool tensor: views share the base's autograd version counter, so an in-place update of one piece
    # (AccumulateGrad's `grad += new` on a parameter gradient that lives in a piece) would invalidate every other piece
    # saved for backward ("modified by an inplace operation").  A fresh tensor set_ onto the pool's storage
    # has its own counter; `off` is a multiple of 256 bytes, so it is a whole number of elements of any dtype.
    piece = torch.empty(0, dtype=dtype, device=device)
    size = tuple(shape) if isinstance(shape, (tuple, list, torch.Size)) else (int(shape),)
    return piece.set_(pool[0].untyped_storage(), off // piece.element_size(), size)


def _timed(name, nbytes):
    return TIMER.launch(name, int(nbytes)) if TIMER is not None else _NO_TIMER


def _as_2d(src):
    if src.dim() == 1:
        return src.reshape(-1, 1), True
    if src.dim() != 2:
        raise NotImplementedError("CSR ops take 1D or 2D source tensors")  # pooling.py:774-777
    return src, False


def _check_ptr(csr_idx):
    if csr_idx.dim() != 1:
        raise ValueError("CSR ops can only be computed over 1D CSR indices")  # pooling.py:771-773
    if csr_idx.dtype != torch.int64:
        raise TypeError(f"CSR pointers must be int64 (LongTensor), got {csr_idx.dtype}")
    return csr_idx.contiguous()


class _SegmentCSR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, csr_idx, reduce):
        lib = _lib.load()
        require_device(src, csr_idx)
        src = src.contiguous()
        n, C = csr_idx.shape[0] - 1, src.shape[1]
        out = torch.empty((n, C), dtype=src.dtype, device=src.device)
        arg = None
        code = _lib.REDUCE_CODE[reduce]
        if code in (_lib.DVA_MAX, _lib.DVA_MIN):
            arg = torch.empty((n, C), dtype=torch.int32, device=src.device)
        with _timed("segment_csr_fwd", (src.shape[0] + n) * C * src.element_size() + n * 8
                    + (n * C * 4 if arg is not None else 0)):
            check(lib.dva_segment_csr_fwd(ptr(src), ptr(csr_idx), ptr(out), ptr(arg), n, C,
                                          dtype_code(src), code, stream_of(src)), "dva_segment_csr_fwd")
        ctx.save_for_backward(csr_idx, arg if arg is not None else csr_idx)
        ctx.meta = (code, src.shape[0], C, arg is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        csr_idx, arg = ctx.saved_tensors
        code, M, C, has_arg = ctx.meta
        gout = gout.contiguous()
        n = csr_idx.shape[0] - 1
        # rows not covered by the pointers (none for a well-formed CSR) keep zero gradient
        gsrc = torch.zeros((M, C), dtype=gout.dtype, device=gout.device)
        with _timed("segment_csr_bwd", (M + n) * C * gout.element_size() + n * 8
                    + (n * C * 4 if has_arg else 0)):
            check(lib.dva_segment_csr_bwd(ptr(gout), ptr(csr_idx), ptr(arg) if has_arg else None,
                                          ptr(gsrc), n, C, dtype_code(gout), code, stream_of(gout)),
                  "dva_segment_csr_bwd")
        return gsrc, None, None


def segment_csr(src, csr_idx, out=None, reduce="sum"):
    """``torch_scatter.segment_csr`` for 1D/2D ``src`` reduced along dim 0. Empty groups give 0."""
    assert out is None, "out= is not supported"
    if reduce not in _lib.REDUCE_CODE:
        raise ValueError(f"Unknown reduce '{reduce}'")
    csr_idx = _check_ptr(csr_idx)
    src2, was_1d = _as_2d(src)
    res = _SegmentCSR.apply(src2, csr_idx, reduce)
    return res.reshape(-1) if was_1d else res


def segment_csr_arg(src, csr_idx, reduce="max"):
    """(values, arg) of a max/min CSR reduction; arg = winning row (int32, -1 for empty groups,
    first row on ties). Not differentiable (used for index selection, pooling.py:135-143)."""
    if reduce not in ("max", "min"):
        raise ValueError("segment_csr_arg supports 'max' and 'min'")
    lib = _lib.load()
    csr_idx = _check_ptr(csr_idx)
    src2, was_1d = _as_2d(src.detach())
    require_device(src2, csr_idx)
    src2 = src2.contiguous()
    n, C = csr_idx.shape[0] - 1, src2.shape[1]
    out = torch.empty((n, C), dtype=src2.dtype, device=src2.device)
    arg = torch.empty((n, C), dtype=torch.int32, device=src2.device)
    check(lib.dva_segment_csr_fwd(ptr(src2), ptr(csr_idx), ptr(out), ptr(arg), n, C,
                                  dtype_code(src2), _lib.REDUCE_CODE[reduce], stream_of(src2)),
          "dva_segment_csr_fwd")
    if was_1d:
        return out.reshape(-1), arg.reshape(-1)
    return out, arg


class _GatherCSR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, csr_idx, n_rows):
        lib = _lib.load()
        require_device(src, csr_idx)
        src = src.contiguous()
        n, C = csr_idx.shape[0] - 1, src.shape[1]
        out = torch.empty((n_rows, C), dtype=src.dtype, device=src.device)
        with _timed("gather_csr", (n_rows + n) * C * src.element_size() + n * 8):
            check(lib.dva_gather_csr(ptr(src), ptr(csr_idx), ptr(out), n, C, dtype_code(src),
                                     stream_of(src)), "dva_gather_csr")
        ctx.save_for_backward(csr_idx)
        return out

    @staticmethod
    def backward(ctx, gout):
        (csr_idx,) = ctx.saved_tensors
        return _SegmentCSR.apply(gout.contiguous(), csr_idx, "sum"), None, None


def num_items(csr_idx):
    """Number of rows a CSR pointer tensor covers (one host sync, like csr.py:140)."""
    return int(csr_idx[-1].item()) if csr_idx.numel() > 0 else 0


def gather_csr(src, csr_idx, n_rows=None):
    """pooling.py:813-841: redistribute group-level rows to the group's elements."""
    if not torch.is_floating_point(src):
        raise ValueError("`gather_csr` can only be computed over tensors with floating point data types.")
    csr_idx = _check_ptr(csr_idx)
    src2, was_1d = _as_2d(src)
    if n_rows is None:
        n_rows = num_items(csr_idx)
    res = _GatherCSR.apply(src2, csr_idx, n_rows)
    return res.reshape(-1) if was_1d else res


def segment_gather_csr(src, csr_idx, reduce="sum"):
    """pooling.py:844-856."""
    n_rows = src.shape[0]
    return gather_csr(segment_csr(src, csr_idx, reduce=reduce), csr_idx, n_rows=n_rows)


class _SegmentSoftmaxCSR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, csr_idx, eps, scaling):
        lib = _lib.load()
        require_device(src, csr_idx)
        src = src.contiguous()
        n, G = csr_idx.shape[0] - 1, src.shape[1]
        out = torch.zeros_like(src)
        check(lib.dva_segment_softmax_csr_fwd(ptr(src), ptr(csr_idx), ptr(out), n, G, int(scaling),
                                              float(eps), stream_of(src)),
              "dva_segment_softmax_csr_fwd")
        ctx.save_for_backward(out, csr_idx)
        ctx.scaling = int(scaling)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        out, csr_idx = ctx.saved_tensors
        gout = gout.contiguous()
        n, G = csr_idx.shape[0] - 1, out.shape[1]
        gsrc = torch.zeros_like(out)
        check(lib.dva_segment_softmax_csr_bwd(ptr(gout), ptr(out), ptr(csr_idx), ptr(gsrc), n, G,
                                              ctx.scaling, stream_of(gout)),
              "dva_segment_softmax_csr_bwd")
        return gsrc, None, None, None


def segment_softmax_csr(src, csr_idx, eps=1e-12, scaling=False):
    """pooling.py:758-810 (same error behaviour)."""
    if not torch.is_floating_point(src):
        raise ValueError(
            "`segment_csr_softmax` can only be computed over tensors with floating point data types.")
    if csr_idx.dim() != 1:
        raise ValueError("`segment_csr_softmax` can only be computed over 1D CSR indices.")
    if src.dim() > 2:
        raise NotImplementedError(
            "`segment_csr_softmax` can only be computed over 1D or 2D source tensors.")
    csr_idx = _check_ptr(csr_idx)
    src2, was_1d = _as_2d(src)
    res = _SegmentSoftmaxCSR.apply(src2.float(), csr_idx, eps, scaling).to(src.dtype)
    return res.reshape(-1) if was_1d else res


class _ViewAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, val, compat, csr_idx, gate_w, gate_b, scaling, eps):
        lib = _lib.load()
        require_device(val, compat, csr_idx, gate_w, gate_b)
        val = val.contiguous()
        compat = compat.contiguous()
        N, (V, C), G = csr_idx.shape[0] - 1, val.shape, compat.shape[1]
        gw = gate_w.detach().reshape(-1).float().contiguous() if gate_w is not None else None
        gb = gate_b.detach().reshape(-1).float().contiguous() if gate_b is not None else None
        if (gw is None) != (gb is None):
            # Gating(weight=.., bias=..) with one of them disabled: neutral element for the other
            gw = gw if gw is not None else torch.ones(G, device=val.device)
            gb = gb if gb is not None else torch.zeros(G, device=val.device)
        out = torch.empty((N, C), dtype=val.dtype, device=val.device)
        att = torch.zeros((V, G), dtype=torch.float32, device=val.device)
        gate = torch.empty((N, G), dtype=torch.float32, device=val.device)
        amax = torch.empty((N, G), dtype=torch.int32, device=val.device)
        es = val.element_size()
        with _timed("view_attention_fwd", V * (C * es + 2 * G * 4) + N * (C * es + 8 + 2 * G * 4)):
            check(lib.dva_view_attention_fwd(ptr(val), ptr(compat), ptr(csr_idx), ptr(gw), ptr(gb),
                                             ptr(out), ptr(att), ptr(gate), ptr(amax), N, V, C, G,
                                             int(scaling), float(eps), dtype_code(val), ATTENTION_ALGO,
                                             stream_of(val)), "dva_view_attention_fwd")
        ctx.save_for_backward(val, compat, csr_idx, att, gate, amax,
                              gw if gw is not None else csr_idx, gb if gb is not None else csr_idx)
        ctx.meta = (int(scaling), gw is not None,
                    None if gate_w is None else gate_w.shape, None if gate_b is None else gate_b.shape)
        ctx.mark_non_differentiable(att, gate)
        return out, att, gate

    @staticmethod
    def backward(ctx, gout, _gatt, _ggate):
        lib = _lib.load()
        val, compat, csr_idx, att, gate, amax, gw, gb = ctx.saved_tensors
        scaling, has_gate, w_shape, b_shape = ctx.meta
        gout = gout.contiguous()
        N, (V, C), G = csr_idx.shape[0] - 1, val.shape, compat.shape[1]
        gval = torch.zeros_like(val)
        gcompat = torch.zeros_like(compat)
        gwb = torch.zeros(2 * G, dtype=torch.float32, device=val.device) if has_gate else None
        es = val.element_size()
        with _timed("view_attention_bwd", V * (2 * C * es + 2 * G * 4) + N * (C * es + 8 + 3 * G * 4)):
            check(lib.dva_view_attention_bwd(ptr(gout), ptr(val), ptr(compat), ptr(att), ptr(gate),
                                             ptr(amax), ptr(csr_idx), ptr(gw) if has_gate else None,
                                             ptr(gb) if has_gate else None, ptr(gval), ptr(gcompat),
                                             ptr(gwb), N, V, C, G, scaling, dtype_code(val),
                                             ATTENTION_ALGO, stream_of(val)), "dva_view_attention_bwd")
        g_w = gwb[:G].reshape(w_shape) if (has_gate and w_shape is not None) else None
        g_b = gwb[G:].reshape(b_shape) if (has_gate and b_shape is not None) else None
        return gval, gcompat, None, g_w, g_b, None, None


def view_attention(val, compat, csr_idx, gate_w=None, gate_b=None, scaling=False, eps=1e-12):
    """Fused tail of GroupBimodalCSRPool / QKVBimodalCSRPool (pooling.py:284-300).

    :param val: [V, C] values (fp32 or bf16)
    :param compat: [V, G] compatibilities (computed in fp32)
    :param csr_idx: LongTensor [N+1]
    :param gate_w, gate_b: Gating parameters of shape [1, G] (both None -> no gating)
    :return: (x_pool [N, C], attentions [V, G], gating [N, G])
    """
    csr_idx = _check_ptr(csr_idx)
    if compat.dim() == 1:
        compat = compat.reshape(-1, 1)
    return _ViewAttention.apply(val, compat.float(), csr_idx, gate_w, gate_b, scaling, eps)


# ---------------------------------------------------------------------------------------------
# gather
# ---------------------------------------------------------------------------------------------

def pack_gather_index(images, atom_ptr, pixels, ratio=1.0):
    """8-byte packed (image, x, y) index of every atom, at feature-map resolution.

    images LongTensor [V], atom_ptr LongTensor [V+1], pixels int16/int32/int64 [P, 2] (w, h);
    ``ratio`` is the mapping -> feature-map downscale (image.py:1916-1980).
    """
    lib = _lib.load()
    require_device(images, atom_ptr, pixels)
    images = images.contiguous()
    atom_ptr = _check_ptr(atom_ptr)
    pixels = pixels.contiguous()
    if pixels.dtype not in (torch.int16, torch.int32, torch.int64):
        raise TypeError(f"pixels must be int16/int32/int64, got {pixels.dtype}")
    V, P = images.shape[0], pixels.shape[0]
    packed = torch.empty(P, dtype=torch.int64, device=pixels.device)
    check(lib.dva_pack_gather_index(ptr(images), ptr(atom_ptr), ptr(pixels), pixels.element_size(),
                                    float(ratio), V, P, ptr(packed), stream_of(pixels)),
          "dva_pack_gather_index")
    return packed


def _nhwc(x):
    """[B,C,H,W] tensor -> contiguous [B,H,W,C] storage (free if x is channels_last)."""
    return x.permute(0, 2, 3, 1).contiguous()


class _GatherNearest(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, packed):
        lib = _lib.load()
        require_device(x, packed)
        B, C, H, W = x.shape
        xl = _nhwc(x)
        P = packed.shape[0]
        out = torch.empty((P, C), dtype=x.dtype, device=x.device)
        es = x.element_size()
        # P*(g*C*s + idx) + P*C*s with g = 1
        with _timed("gather_nearest_fwd", P * (2 * C * es + 8)):
            check(lib.dva_gather_nearest_fwd(ptr(xl), ptr(packed), ptr(out), P, B, H, W, C,
                                             dtype_code(x), stream_of(x)), "dva_gather_nearest_fwd")
        ctx.save_for_backward(packed)
        ctx.meta = (B, C, H, W, x.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        (packed,) = ctx.saved_tensors
        B, C, H, W, dt = ctx.meta
        gout = gout.contiguous()
        P = packed.shape[0]
        if ROWS_GRAD_ALGO == 0:
            # deterministic segmented reduction over the row plan (sorted atoms) instead of fp32 atomics
            row_idx, _, (perm, row_ptr) = gather_row_index(packed, B, H, W, with_counts=False, with_plan=True, split=False)
            gx = torch.empty((B, H, W, C), dtype=torch.float32, device=gout.device)
            with _timed("gather_nearest_bwd", P * (C * gout.element_size() + 12) + B * H * W * C * 4):
                check(lib.dva_gather_rows_sum(ptr(gout), ptr(perm), ptr(row_ptr), None, 0, ptr(gx), B * H * W, P, C,
                                              dtype_code(gout), stream_of(gout)), "dva_gather_rows_sum")
            return gx.permute(0, 3, 1, 2).to(dt), None
        gx = torch.zeros((B, H, W, C), dtype=torch.float32, device=gout.device)
        # P*(C*s + idx) + P*g*C*4*2 (read-modify-write of the fp32 gradient map)
        with _timed("gather_nearest_bwd", P * (C * gout.element_size() + 8) + P * C * 4 * 2):
            check(lib.dva_gather_nearest_bwd(ptr(gout), ptr(packed), ptr(gx), P, B, H, W,
                                             C, dtype_code(gout), stream_of(gout)),
                  "dva_gather_nearest_bwd")
        return gx.permute(0, 3, 1, 2).to(dt), None


def gather_nearest(x, packed_idx):
    """``x[(batch, ..., h, w)]`` on a [B,C,H,W] map (image.py:1285) -> [P, C]."""
    assert x.dim() == 4
    return _GatherNearest.apply(x, packed_idx)


def _anchor_ok(C, dtype):
    vec = 4 if dtype == torch.float32 else 8
    lpr = C // vec
    return C % vec == 0 and lpr > 0 and (lpr & (lpr - 1)) == 0 and lpr <= 64


ANCHOR_WS_BYTES = None     # test hook: workspace budget of bilinear_scatter in bytes (None: from the free device memory)


def _anchor_chunk_images(B, bytes_per_image, device):
    """Images per pass of ``bilinear_scatter`` so that the per-anchor sums stay within the workspace budget."""
    budget = ANCHOR_WS_BYTES
    if budget is None:
        if B * bytes_per_image <= (1 << 30):
            return B                                   # small: no driver query on the common path
        free, _ = torch.cuda.mem_get_info(device)
        budget = max(min(free // 4, 8 << 30), 256 << 20)
    return max(1, min(B, budget // max(bytes_per_image, 1)))


def _geometry(B, H=None, W=None):
    """[(B, H, W), ...] of the feature maps behind a bilinear gather: one triple, or one per SETTING of a multi-setting
    batch (``InterpolatedFeatures.cat``: map rows and anchors of the settings are numbered one after the other)."""
    return [(int(B), int(H), int(W))] if H is not None else [tuple(int(v) for v in g) for g in B]


def n_anchors(geometry):
    """Anchors of a geometry, WITHOUT the dummy anchor (which is the next id)."""
    return sum(b * (h + 1) * (w + 1) for b, h, w in geometry)


def anchor_plan(anchors, B, H=None, W=None):
    """The row plan over the ANCHORS of the views (``dva_gather_bilinear_taps_anchor``): ``(perm, row_ptr)`` with
    sum B (H + 1) (W + 1) + 1 anchors (the last one = views without the 2 x 2 tap structure).  ``B`` may be a list of
    (B, H, W) triples (several settings)."""
    return row_plan(anchors, n_anchors(_geometry(B, H, W)) + 1, with_counts=False, split=False)[0]


def bilinear_scatter(grad, tap_rows, tap_weights, anchors, B, H=None, W=None, bn_backward=None, plan=None):
    """Transpose of the bilinear gather: fp32 [B*H*W, C] = sum over the views and their 4 taps of weight x grad row.
    Views are grouped by ANCHOR (the padded cell of their top-left tap, ``dva_gather_bilinear_taps_anchor``): one
    sort of P keys, every gradient row read once into four per-anchor sums, then a 2 x 2 stencil on the map
    (csrc/attention.hip anchor_rows_sum_kernel, csrc/gather.hip anchor_combine_kernel).  Deterministic.
    ``bn_backward = (z_a, bn_a, sm_a, Y)`` (fused bilinear path): ``grad`` is dy_a (bf16, position order) and the
    BatchNorm_a backward ``G dy_a - K1 - K2 z_a`` is folded into the scatter instead of a pass of its own: with ``Y``
    (the interpolated map rows, position order) at the level of the anchor through the Gram matrix of its tap weights
    (one random row per view), without it from the stored ``z_a`` rows; the dummy-anchor views always use ``z_a``."""
    lib = _lib.load()
    grad = grad.contiguous()
    P, C = grad.shape
    geometry = _geometry(B, H, W)                   # several settings: rows and anchors numbered setting after setting
    n_rows = sum(b * h * w for b, h, w in geometry)
    dummy = n_anchors(geometry)                     # the anchor of views without the 2 x 2 structure
    perm, row_ptr = plan if plan is not None else anchor_plan(anchors, geometry)
    st = stream_of(grad)
    es = grad.element_size()
    if bn_backward is not None:
        z_a, bn_a, sm_a, Y = bn_backward
        if grad.dtype != torch.bfloat16 or z_a.dtype != torch.bfloat16 or z_a.shape != grad.shape or C % 32:
            raise _lib.DvaError("bilinear_scatter(bn_backward): bf16 [V, C] rows, C a multiple of 32", -1)
        z_a = z_a.contiguous()
        if Y is not None:
            if Y.dtype != torch.bfloat16 or Y.shape[1] != C:
                raise _lib.DvaError("bilinear_scatter(bn_backward): Y bf16 [R, C]", -1)
            Y = Y.contiguous()
    two_rows = bn_backward is not None and bn_backward[3] is None
    # The per-anchor sums S [anchors, 4, C] fp32 are 4 x the map gradient: bounded workspace, the images
    # go through in chunks when it would exceed the budget (anchors are image-major, so a chunk of images is a
    # contiguous range of the plan; the dummy anchor's views are added tap by tap by the fix-up below)
    out = torch.empty((n_rows, C), dtype=torch.float32, device=grad.device)
    chunks = [_anchor_chunk_images(b, (h + 1) * (w + 1) * 4 * C * 4, grad.device) for b, h, w in geometry]
    S = torch.empty((max(min(i, b) * (h + 1) * (w + 1) for i, (b, h, w) in zip(chunks, geometry)) + 1, 4, C),
                    dtype=torch.float32, device=grad.device)
    n_views_anchored = max(dummy, 1)
    a_off = r_off = 0
    for imgs, (Bs, Hs, Ws) in zip(chunks, geometry):
        per_image = (Hs + 1) * (Ws + 1)
        for b0 in range(0, Bs, imgs):
            nb = min(imgs, Bs - b0)
            a0, na = a_off + b0 * per_image, nb * per_image
            rp = row_ptr[a0:a0 + na + 1]
            with _timed("bilinear_anchor_sum",
                        P * (C * es * (2 if two_rows else 1) + 20) * na // n_views_anchored + na * 4 * C * 4):
                if bn_backward is None:
                    check(lib.dva_anchor_rows_sum(ptr(grad), ptr(perm), ptr(rp), ptr(tap_weights), ptr(S), na, P, C,
                                                  dtype_code(grad), st), "dva_anchor_rows_sum")
                else:
                    check(lib.dva_anchor_rows_sum_bn(ptr(grad), None if Y is not None else ptr(z_a), ptr(bn_a),
                                                     ptr(sm_a), ptr(perm), ptr(rp), ptr(tap_weights),
                                                     ptr(tap_rows) if Y is not None else None, ptr(Y), ptr(S), na, P, C,
                                                     st), "dva_anchor_rows_sum_bn")
            with _timed("bilinear_anchor_combine", na * 4 * C * 4 + nb * Hs * Ws * C * 4):
                check(lib.dva_anchor_combine(ptr(S), ptr(out[r_off + b0 * Hs * Ws:]), nb, Hs, Ws, C, st),
                      "dva_anchor_combine")
        a_off += Bs * per_image
        r_off += Bs * Hs * Ws
    # views of the dummy anchor, tap by tap: the entries only take the dummy id from (B, H, W) = B (H + 1) (W + 1)
    if bn_backward is None:
        check(lib.dva_anchor_fixup(ptr(grad), ptr(tap_rows), ptr(tap_weights), ptr(anchors), ptr(out), P, dummy, 0, 0, C,
                                   dtype_code(grad), st), "dva_anchor_fixup")
    else:
        check(lib.dva_anchor_fixup_bn(ptr(grad), ptr(z_a), ptr(bn_a), ptr(sm_a), ptr(tap_rows), ptr(tap_weights),
                                      ptr(anchors), ptr(out), P, dummy, 0, 0, C, st), "dva_anchor_fixup_bn")
    return out


class _GatherBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, packed, coords):
        lib = _lib.load()
        require_device(x, packed, coords)
        B, C, H, W = x.shape
        xl = _nhwc(x)
        coords = coords.float().contiguous()
        P = packed.shape[0]
        out = torch.empty((P, C), dtype=x.dtype, device=x.device)
        check(lib.dva_gather_bilinear_fwd(ptr(xl), ptr(packed), ptr(coords), ptr(out), P, B, H, W, C,
                                          dtype_code(x), stream_of(x)), "dva_gather_bilinear_fwd")
        ctx.save_for_backward(packed, coords)
        ctx.meta = (B, C, H, W, x.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        packed, coords = ctx.saved_tensors
        B, C, H, W, dt = ctx.meta
        gout = gout.contiguous()
        P = packed.shape[0]
        if ROWS_GRAD_ALGO == 0 and 4 * P < 2 ** 31:
            # the 4 corner taps of every atom, grouped by map row (row plan), then a weighted segmented
            # reduction: deterministic, no fp32 atomics
            rows4 = torch.empty(4 * P, dtype=torch.int32, device=gout.device)
            w4 = torch.empty(4 * P, dtype=torch.float32, device=gout.device)
            anchors = torch.empty(P, dtype=torch.int32, device=gout.device) if _anchor_ok(C, gout.dtype) else None
            check(lib.dva_gather_bilinear_taps_anchor(ptr(packed), ptr(coords), P, B, H, W, ptr(rows4), ptr(w4),
                                                      ptr(anchors), stream_of(gout)), "dva_gather_bilinear_taps_anchor")
            if anchors is not None:
                # grouped by anchor instead: P keys to sort, every gradient row read once (bilinear_scatter)
                gx = bilinear_scatter(gout, rows4, w4, anchors, B, H, W).view(B, H, W, C)
                return gx.permute(0, 3, 1, 2).to(dt), None, None
            (perm, row_ptr), _ = row_plan(rows4, B * H * W, with_counts=False, split=False)
            gx = torch.empty((B, H, W, C), dtype=torch.float32, device=gout.device)
            with _timed("gather_bilinear_bwd", 4 * P * (C * gout.element_size() + 12) + B * H * W * C * 4):
                check(lib.dva_gather_rows_sum(ptr(gout), ptr(perm), ptr(row_ptr), ptr(w4), 2, ptr(gx), B * H * W,
                                              4 * P, C, dtype_code(gout), stream_of(gout)), "dva_gather_rows_sum")
            return gx.permute(0, 3, 1, 2).to(dt), None, None
        gx = torch.zeros((B, H, W, C), dtype=torch.float32, device=gout.device)
        check(lib.dva_gather_bilinear_bwd(ptr(gout), ptr(packed), ptr(coords), ptr(gx),
                                          packed.shape[0], B, H, W, C, dtype_code(gout),
                                          stream_of(gout)), "dva_gather_bilinear_bwd")
        return gx.permute(0, 3, 1, 2).to(dt), None, None


def gather_bilinear(x, packed_idx, coords):
    """``sparse_interpolation(x, coords, batch)`` with border padding (image.py:105-170)."""
    assert x.dim() == 4
    assert coords.dim() == 2 and coords.shape[1] == 2
    return _GatherBilinear.apply(x, packed_idx, coords)


# ---------------------------------------------------------------------------------------------
# lazy view gather + fused gather-attention  (DESIGN.md "E_mod hoisting")
# ---------------------------------------------------------------------------------------------

# The split plan (round 5, csrc/plan_split.hip): from 1 - 2 M views on (and 512 < rows <= 2^18) ``row_plan`` builds only the
# offset tables of a two-pass radix partition -- row_ptr and counts come out of them -- and the backward runs the two
# scatter passes on the 16-byte view records themselves, so the rows gradient streams its records in plan order.
# DVA_SPLIT_PLAN=0: the permutation plan everywhere (the A/B).
SPLIT_PLAN = os.environ.get("DVA_SPLIT_PLAN", "1") == "1"
# measured crossover (tools/split_threshold.py, profiles/r05_split_threshold.json: plan + rows gradient at C = 64): the split
# form wins from 2 M views on for any row count (0.13 against 0.17 - 0.23 ms) and from 1 M views on maps of >= 2^17 rows
# (0.10 against 0.19 ms); at 0.5 M views its eight launches cost what the records' lines do
SPLIT_PLAN_MIN_VIEWS = 1 << 20


def split_plan_serves(dtype, C):
    """Does a consumer of [R, C] value rows of this dtype take its rows gradient over a SplitPlan?  bf16: the 16-byte
    records of the chain (any C: bucket kernel at C in {32, 64}, the two record passes otherwise); fp32: the 32-byte
    records of the lean attention backward at C in {32, 64} (round 6).  Everything else wants the permutation."""
    return dtype == torch.bfloat16 or (dtype == torch.float32 and C in (32, 64))


def _split_plan_pays(n_views, n_rows):
    return n_views >= SPLIT_PLAN_MIN_VIEWS and (n_views >= 2 * SPLIT_PLAN_MIN_VIEWS or n_rows >= (1 << 17))

# DVA_SPLIT_FUSED=0: pass B + the segmented reduction of attention.hip instead of the bucket kernel (the A/B; also the
# form whose sums equal the permutation plan's bit for bit)
SPLIT_FUSED = os.environ.get("DVA_SPLIT_FUSED", "1") == "1"


def _legacy_row_plan(row_idx, n_rows, with_counts):
    lib = _lib.load()
    V, dev = row_idx.shape[0], row_idx.device
    perm = torch.empty(V, dtype=torch.int32, device=dev)
    row_ptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
    counts = torch.empty(n_rows, dtype=torch.int32, device=dev) if with_counts else None
    nbytes = lib.dva_row_plan_workspace_bytes(V, n_rows)
    if nbytes < 0:
        raise _lib.DvaError("dva_row_plan_workspace_bytes", int(nbytes))
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    with _timed("row_plan", V * 16):
        check(lib.dva_row_plan(ptr(row_idx), V, n_rows, ptr(perm), ptr(row_ptr), ptr(counts), ptr(ws),
                               int(nbytes), stream_of(row_idx)), "dva_row_plan")
    return (perm, row_ptr), counts


class SplitPlan:
    """The backward of the row gather (reference ``core/multimodal/image.py:1262-1287``: ``x[idx]`` in the forward, an
    ``index_add`` of the views' gradients in the backward) for large mappings.
    A row plan without its permutation: ``row_ptr`` and the offset tables of the two scatter passes
    (``dva_plan_split_build``).  ``sort_records`` brings the 16-byte view records of a backward into plan order;
    callers that want ``(perm, row_ptr)`` (unpacking, ``plan[0]``) get the permutation from ``dva_row_plan`` on first
    use -- correct, but it pays the sort the split plan exists to avoid."""

    def __init__(self, row_idx, n_rows, row_ptr, tables):
        self.row_idx, self.n_rows, self.row_ptr, self.tables = row_idx, int(n_rows), row_ptr, tables
        self._perm = None

    LAZY_PERMS = 0      # how often a consumer asked a split plan for its permutation (each one = a full dva_row_plan sort)

    @property
    def perm(self):
        if self._perm is None:
            if SplitPlan.LAZY_PERMS == 0:
                import warnings
                warnings.warn("SplitPlan.perm: a consumer wants (perm, row_ptr) from a split plan -- the permutation is "
                              "being built by a full dva_row_plan sort on top of the split build (correct, but slower "
                              "than asking row_plan(..., split=False) up front); counted in ops.SplitPlan.LAZY_PERMS",
                              RuntimeWarning, stacklevel=3)
            SplitPlan.LAZY_PERMS += 1
            self._perm = _legacy_row_plan(self.row_idx, self.n_rows, False)[0][0]
        return self._perm

    def __iter__(self):
        yield self.perm
        yield self.row_ptr

    def __getitem__(self, i):
        return (self.perm, self.row_ptr)[i] if i in (0, -2) else (None, self.row_ptr)[i]

    def __len__(self):
        return 2

    def sort_records(self, rec, keyed=False, bucket_order=False):
        """rec int32 [V, 4] in view order -> the same storage in plan order (word 3 of a record = its row key).
        ``keyed``: word 3 already holds the row key (the records of ``dva_chain_attn_bwd``).  ``bucket_order``: pass A
        only -- a NEW tensor with the records in bucket order (what ``rows_grad_fused`` consumes)."""
        lib = _lib.load()
        V = self.row_idx.shape[0]
        assert rec.shape == (V, 4) and rec.dtype == torch.int32 and rec.is_contiguous()
        buf = torch.empty_like(rec)
        with _timed("plan_sort_records", V * ((32 if bucket_order else 64) + (0 if keyed else 4))):
            check(lib.dva_plan_split_sort_records(None if keyed else ptr(self.row_idx), ptr(rec), V, self.n_rows,
                                                  ptr(self.row_ptr),
                                                  ptr(self.tables), self.tables.numel(), ptr(buf),
                                                  None if bucket_order else ptr(rec), stream_of(rec)),
                  "dva_plan_split_sort_records")
        return buf if bucket_order else rec

    def rows_grad_f32(self, gout, rec, C, G, stream):
        """fp32 [R, C] rows gradient from the 32-byte view records of ``dva_chain_attn_bwd_f32`` (``rec`` fp32 [V, 8], view
        order) through pass A on those records + the fp32 bucket kernel (round 6), or None where it does not apply."""
        lib = _lib.load()
        V, R = self.row_idx.shape[0], self.n_rows
        if (gout.dtype != torch.float32 or C not in (32, 64) or G not in (1, 2, 4) or (C // 4) % G
                or gout.data_ptr() % 16 or not gout.is_contiguous() or os.environ.get("DVA_PLAN_TILE", "4096") != "4096"):
            return None
        assert rec.shape == (V, 8) and rec.dtype == torch.float32 and rec.is_contiguous()
        brec = torch.empty_like(rec)
        with _timed("plan_sort_records", V * (64 + 4)):
            check(lib.dva_plan_split_sort_records32(ptr(self.row_idx), ptr(rec), V, R, ptr(self.tables),
                                                    self.tables.numel(), ptr(brec), stream), "dva_plan_split_sort_records32")
        g = torch.empty((R, C), dtype=torch.float32, device=gout.device)
        with _timed("view_gather_rows_grad", V * (32 + C * 4) + R * C * 4):
            check(lib.dva_plan_split_rows_grad(ptr(gout), ptr(brec), V, R, ptr(self.tables), self.tables.numel(), ptr(g),
                                               C, G, _lib.DVA_F32, _lib.DVA_F32, stream), "dva_plan_split_rows_grad")
        return g

    def rows_grad_fused(self, gout, rec, C, G, stream):
        """bf16 [R, C] rows gradient from KEYED view-order records through pass A + the bucket kernel
        (``dva_plan_split_rows_grad``: no pass B, no plan-order records), or None where that kernel does not apply."""
        lib = _lib.load()
        V, R = self.row_idx.shape[0], self.n_rows
        # everything dva_plan_split_rows_grad refuses (DVA_ERR_UNSUPPORTED: dtype, C, G, 16-byte alignment of the
        # gradient rows; the records and the output are fresh allocations) is decided here, BEFORE pass A is launched
        if (gout.dtype != torch.bfloat16 or C not in (32, 64) or G not in (1, 2, 4) or (C // 8) % G
                or gout.data_ptr() % 16 or not gout.is_contiguous()):
            return None
        brec = self.sort_records(rec, keyed=True, bucket_order=True)
        g = torch.empty((R, C), dtype=torch.bfloat16, device=gout.device)
        with _timed("view_gather_rows_grad", V * (16 + C * 2) + R * C * 2):
            check(lib.dva_plan_split_rows_grad(ptr(gout), ptr(brec), V, R, ptr(self.tables), self.tables.numel(), ptr(g),
                                               C, G, _lib.DVA_BF16, _lib.DVA_BF16, stream), "dva_plan_split_rows_grad")
        return g


def row_plan(row_idx, n_rows, with_counts=True, split=True):
    """Views grouped by the feature-map row they read: ``(perm, row_ptr)`` int32 (+ ``counts`` int32
    [n_rows]).  Stable, so the order of the views inside a row (and with it every sum over them) is
    deterministic.  Large plans come back as a ``SplitPlan`` (same ``row_ptr`` / ``counts``, no permutation) unless the
    caller knows its consumer wants the permutation (``split=False``: fp32 maps, several atoms per view)."""
    lib = _lib.load()
    require_device(row_idx)
    V, dev = row_idx.shape[0], row_idx.device
    if split and SPLIT_PLAN and _split_plan_pays(V, n_rows) and row_idx.is_contiguous():
        nbytes = int(lib.dva_plan_split_table_bytes(V, n_rows))
        if nbytes > 0:
            row_ptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
            counts = torch.empty(n_rows, dtype=torch.int32, device=dev) if with_counts else None
            tables = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            lows = torch.empty(V, dtype=torch.int16, device=dev)
            with _timed("row_plan", V * 10):      # keys read twice, 2-byte digits written + read
                check(lib.dva_plan_split_build(ptr(row_idx), V, n_rows, ptr(row_ptr), ptr(counts), ptr(tables), nbytes,
                                               ptr(lows), V * 2, stream_of(row_idx)), "dva_plan_split_build")
            return SplitPlan(row_idx, n_rows, row_ptr, tables), counts
    return _legacy_row_plan(row_idx, n_rows, with_counts)


def rows_grad_rec16(gout, plan, rec, R, C, G, out_dtype, stream):
    """Rows gradient from the 16-byte view records of ``dva_chain_attn_bwd`` (``rec`` int32 [V, 4], view order; consumed):
    over the permutation plan, or -- ``SplitPlan`` -- after the records themselves went through the plan's two passes."""
    lib = _lib.load()
    V = rec.shape[0]
    if isinstance(plan, SplitPlan):
        if SPLIT_FUSED and out_dtype == torch.bfloat16:
            fused = plan.rows_grad_fused(gout, rec, C, G, stream)
            if fused is not None:
                return fused
        rec = plan.sort_records(rec, keyed=True)
        perm, row_ptr = None, plan.row_ptr
    else:
        perm, row_ptr = plan
    g = torch.empty((R, C), dtype=out_dtype, device=gout.device)
    with _timed("view_gather_rows_grad", V * ((4 if perm is not None else 0) + 16 + C * 2) + R * (C * 2 + 4)):
        check(lib.dva_view_gather_rows_grad_rec16_to(ptr(gout), ptr(perm), ptr(row_ptr), ptr(rec), ptr(g), dtype_code(g),
                                                     R, V, C, G, _lib.DVA_BF16, stream),
              "dva_view_gather_rows_grad_rec16_to")
    return g


def csr_expand(csr_idx, n_views):
    """int32 [V]: the point (segment) of every view -- the dense form of the CSR pointers."""
    lib = _lib.load()
    require_device(csr_idx)
    vp = torch.empty(n_views, dtype=torch.int32, device=csr_idx.device)
    check(lib.dva_csr_expand(ptr(csr_idx), csr_idx.shape[0] - 1, ptr(vp), stream_of(csr_idx)),
          "dva_csr_expand")
    return vp


def gather_row_index(packed_idx, B, H, W, row_offset=0, with_counts=True, with_plan=False, split=True):
    """Flat row index of every atom into the [B*H*W, C] view of a channels-last map, and the number
    of atoms per row (int32 [B*H*W]) if ``with_counts``.  ``with_plan``: also return the row plan
    (``row_plan``); the counts then come from the plan instead of a histogram with atomics."""
    lib = _lib.load()
    require_device(packed_idx)
    P = packed_idx.shape[0]
    row_idx = torch.empty(P, dtype=torch.int32, device=packed_idx.device)
    hist = with_counts and not with_plan
    counts = torch.zeros(B * H * W, dtype=torch.int32, device=packed_idx.device) if hist else None
    with _timed("gather_row_index", P * 12):
        check(lib.dva_gather_row_index(ptr(packed_idx), P, B, H, W, int(row_offset), ptr(row_idx),
                                       ptr(counts), stream_of(packed_idx)), "dva_gather_row_index")
    if with_plan:
        assert row_offset == 0, "a plan is built over the rows of one map"
        plan, counts = row_plan(row_idx, B * H * W, with_counts, split)
        return row_idx, counts, plan
    return row_idx, counts


def mapping_row_index(images, atom_ptr, pixels, ratio, B, H, W, with_counts=True, with_plan=True, split=True):
    """``gather_row_index(pack_gather_index(images, atom_ptr, pixels, ratio), B, H, W)`` in one pass (no packed index):
    (row_idx, counts, plan) with the counts taken from the row plan."""
    lib = _lib.load()
    require_device(images, atom_ptr, pixels)
    images = images.contiguous()
    atom_ptr = _check_ptr(atom_ptr)
    pixels = pixels.contiguous()
    if pixels.dtype not in (torch.int16, torch.int32, torch.int64):
        raise TypeError(f"pixels must be int16/int32/int64, got {pixels.dtype}")
    V, P = images.shape[0], pixels.shape[0]
    row_idx = torch.empty(P, dtype=torch.int32, device=pixels.device)
    with _timed("gather_row_index", V * 28 + P * 8):
        check(lib.dva_mapping_row_index(ptr(images), ptr(atom_ptr), ptr(pixels), pixels.element_size(), float(ratio),
                                        V, P, B, H, W, ptr(row_idx), stream_of(pixels)), "dva_mapping_row_index")
    if not with_plan:
        return row_idx, None, None
    plan, counts = row_plan(row_idx, B * H * W, with_counts, split)
    return row_idx, counts, plan


class GatheredFeatures:
    """Result of a NEAREST view gather that has not been materialised: ``x_mod[p] = rows[row_idx[p]]``.

    ``rows`` is the [R, C] row view of the channels-last feature map(s) (a differentiable function of
    the 2D encoder output), ``row_idx`` int32 [P], ``counts`` int32 [R] = atoms per row.  Row-wise
    modules (Linear / BatchNorm / activation) commute with the gather, so consumers that understand
    this type apply them to ``rows`` (R << P) and hand ``row_idx`` to the fused gather-attention
    kernel; everything else calls ``materialize()`` and gets the reference's [P, C] tensor.
    ``exact`` tells that every view owns exactly one atom (P == V): the atomic pool is the identity.
    """

    def __init__(self, rows, row_idx, counts, exact, plan=None):
        self.rows, self.row_idx, self.counts, self.exact = rows, row_idx, counts, exact
        self.plan = plan    # (perm, row_ptr) of row_idx, or None: built on demand in backward

    def with_rows(self, rows):
        """Same gather applied to another [R, C'] row tensor (e.g. E_mod(rows))."""
        assert rows.shape[0] == self.rows.shape[0]
        plan = self.plan
        if isinstance(plan, SplitPlan) and not split_plan_serves(rows.dtype, rows.shape[1]):
            plan = None      # the consumer of these rows wants a permutation: built in its backward
        return GatheredFeatures(rows, self.row_idx, self.counts, self.exact, plan)

    @property
    def shape(self):
        return torch.Size((self.row_idx.shape[0], self.rows.shape[1]))

    @property
    def device(self):
        return self.rows.device

    @property
    def dtype(self):
        return self.rows.dtype

    def dim(self):
        return 2

    def materialize(self):
        return gather_rows(self.rows, self.row_idx)

    @staticmethod
    def cat(items, order=None):
        """Concatenate lazily gathered features of several settings (rows are stacked, indices
        offset) and optionally permute the atoms (UnimodalBranch view_cat_sorting)."""
        offs, rows, idx, cnt = 0, [], [], []
        for it in items:
            rows.append(it.rows)
            idx.append(it.row_idx + offs)
            cnt.append(it.counts)
            offs += it.rows.shape[0]
        if len(items) == 1 and order is None:
            return items[0]
        row_idx = torch.cat(idx)
        if order is not None:
            row_idx = row_idx[order]
        return GatheredFeatures(torch.cat(rows, dim=0), row_idx.contiguous(), torch.cat(cnt),
                                all(it.exact for it in items))


def lazy_gather_nearest(x, packed_idx, exact):
    """Lazy counterpart of ``gather_nearest``: no [P, C] tensor is produced."""
    assert x.dim() == 4
    B, C, H, W = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C)   # view when x is channels_last
    # the split plan serves the 16-byte-record rows gradient: bf16 maps, one atom per view
    row_idx, counts, plan = gather_row_index(packed_idx, B, H, W, with_plan=True,
                                             split=bool(exact) and split_plan_serves(x.dtype, C))
    return GatheredFeatures(rows, row_idx, counts, exact, plan)


def lazy_gather_nearest_mapping(x, images, atom_ptr, pixels, ratio, exact):
    """``lazy_gather_nearest(x, pack_gather_index(images, atom_ptr, pixels, ratio), exact)`` without the packed index."""
    assert x.dim() == 4
    B, C, H, W = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C)   # view when x is channels_last
    row_idx, counts, plan = mapping_row_index(images, atom_ptr, pixels, ratio, B, H, W,
                                              split=bool(exact) and split_plan_serves(x.dtype, C))
    return GatheredFeatures(rows, row_idx, counts, exact, plan)


class _GatherSegmentMax(torch.autograd.Function):
    """``segment_csr(rows[row_idx], atom_ptr, 'max')`` without the [P, C] tensor (csrc/gather.hip).  ``plan`` = the row plan
    of the atoms (``row_plan(row_idx, R)[0]``) or None: built in backward."""

    @staticmethod
    def forward(ctx, rows, row_idx, atom_ptr, plan):
        lib = _lib.load()
        require_device(rows, row_idx, atom_ptr)
        rows = rows.contiguous()
        (R, C), V, P = rows.shape, atom_ptr.shape[0] - 1, row_idx.shape[0]
        out = torch.empty((V, C), dtype=rows.dtype, device=rows.device)
        arg = torch.empty((V, C), dtype=torch.int16, device=rows.device)
        es = rows.element_size()
        with _timed("gather_segment_max_fwd", P * (4 + C * es) + V * (8 + C * (es + 2))):
            check(lib.dva_gather_segment_max_fwd(ptr(rows), ptr(row_idx), ptr(atom_ptr), ptr(out), ptr(arg), V, P, R, C,
                                                 dtype_code(rows), stream_of(rows)), "dva_gather_segment_max_fwd")
        ctx.save_for_backward(row_idx, atom_ptr, arg)
        ctx.meta = (R, C, rows.dtype)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        row_idx, atom_ptr, arg = ctx.saved_tensors
        R, C, dt = ctx.meta
        gout = gout.contiguous().to(dt)
        V, P = atom_ptr.shape[0] - 1, row_idx.shape[0]
        es = gout.element_size()
        groups = C // (16 // es)
        if SEGMENT_MAX_ATOMICS or groups & (groups - 1) or groups > 64 or P > 0x7fffffff:
            grows = torch.zeros((R, C), dtype=torch.float32, device=gout.device)
            with _timed("gather_segment_max_bwd", V * (8 + C * (es + 2 + 4)) + R * C * 4):
                check(lib.dva_gather_segment_max_bwd(ptr(gout), ptr(arg), ptr(row_idx), ptr(atom_ptr), None, None, None,
                                                     ptr(grows), V, P, R, C, dtype_code(gout), stream_of(gout)),
                      "dva_gather_segment_max_bwd")
            return grows.to(dt), None, None, None
        perm, row_ptr = ctx.plan if ctx.plan is not None else row_plan(row_idx, R, with_counts=False, split=False)[0]
        voa = csr_expand(atom_ptr, P)
        grows = torch.empty((R, C), dtype=torch.float32, device=gout.device)
        with _timed("gather_segment_max_bwd", P * (16 + C * (es + 2)) + R * (C * 4 + 4)):
            check(lib.dva_gather_segment_max_bwd(ptr(gout), ptr(arg), ptr(row_idx), ptr(atom_ptr), ptr(perm), ptr(row_ptr),
                                                 ptr(voa), ptr(grows), V, P, R, C, dtype_code(gout), stream_of(gout)),
                  "dva_gather_segment_max_bwd")
        return grows.to(dt), None, None, None


# A/B: fp32 atomics instead of the segmented reduction over the atoms' row plan (tests)
SEGMENT_MAX_ATOMICS = False


def _atoms_per_view_ok(atom_ptr):
    """True when no view owns more than 65534 atoms (the uint16 arg offsets of the fused max pool).  One device
    synchronisation per mapping tensor; the verdict is remembered ON the tensor object together with its version
    counter (not under its address: the caching allocator hands the same address to the next mapping), so a
    new or modified pointer tensor is always measured again."""
    memo = getattr(atom_ptr, "_dva_atoms_ok", None)
    if memo is not None and memo[0] == atom_ptr._version:
        return memo[1]
    n = atom_ptr.shape[0] - 1
    ok = n == 0 or int((atom_ptr[1:] - atom_ptr[:-1]).max()) <= 0xfffe
    try:
        atom_ptr._dva_atoms_ok = (atom_ptr._version, ok)
    except AttributeError:       # a tensor subclass without a __dict__: measure every time
        pass
    return ok


LAZY_NONEXACT = os.environ.get("DVA_LAZY_NONEXACT", "1") == "1"


def gather_segment_max_applicable(x_mod, atom_ptr):
    if not (LAZY_NONEXACT and isinstance(x_mod, GatheredFeatures)):
        return False
    C, es = x_mod.rows.shape[1], x_mod.rows.element_size()
    # 16-byte alignment of the row base: a contiguous slice of a feature map with a storage offset that is not a
    # multiple of 16 bytes takes the materialised route instead of DVA_ERR_UNSUPPORTED; non-contiguous
    # rows are copied by the op (fresh, aligned allocation)
    return (x_mod.rows.dtype in (torch.float32, torch.bfloat16) and C % (16 // es) == 0 and C > 0
            and x_mod.rows.is_cuda and (not x_mod.rows.is_contiguous() or x_mod.rows.data_ptr() % 16 == 0)
            and _atoms_per_view_ok(atom_ptr))


def gather_segment_max(x_mod, atom_ptr):
    """Atomic max pool of a lazily gathered NON-exact mapping (several pixels per view): ``GatheredFeatures`` at the atom
    level -> ``GatheredFeatures`` at the VIEW level whose rows are the pooled [V, C] features (identity gather, one view
    per row), so that the view-level pooling stays on its fused path (E_mod on the [V, C] rows -- it does not commute with
    the max --, then the recompute chain).  Reference: modules/multimodal/modules.py:400-407, pooling.py:14-71."""
    atom_ptr = _check_ptr(atom_ptr)
    xv = _GatherSegmentMax.apply(x_mod.rows, x_mod.row_idx.contiguous(), atom_ptr, x_mod.plan)
    V = xv.shape[0]
    ident = torch.arange(V + 1, dtype=torch.int32, device=xv.device)
    return GatheredFeatures(xv, ident[:V], torch.ones(V, dtype=torch.int32, device=xv.device), True, (ident[:V], ident))


class InterpolatedFeatures:
    """Result of a BILINEAR view gather (``sparse_interpolation``, reference image.py:105-170) that has not been
    materialised: ``x_mod[p] = sum_k tap_weights[p, k] * rows[tap_rows[p, k]]`` over the 4 corner taps.

    ``rows`` is the [R, C] row view of the channels-last feature maps (a differentiable function of the 2D encoder
    output), ``tap_rows`` int32 [P, 4] / ``tap_weights`` fp32 [P, 4] the taps of ``dva_gather_bilinear_taps`` (border
    replicated: clamped rows).  A consumer that understands this type (GroupBimodalCSRPool through
    ``fused_bilinear``) evaluates E_mod per view inside its kernels -- the first Linear commutes with the
    interpolation and runs on the R map rows; everything else calls ``materialize()`` and gets the reference's
    [P, C] tensor.  ``exact``: every view owns exactly one atom (P == V): the atomic pool is the identity."""

    def __init__(self, x, packed_idx, coords, tap_rows, tap_weights, anchors, exact):
        self.x, self.packed_idx, self.coords, self.exact = x, packed_idx, coords, exact
        self.tap_rows, self.tap_weights, self.anchors = tap_rows, tap_weights, anchors
        B, C, H, W = x.shape
        self.rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C)      # view when x is channels_last
        self.geometry = [(B, H, W)]     # one (B, H, W) per setting: rows / anchors are numbered setting after setting
        self.parts = self.order = None  # cat(): the per-setting gathers and the view order (materialize goes through them)

    @staticmethod
    def cat(items, order=None):
        """The views of several SETTINGS (feature maps of different sizes, ``ImageData`` = list of
        ``SameSettingImageData``) as one lazy gather, optionally permuted into point order (reference
        modules/multimodal/modules.py:514-525 + core/multimodal/image.py:1549-1588 ``view_cat_sorting``: the reference
        concatenates the materialised [V_s, C] tensors and indexes the result).  The map rows of the settings are stacked
        ([sum R_s, C], differentiable), tap rows and anchors offset into the stacked numbering, so a consumer of taps
        (fused_bilinear) sees one gather; ``materialize()`` still evaluates setting by setting."""
        if len(items) == 1 and order is None:
            return items[0]
        assert all(isinstance(it, InterpolatedFeatures) and it.parts is None for it in items)
        out = InterpolatedFeatures.__new__(InterpolatedFeatures)
        out.x = out.packed_idx = out.coords = None
        out.parts, out.order = list(items), order
        out.exact = all(it.exact for it in items)
        out.geometry = [g for it in items for g in it.geometry]
        out.rows = torch.cat([it.rows for it in items], dim=0)
        # one pass (dva_bilinear_taps_cat): concatenate, offset rows / anchors into the stacked numbering, permute
        import ctypes
        lib = _lib.load()
        S = len(items)
        dev = items[0].tap_rows.device
        V = sum(it.tap_rows.shape[0] for it in items)
        if order is not None:
            order = order.to(torch.int64).contiguous()
            assert order.shape[0] == V
        keep = [(it.tap_rows.contiguous(), it.tap_weights.contiguous(), it.anchors.contiguous()) for it in items]
        arr = lambda k: (ctypes.c_void_p * S)(*[t[k].data_ptr() for t in keep])
        i64 = lambda vals: (ctypes.c_int64 * S)(*vals)
        rows4 = torch.empty((V, 4), dtype=torch.int32, device=dev)
        w4 = torch.empty((V, 4), dtype=torch.float32, device=dev)
        anchors = torch.empty(V, dtype=torch.int32, device=dev)
        with _timed("bilinear_taps_cat", V * (8 + 2 * 36)):
            check(lib.dva_bilinear_taps_cat(S, arr(0), arr(1), arr(2), i64([t[0].shape[0] for t in keep]),
                                            i64([it.rows.shape[0] for it in items]),
                                            i64([n_anchors(it.geometry) for it in items]), ptr(order), V, ptr(rows4),
                                            ptr(w4), ptr(anchors), stream_of(rows4)), "dva_bilinear_taps_cat")
        out.tap_rows, out.tap_weights, out.anchors = rows4, w4, anchors
        return out

    @property
    def shape(self):
        return torch.Size((self.tap_rows.shape[0], self.rows.shape[1]))

    @property
    def device(self):
        return self.rows.device

    @property
    def dtype(self):
        return self.rows.dtype

    def dim(self):
        return 2

    def materialize(self, rows=None):
        """The reference's [P, C] tensor; ``rows`` [R, C']: the same interpolation of another per-row tensor (a linear
        function of the map rows, e.g. the hoisted first Linear of E_mod)."""
        if self.parts is not None:      # several settings: setting by setting, then the view order of cat()
            splits = [None] * len(self.parts) if rows is None else torch.split(rows, [p.rows.shape[0] for p in self.parts])
            out = torch.cat([p.materialize(rows=r) for p, r in zip(self.parts, splits)], dim=0)
            return out if self.order is None else out[self.order]
        if rows is None:
            return gather_bilinear(self.x, self.packed_idx, self.coords)
        B, _, H, W = self.x.shape
        return gather_bilinear(rows.view(B, H, W, rows.shape[1]).permute(0, 3, 1, 2), self.packed_idx, self.coords)


def lazy_gather_bilinear(x, packed_idx, coords, exact):
    """Lazy counterpart of ``gather_bilinear``: the taps are computed, no [P, C] tensor is produced."""
    assert x.dim() == 4 and coords.dim() == 2 and coords.shape[1] == 2
    lib = _lib.load()
    require_device(x, packed_idx, coords)
    B, C, H, W = x.shape
    coords = coords.float().contiguous()
    P = packed_idx.shape[0]
    rows4 = torch.empty((P, 4), dtype=torch.int32, device=x.device)
    w4 = torch.empty((P, 4), dtype=torch.float32, device=x.device)
    anchors = torch.empty(P, dtype=torch.int32, device=x.device)
    with _timed("bilinear_taps", P * (16 + 36)):
        check(lib.dva_gather_bilinear_taps_anchor(ptr(packed_idx), ptr(coords), P, B, H, W, ptr(rows4), ptr(w4),
                                                  ptr(anchors), stream_of(x)), "dva_gather_bilinear_taps_anchor")
    return InterpolatedFeatures(x, packed_idx, coords, rows4, w4, anchors, exact)


LAZY_TYPES = (GatheredFeatures, InterpolatedFeatures)     # what .materialize() turns into the reference's [P, C] tensor


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, row_idx):
        ctx.save_for_backward(row_idx)
        ctx.n_rows = rows.shape[0]
        return rows.index_select(0, row_idx.long())

    @staticmethod
    def backward(ctx, gout):
        (row_idx,) = ctx.saved_tensors
        g = torch.zeros((ctx.n_rows, gout.shape[1]), dtype=torch.float32, device=gout.device)
        g.index_add_(0, row_idx.long(), gout.float())
        return g.to(gout.dtype), None


def gather_rows(rows, row_idx):
    """Materialise ``rows[row_idx]`` (device library op; only used off the fused path)."""
    require_device(rows, row_idx)
    return _GatherRows.apply(rows, row_idx)


def _team_records_ok(C, G, dtype):
    """Mirror of team_geometry() in csrc/attention.hip: the view records are produced by the wavefront-team
    backward kernel only."""
    vec = 4 if dtype == torch.float32 else 8
    if C % vec:
        return False
    lpr = C // vec
    pow2 = lambda x: x > 0 and (x & (x - 1)) == 0
    return pow2(lpr) and lpr <= 64 and pow2(G) and G <= 32 and C % G == 0 and (C // G) % vec == 0


# fp32 rows with four score groups: the lean tile-based attention backward (False: the team kernel; tests A/B)
LEAN_ATTENTION_BWD = True


class _ViewGatherAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, row_idx, compat, csr_idx, gate_w, gate_b, scaling, eps, plan):
        lib = _lib.load()
        require_device(rows, row_idx, compat, csr_idx, gate_w, gate_b)
        ctx.plan = plan
        rows = rows.contiguous()
        compat = compat.contiguous()
        N, V, (R, C), G = csr_idx.shape[0] - 1, row_idx.shape[0], rows.shape, compat.shape[1]
        gw = gate_w.detach().reshape(-1).float().contiguous() if gate_w is not None else None
        gb = gate_b.detach().reshape(-1).float().contiguous() if gate_b is not None else None
        if (gw is None) != (gb is None):
            gw = gw if gw is not None else torch.ones(G, device=rows.device)
            gb = gb if gb is not None else torch.zeros(G, device=rows.device)
        out = torch.empty((N, C), dtype=rows.dtype, device=rows.device)
        att = torch.empty((V, G), dtype=torch.float32, device=rows.device)   # every view belongs to a point
        gate = torch.empty((N, G), dtype=torch.float32, device=rows.device)
        amax = torch.empty((N, G), dtype=torch.int32, device=rows.device)
        es = rows.element_size()
        # SURVEY.md 8(d) "fused view-gather+attention": V*(g*C*s + idx) + scores + N*(C*s + ptr)
        with _timed("view_gather_attention_fwd", V * (C * es + 4 + 2 * G * 4) + N * (C * es + 8 + 2 * G * 4)):
            check(lib.dva_view_gather_attention_fwd(
                ptr(rows), ptr(row_idx), ptr(compat), ptr(csr_idx), ptr(gw), ptr(gb), ptr(out), ptr(att),
                ptr(gate), ptr(amax), N, V, C, G, int(scaling), float(eps), dtype_code(rows),
                ATTENTION_ALGO, stream_of(rows)), "dva_view_gather_attention_fwd")
        ctx.save_for_backward(rows, row_idx, compat, csr_idx, att, gate, amax,
                              gw if gw is not None else csr_idx, gb if gb is not None else csr_idx, out)
        ctx.meta = (int(scaling), gw is not None,
                    None if gate_w is None else gate_w.shape, None if gate_b is None else gate_b.shape)
        ctx.eps = float(eps)
        ctx.mark_non_differentiable(att, gate)
        return out, att, gate

    @staticmethod
    def backward(ctx, gout, _gatt, _ggate):
        lib = _lib.load()
        rows, row_idx, compat, csr_idx, att, gate, amax, gw, gb, out = ctx.saved_tensors
        scaling, has_gate, w_shape, b_shape = ctx.meta
        gout = gout.contiguous()
        N, V, (R, C), G = csr_idx.shape[0] - 1, row_idx.shape[0], rows.shape, compat.shape[1]
        gcompat = torch.empty_like(compat)             # every view is written (it belongs to a point)
        gwb = torch.zeros(2 * G, dtype=torch.float32, device=rows.device) if has_gate else None
        es = rows.element_size()
        need_rows = ctx.needs_input_grad[0]
        use_plan = need_rows and ROWS_GRAD_ALGO == 0
        if (LEAN_ATTENTION_BWD and use_plan and ATTENTION_ALGO != 1 and rows.dtype == torch.float32
                and gout.dtype == torch.float32 and G == 4 and C in (32, 64, 128, 256) and V > 0
                and V * 32 < (1 << 32) - 16 and max(R, N) * C * 4 < (1 << 32) - 16):
            # fp32 rows, four score groups (the no-autocast headline shape): the tile-based attention backward of the
            # chain path (csrc/chain_bwd.hip attn_bwd_kernel<float, ...>: scores in, score gradients + 32-byte view
            # records out) instead of the team kernel: 2.8 -> 2.0 ms at V = 33.5 M
            from .fused_chain import build_tiles
            tiles, n_tiles = build_tiles(csr_idx, V)
            vp = csr_expand(csr_idx, V)
            rec = torch.empty((V, 8), dtype=torch.float32, device=rows.device)
            with _timed("view_gather_attention_bwd", V * (C * 4 + 16 + 8 + 16 + 32) + N * (C * 4 + 8)):
                check(lib.dva_chain_attn_bwd_f32(
                    ptr(compat), ptr(vp), ptr(tiles), ptr(n_tiles), ptr(rows), ptr(row_idx), ptr(csr_idx),
                    ptr(gw) if has_gate else None, ptr(gb) if has_gate else None, ptr(gout), ptr(out), ptr(gcompat),
                    ptr(rec), ptr(gwb), N, V, R, C, G, scaling, ctx.eps, stream_of(rows)), "dva_chain_attn_bwd_f32")
            plan = ctx.plan if ctx.plan is not None else row_plan(row_idx, R, with_counts=False,
                                                                  split=split_plan_serves(rows.dtype, C))[0]
            grows = None
            if isinstance(plan, SplitPlan) and SPLIT_FUSED:
                # round 6: the 32-byte records through pass A of the split plan + the fp32 bucket kernel
                grows = plan.rows_grad_f32(gout, rec, C, G, stream_of(rows))
            if grows is None:
                perm, row_ptr = plan
                grows = torch.empty((R, C), dtype=torch.float32, device=rows.device)
                with _timed("view_gather_rows_grad", V * (4 + 32 + C * es) + R * (C * 4 + 4)):
                    check(lib.dva_view_gather_rows_grad(
                        ptr(gout), ptr(att), ptr(gate) if has_gate else None, None, ptr(perm), ptr(row_ptr), ptr(rec), 8,
                        ptr(grows), R, V, C, G, dtype_code(rows), stream_of(rows)), "dva_view_gather_rows_grad")
            g_w = gwb[:G].reshape(w_shape) if (has_gate and w_shape is not None) else None
            g_b = gwb[G:].reshape(b_shape) if (has_gate and b_shape is not None) else None
            return grows, None, gcompat, None, g_w, g_b, None, None, None
        if (LEAN_ATTENTION_BWD and use_plan and ATTENTION_ALGO != 1 and rows.dtype == torch.bfloat16
                and gout.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and G in (1, 2, 4)
                and C in (32, 64, 128, 256, 512) and V > 0 and V * 16 < (1 << 32) - 16
                and max(R, N) * C * 2 < (1 << 32) - 16):
            # bf16 rows (round 4; QKVBimodalCSRPool on the chain, and every bf16 caller with <= 4 score groups): the
            # chain path's attention backward (scores [V, 4] in, score gradients + 16-byte view records out) and its
            # 16-byte-record rows gradient instead of the team kernels: 1.6 + 1.3 -> 0.9 + 1.3 ms at V = 33.5 M
            from .fused_chain import build_tiles
            tiles, n_tiles = build_tiles(csr_idx, V)
            vp = csr_expand(csr_idx, V)
            if G == 4:
                sc4 = compat
            else:
                sc4 = torch.zeros((V, 4), dtype=torch.float32, device=rows.device)
                sc4[:, :G] = compat
            dc = torch.empty((V, 4), dtype=torch.float32, device=rows.device)
            rec = torch.empty((V, 4), dtype=torch.int32, device=rows.device)
            with _timed("view_gather_attention_bwd", V * (C * 2 + 16 + 8 + 16 + 16) + N * (C * 2 + 8)):
                check(lib.dva_chain_attn_bwd(
                    ptr(sc4), ptr(vp), ptr(tiles), ptr(n_tiles), ptr(rows), ptr(row_idx), ptr(csr_idx),
                    ptr(gw) if has_gate else None, ptr(gb) if has_gate else None, ptr(gout), ptr(out), ptr(dc),
                    ptr(rec), ptr(gwb), N, V, R, C, G, scaling, ctx.eps, stream_of(rows)), "dva_chain_attn_bwd")
            plan = ctx.plan if ctx.plan is not None else row_plan(row_idx, R, with_counts=False)[0]
            # bf16: rounded where it is summed
            grows = rows_grad_rec16(gout, plan, rec, R, C, G, rows.dtype, stream_of(rows))
            g_w = gwb[:G].reshape(w_shape) if (has_gate and w_shape is not None) else None
            g_b = gwb[G:].reshape(b_shape) if (has_gate and b_shape is not None) else None
            gcompat = dc if G == 4 else dc[:, :G].contiguous()
            return grows, None, gcompat, None, g_w, g_b, None, None, None
        if need_rows and not use_plan:
            grows = torch.zeros((R, C), dtype=torch.float32, device=rows.device)
        else:
            grows = None
        # per-view records (point id | gate * attention per group) for the rows-gradient pass: one 32-byte
        # sector per view instead of scattered reads of view_point, att and gate
        rs = ((G + 1 + 7) // 8) * 8
        rec = None
        if use_plan and ATTENTION_ALGO != 1 and _team_records_ok(C, G, rows.dtype):
            rec = torch.empty((V, rs), dtype=torch.float32, device=rows.device)
        with _timed("view_gather_attention_bwd",
                    V * (C * es + 4 + 2 * G * 4 + (C * 4 * 2 if grows is not None else 0)
                         + (rs * 4 if rec is not None else 0))
                    + N * (C * es + 8 + 3 * G * 4)):
            check(lib.dva_view_gather_attention_bwd(
                ptr(gout), ptr(rows), ptr(row_idx), ptr(compat), ptr(att), ptr(gate), ptr(amax),
                ptr(csr_idx), ptr(gw) if has_gate else None, ptr(gb) if has_gate else None, ptr(grows),
                ptr(gcompat), ptr(gwb), ptr(rec), rs, N, V, C, G, scaling, dtype_code(rows), ATTENTION_ALGO,
                stream_of(rows)), "dva_view_gather_attention_bwd")
        if use_plan:
            plan = ctx.plan if ctx.plan is not None else row_plan(row_idx, R, with_counts=False, split=False)[0]
            perm, row_ptr = plan
            vp = csr_expand(csr_idx, V) if rec is None else None
            grows = torch.empty((R, C), dtype=torch.float32, device=rows.device)
            # per view: perm + record (or view_point + scores) + grad_out row; per row: fp32 row written
            with _timed("view_gather_rows_grad", V * (4 + (rs * 4 if rec is not None else 4 + 2 * G * 4) + C * es)
                        + R * (C * 4 + 4)):
                check(lib.dva_view_gather_rows_grad(
                    ptr(gout), ptr(att), ptr(gate) if has_gate else None, ptr(vp), ptr(perm), ptr(row_ptr),
                    ptr(rec), rs, ptr(grows), R, V, C, G, dtype_code(rows), stream_of(rows)),
                    "dva_view_gather_rows_grad")
        g_w = gwb[:G].reshape(w_shape) if (has_gate and w_shape is not None) else None
        g_b = gwb[G:].reshape(b_shape) if (has_gate and b_shape is not None) else None
        return (grows.to(rows.dtype) if grows is not None else None), None, gcompat, None, g_w, g_b, \
            None, None, None


def view_gather_attention(rows, row_idx, compat, csr_idx, gate_w=None, gate_b=None, scaling=False,
                          eps=1e-12, plan=None):
    """``view_attention`` with the view gather fused in: the value of view v is ``rows[row_idx[v]]``.
    ``plan`` = ``row_plan(row_idx, R)[0]`` if the caller already has it (else built in backward)."""
    csr_idx = _check_ptr(csr_idx)
    if compat.dim() == 1:
        compat = compat.reshape(-1, 1)
    return _ViewGatherAttention.apply(rows, row_idx.contiguous(), compat.float(), csr_idx, gate_w, gate_b,
                                      scaling, eps, plan)


# ---------------------------------------------------------------------------------------------
# E_mod on the feature-map rows: tall-skinny Linear + weighted BatchNorm + LeakyReLU
# ---------------------------------------------------------------------------------------------

def _xty_splitk(a, b, splits=64):
    """``a.T @ b`` for tall operands [R, O], [R, I] -> fp32 [O, I].  A plain GEMM call runs this
    K = R reduction in ONE 64x64 tile (one workgroup, 0.5 ms at R = 262144); ``splits`` batched
    products of R / splits rows fill the device, the partial sums are added in fp32."""
    R = a.shape[0]
    if R < 8192 or R % splits:
        return a.float().t() @ b.float()
    part = torch.bmm(a.view(splits, R // splits, -1).transpose(1, 2), b.view(splits, R // splits, -1))
    return part.sum(0, dtype=torch.float32)        # one reduction kernel (fp32 accumulation), no fp32 copy of `part`


class _TallLinear(torch.autograd.Function):
    """``F.linear`` whose weight gradient is a split-K product (same forward, same autocast rule)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else x.dtype
        xc, wc = x.to(dt), weight.to(dt)
        ctx.save_for_backward(xc, wc)
        ctx.meta = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return torch.nn.functional.linear(xc, wc, None if bias is None else bias.to(dt))

    @staticmethod
    def backward(ctx, gy):
        xc, wc = ctx.saved_tensors
        x_dt, w_dt, b_dt = ctx.meta
        gy = gy.contiguous()
        gx = (gy @ wc).to(x_dt) if ctx.needs_input_grad[0] else None
        gw = _xty_splitk(gy, xc).to(w_dt) if ctx.needs_input_grad[1] else None
        gb = gy.float().sum(0).to(b_dt) if (b_dt is not None and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


def tall_linear(x, weight, bias=None):
    return _TallLinear.apply(x, weight, bias)


class _RowBNAct(torch.autograd.Function):
    """``leaky(BatchNorm(y))`` on map rows with the statistics weighted by ``counts`` (views per row).
    ``mean`` / ``invstd`` are given (batch statistics from ``rowbn_stats`` or running statistics);
    with ``batch_stats`` the backward includes the statistics terms."""

    @staticmethod
    def forward(ctx, y, counts, gamma, beta, mean, invstd, n, batch_stats, slope, bn_tab=None):
        lib = _lib.load()
        require_device(y)
        y = y.contiguous()
        R, C = y.shape
        if bn_tab is None:
            g = gamma.detach().float() if gamma is not None else torch.ones(C, device=y.device)
            b = beta.detach().float() if beta is not None else torch.zeros(C, device=y.device)
            bn_tab = torch.stack([mean.float(), invstd.float(), g, b]).contiguous()
        bn = bn_tab
        out = torch.empty_like(y)
        with _timed("rowbn_apply", R * C * 2 * y.element_size()):
            check(lib.dva_rowbn_apply(ptr(y), ptr(bn), ptr(out), R, C, float(slope), dtype_code(y),
                                      stream_of(y)), "dva_rowbn_apply")
        ctx.save_for_backward(y, bn, *([counts] if counts is not None else []))
        # the same clamped normaliser as the forward's statistics (bn_table): a sample without views has n = 0
        ctx.meta = (max(float(n), 1.0), bool(batch_stats), float(slope), gamma is not None, beta is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        y, bn = ctx.saved_tensors[:2]
        counts = ctx.saved_tensors[2] if len(ctx.saved_tensors) > 2 else None
        n, batch_stats, slope, has_g, has_b = ctx.meta
        R, C = y.shape
        gout = gout.contiguous().to(y.dtype)
        sums = zeros_small(2 * C, torch.float64, y.device)
        with _timed("rowbn_bwd_stats", R * C * 2 * y.element_size()):
            check(lib.dva_rowbn_bwd_stats(ptr(gout), ptr(y), ptr(bn), ptr(sums), R, C, slope, dtype_code(y),
                                          stream_of(y)), "dva_rowbn_bwd_stats")
        small = torch.empty(4 * C, dtype=torch.float32, device=y.device)    # S/n [2C] | d gamma [C] | d beta [C]
        sm, dg, db = small[:2 * C], small[2 * C:3 * C], small[3 * C:]
        check(lib.dva_bn_bwd_consts(ptr(sums), None, (1.0 / n) if batch_stats else 0.0, 0, ptr(sm), ptr(dg), ptr(db),
                                    C, stream_of(y)), "dva_bn_bwd_consts")
        dy = torch.empty_like(y)
        with _timed("rowbn_bwd_apply", R * C * 3 * y.element_size()):
            check(lib.dva_rowbn_bwd_apply(ptr(gout), ptr(y), ptr(counts), ptr(bn), ptr(sm), ptr(dy), R, C, slope,
                                          dtype_code(y), stream_of(y)), "dva_rowbn_bwd_apply")
        return dy, None, dg if has_g else None, db if has_b else None, None, None, None, None, None, None


def rowbn_sums(y, counts):
    """sum_r counts_r y_r | sum_r counts_r y_r^2 as one float64 [2C] tensor."""
    lib = _lib.load()
    require_device(y)
    y = y.contiguous()
    R, C = y.shape
    sums = zeros_small(2 * C, torch.float64, y.device)
    with _timed("rowbn_stats", R * (C * y.element_size() + 4)):
        check(lib.dva_rowbn_stats(ptr(y), ptr(counts), ptr(sums), R, C, dtype_code(y), stream_of(y)),
              "dva_rowbn_stats")
    return sums


def bn_table(sums, n, bn, batch_stats):
    """fp32 [4, C] = mean | invstd | gamma | beta of an nn.BatchNorm1d from the float64 sums of its input over ``n``
    rows (or from the running statistics), running statistics updated as the module does: one launch."""
    lib = _lib.load()
    C = bn.num_features
    dev = (bn.weight if bn.affine else (sums if sums is not None else bn.running_mean)).device
    out = torch.empty((4, C), dtype=torch.float32, device=dev)
    update = batch_stats and bn.training and bn.track_running_stats
    check(lib.dva_bn_finalize(ptr(sums), float(max(n, 1.0)), ptr(bn.running_mean if (update or not batch_stats) else None),
                              ptr(bn.running_var if (update or not batch_stats) else None),
                              ptr(bn.num_batches_tracked if update else None),
                              ptr(bn.weight.detach()) if bn.affine else None,
                              ptr(bn.bias.detach()) if bn.affine else None,
                              float(bn.momentum), float(bn.eps), 1 if batch_stats else 0, C, ptr(out),
                              stream_of(out)), "dva_bn_finalize")
    return out


def rowbn_stats(y, counts):
    """(sum_r counts_r y_r, sum_r counts_r y_r^2) as float64 [C] each."""
    sums = rowbn_sums(y, counts)
    C = y.shape[1]
    return sums[:C], sums[C:]


def rowbn_act(y, counts, gamma, beta, mean, invstd, n, batch_stats, slope, bn_tab=None):
    return _RowBNAct.apply(y, counts, gamma, beta, mean, invstd, n, batch_stats, slope, bn_tab)


# ---------------------------------------------------------------------------------------------
# voxel parent index after a strided sparse 3D convolution (modules.py:176-198)
# ---------------------------------------------------------------------------------------------

def voxel_parent_index(in_coords, out_coords, stride_out, batch_col=3):
    """``idx[i] = j`` with ``out_coords[j] == floor_to_stride(in_coords[i])`` (batch column untouched),
    ``-1`` when absent: the torchsparse ``sphashquery(sphash(in), sphash(out))`` of the reference."""
    lib = _lib.load()
    require_device(in_coords, out_coords)
    assert in_coords.dim() == 2 and in_coords.shape[1] == 4 and out_coords.dim() == 2 and out_coords.shape[1] == 4
    ic = in_coords.to(torch.int32).contiguous()
    oc = out_coords.to(torch.int32).contiguous()
    n_in, n_out = ic.shape[0], oc.shape[0]
    idx = torch.empty(n_in, dtype=torch.int64, device=ic.device)
    nbytes = lib.dva_voxel_parent_workspace_bytes(n_out)
    if nbytes < 0:
        raise _lib.DvaError("dva_voxel_parent_workspace_bytes", int(nbytes))
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=ic.device)
    with _timed("voxel_parent_index", n_in * 24 + n_out * 24):
        check(lib.dva_voxel_parent_index(ptr(ic), n_in, ptr(oc), n_out, int(stride_out), int(batch_col), ptr(idx),
                                         ptr(ws), int(nbytes), stream_of(ic)), "dva_voxel_parent_index")
    return idx


# ---------------------------------------------------------------------------------------------
# sparse 3D convolution on voxel tensors (modules/SparseConv3d over torchsparse 1.1.0 Conv3d)
# ---------------------------------------------------------------------------------------------

def voxel_kernel_map(src_coords, dst_coords, offsets):
    """``nbr[k, j] = i`` with ``src_coords[i] == dst_coords[j] + offsets[k]`` (spatial columns; equal batch
    column), ``-1`` when absent.  int32 [K, n_dst]."""
    lib = _lib.load()
    require_device(src_coords, dst_coords)
    assert src_coords.dim() == 2 and src_coords.shape[1] == 4 and dst_coords.dim() == 2 and dst_coords.shape[1] == 4
    sc = src_coords.to(torch.int32).contiguous()
    dc = dst_coords.to(torch.int32).contiguous()
    off = torch.as_tensor(offsets, dtype=torch.int32).reshape(-1, 3).to(sc.device).contiguous()
    K, n_src, n_dst = off.shape[0], sc.shape[0], dc.shape[0]
    nbr = torch.empty((K, n_dst), dtype=torch.int32, device=sc.device)
    nbytes = lib.dva_voxel_parent_workspace_bytes(n_src)
    if nbytes < 0:
        raise _lib.DvaError("dva_voxel_parent_workspace_bytes", int(nbytes))
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=sc.device)
    with _timed("voxel_kernel_map", n_src * 24 + K * n_dst * 24):
        check(lib.dva_voxel_kernel_map(ptr(sc), n_src, ptr(dc), n_dst, ptr(off), K, ptr(nbr), ptr(ws), int(nbytes),
                                       stream_of(sc)), "dva_voxel_kernel_map")
    return nbr


SPARSE_CONV_TIMER_SHAPES = False   # bench tools: label the KernelTimer entries with the layer shape


def _pad16(c):
    return (c + 15) // 16 * 16


def _sparse_conv_apply(x, nbr, W, bias, n_dst, mode):
    """out[j] = bias + sum_k x[nbr[k, j]] @ (W[k] if mode == 0 else W[k].T); channel counts padded to 16."""
    lib = _lib.load()
    K = nbr.shape[0]
    cin, cout = (W.shape[1], W.shape[2]) if mode == 0 else (W.shape[2], W.shape[1])
    assert x.shape[1] == cin, f"features have {x.shape[1]} channels, the kernel expects {cin}"
    if x.shape[0] == 0 or n_dst == 0:
        out = torch.zeros((n_dst, cout), dtype=x.dtype, device=x.device)
        return out if bias is None else out + bias.to(x.dtype)
    cin_p, cout_p = _pad16(cin), _pad16(cout)
    Wf = W.detach().float()
    if cin_p != cin or cout_p != cout:
        pa, pb = (cin_p - cin, cout_p - cout) if mode == 0 else (cout_p - cout, cin_p - cin)
        Wf = torch.nn.functional.pad(Wf, (0, pb, 0, pa))
        x = torch.nn.functional.pad(x, (0, cin_p - cin))
        if bias is not None:
            bias = torch.nn.functional.pad(bias.detach().float(), (0, cout_p - cout))
    x, Wf = x.contiguous(), Wf.contiguous()
    bias = None if bias is None else bias.detach().float().contiguous()
    out = torch.empty((n_dst, cout_p), dtype=x.dtype, device=x.device)
    code = dtype_code(x)
    nbytes = int(lib.dva_sparse_conv_workspace_bytes(K, cin_p, cout_p, code))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    s = x.element_size()
    with _timed(f"sparse_conv_apply{SPARSE_CONV_TIMER_SHAPES and f'[{K}x{cin_p}->{cout_p}@{n_dst}]' or ''}",
                n_dst * (K * 4 + cout_p * s) + x.shape[0] * cin_p * s):
        check(lib.dva_sparse_conv_apply(ptr(x), ptr(nbr), ptr(Wf), ptr(bias), ptr(out), x.shape[0], n_dst, K, cin_p,
                                        cout_p, mode, code, ptr(ws), nbytes, stream_of(x)), "dva_sparse_conv_apply")
    return out if cout_p == cout else out[:, :cout].contiguous()


def _sparse_conv_wgrad(x, nbr, gout, cin, cout):
    lib = _lib.load()
    K, n_dst = nbr.shape
    cin_p, cout_p = _pad16(cin), _pad16(cout)
    if cin_p != cin:
        x = torch.nn.functional.pad(x, (0, cin_p - cin))
    if cout_p != cout:
        gout = torch.nn.functional.pad(gout, (0, cout_p - cout))
    x, gout = x.contiguous(), gout.contiguous()
    gW = torch.empty((K, cin_p, cout_p), dtype=torch.float32, device=x.device)
    s = x.element_size()
    with _timed(f"sparse_conv_wgrad{SPARSE_CONV_TIMER_SHAPES and f'[{K}x{cin_p}->{cout_p}@{n_dst}]' or ''}",
                n_dst * (K * 4 + cout_p * s) + x.shape[0] * cin_p * s):
        check(lib.dva_sparse_conv_wgrad(ptr(x), ptr(nbr), ptr(gout), ptr(gW), x.shape[0], n_dst, K, cin_p, cout_p,
                                        dtype_code(x), stream_of(x)), "dva_sparse_conv_wgrad")
    return gW if (cin_p == cin and cout_p == cout) else gW[:, :cin, :cout].contiguous()


class _SparseConv(torch.autograd.Function):
    """out = bias + sum_k x[nbr[k]] @ W[k].  ``nbr`` [K, n_dst] maps destination voxels to source rows, ``nbr_t``
    [K, n_src] is the transposed map (source voxel -> destination row under the same offset)."""

    @staticmethod
    def forward(ctx, x, W, bias, nbr, nbr_t):
        require_device(x, W, nbr, nbr_t)
        ctx.save_for_backward(x, W, nbr, nbr_t)
        ctx.has_bias = bias is not None
        return _sparse_conv_apply(x, nbr, W, bias, nbr.shape[1], 0)

    @staticmethod
    def backward(ctx, gout):
        x, W, nbr, nbr_t = ctx.saved_tensors
        gout = gout.contiguous().to(x.dtype)
        gx = gW = gb = None
        if ctx.needs_input_grad[0]:
            gx = _sparse_conv_apply(gout, nbr_t, W, None, x.shape[0], 1)
        if ctx.needs_input_grad[1]:
            gW = _sparse_conv_wgrad(x, nbr, gout, W.shape[1], W.shape[2]).to(W.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gout.float().sum(0)
        return gx, gW, gb, None, None


def sparse_conv(x, W, bias, nbr, nbr_t):
    """Sparse convolution over a kernel map pair (see ``voxel_kernel_map``): x [n_src, Cin] (fp32 or bf16),
    W fp32 [K, Cin, Cout], bias [Cout] or None -> [n_dst, Cout].  Under ``torch.autocast`` the features are
    taken in the autocast dtype (the weights stay fp32 masters, rounded to bf16 inside the kernel)."""
    if torch.is_autocast_enabled() and x.is_cuda:
        x = x.to(torch.get_autocast_gpu_dtype())
    return _SparseConv.apply(x, W, bias, nbr, nbr_t)


# ---------------------------------------------------------------------------------------------
# neighbourhood-based mapping features (data_transform/multimodal/image.py:431-612)
# ---------------------------------------------------------------------------------------------

def knn(xyz, k, cell=None):
    """Exact k nearest neighbours of every point among all points (itself first): ``(neighbors int32 [n, k],
    dist2 fp32 [n, k])``, ascending by (squared fp32 distance, index)."""
    lib = _lib.load()
    require_device(xyz)
    xyz = xyz.float().contiguous()
    n = xyz.shape[0]
    assert xyz.dim() == 2 and xyz.shape[1] == 3
    if not 0 < k <= 128:
        raise ValueError(f"ops.knn: k = {k} neighbours requested, the kernel keeps at most 128 candidates per query "
                         f"(csrc/knn.hip); the reference's settings are k <= 75")
    nbr = torch.empty((n, k), dtype=torch.int32, device=xyz.device)
    d2 = torch.empty((n, k), dtype=torch.float32, device=xyz.device)
    if n == 0:
        return nbr, d2
    lo, hi = xyz.min(0).values, xyz.max(0).values
    bbox = torch.cat([lo, hi]).contiguous()
    ext = float((hi - lo).max())                   # host sync: this is a preprocessing transform
    if cell is None:
        # a cell that holds a few points of a surface-like cloud: mean spacing x (k/4)^(1/3)
        dims = (hi - lo) > 1e-6 * max(ext, 1e-30)
        nd = int(dims.sum())
        if nd in (1, 2):
            # a PLANAR (or collinear) cloud -- BiasuttiVisibility searches image-plane points (x, y, 0): the volume rule
            # would give a cell far below the point spacing (z extent clamped to 1e-6) and every query would be retried
            # on coarser grids (4 - 6 full launches): mean spacing in the occupied dimensions x (k/4)^(1/nd)
            area = float((hi - lo)[dims].prod())
            cell = (area / n) ** (1 / nd) * max(1.0, (k / 4) ** (1 / nd))
        else:
            vol = float((hi - lo).clamp(min=1e-6).prod())
            cell = (vol / n) ** (1 / 3) * max(1.0, (k / 4) ** (1 / 3))
    cell = max(float(cell), ext / (1 << 19), 1e-12)
    nbytes = lib.dva_knn_workspace_bytes(n)
    if nbytes < 0:
        raise _lib.DvaError("dva_knn_workspace_bytes", int(nbytes))
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=xyz.device)
    done = torch.zeros(n, dtype=torch.uint8, device=xyz.device)
    # levels: a query that is not provably complete within 2 shells of cells (sparse region, outlier) is
    # retried on a 4x coarser grid; the last level covers the whole cloud
    shells = 2
    with _timed("knn", n * (12 + k * 8)):
        while True:
            last = cell * shells >= ext + cell
            check(lib.dva_knn(ptr(xyz), n, ptr(bbox), float(cell), int(k), (1 << 30) if last else shells, ptr(done),
                              ptr(nbr), ptr(d2), ptr(ws), int(nbytes), stream_of(xyz)), "dva_knn")
            if last:
                break
            cell *= 4.0
    return nbr, d2


def view_occlusion(csr_idx, images, neighbors, k_list, n_images):
    """fp32 [V, len(k_list)]: ratio of the k nearest neighbours (and the point itself) seen by the view's image."""
    lib = _lib.load()
    require_device(csr_idx, images, neighbors)
    V, N, k = images.shape[0], csr_idx.shape[0] - 1, neighbors.shape[1]
    kl = torch.tensor(sorted(int(x) for x in k_list), dtype=torch.int32, device=images.device)
    assert int(kl[-1]) <= k
    out = torch.empty((V, kl.shape[0]), dtype=torch.float32, device=images.device)
    if V == 0:
        return out
    vp = csr_expand(csr_idx, V)
    bits = torch.empty(N * ((n_images + 63) // 64), dtype=torch.int64, device=images.device)
    check(lib.dva_view_occlusion(ptr(vp), ptr(images.contiguous()), V, N, int(n_images), ptr(neighbors.contiguous()),
                                 k, ptr(kl), kl.shape[0], ptr(bits), ptr(out), stream_of(images)),
          "dva_view_occlusion")
    return out
