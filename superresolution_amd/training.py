"""Pre-train step operators on the HIP path (``/root/reference/train.py:90-97``).

The reference trains with autograd: ``loss = weighted_l1(sr, hr) + 0.1 * (1 - ssim(sr, hr))``, ``loss.backward()``,
``AdamW.step()`` (``train.py:93-97,270``).  Here the same arithmetic is composed from the backward-pass operators of
``libsrgen.so`` (``include/srgen.h``, "Pre-train step operators"): there is no autograd graph and no eager-PyTorch
fallback -- torch supplies device memory and the stream only.  First slice: the residual trunk
(``Generator.py:30-42``), the gradient of the pixel losses, and AdamW.  Activations are fp32 NHWC ``(n, 64, 64, 96)``.

fp32 throughout: the reference runs this step under ``torch.autocast(bfloat16)`` (``train.py:91``), which rounds every
conv to bf16; the fp32 operators here are the stricter computation of the same step.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from typing import Dict, Optional, Tuple

import torch

from . import _lib

WINO_FLOATS = 16 * 96 * 96           # Winograd weight image of one 96 -> 96 conv
SPLIT_HALFS = 2 * 9 * 96 * 96         # split-f16 weight image of one 96 -> 96 conv: (hi, lo) per weight
AMAX_SLOTS = 1024                     # SRGEN_AMAX_SLOTS (include/srgen.h)
BF16_IMAGE_ELEMS = 9 * 96 * 96        # bf16 weight image of one 96 -> 96 conv (the eval forward's bf16 kernel)
SPLIT_TAIL_HALFS = 64                 # ... and its 128-byte tail: 32 partial maxima of |w| (the weights' scale)
STEM_SPLIT_BYTES = 9 * 3 * 3 * 2 * 64 * 16 + 128     # split-f16 weight image of a 9x9 conv against a 3-channel image


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


# ---- per-operator-class timing (bench.py: which kernel class a training step spends its time in) -----------------------
# Inside ``profile_ops()`` every operator call below is bracketed by a HIP event pair on the stream it launches on; the
# block yields {class: [ms, calls, work]} (work = executed-independent algorithmic FLOPs for the MFMA-bound classes, bytes
# moved for the HBM-bound ones), filled when the block ends.  Use with ONE stream (PretrainStep(...).side = None): on two
# streams the classes overlap and their times do not add up to the step.
_PROFILE = [None]


@contextlib.contextmanager
def profile_ops():
    rec, out = [], {}
    old = _PROFILE[0]
    _PROFILE[0] = rec
    try:
        yield out
    finally:
        _PROFILE[0] = old
        torch.cuda.synchronize()
        for cls, e0, e1, work in rec:
            a = out.setdefault(cls, [0.0, 0, 0.0])
            a[0] += e0.elapsed_time(e1)
            a[1] += 1
            a[2] += work


def _timed(cls: str, work=lambda *a, **k: 0.0):
    def deco(fn):
        import functools

        @functools.wraps(fn)
        def wrapped(*a, **k):
            rec = _PROFILE[0]
            if rec is None:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            rec.append((cls, e0, e1, float(work(*a, **k))))
            return r
        return wrapped
    return deco


def _conv_flops(x, wpack, bias=None, res=None, cout_blocks=1, **k):
    return 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * 96 * 96 * 9 * cout_blocks


def _bytes_of(n_tensors):
    return lambda c, *a, **k: float(n_tensors * c.numel() * c.element_size())


def _chk_nhwc(name: str, t: torch.Tensor, like: Optional[torch.Tensor] = None) -> None:
    if not t.is_cuda or t.dtype not in (torch.float32, torch.bfloat16) or not t.is_contiguous() or t.dim() != 4 \
            or t.shape[3] != 96 or t.shape[1] != t.shape[2]:
        raise RuntimeError(f"{name}: expected a contiguous fp32 (or, in the 16-bit step, bf16) ROCm tensor (n, hw, hw, 96), got "
                           f"{tuple(t.shape)} {t.dtype} on {t.device}; there is no CPU fallback")
    if like is not None and t.dtype != like.dtype:
        raise RuntimeError(f"{name} is {t.dtype} beside a {like.dtype} operand: the operators do not mix storage types")


def _bf(t: torch.Tensor) -> bool:
    return t.dtype == torch.bfloat16


class Workspace:
    """Scratch buffers of the training operators for one device (grown on demand, re-used across calls)."""

    def __init__(self, device: torch.device):
        self.device = device
        self._buf: Dict[str, torch.Tensor] = {}

    def get(self, key: str, nbytes: int) -> torch.Tensor:
        b = self._buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._buf[key] = b
        return b


def wino_pack(w: torch.Tensor, mode: int = 0, block: int = 0) -> torch.Tensor:
    """Winograd weight image of a (cout, 96, 3, 3) fp32 weight on its device: forward (mode 0) or the data-gradient
    conv of cout block ``block`` (mode 1)."""
    if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 4 or tuple(w.shape[1:]) != (96, 3, 3):
        raise RuntimeError(f"expected a ROCm fp32 (cout, 96, 3, 3) weight, got {tuple(w.shape)} on {w.device}")
    wc = w.detach().contiguous()
    cb = wc.shape[0] // 96 if mode == 0 else 1
    out = torch.empty(cb * WINO_FLOATS, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.lib().srgen_wino_pack(wc.data_ptr(), wc.shape[0], mode, block, out.data_ptr(), _stream(w)))
    return out


def split_pack(w: torch.Tensor, mode: int = 0, block: int = 0) -> torch.Tensor:
    """Weight image of the split-f16 3x3 kernel (an fp16 tensor: high and low parts of every weight) of a (cout, 96, 3, 3)
    fp32 weight; modes as ``wino_pack``."""
    if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 4 or tuple(w.shape[1:]) != (96, 3, 3):
        raise RuntimeError(f"expected a ROCm fp32 (cout, 96, 3, 3) weight, got {tuple(w.shape)} on {w.device}")
    wc = w.detach().contiguous()
    cb = wc.shape[0] // 96 if mode == 0 else 1
    out = torch.empty(cb * SPLIT_HALFS + SPLIT_TAIL_HALFS, dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.lib().srgen_split_pack(wc.data_ptr(), wc.shape[0], mode, block, out.data_ptr(), _stream(w)))
    return out


def conv_path() -> str:
    """The fp32 3x3 kernel the training operators use: ``SRGEN_F32_CONV3X3`` = split (default) | wino (``direct`` has no
    device-side weight packing: the Winograd kernel stands in for it)."""
    v = os.environ.get("SRGEN_F32_CONV3X3") or "split"
    if v not in ("split", "wino", "direct"):
        raise RuntimeError(f"SRGEN_F32_CONV3X3='{v}': expected split, wino or direct")
    return "split" if v == "split" else "wino"


def wgrad_path() -> str:
    """The 3x3 weight-gradient kernel: ``SRGEN_WGRAD`` = split (default when the convs run the split path: products from
    f16 splits, srgen_wgrad_f32s.hip) | wino (Winograd domain on the fp32 MFMA) | direct (9-tap GEMM on the fp32 MFMA;
    read by the library itself)."""
    v = os.environ.get("SRGEN_WGRAD") or ("split" if conv_path() == "split" else "wino")
    if v not in ("split", "wino", "direct"):
        raise RuntimeError(f"SRGEN_WGRAD='{v}': expected split, wino or direct")
    return v


# ---- precision of the 3x3 convs (forward, data gradient, weight gradient) ---------------------------------------------
# "fp32": every product to fp32 accuracy (split path: three f16 MFMAs per product; or the exact-fp32 MFMA kernels).
# "f16":  the operands rounded to f16 after the power-of-two range scaling, ONE MFMA per product, fp32 accumulation,
#         fp32 tensors -- the counterpart of the reference's `with torch.autocast('cuda', dtype=torch.bfloat16)` around the
#         generator call (train.py:91-92,148-149,166-167), where every conv runs on 16-bit operands.  f16 keeps 11
#         significant bits against bf16's 8, so this mode sits between the reference's autocast step and its fp32 step
#         (tests/test_gpu_training.py: closer to the fp32 gradients than the reference's own autocast gradients are).
# "bf16": the 16-bit step -- activations and gradients are bf16 TENSORS (half the bytes of every BatchNorm / PReLU pass, no
#         conversion in the convs' staging), bf16 MFMA products, fp32 accumulation: the storage and product types of the
#         reference's autocast step itself.  Parameters, statistics, parameter gradients and the images stay fp32.
_CONV_PRECISION = ["fp32"]


@contextlib.contextmanager
def conv_precision(precision: str):
    """Within the block ``conv3x3`` / ``conv3x3_wgrad`` run at ``precision`` ("fp32" | "f16" | "bf16"; the 16-bit ones on the
    split path only), and ``GeneratorGrad.forward`` keeps its activations in fp32 ("fp32", "f16") or bf16 ("bf16")."""
    if precision not in ("fp32", "f16", "bf16"):
        raise RuntimeError(f"conv precision '{precision}': expected fp32, f16 or bf16")
    if precision != "fp32" and conv_path() != "split":
        raise RuntimeError("16-bit conv products exist on the split path only (SRGEN_F32_CONV3X3 unset or 'split')")
    old = _CONV_PRECISION[0]
    _CONV_PRECISION[0] = precision
    try:
        yield
    finally:
        _CONV_PRECISION[0] = old


def resolve_precision(precision: str) -> str:
    """``"auto"``: inside ``torch.autocast`` for ROCm devices (the way the reference runs its training forward, train.py:91)
    the 16-bit arithmetic of the autocast dtype -- "bf16" (bf16 tensors and products: what autocast(bfloat16) itself
    computes) or, for float16, "f16" (f16 products on fp32 tensors) -- when the split path is active; else "fp32"."""
    if precision != "auto":
        return precision
    if torch.is_autocast_enabled("cuda") and conv_path() == "split":
        dt = torch.get_autocast_dtype("cuda")
        if dt == torch.bfloat16:
            return "bf16"
        if dt == torch.float16:
            return "f16"
    return "fp32"


def split_pack_many(jobs) -> list:
    """``split_pack`` for a list of ``(weight, mode, block)`` in two launches (``srgen_split_pack_many``): the images, in
    order, as views of one fp16 buffer."""
    ws = []
    for w, mode, block in jobs:
        if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 4 or tuple(w.shape[1:]) != (96, 3, 3):
            raise RuntimeError(f"expected a ROCm fp32 (cout, 96, 3, 3) weight, got {tuple(w.shape)} on {w.device}")
        ws.append(w.detach().contiguous())
    dev = ws[0].device
    sizes = [(w.shape[0] // 96 if mode == 0 else 1) * SPLIT_HALFS + SPLIT_TAIL_HALFS for w, (_, mode, _) in zip(ws, jobs)]
    buf = torch.empty(sum(sizes), dtype=torch.float16, device=dev)         # every image size is a multiple of 16 bytes
    outs, off = [], 0
    for n in sizes:
        outs.append(buf[off:off + n])
        off += n
    done = 0
    with torch.cuda.device(dev):
        while done < len(jobs):                                            # the C entry point takes up to 96 jobs
            k = min(96, len(jobs) - done)
            sl = slice(done, done + k)
            vp, ci = ctypes.c_void_p, ctypes.c_int
            _lib.check(_lib.lib().srgen_split_pack_many(
                k, (vp * k)(*[w.data_ptr() for w in ws[sl]]), (ci * k)(*[w.shape[0] for w in ws[sl]]),
                (ci * k)(*[m for _, m, _ in jobs[sl]]), (ci * k)(*[b for _, _, b in jobs[sl]]),
                (vp * k)(*[o.data_ptr() for o in outs[sl]]), _stream(ws[0])))
            done += k
    return outs


def bf16_pack_many(jobs) -> list:
    """Weight images of the bf16 3x3 kernel for a list of ``(weight, mode, block)`` in one launch
    (``srgen_bf16_pack_many``), as views of one bf16 buffer."""
    ws = []
    for w, mode, block in jobs:
        if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 4 or tuple(w.shape[1:]) != (96, 3, 3):
            raise RuntimeError(f"expected a ROCm fp32 (cout, 96, 3, 3) weight, got {tuple(w.shape)} on {w.device}")
        ws.append(w.detach().contiguous())
    dev = ws[0].device
    sizes = [(w.shape[0] // 96 if mode == 0 else 1) * BF16_IMAGE_ELEMS for w, (_, mode, _) in zip(ws, jobs)]
    buf = torch.empty(sum(sizes), dtype=torch.bfloat16, device=dev)
    outs, off = [], 0
    for n in sizes:
        outs.append(buf[off:off + n])
        off += n
    done = 0
    with torch.cuda.device(dev):
        while done < len(jobs):
            k = min(96, len(jobs) - done)
            sl = slice(done, done + k)
            vp, ci = ctypes.c_void_p, ctypes.c_int
            _lib.check(_lib.lib().srgen_bf16_pack_many(
                k, (vp * k)(*[w.data_ptr() for w in ws[sl]]), (ci * k)(*[w.shape[0] for w in ws[sl]]),
                (ci * k)(*[m for _, m, _ in jobs[sl]]), (ci * k)(*[b for _, _, b in jobs[sl]]),
                (vp * k)(*[o.data_ptr() for o in outs[sl]]), _stream(ws[0])))
            done += k
    return outs


# Weight images made ahead of a step: inside ``packed_weights(jobs)`` ``conv_pack`` serves the images of the listed (weight,
# mode, block) from ONE batched pack (two launches instead of two per image: 146 launches of ~5 us in a pre-train step).
# The weights must not change inside the block (they do not: the optimizer steps after the backward).
_PACKED = {}


@contextlib.contextmanager
def packed_weights(jobs, outs=None):
    """``outs``: images a former ``packed_weights(jobs)`` block made (it yields them) -- installed again without packing,
    e.g. by a backward pass that runs after the block of its forward has ended."""
    if conv_path() != "split" or not jobs:
        yield None
        return
    if outs is None:
        outs = bf16_pack_many(jobs) if _CONV_PRECISION[0] == "bf16" else split_pack_many(jobs)
    old = dict(_PACKED)
    for (w, mode, block), o in zip(jobs, outs):
        _PACKED[(w.data_ptr(), mode, block)] = o
    try:
        yield outs
    finally:
        _PACKED.clear()
        _PACKED.update(old)


def conv_pack(w: torch.Tensor, mode: int = 0, block: int = 0) -> torch.Tensor:
    """The weight image ``conv3x3`` wants for the selected kernel (``conv_path``); inside ``packed_weights`` the image
    made there."""
    hit = _PACKED.get((w.data_ptr(), mode, block))
    if hit is not None:
        return hit
    if _CONV_PRECISION[0] == "bf16":
        return bf16_pack_many([(w, mode, block)])[0]
    return split_pack(w, mode, block) if conv_path() == "split" else wino_pack(w, mode, block)


def absmax(x: torch.Tensor) -> torch.Tensor:
    """``AMAX_SLOTS`` partial maxima of |x| on the device (their maximum is max |x|): what the split conv kernel takes
    as the bound of a gradient input.  One launch, no host synchronisation."""
    if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous():
        raise RuntimeError("absmax: a contiguous fp32 ROCm tensor; there is no CPU fallback")
    out = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().srgen_absmax(x.data_ptr(), x.numel(), out.data_ptr(), _stream(x)))
    return out


@_timed("conv3x3 forward / data gradient", _conv_flops)
def conv3x3(x: torch.Tensor, wpack: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
            cout_blocks: int = 1, shuffle: bool = False, scale_input: bool = False) -> torch.Tensor:
    """``conv(x) + bias [+ res]`` on NHWC fp32 (``F.conv2d(..., padding=1)``); with a mode-1 image: the data gradient.
    The image's dtype picks the kernel: fp16 (``split_pack``) the split-f16 one, fp32 (``wino_pack``) Winograd.
    ``scale_input``: the input is a gradient (any magnitude): the split kernel scales it by a power of two taken from
    ``absmax(x)`` before it forms the f16 parts (activations of order 1 need none; Winograd ignores the flag)."""
    _chk_nhwc("x", x)
    n, hw = x.shape[0], x.shape[1]
    oh = 2 * hw if shuffle else hw
    if wpack.dtype == torch.bfloat16:                              # the 16-bit step: bf16 tensors, the eval forward's bf16 kernel
        if not _bf(x) or (res is not None and not _bf(res)):
            raise RuntimeError("a bf16 weight image multiplies bf16 tensors (conv_precision('bf16'))")
        if wpack.numel() != cout_blocks * BF16_IMAGE_ELEMS or not wpack.is_contiguous():
            raise RuntimeError(f"weight image of {wpack.numel()} bf16 values does not fit {cout_blocks} cout block(s)")
        out = torch.empty((n, oh, oh, 96), dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().srgen_conv3x3_nhwc_bf16(x.data_ptr(), wpack.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                          res.data_ptr() if res is not None else None, out.data_ptr(), n, hw,
                                                          cout_blocks, 1 if shuffle else 0, _stream(x)))
        return out
    if _bf(x):
        raise RuntimeError("a bf16 tensor needs a bf16 weight image (conv_precision('bf16'))")
    split = wpack.dtype == torch.float16
    want = cout_blocks * SPLIT_HALFS + SPLIT_TAIL_HALFS if split else cout_blocks * WINO_FLOATS
    if wpack.numel() != want or not wpack.is_contiguous():
        raise RuntimeError(f"weight image of {wpack.numel()} {wpack.dtype} values does not fit {cout_blocks} cout block(s)")
    out = torch.empty((n, oh, oh, 96), dtype=torch.float32, device=x.device)
    pb = bias.data_ptr() if bias is not None else None
    pr = res.data_ptr() if res is not None else None
    with torch.cuda.device(x.device):
        if split:
            amax = None
            if scale_input:                                       # the producer's partial maxima, or a pass of our own
                amax = getattr(x, "amax_slots", None)
                if amax is None:
                    amax = absmax(x)
            op = _lib.lib().srgen_conv3x3_nhwc_f16 if _CONV_PRECISION[0] == "f16" else _lib.lib().srgen_conv3x3_nhwc_split
            _lib.check(op(x.data_ptr(), wpack.data_ptr(), pb, pr, out.data_ptr(), n, hw, cout_blocks, 1 if shuffle else 0,
                          amax.data_ptr() if amax is not None else None, _stream(x)))
        else:
            _lib.check(_lib.lib().srgen_conv3x3_nhwc(x.data_ptr(), wpack.data_ptr(), pb, pr, out.data_ptr(), n, hw,
                                                     cout_blocks, 1 if shuffle else 0, _stream(x)))
    return out


def _out(t: Optional[torch.Tensor], shape, dev) -> torch.Tensor:
    """``t`` (a contiguous fp32 gradient view to write into, e.g. ``p.grad``) or a fresh tensor of ``shape``."""
    if t is None:
        return torch.empty(shape, dtype=torch.float32, device=dev)
    if not t.is_contiguous() or t.dtype != torch.float32 or t.numel() != int(torch.Size(shape).numel()):
        raise RuntimeError(f"gradient output must be a contiguous fp32 tensor of {tuple(shape)}, got {tuple(t.shape)} {t.dtype}")
    return t


# ---- weight gradients on a second HIP stream ------------------------------------------------------------------------------
# In a backward pass the weight gradient of a conv and everything downstream of its data gradient are independent, and
# both chains leave most of the chip idle (staging- and latency-bound kernels, HBM-bound BatchNorm passes).  Inside
# ``wgrad_stream(side)`` every ``conv3x3_wgrad`` call is enqueued on ``side`` (they share one scratch slab, so they must
# stay in one queue) behind what the current stream holds so far.  A call that writes into caller-owned outputs (``p.grad``
# views) lets the current stream go on with the data gradient at once; a call that returns fresh tensors makes the current
# stream wait for them.  The streams are joined when the block ends.  Same kernels, same order per buffer: same results.
_WGRAD_SIDE = [None]


@contextlib.contextmanager
def wgrad_stream(side: Optional["torch.cuda.Stream"]):
    old = _WGRAD_SIDE[0]
    _WGRAD_SIDE[0] = side
    try:
        yield
    finally:
        _WGRAD_SIDE[0] = old
        if side is not None:
            torch.cuda.current_stream(side.device).wait_stream(side)


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, ws: Workspace, dw: Optional[torch.Tensor] = None,
                  db: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``_conv3x3_wgrad`` -- on the side stream of ``wgrad_stream`` when the outputs are the caller's."""
    side = _WGRAD_SIDE[0]
    if side is None:
        return _conv3x3_wgrad(x, dy, ws, dw, db)
    fresh = dw is None or db is None
    dw = _out(dw, (96, 96, 3, 3), x.device)                         # (allocated on the caller's stream)
    db = _out(db, (96,), x.device)
    main = torch.cuda.current_stream(x.device)
    side.wait_stream(main)                                          # dy (and its partial maxima) are ready behind this
    with torch.cuda.stream(side):
        out = _conv3x3_wgrad(x, dy, ws, dw, db)
    for t in (x, dy, getattr(dy, "amax_slots", None)):              # the allocator must not hand these out before `side` is done
        if t is not None:
            t.record_stream(side)
    if fresh:
        main.wait_stream(side)                                      # the caller reads dw / db next
    return out


@_timed("conv3x3 weight gradient", lambda x, *a, **k: 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * 96 * 96 * 9)
def _conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, ws: Workspace, dw: Optional[torch.Tensor] = None,
                   db: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dw (96, 96, 3, 3), dbias (96,)) of ``F.conv2d(x, w, b, padding=1)`` given ``dy``; NHWC fp32 operands.  ``dw`` /
    ``db``: write into these (``p.grad`` views) instead of fresh tensors."""
    _chk_nhwc("x", x)
    _chk_nhwc("dy", dy, x)
    n, hw = x.shape[0], x.shape[1]
    L = _lib.lib()
    need = ctypes.c_size_t()
    _lib.check(L.srgen_conv3x3_wgrad_scratch_bytes(n, hw, ctypes.byref(need)))
    scratch = ws.get("wgrad", need.value)
    dw = _out(dw, (96, 96, 3, 3), x.device)
    db = _out(db, (96,), x.device)
    with torch.cuda.device(x.device):
        if _bf(x):
            _lib.check(L.srgen_conv3x3_wgrad_bf16(x.data_ptr(), dy.data_ptr(), n, hw, dw.data_ptr(), db.data_ptr(),
                                                  scratch.data_ptr(), scratch.numel(), _stream(x)))
        elif wgrad_path() == "split" and hw % 64 == 0:
            # the split kernel scales the gradient into f16's range: its producer's partial maxima, or a pass of our own
            amax = getattr(dy, "amax_slots", None)
            if amax is None:
                amax = absmax(dy)
            op = L.srgen_conv3x3_wgrad_f16 if _CONV_PRECISION[0] == "f16" else L.srgen_conv3x3_wgrad_split
            _lib.check(op(x.data_ptr(), dy.data_ptr(), n, hw, amax.data_ptr(), dw.data_ptr(), db.data_ptr(),
                          scratch.data_ptr(), scratch.numel(), _stream(x)))
        else:
            _lib.check(L.srgen_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), n, hw, dw.data_ptr(), db.data_ptr(),
                                             scratch.data_ptr(), scratch.numel(), _stream(x)))
    return dw, db


def _bn_scratch(ws: Workspace) -> torch.Tensor:
    need = ctypes.c_size_t()
    _lib.check(_lib.lib().srgen_bn_train_scratch_bytes(ctypes.byref(need)))
    return ws.get("bn", need.value)


@_timed("BatchNorm forward (statistics + apply)", _bytes_of(3))
def bn_train_fwd(c: torch.Tensor, gamma, beta, ws: Workspace, slope=None, res=None, res_factor: float = 0.0):
    """(out, batch_stats): ``res + res_factor * PReLU(bn(c))`` with batch statistics (``nn.BatchNorm2d`` in train mode)."""
    _chk_nhwc("c", c)
    out = torch.empty_like(c)
    stats = torch.empty((2, 96), dtype=torch.float32, device=c.device)
    scratch = _bn_scratch(ws)
    npix = c.numel() // 96
    if res is not None:
        _chk_nhwc("res", res, c)
    op = _lib.lib().srgen_bn_train_fwd_bf16 if _bf(c) else _lib.lib().srgen_bn_train_fwd
    with torch.cuda.device(c.device):
        _lib.check(op(c.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                 slope.data_ptr() if slope is not None else None,
                                                 res.data_ptr() if res is not None else None, float(res_factor),
                                                 stats.data_ptr(), npix, scratch.data_ptr(), scratch.numel(), _stream(c)))
    return out, stats


@_timed("BatchNorm backward", _bytes_of(5))
def bn_train_bwd(c, dout, stats, gamma, beta, ws: Workspace, slope=None, dout_scale: float = 1.0, dgamma=None, dbeta=None,
                 dslope=None):
    """(dc, dgamma, dbeta, dslope | None) of ``PReLU(bn(c))`` in train mode given ``dout * dout_scale``.  ``dgamma`` /
    ``dbeta`` / ``dslope``: write into these (``p.grad`` views) instead of fresh tensors."""
    _chk_nhwc("c", c)
    _chk_nhwc("dout", dout, c)
    dev = c.device
    dc = torch.empty_like(c)
    dgamma, dbeta = _out(dgamma, (96,), dev), _out(dbeta, (96,), dev)
    dslope = _out(dslope, (96,), dev) if slope is not None else None
    scratch = _bn_scratch(ws)
    # the split-f16 data-gradient conv that reads dc next wants a bound of |dc| (conv3x3: scale_input): the apply kernel
    # leaves its partial maxima here, carried on the tensor as ``dc.amax_slots``
    amax = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=dev) if conv_path() == "split" and not _bf(c) else None
    with torch.cuda.device(dev):
        if _bf(c):
            _lib.check(_lib.lib().srgen_bn_train_bwd_bf16(c.data_ptr(), dout.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                                                          beta.data_ptr(), slope.data_ptr() if slope is not None else None,
                                                          float(dout_scale), dc.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                          dslope.data_ptr() if dslope is not None else None, c.numel() // 96,
                                                          scratch.data_ptr(), scratch.numel(), _stream(c)))
            return dc, dgamma, dbeta, dslope
        _lib.check(_lib.lib().srgen_bn_train_bwd(c.data_ptr(), dout.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                                                 beta.data_ptr(), slope.data_ptr() if slope is not None else None,
                                                 float(dout_scale), dc.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                 dslope.data_ptr() if dslope is not None else None, c.numel() // 96,
                                                 scratch.data_ptr(), scratch.numel(),
                                                 amax.data_ptr() if amax is not None else None, _stream(c)))
    if amax is not None:
        dc.amax_slots = amax
    return dc, dgamma, dbeta, dslope


@_timed("loss gradient")
def pixel_loss_grad(sr: torch.Tensor, hr: torch.Tensor, ws: Workspace, w_l1: float = 1.0, w_ssim: float = 0.1) -> torch.Tensor:
    """d/dsr of ``w_l1 * weighted_l1(sr, hr) + w_ssim * (1 - ssim(sr, hr))`` (train.py:93; 0.3 / 1.0 at train.py:170-172)."""
    for name, t in (("sr", sr), ("hr", hr)):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
            raise RuntimeError(f"{name}: expected a ROCm fp32 (n, c, h, w) tensor; there is no CPU fallback")
    if sr.shape != hr.shape:
        raise RuntimeError(f"sr {tuple(sr.shape)} and hr {tuple(hr.shape)} differ")
    src, hrc = sr.detach().contiguous(), hr.detach().contiguous()
    n, c, h, w = src.shape
    scratch = ws.get("loss", 3 * src.numel() * 4)
    dsr = torch.empty_like(src)
    with torch.cuda.device(src.device):
        _lib.check(_lib.lib().srgen_pixel_loss_grad(src.data_ptr(), hrc.data_ptr(), n, c, h, w, float(w_l1), float(w_ssim),
                                                    dsr.data_ptr(), scratch.data_ptr(), scratch.numel(), _stream(src)))
    return dsr


class FlatAdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` (train.py:270: lr 1e-4, betas (0.9, 0.999), eps 1e-8, weight decay 0.01) over ONE flat fp32
    buffer: the parameters are re-pointed at views of it, so one kernel launch updates every tensor and the data-parallel
    gradient all-reduce of the training configurations is one collective over ``flat_grad`` (12.86 MB for the generator).
    A ``torch.optim.Optimizer`` with one parameter group, so the reference's ``ReduceLROnPlateau`` (train.py:76-79) and
    ``optimizer.param_groups[0]['lr']`` (train.py:106) work on it unchanged; ``state_dict`` holds the flat moments."""

    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01):
        plist = [p for p in params]
        if not plist or not all(p.is_cuda and p.dtype == torch.float32 for p in plist):
            raise RuntimeError("FlatAdamW needs fp32 parameters on a ROCm device; there is no CPU fallback")
        super().__init__(plist, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.params = plist
        dev = plist[0].device
        n = sum(p.numel() for p in plist)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        off = 0
        self._grad_views = []
        with torch.no_grad():
            for p in plist:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.flat_grad[off:off + k].view(p.shape)
                self._grad_views.append(p.grad)
                off += k
        self.steps = 0

    def zero_grad(self, set_to_none: bool = False) -> None:      # the gradients are views of flat_grad: never None
        self.flat_grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FlatAdamW.step takes no closure")
        g = self.param_groups[0]
        self.steps += 1
        with torch.cuda.device(self.flat.device):
            _lib.check(_lib.lib().srgen_adamw(self.flat.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), self.flat.numel(), float(g["lr"]),
                                              float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                              float(g["weight_decay"]), self.steps, _stream(self.flat)))
        # the kernel wrote the parameters behind torch's back: bump their version counters, which is what the Generator /
        # Discriminator mirrors watch to re-fold and re-upload their weights before the next eval forward
        torch.autograd.graph.increment_version(self.params)

    def restore_grad_views(self) -> None:
        """Re-point every ``p.grad`` at its view of ``flat_grad`` where it is something else (``nn.Module.zero_grad()`` /
        ``zero_grad(set_to_none=True)`` of a stock optimizer set them to None; a gradient tensor someone assigned is
        copied in).  One identity check per parameter when nothing changed."""
        views = self._grad_views
        if all(p.grad is v for p, v in zip(self.params, views)):
            return
        for p, v in zip(self.params, views):
            if p.grad is not v:
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v

    def state_dict(self) -> dict:
        """``torch.optim.AdamW.state_dict()``'s layout -- per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` under
        ``state`` and the hyper-parameters under ``param_groups`` -- so that the ``opt_g`` / ``opt_d`` entries of the
        reference's ``epoch_XXX.pt`` checkpoints (train.py:219-225) can be exchanged with a stock AdamW in both directions."""
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(self.params)))
        state, off = {}, 0
        if self.steps > 0:
            for i, p in enumerate(self.params):
                k = p.numel()
                state[i] = {"step": torch.tensor(float(self.steps)),
                            "exp_avg": self.exp_avg[off:off + k].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + k].view(p.shape).clone()}
                off += k
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd: dict) -> None:
        """Takes ``torch.optim.AdamW.state_dict()`` (and what ``state_dict`` above returns); the flat layout of earlier
        versions of this class ({'step', 'exp_avg', 'exp_avg_sq', 'param_groups'}) is still read."""
        if "state" not in sd:                                            # (the former private layout)
            if sd["exp_avg"].numel() != self.flat.numel():
                raise ValueError(f"optimizer state has {sd['exp_avg'].numel()} entries, the parameters {self.flat.numel()}")
            self.steps = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.param_groups[0].update({k: v for k, v in sd["param_groups"][0].items() if k != "params"})
            return
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(groups) != 1 or len(ids) != len(self.params):
            raise ValueError(f"optimizer state describes {len(ids)} parameters in {len(groups)} group(s); this optimizer "
                             f"holds {len(self.params)} in one")
        state = sd["state"]
        steps = {int(float(st["step"])) for st in state.values()}
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): one flat AdamW has one step count")
        self.steps = steps.pop() if steps else 0
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        off = 0
        for i, p in zip(ids, self.params):
            k = p.numel()
            st = state.get(i)
            if st is not None:
                if st["exp_avg"].numel() != k:
                    raise ValueError(f"optimizer state of parameter {i} has {st['exp_avg'].numel()} entries, the parameter {k}")
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            off += k
        self.param_groups[0].update({k: v for k, v in groups[0].items() if k != "params"})


# ---- the residual block (Generator.py:30-42) in train mode: forward with saved tensors, backward ----------------------
def resblock_forward(x: torch.Tensor, block: torch.nn.Module, ws: Workspace):
    """``x + 0.1 * block(x)`` with batch-statistics BatchNorm.  ``block`` is a ``ResBlock`` parameter container
    (keys ``block.{0,3}.{weight,bias}``, ``block.{1,4}.{weight,bias}``, ``block.2.weight``).  Returns the output and
    what the backward needs: the two raw conv outputs, conv2's input, and the batch statistics."""
    seq = block.block
    c1 = conv3x3(x, conv_pack(seq[0].weight), seq[0].bias.detach())
    t, st1 = bn_train_fwd(c1, seq[1].weight.detach(), seq[1].bias.detach(), ws, slope=seq[2].weight.detach())
    c2 = conv3x3(t, conv_pack(seq[3].weight), seq[3].bias.detach())
    out, st2 = bn_train_fwd(c2, seq[4].weight.detach(), seq[4].bias.detach(), ws, res=x, res_factor=0.1)
    return out, {"x": x, "c1": c1, "t": t, "c2": c2, "st1": st1, "st2": st2}


def resblock_backward(dout: torch.Tensor, saved: dict, block: torch.nn.Module, ws: Workspace, into: Optional[dict] = None):
    """(dx, {state_dict key -> gradient}) given ``dout`` = gradient of the block output.  ``into``: {key -> tensor} the
    gradients are written to directly (the ``p.grad`` views of the step) instead of fresh tensors."""
    seq = block.block
    g: Dict[str, torch.Tensor] = {}
    d = (lambda k: into[k]) if into is not None else (lambda k: None)
    # out = x + 0.1 * bn2(c2)
    dc2, g["block.4.weight"], g["block.4.bias"], _ = bn_train_bwd(saved["c2"], dout, saved["st2"], seq[4].weight.detach(),
                                                                   seq[4].bias.detach(), ws, dout_scale=0.1,
                                                                   dgamma=d("block.4.weight"), dbeta=d("block.4.bias"))
    g["block.3.weight"], g["block.3.bias"] = conv3x3_wgrad(saved["t"], dc2, ws, d("block.3.weight"), d("block.3.bias"))
    dt = conv3x3(dc2, conv_pack(seq[3].weight, mode=1), scale_input=True)
    # t = PReLU(bn1(c1))
    dc1, g["block.1.weight"], g["block.1.bias"], g["block.2.weight"] = bn_train_bwd(
        saved["c1"], dt, saved["st1"], seq[1].weight.detach(), seq[1].bias.detach(), ws, slope=seq[2].weight.detach(),
        dgamma=d("block.1.weight"), dbeta=d("block.1.bias"), dslope=d("block.2.weight"))
    g["block.0.weight"], g["block.0.bias"] = conv3x3_wgrad(saved["x"], dc1, ws, d("block.0.weight"), d("block.0.bias"))
    dx = conv3x3(dc1, conv_pack(seq[0].weight, mode=1), res=dout, scale_input=True)          # + the identity branch
    return dx, g


# ---- remaining layer types ------------------------------------------------------------------------------------------
@_timed("scalar PReLU", _bytes_of(2))
def prelu_scalar(x: torch.Tensor, slope: torch.Tensor) -> torch.Tensor:
    """``nn.PReLU()`` with one slope (Generator.py:51, 9, 70), out of place (the pre-activation is kept for the backward)."""
    y = torch.empty_like(x)
    op = _lib.lib().srgen_prelu_scalar_fwd_bf16 if _bf(x) else _lib.lib().srgen_prelu_scalar_fwd
    with torch.cuda.device(x.device):
        _lib.check(op(x.data_ptr(), slope.data_ptr(), y.data_ptr(), x.numel(), _stream(x)))
    return y


@_timed("scalar PReLU", _bytes_of(3))
def prelu_scalar_bwd(x, dy, slope, ws: Workspace, dy2=None, unshuffle: bool = False):
    """(dx, dslope (1,)) given the pre-activation ``x`` and ``dy`` (+ ``dy2``, a second consumer's gradient).
    ``unshuffle`` (the upsample block: PixelShuffle, then PReLU): ``dx`` comes back as (2, 2, n, h/2, w/2, 96) with
    ``dx[i, j, n, y, x] =`` the gradient at pixel (2y + i, 2x + j) -- the PixelShuffle's backward folded into the store."""
    need = ctypes.c_size_t()
    _lib.check(_lib.lib().srgen_prelu_scratch_bytes(ctypes.byref(need)))
    scratch = ws.get("prelu", need.value)
    dslope = torch.empty(1, dtype=torch.float32, device=x.device)
    if dy.dtype != x.dtype or (dy2 is not None and dy2.dtype != x.dtype):
        raise RuntimeError("prelu_scalar_bwd: operands of one storage type")
    L = _lib.lib()
    if unshuffle:
        n, h, w, c = x.shape
        if dy2 is not None or h != w or (h & 1) or c != 96 or not x.is_contiguous() or not dy.is_contiguous():
            raise RuntimeError("prelu_scalar_bwd(unshuffle=True): contiguous (n, 2k, 2k, 96) operands, no second gradient")
        dx = torch.empty((2, 2, n, h // 2, w // 2, 96), dtype=x.dtype, device=x.device)
        op = L.srgen_prelu_scalar_bwd_unshuffle_bf16 if _bf(x) else L.srgen_prelu_scalar_bwd_unshuffle
        with torch.cuda.device(x.device):
            _lib.check(op(x.data_ptr(), dy.data_ptr(), slope.data_ptr(), dx.data_ptr(), dslope.data_ptr(), n, h,
                          scratch.data_ptr(), scratch.numel(), _stream(x)))
        return dx, dslope
    dx = torch.empty_like(x)
    op = L.srgen_prelu_scalar_bwd_bf16 if _bf(x) else L.srgen_prelu_scalar_bwd
    with torch.cuda.device(x.device):
        _lib.check(op(x.data_ptr(), dy.data_ptr(), dy2.data_ptr() if dy2 is not None else None,
                                                     slope.data_ptr(), dx.data_ptr(), dslope.data_ptr(), x.numel(),
                                                     scratch.data_ptr(), scratch.numel(), _stream(x)))
    return dx, dslope


@_timed("9x9 stem / head")
def conv9x9_c3(img: torch.Tensor, wm: torch.Tensor, bias, out_hw: int, flip: bool,
               out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """out (n, out_hw, out_hw, 96) NHWC = bias + sum over the 243 taps of a 3-channel NCHW image times ``wm`` [243][96].
    ``out_dtype`` bfloat16 (the 16-bit step): products of the operands rounded to f16 (one MFMA per product: the arithmetic
    class of the reference's autocast convs, 11 significant bits instead of bf16's 8), fp32 sums rounded once when stored."""
    n, c, ih, iw = img.shape
    assert c == 3 and img.is_contiguous() and wm.shape == (243, 96) and wm.is_contiguous()
    out = torch.empty((n, out_hw, out_hw, 96), dtype=out_dtype, device=img.device)
    pb = bias.data_ptr() if bias is not None else None
    with torch.cuda.device(img.device):
        if out_dtype == torch.bfloat16:
            if conv_path() != "split" or out_hw % 64:
                raise RuntimeError("the bf16-output 9x9 conv exists on the split path, for outputs that are multiples of 64")
            pack = torch.empty(STEM_SPLIT_BYTES, dtype=torch.uint8, device=img.device)
            _lib.check(_lib.lib().srgen_conv9x9_c3_nhwc_split_bf16(img.data_ptr(), wm.data_ptr(), pb, out.data_ptr(), n, ih, iw,
                                                                   out_hw, out_hw, 1 if flip else 0, pack.data_ptr(), pack.numel(),
                                                                   _stream(img)))
        elif conv_path() == "split" and out_hw % 64 == 0:
            # the eval stem's kernel (three f16 MFMAs per product, the input tile scaled by its own maximum, so gradients
            # of any magnitude keep fp32 accuracy); its weight image is made from wm on the device
            pack = torch.empty(STEM_SPLIT_BYTES, dtype=torch.uint8, device=img.device)
            _lib.check(_lib.lib().srgen_conv9x9_c3_nhwc_split(img.data_ptr(), wm.data_ptr(), pb, out.data_ptr(), n, ih, iw, out_hw,
                                                              out_hw, 1 if flip else 0, pack.data_ptr(), pack.numel(), _stream(img)))
        else:
            _lib.check(_lib.lib().srgen_conv9x9_c3_nhwc(img.data_ptr(), wm.data_ptr(), pb, out.data_ptr(), n, ih, iw, out_hw, out_hw,
                                                        1 if flip else 0, _stream(img)))
    return out


@_timed("9x9 stem / head")
def conv9x9_c3_wgrad(act: torch.Tensor, img: torch.Tensor, ws: Workspace, flip: bool, want_bias: bool):
    """(r (96, 243), dbias (96,) | None): r[c][j] = sum over pixels act[..., c] * tap_j(img)."""
    _chk_nhwc("act", act)
    n, ah = act.shape[0], act.shape[1]
    _, c, ih, iw = img.shape
    assert c == 3 and img.is_contiguous()
    L = _lib.lib()
    need = ctypes.c_size_t()
    _lib.check(L.srgen_conv9x9_c3_wgrad_scratch_bytes(n, ah, ctypes.byref(need)))
    scratch = ws.get("wgrad9", need.value)
    r = torch.empty((96, 243), dtype=torch.float32, device=act.device)
    db = torch.empty(96, dtype=torch.float32, device=act.device) if want_bias else None
    op = L.srgen_conv9x9_c3_wgrad_bf16 if _bf(act) else L.srgen_conv9x9_c3_wgrad
    with torch.cuda.device(act.device):
        _lib.check(op(act.data_ptr(), img.data_ptr(), n, ah, ah, ih, iw, 1 if flip else 0, r.data_ptr(),
                                            db.data_ptr() if db is not None else None, scratch.data_ptr(), scratch.numel(),
                                            _stream(act)))
    return r, db


def head_pack(w: torch.Tensor) -> torch.Tensor:
    """Weight image of the head kernel from the (3, 96, 9, 9) weight, on the device: float2 units
    [ky 9][c8 12][u 2][h 2][n 32] with ci = 8 c8 + 4 u + 2 h + e and n = 3 kx + co (27 of 32 used).  Pure data movement."""
    t = w.detach().permute(2, 1, 3, 0).reshape(9, 12, 2, 2, 2, 27)            # ky, (c8, u, h, e), (kx, co)
    t = t.permute(0, 1, 2, 3, 5, 4)                                             # ky, c8, u, h, n, e
    out = torch.zeros((9, 12, 2, 2, 32, 2), dtype=torch.float32, device=w.device)
    out[:, :, :, :, :27, :] = t
    return out


def head_pack_split(w: torch.Tensor) -> torch.Tensor:
    """Weight image of the split-path head kernel (srgen_head_f32s.hip) from the (3, 96, 9, 9) weight, on the device:
    f16 units of 8 [chunk 6][ky 9][hi | lo][lane 64], lane = 32 h + n, ci = 16 chunk + 8 h + e, n = 3 kx + co, of
    w * 2^k with max |w| * 2^k in [2^13, 2^14) (split_exp of srgen_internal.h: frexp gives the same exponent), then the
    128-byte tail whose first float is max |w|."""
    wd = w.detach()
    amax = wd.abs().max()
    _, e = torch.frexp(amax)                                  # amax = m * 2^e, m in [0.5, 1): floor(log2 amax) = e - 1
    k = torch.where(amax > 0, 13 - (e - 1), torch.zeros_like(e)).clamp(-60, 60)
    ws = torch.ldexp(wd, k)
    hi = ws.to(torch.float16)
    lo = (ws - hi.to(torch.float32)).to(torch.float16)
    img = torch.zeros((6, 9, 2, 2, 32, 8), dtype=torch.float16, device=w.device)         # chunk, ky, hl, h, n, e
    for hl, part in enumerate((hi, lo)):
        t = part.permute(2, 1, 3, 0).reshape(9, 6, 2, 8, 27)                              # ky, chunk, h, e, (kx, co)
        img[:, :, hl, :, :27, :] = t.permute(1, 0, 2, 4, 3)                               # chunk, ky, h, n, e
    tail = torch.zeros(32, dtype=torch.float32, device=w.device)
    tail[0] = amax
    return torch.cat([img.reshape(-1), tail.view(torch.float16)])


def head_pack_bf16(w: torch.Tensor) -> torch.Tensor:
    """Weight image of the bf16 head kernel (srgen_kernels_bf16.hip) from the (3, 96, 9, 9) weight, on the device: bf16 units
    of 8 [ky 9][g 6][h 2][n 32], ci = 16 g + 8 h + e, n = 3 kx + co (27 of 32 used).  Data movement and one rounding."""
    t = w.detach().permute(2, 1, 3, 0).reshape(9, 6, 2, 8, 27)                 # ky, (g, h, e), (kx, co)
    img = torch.zeros((9, 6, 2, 32, 8), dtype=torch.bfloat16, device=w.device)
    img[:, :, :, :27, :] = t.permute(0, 1, 2, 4, 3).to(torch.bfloat16)
    return img


@_timed("9x9 stem / head")
def head_forward(v: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``F.conv2d(v, w, b, padding=4)[:, :, :125, :125]`` (Generator.py:71,80): NHWC (n, 128, 128, 96) -> NCHW (n, 3, 125, 125)."""
    _chk_nhwc("v", v)
    n = v.shape[0]
    y = torch.empty((n, 3, 125, 125), dtype=torch.float32, device=v.device)
    if _bf(v):
        pack = head_pack_bf16(w)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().srgen_head_nhwc_bf16(v.data_ptr(), pack.data_ptr(), b.detach().data_ptr(), y.data_ptr(), n,
                                                       _stream(v)))
        return y
    split = conv_path() == "split"
    pack = head_pack_split(w) if split else head_pack(w)
    fn = _lib.lib().srgen_head_nhwc_split if split else _lib.lib().srgen_head_nhwc
    with torch.cuda.device(v.device):
        _lib.check(fn(v.data_ptr(), pack.data_ptr(), b.detach().data_ptr(), y.data_ptr(), n, _stream(v)))
    return y


@_timed("channel attention", _bytes_of(3))
def chan_attn_forward(x: torch.Tensor, ca: torch.nn.Module):
    """``x * sigmoid(fc(mean_hw(x)))`` (Generator.py:15-27): (y, saved)."""
    _chk_nhwc("x", x)
    n = x.shape[0]
    y = x.clone()                                     # the kernel scales in place; the backward needs the unscaled input
    pool = torch.empty((n, 16, 96), dtype=torch.float32, device=x.device)
    scale = torch.empty((n, 96), dtype=torch.float32, device=x.device)
    fc = ca.fc
    op = _lib.lib().srgen_chan_attn_nhwc_bf16 if _bf(x) else _lib.lib().srgen_chan_attn_nhwc
    with torch.cuda.device(x.device):
        _lib.check(op(y.data_ptr(), fc[0].weight.data_ptr(), fc[0].bias.data_ptr(),
                                                   fc[2].weight.data_ptr(), fc[2].bias.data_ptr(), pool.data_ptr(),
                                                   scale.data_ptr(), n, _stream(x)))
    return y, {"x": x, "pool": pool, "scale": scale}


@_timed("channel attention", _bytes_of(3))
def chan_attn_backward(dy: torch.Tensor, saved: dict, ca: torch.nn.Module, ws: Workspace, into: Optional[dict] = None):
    x = saved["x"]
    n, dev = x.shape[0], x.device
    L = _lib.lib()
    need = ctypes.c_size_t()
    _lib.check(L.srgen_chan_attn_bwd_scratch_bytes(n, ctypes.byref(need)))
    scratch = ws.get("ca", need.value)
    fc = ca.fc
    dx = torch.empty_like(x)
    d = (lambda k: into[k]) if into is not None else (lambda k: None)
    g = {"fc.0.weight": _out(d("fc.0.weight"), (12, 96, 1, 1), dev), "fc.0.bias": _out(d("fc.0.bias"), (12,), dev),
         "fc.2.weight": _out(d("fc.2.weight"), (96, 12, 1, 1), dev), "fc.2.bias": _out(d("fc.2.bias"), (96,), dev)}
    if dy.dtype != x.dtype:
        raise RuntimeError("chan_attn_backward: operands of one storage type")
    op = L.srgen_chan_attn_bwd_bf16 if _bf(x) else L.srgen_chan_attn_bwd
    with torch.cuda.device(dev):
        _lib.check(op(x.data_ptr(), dy.data_ptr(), saved["pool"].data_ptr(), saved["scale"].data_ptr(), n,
                                         fc[0].weight.data_ptr(), fc[0].bias.data_ptr(), fc[2].weight.data_ptr(),
                                         fc[2].bias.data_ptr(), dx.data_ptr(), g["fc.0.weight"].data_ptr(),
                                         g["fc.0.bias"].data_ptr(), g["fc.2.weight"].data_ptr(), g["fc.2.bias"].data_ptr(),
                                         scratch.data_ptr(), scratch.numel(), _stream(x)))
    return dx, g


def allreduce_mean(buf: torch.Tensor, group=None) -> None:
    """Data-parallel gradient averaging: ONE all-reduce (SUM, then / world size) of a flat gradient buffer, enqueued on the
    caller's stream behind the backward pass that fills it -- the same code for RCCL ("nccl") and for host-staged backends
    (gloo stages through pinned memory on a stream of its own behind an event of the caller's stream).  No-op outside
    ``torch.distributed`` or with one rank."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        buf.div_(dist.get_world_size(group))


# ---- the whole generator: forward in train mode with saved tensors, backward, optimizer step ------------------------------
class GeneratorGrad:
    """Train-mode forward of the drop-in ``Generator`` with the tensors its backward needs, and that backward, composed from
    the HIP operators above (``Generator.py:45-80`` under ``train.py:92-94``).  ``backward`` writes every parameter gradient
    into tensors the caller names (``p.grad`` views of a flat buffer for ``PretrainStep``; a fresh flat buffer for the
    ``torch.autograd.Function`` behind ``Generator.forward``, superresolution_amd/autograd.py)."""

    def __init__(self, generator):
        p0 = next(generator.parameters())
        if not p0.is_cuda:
            raise RuntimeError("the training operators need the generator on a ROCm device; there is no CPU fallback")
        if generator.num_blocks != 16 or generator.base_ch != 96 or generator.in_ch != 3:
            raise NotImplementedError("the training operators are built for Generator(16, 3, 96)")
        self.g = generator
        self.ws = Workspace(p0.device)
        # the 3x3 weight gradients of a backward pass run on a stream of their own beside the data-gradient chain
        # (SRGEN_WGRAD_STREAM=0: everything on the caller's stream)
        self.side = torch.cuda.Stream(p0.device) if os.environ.get("SRGEN_WGRAD_STREAM", "1") != "0" else None

    def pack_jobs(self, backward: bool = True) -> list:
        """(weight, mode, block) of every 3x3 weight image a step uses: forward images, and the data-gradient images."""
        g = self.g
        ws = [m.block[i].weight for m in g.blocks if hasattr(m, "block") for i in (0, 3)] + [g.mid[0].weight, g.output[0].weight]
        jobs = [(w, 0, 0) for w in ws] + [(g.upsample.conv.weight, 0, 0)]
        if backward:
            jobs += [(w, 1, 0) for w in ws] + [(g.upsample.conv.weight, 1, b) for b in range(4)]
        return jobs

    # -- forward ------------------------------------------------------------------------------------------------------
    def forward(self, lr_img: torch.Tensor):
        g, ws = self.g, self.ws
        x = lr_img.detach().to(torch.float32).contiguous()
        S = {"lr": x, "blocks": []}
        stats = []                                                           # batch statistics in module order
        w = g.input[0].weight
        act_dtype = torch.bfloat16 if _CONV_PRECISION[0] == "bf16" else torch.float32
        S["x1_pre"] = conv9x9_c3(x, w.detach().permute(1, 2, 3, 0).reshape(243, 96).contiguous(), g.input[0].bias.detach(), 64, False,
                                 out_dtype=act_dtype)
        x1 = prelu_scalar(S["x1_pre"], g.input[1].weight.detach())
        cur = x1
        for m in g.blocks:
            if hasattr(m, "block"):
                cur, saved = resblock_forward(cur, m, ws)
                stats += [saved["st1"], saved["st2"]]
            else:
                cur, saved = chan_attn_forward(cur, m)
            S["blocks"].append(saved)
        S["mid_in"] = cur
        S["mid_c"] = conv3x3(cur, conv_pack(g.mid[0].weight), g.mid[0].bias.detach())
        m_out, S["mid_st"] = bn_train_fwd(S["mid_c"], g.mid[1].weight.detach(), g.mid[1].bias.detach(), ws, res=x1, res_factor=1.0)
        stats.append(S["mid_st"])
        S["m"] = m_out
        up = g.upsample
        bias_p = up.conv.bias.detach().view(96, 4).t().contiguous().view(-1)      # packed block b holds channels 4c + b
        S["u_pre"] = conv3x3(m_out, conv_pack(up.conv.weight), bias_p, cout_blocks=4, shuffle=True)
        S["u"] = prelu_scalar(S["u_pre"], up.act.weight.detach())
        S["v_pre"] = conv3x3(S["u"], conv_pack(g.output[0].weight), g.output[0].bias.detach())
        S["v"] = prelu_scalar(S["v_pre"], g.output[1].weight.detach())
        y = head_forward(S["v"], g.output[2].weight, g.output[2].bias)
        S["stats"] = stats
        return y, S

    # -- backward -----------------------------------------------------------------------------------------------------
    def backward(self, dsr: torch.Tensor, S: dict, G: Optional[Dict[str, torch.Tensor]] = None) -> None:
        """Writes the gradient of every parameter into ``G[name]`` (contiguous fp32 tensors of the parameters' shapes;
        default: the parameters' ``.grad``); on return the caller's stream holds all of them (the side stream is joined)."""
        if G is None:
            G = {}
            for k, p in self.g.named_parameters():
                if p.grad is None:
                    raise RuntimeError(f"{k}.grad is None: the step writes into the gradient views its optimizer made "
                                       "(zero_grad(set_to_none=True) on the module drops them; use optimizer.zero_grad())")
                G[k] = p.grad
        with wgrad_stream(self.side):
            self._backward(dsr, S, G)

    def _backward(self, dsr: torch.Tensor, S: dict, G: Dict[str, torch.Tensor]) -> None:
        g, ws = self.g, self.ws

        def put(key, grad):
            G[key].copy_(grad.view_as(G[key]))

        # head: 9x9, 96 -> 3 (Generator.py:71), crop folded in (d out is 125 x 125)
        w = g.output[2].weight.detach()
        side = _WGRAD_SIDE[0]
        if side is not None:                              # the head's weight gradient beside its data gradient
            side.wait_stream(torch.cuda.current_stream(dsr.device))
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            r, _ = conv9x9_c3_wgrad(S["v"], dsr, ws, flip=True, want_bias=False)      # r[ci][(co,ky,kx)]
            put("output.2.weight", r.view(96, 3, 9, 9).permute(1, 0, 2, 3).contiguous())
            put("output.2.bias", dsr.sum(dim=(0, 2, 3)))
        if side is not None:
            S["v"].record_stream(side)
            dsr.record_stream(side)
        dv = conv9x9_c3(dsr, w.permute(0, 2, 3, 1).reshape(243, 96).contiguous(), None, 128, True, out_dtype=S["v"].dtype)
        # output[0..1]: conv3x3 at 128 x 128 + scalar PReLU (Generator.py:69-70)
        dv_pre, ds = prelu_scalar_bwd(S["v_pre"], dv, g.output[1].weight.detach(), ws)
        put("output.1.weight", ds)
        conv3x3_wgrad(S["u"], dv_pre, ws, G["output.0.weight"], G["output.0.bias"])
        du = conv3x3(dv_pre, conv_pack(g.output[0].weight, mode=1), scale_input=True)
        # upsample: conv 96 -> 384, PixelShuffle(2), scalar PReLU (Generator.py:4-12)
        up = g.upsample
        planes, ds = prelu_scalar_bwd(S["u_pre"], du, up.act.weight.detach(), ws, unshuffle=True)     # [i][j] -> (n, 64, 64, 96)
        put("upsample.act.weight", ds)
        gw, gb = G["upsample.conv.weight"].view(96, 4, 96, 3, 3), G["upsample.conv.bias"].view(96, 4)
        dm = None
        for b in range(4):
            pl = planes[b >> 1, b & 1]
            dw, db = conv3x3_wgrad(S["m"], pl, ws)
            gw[:, b].copy_(dw)
            gb[:, b].copy_(db)
            dm = conv3x3(pl, conv_pack(up.conv.weight, mode=1, block=b), res=dm, scale_input=True)
        # mid conv + BatchNorm, long skip x1 + x3 (Generator.py:61-64,78)
        dc, _, _, _ = bn_train_bwd(S["mid_c"], dm, S["mid_st"], g.mid[1].weight.detach(), g.mid[1].bias.detach(), ws,
                                   dgamma=G["mid.1.weight"], dbeta=G["mid.1.bias"])
        conv3x3_wgrad(S["mid_in"], dc, ws, G["mid.0.weight"], G["mid.0.bias"])
        dcur = conv3x3(dc, conv_pack(g.mid[0].weight, mode=1), scale_input=True)
        # trunk, last module first (Generator.py:55-59)
        for idx in range(len(g.blocks) - 1, -1, -1):
            m, saved = g.blocks[idx], S["blocks"][idx]
            into = {k[len(f"blocks.{idx}."):]: t for k, t in G.items() if k.startswith(f"blocks.{idx}.")}
            if hasattr(m, "block"):                       # gradients are written straight into the caller's tensors
                dcur, _ = resblock_backward(dcur, saved, m, ws, into)
            else:
                dcur, _ = chan_attn_backward(dcur, saved, m, ws, into)
        # stem: x1 feeds the trunk and the long skip (gradient dm)
        dx1_pre, ds = prelu_scalar_bwd(S["x1_pre"], dcur, g.input[1].weight.detach(), ws, dy2=dm)
        put("input.1.weight", ds)
        if side is not None:                              # (the 9x9 scratch slab was last used on the side stream)
            torch.cuda.current_stream(dsr.device).wait_stream(side)
        r, db = conv9x9_c3_wgrad(dx1_pre, S["lr"], ws, flip=False, want_bias=True)
        put("input.0.weight", r)
        put("input.0.bias", db)

    def _update_running_stats(self, S: dict, n: int) -> None:
        bns = self.g._bn_modules()
        m = n * 64 * 64
        with torch.no_grad():
            means, varis = [bn.running_mean for bn in bns], [bn.running_var for bn in bns]
            torch._foreach_mul_(means, 0.9)
            torch._foreach_add_(means, [st[0] for st in S["stats"]], alpha=0.1)
            torch._foreach_mul_(varis, 0.9)
            torch._foreach_add_(varis, [st[1] for st in S["stats"]], alpha=0.1 * m / max(m - 1, 1))
            torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)



class PretrainStep(GeneratorGrad):
    """One pre-train step of the reference (``train.py:90-97``) for the drop-in ``Generator`` on one GPU:

        sr = generator(lr)                                  # train mode: batch-statistics BatchNorm (Generator.py:74-80)
        loss = weighted_l1(sr, hr) + 0.1 * (1 - ssim(sr, hr))
        loss.backward(); optimizer.step()                   # AdamW(lr 1e-4), train.py:270

    as ONE call with no autograd graph (the reference's own loop body also runs unchanged on the drop-in modules, through
    superresolution_amd/autograd.py; this object is the fused form: loss and loss gradient from two kernels, gradients
    written straight into one flat buffer).  Gradients land in ``p.grad`` (views of ``optimizer.flat_grad``, so the
    reference's ``grad_norm`` (train.py:23-28) and a data-parallel all-reduce of ONE 12.86 MB buffer work unchanged),
    BatchNorm running statistics are updated as ``nn.BatchNorm2d`` does.  ``precision`` picks the arithmetic: "fp32"
    (fp32 tensors, fp32-accurate 3x3 conv products), "f16" (fp32 tensors, 16-bit operands in the 3x3 convs, fp32
    accumulation), "bf16" (the 16-bit step: bf16 activation and gradient tensors, bf16 products, fp32 accumulation -- the
    storage and product types of the reference's own autocast step) or "auto" (default: the autocast dtype's arithmetic --
    "bf16" inside ``torch.autocast(dtype=torch.bfloat16)``, as train.py:91 calls the generator, "f16" inside float16 autocast, else
    "fp32"; see ``resolve_precision``).  With
    ``world_size > 1`` (one process per GPU, ``torch.distributed`` initialised) ``step`` averages ``flat_grad`` over the
    ranks with one all-reduce before the update, as DistributedDataParallel would."""

    def __init__(self, generator, lr: float = 1e-4, ssim_weight: float = 0.1, l1_weight: float = 1.0, group=None,
                 precision: str = "auto"):
        if precision not in ("auto", "fp32", "f16", "bf16"):
            raise RuntimeError(f"precision '{precision}': expected auto, fp32, f16 or bf16")
        self.precision = precision
        super().__init__(generator)
        self.optimizer = FlatAdamW(generator.parameters(), lr=lr)
        self.ssim_weight, self.l1_weight, self.group = ssim_weight, l1_weight, group
        self.last_metrics = None

    def forward_backward(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> torch.Tensor:
        """Fills every ``p.grad`` and returns the loss (0-dim device tensor)."""
        from .functional import pixel_losses
        if not self.g.training:
            raise RuntimeError("PretrainStep runs the generator in train mode (train.py:81): call generator.train()")
        hr = hr_img.detach().to(torch.float32).contiguous()
        self.optimizer.restore_grad_views()                # (a module-level zero_grad() may have dropped them)
        with conv_precision(resolve_precision(self.precision)), packed_weights(self.pack_jobs()):
            sr, S = self.forward(lr_img)
            vals = pixel_losses(sr, hr)                                     # [psnr, ssim, l1, weighted_l1]
            loss = self.l1_weight * vals[3] + self.ssim_weight * (1.0 - vals[1])
            dsr = pixel_loss_grad(sr, hr, self.ws, self.l1_weight, self.ssim_weight)
            self.backward(dsr, S)
        self._update_running_stats(S, sr.shape[0])
        self.last_metrics = vals
        return loss

    def step(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> torch.Tensor:
        self.optimizer.zero_grad()
        loss = self.forward_backward(lr_img, hr_img)
        allreduce_mean(self.optimizer.flat_grad, self.group)                    # 12.86 MB, one collective
        self.optimizer.step()
        return loss
