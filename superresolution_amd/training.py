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

import ctypes
from typing import Dict, Optional, Tuple

import torch

from . import _lib

WINO_FLOATS = 16 * 96 * 96           # Winograd weight image of one 96 -> 96 conv


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk_nhwc(name: str, t: torch.Tensor) -> None:
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 4 or t.shape[3] != 96 \
            or t.shape[1] != t.shape[2]:
        raise RuntimeError(f"{name}: expected a contiguous fp32 ROCm tensor (n, hw, hw, 96), got {tuple(t.shape)} "
                           f"{t.dtype} on {t.device}; there is no CPU fallback")


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


def conv3x3(x: torch.Tensor, wpack: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
            cout_blocks: int = 1, shuffle: bool = False) -> torch.Tensor:
    """``conv(x) + bias [+ res]`` on NHWC fp32 (``F.conv2d(..., padding=1)``); with a mode-1 image: the data gradient."""
    _chk_nhwc("x", x)
    n, hw = x.shape[0], x.shape[1]
    oh = 2 * hw if shuffle else hw
    out = torch.empty((n, oh, oh, 96), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().srgen_conv3x3_nhwc(x.data_ptr(), wpack.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                 res.data_ptr() if res is not None else None, out.data_ptr(), n, hw, cout_blocks,
                                                 1 if shuffle else 0, _stream(x)))
    return out


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, ws: Workspace) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dw (96, 96, 3, 3), dbias (96,)) of ``F.conv2d(x, w, b, padding=1)`` given ``dy``; NHWC fp32 operands."""
    _chk_nhwc("x", x)
    _chk_nhwc("dy", dy)
    n, hw = x.shape[0], x.shape[1]
    L = _lib.lib()
    need = ctypes.c_size_t()
    _lib.check(L.srgen_conv3x3_wgrad_scratch_bytes(n, hw, ctypes.byref(need)))
    scratch = ws.get("wgrad", need.value)
    dw = torch.empty((96, 96, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty(96, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.srgen_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), n, hw, dw.data_ptr(), db.data_ptr(), scratch.data_ptr(),
                                         scratch.numel(), _stream(x)))
    return dw, db


def _bn_scratch(ws: Workspace) -> torch.Tensor:
    need = ctypes.c_size_t()
    _lib.check(_lib.lib().srgen_bn_train_scratch_bytes(ctypes.byref(need)))
    return ws.get("bn", need.value)


def bn_train_fwd(c: torch.Tensor, gamma, beta, ws: Workspace, slope=None, res=None, res_factor: float = 0.0):
    """(out, batch_stats): ``res + res_factor * PReLU(bn(c))`` with batch statistics (``nn.BatchNorm2d`` in train mode)."""
    _chk_nhwc("c", c)
    out = torch.empty_like(c)
    stats = torch.empty((2, 96), dtype=torch.float32, device=c.device)
    scratch = _bn_scratch(ws)
    npix = c.numel() // 96
    with torch.cuda.device(c.device):
        _lib.check(_lib.lib().srgen_bn_train_fwd(c.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                 slope.data_ptr() if slope is not None else None,
                                                 res.data_ptr() if res is not None else None, float(res_factor),
                                                 stats.data_ptr(), npix, scratch.data_ptr(), scratch.numel(), _stream(c)))
    return out, stats


def bn_train_bwd(c, dout, stats, gamma, beta, ws: Workspace, slope=None, dout_scale: float = 1.0):
    """(dc, dgamma, dbeta, dslope | None) of ``PReLU(bn(c))`` in train mode given ``dout * dout_scale``."""
    _chk_nhwc("c", c)
    _chk_nhwc("dout", dout)
    dev = c.device
    dc = torch.empty_like(c)
    dgamma, dbeta = torch.empty(96, dtype=torch.float32, device=dev), torch.empty(96, dtype=torch.float32, device=dev)
    dslope = torch.empty(96, dtype=torch.float32, device=dev) if slope is not None else None
    scratch = _bn_scratch(ws)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().srgen_bn_train_bwd(c.data_ptr(), dout.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                                                 beta.data_ptr(), slope.data_ptr() if slope is not None else None,
                                                 float(dout_scale), dc.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                 dslope.data_ptr() if dslope is not None else None, c.numel() // 96,
                                                 scratch.data_ptr(), scratch.numel(), _stream(c)))
    return dc, dgamma, dbeta, dslope


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


class FlatAdamW:
    """``torch.optim.AdamW`` (train.py:270: lr 1e-4, betas (0.9, 0.999), eps 1e-8, weight decay 0.01) over ONE flat fp32
    buffer: the parameters are re-pointed at views of it, so one kernel launch updates every tensor and the data-parallel
    gradient all-reduce of the training configurations is one collective over ``flat_grad`` (12.86 MB for the generator)."""

    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01):
        self.params = [p for p in params]
        if not self.params or not all(p.is_cuda and p.dtype == torch.float32 for p in self.params):
            raise RuntimeError("FlatAdamW needs fp32 parameters on a ROCm device; there is no CPU fallback")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.flat_grad[off:off + k].view(p.shape)
                off += k
        self.lr, self.betas, self.eps, self.weight_decay, self.steps = lr, betas, eps, weight_decay, 0

    def zero_grad(self) -> None:
        self.flat_grad.zero_()

    def step(self) -> None:
        self.steps += 1
        with torch.cuda.device(self.flat.device):
            _lib.check(_lib.lib().srgen_adamw(self.flat.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), self.flat.numel(), float(self.lr),
                                              float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                              float(self.weight_decay), self.steps, _stream(self.flat)))
        for p in self.params:               # in-place kernel update: tell torch the storage changed (weight re-upload)
            p.data = p.data


# ---- the residual block (Generator.py:30-42) in train mode: forward with saved tensors, backward ----------------------
def resblock_forward(x: torch.Tensor, block: torch.nn.Module, ws: Workspace):
    """``x + 0.1 * block(x)`` with batch-statistics BatchNorm.  ``block`` is a ``ResBlock`` parameter container
    (keys ``block.{0,3}.{weight,bias}``, ``block.{1,4}.{weight,bias}``, ``block.2.weight``).  Returns the output and
    what the backward needs: the two raw conv outputs, conv2's input, and the batch statistics."""
    seq = block.block
    c1 = conv3x3(x, wino_pack(seq[0].weight), seq[0].bias.detach())
    t, st1 = bn_train_fwd(c1, seq[1].weight.detach(), seq[1].bias.detach(), ws, slope=seq[2].weight.detach())
    c2 = conv3x3(t, wino_pack(seq[3].weight), seq[3].bias.detach())
    out, st2 = bn_train_fwd(c2, seq[4].weight.detach(), seq[4].bias.detach(), ws, res=x, res_factor=0.1)
    return out, {"x": x, "c1": c1, "t": t, "c2": c2, "st1": st1, "st2": st2}


def resblock_backward(dout: torch.Tensor, saved: dict, block: torch.nn.Module, ws: Workspace):
    """(dx, {state_dict key -> gradient}) given ``dout`` = gradient of the block output."""
    seq = block.block
    g: Dict[str, torch.Tensor] = {}
    # out = x + 0.1 * bn2(c2)
    dc2, g["block.4.weight"], g["block.4.bias"], _ = bn_train_bwd(saved["c2"], dout, saved["st2"], seq[4].weight.detach(),
                                                                   seq[4].bias.detach(), ws, dout_scale=0.1)
    g["block.3.weight"], g["block.3.bias"] = conv3x3_wgrad(saved["t"], dc2, ws)
    dt = conv3x3(dc2, wino_pack(seq[3].weight, mode=1))
    # t = PReLU(bn1(c1))
    dc1, g["block.1.weight"], g["block.1.bias"], g["block.2.weight"] = bn_train_bwd(
        saved["c1"], dt, saved["st1"], seq[1].weight.detach(), seq[1].bias.detach(), ws, slope=seq[2].weight.detach())
    g["block.0.weight"], g["block.0.bias"] = conv3x3_wgrad(saved["x"], dc1, ws)
    dx = conv3x3(dc1, wino_pack(seq[0].weight, mode=1), res=dout)          # + the identity branch
    return dx, g
