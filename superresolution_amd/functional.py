"""Stand-alone HIP ops around the generator forward (include/srgen.h):

* :func:`bilinear_resize` -- the optional 128->125 resize BASELINE.json's north_star names.  The
  reference forward does NOT resize, it crops (``Generator.py:80``), so this op is
  never wired into ``Generator.forward``; it matches ``F.interpolate(mode='bilinear',
  align_corners=False)``.
* :func:`batch_metrics` -- PSNR / SSIM / L1 of one batch as ``validate()`` computes them
  (``train.py:66-68``, ``utils.py:5-41``), returned as a 3-element device tensor (no ``.item()``
  sync), ready for the RCCL SUM all-reduce of ``superresolution_amd.evaluate``.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda_f32(name: str, t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a ROCm device tensor (got {t.device}); there is no CPU fallback")
    return t.detach().to(torch.float32).contiguous()


def bilinear_resize(x: torch.Tensor, size=(125, 125)) -> torch.Tensor:
    x = _need_cuda_f32("x", x)
    if x.dim() != 4:
        raise RuntimeError(f"expected (N, C, H, W), got {tuple(x.shape)}")
    n, c, h, w = x.shape
    oh, ow = int(size[0]), int(size[1])
    y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().srgen_bilinear_resize(x.data_ptr(), y.data_ptr(), n * c, h, w, oh, ow, _stream(x)))
    return y


def batch_metrics(sr: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
    """float32 device tensor [psnr, ssim, l1] over the whole batch (max_val = 1)."""
    sr, hr = _need_cuda_f32("sr", sr), _need_cuda_f32("hr", hr)
    if sr.shape != hr.shape or sr.dim() != 4:
        raise RuntimeError(f"sr {tuple(sr.shape)} and hr {tuple(hr.shape)} must be equal 4D shapes")
    n, c, h, w = sr.shape
    L = _lib.lib()
    need = ctypes.c_size_t()
    _lib.check(L.srgen_metrics_scratch_bytes(n, c, h, w, ctypes.byref(need)))
    scratch = torch.empty(need.value, dtype=torch.uint8, device=sr.device)
    out = torch.empty(3, dtype=torch.float32, device=sr.device)
    with torch.cuda.device(sr.device):
        _lib.check(L.srgen_metrics(sr.data_ptr(), hr.data_ptr(), n, c, h, w, out.data_ptr(), scratch.data_ptr(),
                                   scratch.numel(), _stream(sr)))
    return out


def pixel_losses(sr: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
    """float32 device tensor [psnr, ssim, l1, weighted_l1] over the whole batch: ``batch_metrics`` plus
    the reference's ``weighted_l1`` (``train.py:18-20``) from the same pass.  With it the pixel terms of
    the training losses are ``out[3] + w * (1 - out[1])`` (``train.py:94,170-172``).  Forward values only."""
    sr, hr = _need_cuda_f32("sr", sr), _need_cuda_f32("hr", hr)
    if sr.shape != hr.shape or sr.dim() != 4:
        raise RuntimeError(f"sr {tuple(sr.shape)} and hr {tuple(hr.shape)} must be equal 4D shapes")
    n, c, h, w = sr.shape
    L = _lib.lib()
    need = ctypes.c_size_t()
    _lib.check(L.srgen_metrics_scratch_bytes(n, c, h, w, ctypes.byref(need)))
    scratch = torch.empty(need.value, dtype=torch.uint8, device=sr.device)
    out = torch.empty(4, dtype=torch.float32, device=sr.device)
    with torch.cuda.device(sr.device):
        _lib.check(L.srgen_pixel_losses(sr.data_ptr(), hr.data_ptr(), n, c, h, w, out.data_ptr(),
                                        scratch.data_ptr(), scratch.numel(), _stream(sr)))
    return out


def weighted_l1(sr: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
    """``train.py:18-20`` as a 0-dim device tensor (forward value)."""
    return pixel_losses(sr, hr)[3]
