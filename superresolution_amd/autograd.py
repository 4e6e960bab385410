"""``torch.autograd.Function`` wrappers that put the HIP training operators behind the drop-in modules, so that the
reference's own training loops run on them with only the import line changed:

    pre-train (``/root/reference/train.py:90-97``)          GAN phase (``train.py:145-176``)
        optimizer.zero_grad()                                   fake = generator(lr)            # no_grad, train mode
        with torch.autocast('cuda', torch.bfloat16):            d_loss = criterion.discriminator_loss(D(hr), D(fake))
            sr = generator(lr)                                  d_loss.backward(); opt_d.step()
            loss = weighted_l1(sr, hr) + w * (1 - ssim(..))     g_loss = w_a * criterion.generator_loss(real.detach(), D(G(lr)))
        loss.backward(); grad_norm(generator)                          + w_p * weighted_l1 + w_s * (1 - ssim)
        optimizer.step()                                        g_loss.backward(); clip_grad_norm_(...); opt_g.step()

``loss.backward()`` reaches ``backward`` below with the gradient at the module's output; the existing operator chains
(``training.GeneratorGrad``, ``gan_training.DiscriminatorGrad``, ``ragan_loss_grad``, ``pixel_loss_grad``) do the rest and
the parameter gradients go back to autograd as views of one flat buffer, which it stores in (or adds to) ``p.grad`` -- stock
``torch.optim.AdamW``, ``zero_grad(set_to_none=True)``, ``clip_grad_norm_`` and ``grad_norm`` (train.py:23-28) all work.
The loss expressions themselves (``weighted_l1`` is defined in train.py, ``ssim`` in utils.py) stay what the caller wrote:
plain torch ops, differentiated by torch; ``superresolution_amd.ssim`` / ``weighted_l1`` are the fused alternatives.

No eager fallback: every op here needs ROCm tensors and ``libsrgen.so``.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import training as T


def _flat_views(params):
    """One fresh flat fp32 buffer and a view of it per parameter (in order)."""
    n = sum(p.numel() for p in params)
    flat = torch.empty(n, dtype=torch.float32, device=params[0].device)
    views, off = [], 0
    for p in params:
        views.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    return flat, views


class GeneratorTrainFn(torch.autograd.Function):
    """``Generator.forward`` in train mode under autograd (train.py:92,167)."""

    @staticmethod
    def forward(ctx, engine: T.GeneratorGrad, precision: str, names, x: torch.Tensor, *params):
        with T.conv_precision(precision), T.packed_weights(engine.pack_jobs()) as packs:
            y, S = engine.forward(x)
        engine._update_running_stats(S, y.shape[0])          # nn.BatchNorm2d moves its running statistics in forward
        ctx.engine, ctx.precision, ctx.names, ctx.S, ctx.packs = engine, precision, names, S, packs
        ctx.param_versions = [p._version for p in params]
        ctx.params = params
        return y

    @staticmethod
    def backward(ctx, dy: torch.Tensor):
        engine, S = ctx.engine, ctx.S
        if S is None:
            raise RuntimeError("Trying to backward through the generator a second time: its saved tensors were freed "
                               "(retain_graph is not supported by the HIP training operators)")
        if [p._version for p in ctx.params] != ctx.param_versions:
            raise RuntimeError("a generator parameter was modified in place between forward and backward")
        _, views = _flat_views(ctx.params)
        G = dict(zip(ctx.names, views))
        dsr = dy.detach().to(torch.float32).contiguous()
        with T.conv_precision(ctx.precision), T.packed_weights(engine.pack_jobs(), ctx.packs):
            engine.backward(dsr, S, G)
        ctx.S = ctx.packs = None                              # free the saved activations (2.4 GB at 64 images)
        grads = [v if need else None for v, need in zip(views, ctx.needs_input_grad[4:])]
        return (None, None, None, None, *grads)


def generator_train_forward(gen, engine: T.GeneratorGrad, x: torch.Tensor, precision: str) -> torch.Tensor:
    names = [k for k, _ in gen.named_parameters()]
    params = [p for _, p in gen.named_parameters()]
    return GeneratorTrainFn.apply(engine, precision, names, x, *params)


class DiscriminatorFn(torch.autograd.Function):
    """``Discriminator.forward`` under autograd (train.py:152-153,168): logits with the activations saved; backward gives
    the gradient of the ten convs' parameters and, when the input asks for it (the generator half-step), of the image."""

    @staticmethod
    def forward(ctx, dgrad, names, x: torch.Tensor, *params):
        logits, saved = dgrad.forward(x)
        ctx.dgrad, ctx.names, ctx.saved, ctx.params = dgrad, names, saved, params
        return logits

    @staticmethod
    def backward(ctx, dlogits: torch.Tensor):
        if ctx.saved is None:
            raise RuntimeError("Trying to backward through the discriminator a second time: its saved activations were "
                               "freed (retain_graph is not supported by the HIP training operators)")
        want_w = any(ctx.needs_input_grad[3:])
        want_x = ctx.needs_input_grad[2]
        G = None
        views = [None] * len(ctx.params)
        if want_w:
            flat, views = _flat_views(ctx.params)
            G = dict(zip(ctx.names, views))
        dimg = ctx.dgrad.backward(dlogits, ctx.saved, want_wgrad=want_w, want_dimg=want_x, into=G)
        ctx.saved = None
        grads = [v if need else None for v, need in zip(views, ctx.needs_input_grad[3:])]
        return (None, None, dimg, *grads)


class RaganLossFn(torch.autograd.Function):
    """Both ``RelativisticAverageLoss`` values (Discriminator.py:53-72) from one launch; ``which`` = 0 discriminator
    loss (targets real_label / fake_label), 1 generator loss (targets 0 / 1)."""

    @staticmethod
    def forward(ctx, real: torch.Tensor, fake: torch.Tensor, real_label: float, fake_label: float, which: int):
        from .discriminator import ragan_both
        r, f = real.detach().float().contiguous(), fake.detach().float().contiguous()
        out = ragan_both(r, f, real_label, fake_label)
        ctx.r, ctx.f = r, f
        ctx.targets = (real_label, fake_label) if which == 0 else (0.0, 1.0)
        ctx.in_dtypes = (real.dtype, fake.dtype)
        ctx.shapes = (real.shape, fake.shape)
        return out[which].clone()

    @staticmethod
    def backward(ctx, dloss: torch.Tensor):
        from .gan_training import ragan_loss_grad
        dr, df = ragan_loss_grad(ctx.r, ctx.f, ctx.targets[0], ctx.targets[1], 1.0, want_real=ctx.needs_input_grad[0])
        g = dloss.detach().to(torch.float32)
        out_r = (dr * g).view(ctx.shapes[0]).to(ctx.in_dtypes[0]) if ctx.needs_input_grad[0] else None
        out_f = (df * g).view(ctx.shapes[1]).to(ctx.in_dtypes[1]) if ctx.needs_input_grad[1] else None
        return out_r, out_f, None, None, None


class PixelLossFn(torch.autograd.Function):
    """``w_l1 * weighted_l1(sr, hr) + w_ssim * (1 - ssim(sr, hr))`` (train.py:18-20,93,170-172; utils.py:21-41) as one
    differentiable scalar: value from the metrics kernel, gradient from ``pixel_loss_grad``."""

    @staticmethod
    def forward(ctx, sr: torch.Tensor, hr: torch.Tensor, w_l1: float, w_ssim: float):
        from .functional import pixel_losses
        s, h = sr.detach().to(torch.float32).contiguous(), hr.detach().to(torch.float32).contiguous()
        vals = pixel_losses(s, h)
        ctx.s, ctx.h, ctx.w, ctx.in_dtype = s, h, (w_l1, w_ssim), sr.dtype
        return w_l1 * vals[3] + w_ssim * (1.0 - vals[1])

    @staticmethod
    def backward(ctx, dloss: torch.Tensor):
        ws = T.Workspace(ctx.s.device)
        d = T.pixel_loss_grad(ctx.s, ctx.h, ws, ctx.w[0], ctx.w[1])
        return (d * dloss.detach().to(torch.float32)).to(ctx.in_dtype), None, None, None


def pixel_loss(sr: torch.Tensor, hr: torch.Tensor, w_l1: float = 1.0, w_ssim: float = 0.1) -> torch.Tensor:
    """The pixel terms of both training losses as ONE differentiable op (two kernels forward, two backward) instead of
    the ~40 eager torch ops of ``weighted_l1(sr, hr) + w * (1 - ssim(sr, hr))``; defaults = the pre-train loss."""
    if not sr.is_cuda or not hr.is_cuda:
        raise RuntimeError("pixel_loss: ROCm tensors only; there is no CPU fallback")
    return PixelLossFn.apply(sr, hr, float(w_l1), float(w_ssim))
