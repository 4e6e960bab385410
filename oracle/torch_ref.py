"""oracle/torch_ref.py -- TEST INFRASTRUCTURE ONLY (never imported by superresolution_amd/).

The reference's forward restated with stock ``torch.nn.functional`` ops driven from a
state_dict -- the same ATen CPU kernels ``/root/reference/Generator.py`` dispatches to,
without importing it (the reference's files do not exist on the GPU box).  Used (a) as the
second, independent checker next to the C oracle, (b) as ``bench.py``'s ``cpu_baseline``
("port": stock torch CPU conv path on the box's host cores).

Line references are to /root/reference/Generator.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training=False, momentum=0.1, eps=1e-5)   # :35,38,63


@torch.no_grad()
def generator_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor,
                      taps: Optional[dict] = None, num_blocks: int = 16) -> torch.Tensor:
    def tap(name, v):
        if taps is not None:
            taps[name] = v

    x1 = F.prelu(F.conv2d(x, sd["input.0.weight"], sd["input.0.bias"], padding=4),
                 sd["input.1.weight"])                                              # :49-52
    tap("stem", x1)
    cur, idx = x1, 0
    for i in range(num_blocks):
        p = f"blocks.{idx}.block"
        t = F.conv2d(cur, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=1)        # :34
        t = F.prelu(_bn(t, sd, p + ".1"), sd[p + ".2.weight"])                      # :35-36
        t = _bn(F.conv2d(t, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1), sd, p + ".4")
        cur = cur + 0.1 * t                                                         # :42
        if idx in (0, 7, 16):
            tap(f"block{idx}", cur)
        idx += 1
        if i in (7, 15):
            p = f"blocks.{idx}.fc"
            pooled = F.adaptive_avg_pool2d(cur, 1)                                  # :18
            sc = torch.sigmoid(F.conv2d(F.relu(F.conv2d(pooled, sd[p + ".0.weight"],
                                                        sd[p + ".0.bias"])),
                                        sd[p + ".2.weight"], sd[p + ".2.bias"]))    # :19-24
            cur = cur * sc                                                          # :27
            tap(f"ca{idx}", cur)
            tap(f"ca{idx}_scale", sc.flatten(1))
            idx += 1
    x3 = _bn(F.conv2d(cur, sd["mid.0.weight"], sd["mid.0.bias"], padding=1), sd, "mid.1")
    s = x1 + x3                                                                     # :78
    tap("skip", s)
    u = F.prelu(F.pixel_shuffle(F.conv2d(s, sd["upsample.conv.weight"],
                                         sd["upsample.conv.bias"], padding=1), 2),
                sd["upsample.act.weight"])                                          # :12
    tap("upsample", u)
    o = F.prelu(F.conv2d(u, sd["output.0.weight"], sd["output.0.bias"], padding=1),
                sd["output.1.weight"])                                              # :69-70
    tap("output1", o)
    o = F.conv2d(o, sd["output.2.weight"], sd["output.2.bias"], padding=4)          # :71
    tap("output2_full", o)
    return o[:, :, :125, :125]                                                      # :80


def psnr(sr, hr, max_val=1.0):
    """utils.py:5-9."""
    mse = F.mse_loss(sr, hr)
    if mse == 0:
        return torch.tensor(100.0)
    return 10 * torch.log10(max_val ** 2 / mse)


def ssim(sr, hr, window_size=11, sigma=1.5):
    """utils.py:12-41 (zero-padded depthwise gaussian, den clamp 1e-8, global mean)."""
    c = sr.size(1)
    coords = torch.arange(window_size, dtype=torch.float32, device=sr.device) - window_size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    k = g.outer(g).view(1, 1, window_size, window_size).repeat(c, 1, 1, 1)
    pad = window_size // 2
    conv = lambda t: F.conv2d(t, k, padding=pad, groups=c)
    mx, my = conv(sr), conv(hr)
    sxx, syy, sxy = conv(sr * sr) - mx * mx, conv(hr * hr) - my * my, conv(sr * hr) - mx * my
    num = (2 * mx * my + 0.01 ** 2) * (2 * sxy + 0.03 ** 2)
    den = (mx * mx + my * my + 0.01 ** 2) * (sxx + syy + 0.03 ** 2)
    return (num / den.clamp(min=1e-8)).mean()


# ---- training steps on the CPU with torch autograd: bench.py's cpu_baseline for BASELINE configs[3] / [4] ----------------
def generator_forward_train(P: Dict[str, torch.Tensor], x: torch.Tensor, num_blocks: int = 16) -> torch.Tensor:
    """Generator.py:74-80 in TRAIN mode (batch-statistics BatchNorm, no running-statistics update) on a dict of
    parameters that may require grad -- the graph train.py:92 builds."""
    def bn(t, p):
        return F.batch_norm(t, None, None, P[p + ".weight"], P[p + ".bias"], training=True, eps=1e-5)

    x1 = F.prelu(F.conv2d(x, P["input.0.weight"], P["input.0.bias"], padding=4), P["input.1.weight"])
    cur, idx = x1, 0
    for i in range(num_blocks):
        p = f"blocks.{idx}.block"
        t = F.prelu(bn(F.conv2d(cur, P[p + ".0.weight"], P[p + ".0.bias"], padding=1), p + ".1"), P[p + ".2.weight"])
        t = bn(F.conv2d(t, P[p + ".3.weight"], P[p + ".3.bias"], padding=1), p + ".4")
        cur = cur + 0.1 * t
        idx += 1
        if i in (7, 15):
            p = f"blocks.{idx}.fc"
            sc = torch.sigmoid(F.conv2d(F.relu(F.conv2d(F.adaptive_avg_pool2d(cur, 1), P[p + ".0.weight"], P[p + ".0.bias"])),
                                        P[p + ".2.weight"], P[p + ".2.bias"]))
            cur = cur * sc
            idx += 1
    s = x1 + bn(F.conv2d(cur, P["mid.0.weight"], P["mid.0.bias"], padding=1), "mid.1")
    u = F.prelu(F.pixel_shuffle(F.conv2d(s, P["upsample.conv.weight"], P["upsample.conv.bias"], padding=1), 2),
                P["upsample.act.weight"])
    o = F.prelu(F.conv2d(u, P["output.0.weight"], P["output.0.bias"], padding=1), P["output.1.weight"])
    return F.conv2d(o, P["output.2.weight"], P["output.2.bias"], padding=4)[:, :, :125, :125]


def discriminator_forward(D: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """Discriminator.py:22-38,49-50 on a dict of its 20 parameters."""
    keys = ["backbone.0"] + [f"backbone.{k}.block.0" for k in range(2, 9)] + ["head.0", "head.2"]
    strides = [1, 2, 1, 2, 1, 2, 1, 2, 1, 1]
    h = x
    for i, (k, st) in enumerate(zip(keys, strides)):
        h = F.conv2d(h, D[k + ".weight"], D[k + ".bias"], stride=st, padding=1)
        if i < 9:
            h = F.leaky_relu(h, 0.2)
    return h


def weighted_l1(sr, hr):
    """train.py:18-20."""
    return (((hr > 0.01).float() * 9 + 1) * (sr - hr).abs()).mean()


def _ragan(real, fake, t_real, t_fake):
    """Discriminator.py:53-72."""
    bce = F.binary_cross_entropy_with_logits
    rf, fr = real - fake.mean(), fake - real.mean()
    return (bce(rf, torch.full_like(rf, t_real)) + bce(fr, torch.full_like(fr, t_fake))) / 2


def pretrain_step(P, opt, lr, hr):
    """train.py:90-97 (fp32): one optimizer step; returns the loss."""
    opt.zero_grad()
    sr = generator_forward_train(P, lr)
    loss = weighted_l1(sr, hr) + 0.1 * (1 - ssim(sr, hr))
    loss.backward()
    opt.step()
    return float(loss.detach())


def gan_iteration(P, D, opt_g, opt_d, lr, hr, update_d=True):
    """train.py:145-176 (fp32): one iteration; returns g_loss."""
    if update_d:
        opt_d.zero_grad()
        with torch.no_grad():
            fake = generator_forward_train(P, lr)
        real_logits = discriminator_forward(D, hr)
        d_loss = _ragan(real_logits, discriminator_forward(D, fake), 0.9, 0.1)
        d_loss.backward()
        opt_d.step()
    else:
        with torch.no_grad():
            real_logits = discriminator_forward(D, hr)
    opt_g.zero_grad()
    fake = generator_forward_train(P, lr)
    g_loss = (0.005 * _ragan(real_logits.detach(), discriminator_forward(D, fake), 0.0, 1.0) + weighted_l1(fake, hr)
              + 0.3 * (1 - ssim(fake, hr)))
    g_loss.backward()
    torch.nn.utils.clip_grad_norm_(list(P.values()), 10.0)
    opt_g.step()
    return float(g_loss.detach())
