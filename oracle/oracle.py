"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by superresolution_amd/).

Python driver over ``liboracle.so`` (``srgen_oracle.c``): composes the C ops layer by layer
in exactly the reference's order -- ``Generator.forward`` (``/root/reference/Generator.py:
74-80``) and the sub-modules it chains (``:4-42``) -- from a plain ``{key: ndarray}``
state_dict with the reference's 266 keys.  No folding, no fusion: conv, then BN, then
PReLU, ... so that the HIP path's fold algebra is checked against an independent chain.

Pinned against the real reference by ``tests/golden`` (see ``make_golden.py``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[ctypes.CDLL] = None

_F = ctypes.POINTER(ctypes.c_float)
_D = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "srgen_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        i = ctypes.c_int
        L.oracle_conv2d.argtypes = [_F, _F, _F, _F, i, i, i, i, i, i]
        L.oracle_conv2d.restype = i
        L.oracle_batchnorm_eval.argtypes = [_F, _F, _F, _F, _F, _F, ctypes.c_double, i, i, i]
        L.oracle_batchnorm_eval.restype = None
        L.oracle_prelu.argtypes = [_F, _F, _F, i, i, i, i]
        L.oracle_prelu.restype = None
        L.oracle_axpy.argtypes = [_F, _F, ctypes.c_float, _F, ctypes.c_size_t]
        L.oracle_axpy.restype = None
        L.oracle_pixel_shuffle.argtypes = [_F, _F, i, i, i, i, i]
        L.oracle_pixel_shuffle.restype = None
        L.oracle_channel_attention.argtypes = [_F, _F, _F, _F, _F, _F, _F, i, i, i, i]
        L.oracle_channel_attention.restype = i
        L.oracle_crop.argtypes = [_F, _F, i, i, i, i, i, i]
        L.oracle_crop.restype = None
        L.oracle_bilinear_resize.argtypes = [_F, _F, i, i, i, i, i, i]
        L.oracle_bilinear_resize.restype = None
        L.oracle_metrics.argtypes = [_F, _F, i, i, i, i, _D, _D, _D]
        L.oracle_metrics.restype = None
        _LIB = L
    return _LIB


def _p(a: Optional[np.ndarray]):
    if a is None:
        return ctypes.cast(None, _F)
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_F)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


# ---- single ops (each cites the reference line it restates) ---------------------------

def conv2d(x, w, b=None) -> np.ndarray:
    """nn.Conv2d(cin, cout, k, padding=k//2) -- Generator.py:7,34,37,50,62,69,71."""
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    n, cin, h, wd = x.shape
    cout, cin2, k, k2 = w.shape
    assert cin == cin2 and k == k2
    y = np.empty((n, cout, h, wd), np.float32)
    rc = lib().oracle_conv2d(_p(x), _p(w), _p(b), _p(y), n, cin, cout, h, wd, k)
    assert rc == 0
    return y


def batchnorm_eval(x, gamma, beta, mean, var, eps=1e-5) -> np.ndarray:
    """nn.BatchNorm2d.eval() -- Generator.py:35,38,63."""
    x = _f32(x)
    n, c, h, w = x.shape
    y = np.empty_like(x)
    lib().oracle_batchnorm_eval(_p(x), _p(y), _p(_f32(gamma)), _p(_f32(beta)), _p(_f32(mean)),
                                _p(_f32(var)), float(eps), n, c, h * w)
    return y


def prelu(x, slope) -> np.ndarray:
    """nn.PReLU() / nn.PReLU(ch) -- Generator.py:9,36,51,70."""
    x, slope = _f32(x), _f32(slope).reshape(-1)
    n, c, h, w = x.shape
    assert slope.size in (1, c)
    y = np.empty_like(x)
    lib().oracle_prelu(_p(x), _p(y), _p(slope), int(slope.size), n, c, h * w)
    return y


def axpy(a, b, s: float) -> np.ndarray:
    """a + s*b -- ResBlock.forward Generator.py:42 (s=0.1), long skip :78 (s=1)."""
    a, b = _f32(a), _f32(b)
    y = np.empty_like(a)
    lib().oracle_axpy(_p(a), _p(b), float(s), _p(y), a.size)
    return y


def pixel_shuffle(x, r=2) -> np.ndarray:
    """nn.PixelShuffle(r) -- Generator.py:8."""
    x = _f32(x)
    n, crr, h, w = x.shape
    c = crr // (r * r)
    y = np.empty((n, c, h * r, w * r), np.float32)
    lib().oracle_pixel_shuffle(_p(x), _p(y), n, c, h, w, r)
    return y


def channel_attention(x, w1, b1, w2, b2):
    """ChannelAttention.forward -- Generator.py:15-27.  Returns (y, scale[n,c])."""
    x = _f32(x)
    n, c, h, w = x.shape
    w1, w2 = _f32(w1).reshape(-1, c), _f32(w2)
    cr = w1.shape[0]
    w2 = w2.reshape(c, cr)
    y = np.empty_like(x)
    sc = np.empty((n, c), np.float32)
    rc = lib().oracle_channel_attention(_p(x), _p(y), _p(w1), _p(_f32(b1)), _p(w2), _p(_f32(b2)),
                                        _p(sc), n, c, cr, h * w)
    assert rc == 0
    return y, sc


def crop(x, oh=125, ow=125) -> np.ndarray:
    """out[:, :, :125, :125] -- Generator.py:80."""
    x = _f32(x)
    n, c, h, w = x.shape
    y = np.empty((n, c, oh, ow), np.float32)
    lib().oracle_crop(_p(x), _p(y), n, c, h, w, oh, ow)
    return y


def bilinear_resize(x, oh=125, ow=125) -> np.ndarray:
    """F.interpolate(x,(oh,ow),mode='bilinear',align_corners=False) (optional op, not on
    the reference forward path: SURVEY.md D1)."""
    x = _f32(x)
    n, c, h, w = x.shape
    y = np.empty((n, c, oh, ow), np.float32)
    lib().oracle_bilinear_resize(_p(x), _p(y), n, c, h, w, oh, ow)
    return y


def metrics(sr, hr):
    """(psnr, ssim, l1) over the whole batch -- utils.py:5-41, train.py:61-68."""
    sr, hr = _f32(sr), _f32(hr)
    n, c, h, w = sr.shape
    ps, ss, l1 = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib().oracle_metrics(_p(sr), _p(hr), n, c, h, w, ctypes.byref(ps), ctypes.byref(ss),
                         ctypes.byref(l1))
    return ps.value, ss.value, l1.value


# ---- composition: Generator.forward, Generator.py:74-80 ------------------------------

TAP_NAMES = ("stem", "block0", "block7", "ca8", "block16", "ca17", "skip", "upsample",
             "output1", "output2_full")


def _bn_args(sd, prefix):
    return (sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"],
            sd[prefix + ".running_var"])


def generator_forward(sd: Dict[str, np.ndarray], x, taps: Optional[dict] = None,
                      num_blocks: int = 16) -> np.ndarray:
    """fp32 eval forward of the reference Generator from a numpy state_dict.

    ``taps`` (optional dict) receives the intermediate NCHW tensors named in TAP_NAMES plus
    ``ca8_scale`` / ``ca17_scale``.
    """
    sd = {k: np.asarray(v) for k, v in sd.items()}
    x = _f32(x)

    def tap(name, v):
        if taps is not None:
            taps[name] = v

    # input: Conv2d(3,96,9,pad 4) + PReLU()                      Generator.py:49-52,75
    x1 = prelu(conv2d(x, sd["input.0.weight"], sd["input.0.bias"]), sd["input.1.weight"])
    tap("stem", x1)
    # blocks: 16 x ResBlock, ChannelAttention after block 7 and 15    Generator.py:54-59,76
    cur = x1
    idx = 0
    for i in range(num_blocks):
        p = f"blocks.{idx}.block"
        t = conv2d(cur, sd[p + ".0.weight"], sd[p + ".0.bias"])      # :34
        t = batchnorm_eval(t, *_bn_args(sd, p + ".1"))               # :35
        t = prelu(t, sd[p + ".2.weight"])                            # :36
        t = conv2d(t, sd[p + ".3.weight"], sd[p + ".3.bias"])        # :37
        t = batchnorm_eval(t, *_bn_args(sd, p + ".4"))               # :38
        cur = axpy(cur, t, 0.1)                                      # :42
        if idx in (0, 7, 16):
            tap(f"block{idx}", cur)
        idx += 1
        if i in (7, 15):
            p = f"blocks.{idx}.fc"
            cur, sc = channel_attention(cur, sd[p + ".0.weight"], sd[p + ".0.bias"],
                                        sd[p + ".2.weight"], sd[p + ".2.bias"])   # :26-27
            tap(f"ca{idx}", cur)
            tap(f"ca{idx}_scale", sc)
            idx += 1
    # mid: Conv2d + BN, long skip                                 Generator.py:61-64,77-78
    x3 = batchnorm_eval(conv2d(cur, sd["mid.0.weight"], sd["mid.0.bias"]), *_bn_args(sd, "mid.1"))
    s = axpy(x1, x3, 1.0)
    tap("skip", s)
    # upsample: conv 96->384, PixelShuffle(2), PReLU()            Generator.py:4-12,78
    u = prelu(pixel_shuffle(conv2d(s, sd["upsample.conv.weight"], sd["upsample.conv.bias"]), 2),
              sd["upsample.act.weight"])
    tap("upsample", u)
    # output: conv3x3 + PReLU() + conv9x9 96->3                   Generator.py:68-72,79
    o = prelu(conv2d(u, sd["output.0.weight"], sd["output.0.bias"]), sd["output.1.weight"])
    tap("output1", o)
    o = conv2d(o, sd["output.2.weight"], sd["output.2.bias"])
    tap("output2_full", o)
    return crop(o, 125, 125)                                         # :80


# ---- composition: Discriminator.forward, Discriminator.py:17-50 -----------------------

DISC_LAYERS = (("backbone.0", 1), ("backbone.2.block.0", 2), ("backbone.3.block.0", 1),
               ("backbone.4.block.0", 2), ("backbone.5.block.0", 1), ("backbone.6.block.0", 2),
               ("backbone.7.block.0", 1), ("backbone.8.block.0", 2), ("head.0", 1), ("head.2", 1))


def leaky_relu(x, slope=0.2) -> np.ndarray:
    """nn.LeakyReLU(0.2) -- Discriminator.py:10,24,36."""
    x = _f32(x)
    return np.where(x > 0, x, x * np.float32(slope)).astype(np.float32)


def discriminator_forward(sd: Dict[str, np.ndarray], x, taps: Optional[list] = None) -> np.ndarray:
    """fp32 forward of the reference Discriminator from a numpy state_dict (20 keys).

    A 3x3 / pad 1 / stride 2 conv (Discriminator.py:9) is the stride-1 conv sampled at even output
    positions, so the strided layers run oracle_conv2d and keep [::2, ::2].  ``taps`` (optional
    list) receives the ten post-activation conv outputs."""
    cur = _f32(x)
    for i, (key, stride) in enumerate(DISC_LAYERS):
        cur = conv2d(cur, sd[key + ".weight"], sd[key + ".bias"])
        if stride == 2:
            cur = np.ascontiguousarray(cur[:, :, ::2, ::2])
        if i != len(DISC_LAYERS) - 1:                       # no activation after head.2 (:37)
            cur = leaky_relu(cur, 0.2)
        if taps is not None:
            taps.append(cur)
    return cur


def ragan_losses(real, fake, real_label=0.9, fake_label=0.1):
    """(discriminator_loss, generator_loss) of RelativisticAverageLoss -- Discriminator.py:53-72,
    in fp64.  BCEWithLogits(x, t) = mean(max(x,0) - x*t + log1p(exp(-|x|)))."""
    r, f = np.asarray(real, np.float64).ravel(), np.asarray(fake, np.float64).ravel()

    def bce(x, t):
        return float(np.mean(np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))))

    rf, fr = r - f.mean(), f - r.mean()                     # :61-62, :68-69
    return ((bce(rf, real_label) + bce(fr, fake_label)) / 2,      # :63-65
            (bce(rf, 0.0) + bce(fr, 1.0)) / 2)                    # :70-72


def weighted_l1(sr, hr) -> float:
    """train.py:18-20: weight = (hr > 0.01).float() * 9 + 1; (weight * (sr - hr).abs()).mean(), in fp64."""
    sr, hr = np.asarray(sr, np.float64), np.asarray(hr, np.float64)
    weight = (np.asarray(hr, np.float32) > np.float32(0.01)).astype(np.float64) * 9 + 1
    return float((weight * np.abs(sr - hr)).mean())


# ---- train-mode forward (batch-statistics BatchNorm): Generator.train()(x), train.py:92 ----

def batchnorm_train(x, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1):
    """nn.BatchNorm2d in training mode (Generator.py:35,38,63): normalise with the batch mean and the
    BIASED batch variance over (N, H, W); running statistics move by `momentum` towards the batch mean and
    the UNBIASED batch variance.  fp64 statistics.  Returns (y, new_running_mean, new_running_var)."""
    x = _f32(x)
    n = x.shape[0] * x.shape[2] * x.shape[3]
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=(0, 2, 3))
    var = x64.var(axis=(0, 2, 3))
    y = (x64 - mean[None, :, None, None]) / np.sqrt(var + eps)[None, :, None, None]
    y = y * np.asarray(gamma, np.float64)[None, :, None, None] + np.asarray(beta, np.float64)[None, :, None, None]
    new_mean = (1 - momentum) * np.asarray(running_mean, np.float64) + momentum * mean
    new_var = (1 - momentum) * np.asarray(running_var, np.float64) + momentum * var * (n / max(n - 1, 1))
    return y.astype(np.float32), new_mean.astype(np.float32), new_var.astype(np.float32)


def generator_forward_train(sd: Dict[str, np.ndarray], x, num_blocks: int = 16):
    """fp32 TRAIN-mode forward of the reference Generator.  Returns (y, {bn prefix: (running_mean,
    running_var)}) with the statistics as they stand after this forward."""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    x = _f32(x)
    new_stats = {}

    def bn(t, prefix):
        y, m, v = batchnorm_train(t, *_bn_args(sd, prefix))
        new_stats[prefix] = (m, v)
        return y

    x1 = prelu(conv2d(x, sd["input.0.weight"], sd["input.0.bias"]), sd["input.1.weight"])
    cur = x1
    idx = 0
    for i in range(num_blocks):
        p = f"blocks.{idx}.block"
        t = prelu(bn(conv2d(cur, sd[p + ".0.weight"], sd[p + ".0.bias"]), p + ".1"), sd[p + ".2.weight"])
        t = bn(conv2d(t, sd[p + ".3.weight"], sd[p + ".3.bias"]), p + ".4")
        cur = axpy(cur, t, 0.1)
        idx += 1
        if i in (7, 15):
            p = f"blocks.{idx}.fc"
            cur, _ = channel_attention(cur, sd[p + ".0.weight"], sd[p + ".0.bias"],
                                       sd[p + ".2.weight"], sd[p + ".2.bias"])
            idx += 1
    x3 = bn(conv2d(cur, sd["mid.0.weight"], sd["mid.0.bias"]), "mid.1")
    s = axpy(x1, x3, 1.0)
    u = prelu(pixel_shuffle(conv2d(s, sd["upsample.conv.weight"], sd["upsample.conv.bias"]), 2),
              sd["upsample.act.weight"])
    o = prelu(conv2d(u, sd["output.0.weight"], sd["output.0.bias"]), sd["output.1.weight"])
    o = conv2d(o, sd["output.2.weight"], sd["output.2.bias"])
    return crop(o, 125, 125), new_stats
