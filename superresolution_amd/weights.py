"""Deterministic synthetic weights and inputs for the SRGAN generator.

The reference ships no trained checkpoint (``/root/reference/.MISSING_LARGE_BLOBS``), so
every parity vector in this repo is produced from *synthetic* weights.  They are made by
an integer hash (splitmix64 finaliser) keyed by ``(seed, state_dict key, flat index)``
and NOT by torch's RNG, so the GPU box regenerates exactly the tensors the golden
fixtures were made from without shipping 12.9 MB of weights.

BatchNorm statistics, affine parameters and PReLU slopes are deliberately non-trivial
(random-init BN is the identity and would hide BN-fold / per-channel-slope bugs):

    conv / fc weight, bias   U(-1/sqrt(fan_in), +1/sqrt(fan_in))
    BN running_mean, bias    N(0, 0.1^2)
    BN running_var, weight   U(0.5, 1.5)
    PReLU slope              U(0.05, 0.45)

Shapes and key names follow the reference module tree (``Generator.py:45-72``).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        x = x ^ (x >> np.uint64(31))
    return x


def hash_uniform(seed: int, key: str, n: int, stream: int = 0) -> np.ndarray:
    """n float64 values in [0, 1), a pure function of (seed, key, stream, index)."""
    base = (_fnv1a64(key) ^ ((seed * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF)
            ^ ((stream * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(_splitmix64(idx ^ np.uint64(base)) + np.uint64(base))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hash_normal(seed: int, key: str, n: int) -> np.ndarray:
    u1 = hash_uniform(seed, key, n, stream=1)
    u2 = hash_uniform(seed, key, n, stream=2)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def generator_state_shapes(num_blocks: int = 16, in_ch: int = 3, base_ch: int = 96) -> "OrderedDict[str, tuple]":
    """state_dict key -> shape, in the reference's registration order (Generator.py:45-72)."""
    C = base_ch
    sh: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(prefix, co, ci, k):
        sh[prefix + ".weight"] = (co, ci, k, k)
        sh[prefix + ".bias"] = (co,)

    def bn(prefix):
        sh[prefix + ".weight"] = (C,)
        sh[prefix + ".bias"] = (C,)
        sh[prefix + ".running_mean"] = (C,)
        sh[prefix + ".running_var"] = (C,)
        sh[prefix + ".num_batches_tracked"] = ()

    conv("input.0", C, in_ch, 9)
    sh["input.1.weight"] = (1,)
    idx = 0
    for i in range(num_blocks):
        p = f"blocks.{idx}.block"
        conv(p + ".0", C, C, 3)
        bn(p + ".1")
        sh[p + ".2.weight"] = (C,)
        conv(p + ".3", C, C, 3)
        bn(p + ".4")
        idx += 1
        if i in (7, 15):
            p = f"blocks.{idx}.fc"
            conv(p + ".0", C // 8, C, 1)
            conv(p + ".2", C, C // 8, 1)
            idx += 1
    conv("mid.0", C, C, 3)
    bn("mid.1")
    conv("upsample.conv", 4 * C, C, 3)
    sh["upsample.act.weight"] = (1,)
    conv("output.0", C, C, 3)
    sh["output.1.weight"] = (1,)
    conv("output.2", in_ch, C, 9)
    return sh


def _is_bn_key(key: str, shapes) -> bool:
    return key.rsplit(".", 1)[0] + ".running_mean" in shapes


def synthetic_state_dict_numpy(seed: int = 0, num_blocks: int = 16, in_ch: int = 3,
                               base_ch: int = 96) -> "OrderedDict[str, np.ndarray]":
    """Synthetic fp32 state_dict (numpy) with the reference's 266 keys for the defaults."""
    shapes = generator_state_shapes(num_blocks, in_ch, base_ch)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in shapes.items():
        n = int(np.prod(shape)) if shape else 1
        leaf = key.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            out[key] = np.array(1000, dtype=np.int64)
            continue
        if _is_bn_key(key, shapes):
            if leaf in ("weight", "running_var"):
                v = 0.5 + hash_uniform(seed, key, n)
            else:
                v = 0.1 * hash_normal(seed, key, n)
        elif len(shape) == 4:                       # conv / 1x1 fc weight
            fan_in = shape[1] * shape[2] * shape[3]
            v = (2.0 * hash_uniform(seed, key, n) - 1.0) / np.sqrt(fan_in)
        elif leaf == "bias":                        # conv bias: fan_in of its weight
            wshape = shapes[key.rsplit(".", 1)[0] + ".weight"]
            fan_in = wshape[1] * wshape[2] * wshape[3]
            v = (2.0 * hash_uniform(seed, key, n) - 1.0) / np.sqrt(fan_in)
        else:                                       # PReLU slope
            v = 0.05 + 0.4 * hash_uniform(seed, key, n)
        out[key] = v.astype(np.float32).reshape(shape)
    return out


def synthetic_state_dict(seed: int = 0, **kw):
    """Same as :func:`synthetic_state_dict_numpy` but as an OrderedDict of torch tensors."""
    import torch
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)))
                       for k, v in synthetic_state_dict_numpy(seed, **kw).items())


def synthetic_lr_batch(n: int, seed: int = 0, kind: str = "dense", in_ch: int = 3,
                       hw: int = 64) -> np.ndarray:
    """(n, in_ch, hw, hw) fp32 in [0, 1].

    ``dense``: U[0,1).  ``sparse``: ~97 % exact zeros with a few hot pixels, mimicking the
    p99.5-normalised, clamped jet images of ``dataset.py:52-54``.
    """
    m = n * in_ch * hw * hw
    u = hash_uniform(seed, f"lr.{kind}", m)
    if kind == "dense":
        x = u
    elif kind == "sparse":
        gate = hash_uniform(seed, "lr.sparse.gate", m)
        x = np.where(gate > 0.97, np.minimum(1.0, u * 1.5), 0.0)
    else:
        raise ValueError(f"unknown input kind {kind!r}")
    return x.astype(np.float32).reshape(n, in_ch, hw, hw)


def synthetic_hr_batch(n: int, seed: int = 1, in_ch: int = 3, hw: int = 125) -> np.ndarray:
    """(n, in_ch, hw, hw) fp32 U[0,1) target used only to form PSNR/SSIM numbers."""
    m = n * in_ch * hw * hw
    return hash_uniform(seed, "hr.dense", m).astype(np.float32).reshape(n, in_ch, hw, hw)


def write_synthetic_chunks(tensor_dir: str, num_files: int = 10, seed: int = 5):
    """Writes `num_files` chunk files in the reference's format (scripts/convert_to_pt.py:39:
    {'lr','hr','num_samples'}) plus normalization_stats.pt, deterministically from the integer
    hash: sparse non-negative "jet image" values with a long tail, ragged sample counts (3..6),
    down-sized hr (3,25,25) so the fixture stays small -- the dataset logic does not look at
    spatial shapes.  Used by tests/golden/make_golden.py (to run the real reference dataset over
    them) and by the tests (to run the mirror over the same bytes)."""
    import os
    import torch
    os.makedirs(tensor_dir, exist_ok=True)
    for f in range(num_files):
        n = 3 + (f * 7 + seed) % 4
        lr = hash_uniform(seed, f"chunk{f}.lr", n * 3 * 64 * 64).reshape(n, 3, 64, 64)
        hr = hash_uniform(seed, f"chunk{f}.hr", n * 3 * 25 * 25).reshape(n, 3, 25, 25)
        # ~90 % zeros, the rest up to ~3x the p99.5 scale so the clamp at 1 is exercised
        lr = np.where(lr > 0.9, (lr - 0.9) * 30.0, 0.0).astype(np.float32)
        hr = np.where(hr > 0.9, (hr - 0.9) * 30.0, 0.0).astype(np.float32)
        torch.save({"lr": torch.from_numpy(lr), "hr": torch.from_numpy(hr), "num_samples": n},
                   os.path.join(tensor_dir, f"synthetic_run{f // 4}_chunk{f % 4:03d}.pt"))
    torch.save({"lr_max": 3.0, "hr_max": 3.0, "lr_p995": 1.25, "hr_p995": 1.37},
               os.path.join(tensor_dir, "normalization_stats.pt"))
