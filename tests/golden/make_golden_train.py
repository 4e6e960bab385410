#!/usr/bin/env python3
"""Generates tests/golden/generator_train.npz by running the REAL reference Generator in TRAIN mode
(batch-statistics BatchNorm, running-statistics update; train.py:92 without the backward) in the build
container:

    python tests/golden/make_golden_train.py

Same conventions as make_golden.py (reference loaded by file path, only data is stored).  Contents:
  y1, y2          outputs of two consecutive train-mode forwards (no_grad) on two different batches of 4
                  (3 dense + 1 sparse synthetic images, seeds 2 and 3), synthetic weights seed 0
  bn_keys         the 33 BatchNorm prefixes in module order
  mean1, var1     running_mean / running_var of every BatchNorm after the first forward  (33, 96)
  mean2, var2     ... after the second
  nbt             num_batches_tracked after the second forward

`python tests/golden/make_golden_train.py bf16` writes generator_train_bf16.npz instead: the reference's train-mode
forward of the seed-2 batch under torch.autocast(bfloat16) -- the way train.py:91-92,148-149,166-167 call it
(CPU autocast here: no GPU in the build container) -- as float32 values of the bf16 output (y1_bf16_autocast).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from superresolution_amd import weights as W  # noqa: E402


def load_ref(name, fname):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def train_batch(seed):
    return np.concatenate([W.synthetic_lr_batch(3, seed, "dense"), W.synthetic_lr_batch(1, seed, "sparse")], 0)


def main():
    torch.set_num_threads(8)
    G = load_ref("ref_generator", "Generator.py")
    g = G.Generator()
    g.load_state_dict(W.synthetic_state_dict(0), strict=True)
    g.train()
    keys = [k[:-len(".running_mean")] for k in g.state_dict() if k.endswith(".running_mean")]
    out = {"bn_keys": np.array(keys)}

    def stats():
        sd = g.state_dict()
        return (np.stack([sd[k + ".running_mean"].numpy().copy() for k in keys]),
                np.stack([sd[k + ".running_var"].numpy().copy() for k in keys]))

    with torch.no_grad():
        out["y1"] = g(torch.from_numpy(train_batch(2))).numpy().copy()
        out["mean1"], out["var1"] = stats()
        out["y2"] = g(torch.from_numpy(train_batch(3))).numpy().copy()
        out["mean2"], out["var2"] = stats()
    out["nbt"] = np.array([int(g.state_dict()[k + ".num_batches_tracked"]) for k in keys])
    np.savez_compressed(os.path.join(OUT, "generator_train.npz"), **out)
    print("wrote generator_train.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


def main_bf16():
    torch.set_num_threads(8)
    G = load_ref("ref_generator", "Generator.py")
    g = G.Generator()
    g.load_state_dict(W.synthetic_state_dict(0), strict=True)
    g.train()
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        y = g(torch.from_numpy(train_batch(2)))
    assert y.dtype == torch.bfloat16
    np.savez_compressed(os.path.join(OUT, "generator_train_bf16.npz"), y1_bf16_autocast=y.float().numpy())
    print("wrote generator_train_bf16.npz", tuple(y.shape))


if __name__ == "__main__":
    main_bf16() if sys.argv[1:] == ["bf16"] else main()
