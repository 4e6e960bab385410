#!/usr/bin/env python3
"""Discriminator forward + RaGAN loss at the reference's GAN-phase batch (config.py:8, batch 256 of
125x125): whole-step timing with HIP events on torch's current stream (the stream the library
launches on).  A 'step' is what train.py:152-154 evaluates per D update: D(hr), D(fake), the loss.
Prints one JSON line; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# MACs per image of each conv at 125x125 (Discriminator.py:22-38): out pixels * 9 * cin * cout
LAYERS = [(125, 3, 64), (63, 64, 64), (63, 64, 128), (32, 128, 128), (32, 128, 256), (16, 256, 256),
          (16, 256, 512), (8, 512, 512), (8, 512, 256), (8, 256, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    from superresolution_amd import Discriminator, RelativisticAverageLoss
    torch.manual_seed(0)
    d = Discriminator().cuda()
    crit = RelativisticAverageLoss()
    g = torch.Generator(device="cuda").manual_seed(1)
    hr = torch.rand((a.batch, 3, 125, 125), device="cuda", generator=g)
    sr = torch.rand((a.batch, 3, 125, 125), device="cuda", generator=g)

    def step():
        with torch.no_grad():
            dr, df = d(hr), d(sr)
            return crit.both_losses(dr, df)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        ld, lg = step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    macs = sum(s * s * 9 * ci * co for s, ci, co in LAYERS)
    flops = 2.0 * macs * 2 * a.batch
    print(json.dumps({"metric": "discriminator_images_per_sec", "value": 2 * a.batch / (ms * 1e-3),
                      "unit": "images/s", "ms_per_step": ms, "batch": a.batch, "steps": a.steps,
                      "dtype": "f32", "gmac_per_image": macs / 1e9, "tflops": flops / (ms * 1e-3) / 1e12,
                      "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": 157.3,
                                   "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / 157.3,
                                   "note": "whole step (stem, tail and loss launches included) against the nominal "
                                           "fp32 MFMA peak; per-layer figures: tools/disc_layers.py on a rocprofv3 "
                                           "kernel trace (profiles/r01_disc_layers.txt)"},
                      "loss_d": ld.item(), "loss_g": lg.item(),
                      "config": {"workload": "D(hr), D(sr) + RaGAN losses, batch 256 of 3x125x125 (train.py:152-154)"}}))


if __name__ == "__main__":
    main()
