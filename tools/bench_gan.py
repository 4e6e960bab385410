#!/usr/bin/env python3
"""tools/bench_gan.py -- BASELINE.json configs[4]: one GAN iteration (train.py:142-182: discriminator step + generator step)
at 64 images per GPU, fp32, synthetic data (GanStep of superresolution_amd/gan_training.py).  Prints one JSON line:
images/s, ms per iteration (HIP-event median) and the split between the two half-steps.

    python tools/bench_gan.py [--batch 64] [--steps 8] [--warmup 2]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", choices=["fp32", "f16"], default="fp32", help="generator 3x3 conv products")
    args = ap.parse_args()
    import torch
    from superresolution_amd import Discriminator, Generator, weights as W
    from superresolution_amd.gan_training import GanStep
    dev = torch.device("cuda", 0)
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.to(dev).train()
    torch.manual_seed(3)
    d = Discriminator().to(dev).train()
    step = GanStep(g, d, precision=args.precision)
    gen = torch.Generator(device="cpu").manual_seed(7)
    lr = torch.rand((args.batch, 3, 64, 64), generator=gen).to(dev)
    hr = torch.rand((args.batch, 3, 125, 125), generator=gen).to(dev)
    for _ in range(args.warmup):
        step.step(lr, hr, True)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ev[0].record()
    for i in range(args.steps):
        out = step.step(lr, hr, True)
        ev[i + 1].record()
    ev[-1].synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    med = ms[len(ms) // 2]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    d_loss, real_logits = step.d_step(lr, hr)
    e[1].record()
    step.g_step(lr, hr, real_logits)
    e[2].record()
    e[2].synchronize()
    print(json.dumps({"metric": "GAN iteration images/s (configs[4], one GPU, discriminator updated every iteration)",
                      "value": round(args.batch / (med * 1e-3), 1), "unit": "images/s", "batch": args.batch, "dtype": "f32",
                      "median_ms_per_iteration": round(med, 3), "min_ms": round(ms[0], 3), "steps": args.steps,
                      "phases_ms": {"d_step (G forward no-grad, D fwd+bwd on 2n images, AdamW)": round(e[0].elapsed_time(e[1]), 3),
                                    "g_step (G fwd, D fwd + data gradient, G bwd, clip, AdamW)": round(e[1].elapsed_time(e[2]), 3)},
                      "d_loss": float(out["d_loss"]), "g_loss": float(out["g_loss"]),
                      "note": "fp32; the reference runs both half-steps under bf16 autocast and updates D every 3rd iteration"}))


if __name__ == "__main__":
    main()
