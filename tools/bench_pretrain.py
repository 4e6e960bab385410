#!/usr/bin/env python3
"""tools/bench_pretrain.py -- BASELINE.json configs[3]: one pre-train step (generator forward in train mode, loss,
backward, AdamW) at 64 images per GPU, fp32, synthetic data (PretrainStep of superresolution_amd/training.py).
Prints one JSON line: images/s, ms per step, and the share of each phase measured with HIP events on the launch stream.

    python tools/bench_pretrain.py [--batch 64] [--steps 10] [--warmup 3]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bench_ops(torch, T, dev, batch):
    """HIP-event median of each trunk operator at (batch, 64, 64, 96): one JSON line of microseconds."""
    gen = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn((batch, 64, 64, 96), generator=gen).to(dev)
    dy = torch.randn((batch, 64, 64, 96), generator=gen).to(dev)
    w = (torch.randn((96, 96, 3, 3), generator=gen) * 0.03).to(dev)
    gamma, beta = torch.rand(96, generator=gen).to(dev) + 0.5, torch.randn(96, generator=gen).to(dev) * 0.1
    slope = torch.full((96,), 0.25, device=dev)
    ws = T.Workspace(dev)
    wp, wd = T.conv_pack(w, 0), T.conv_pack(w, 1)
    _, stats = T.bn_train_fwd(x, gamma, beta, ws, slope=slope)
    ops = {"conv3x3_fwd": lambda: T.conv3x3(x, wp), "conv3x3_dgrad": lambda: T.conv3x3(dy, wd),
           "conv3x3_wgrad": lambda: T.conv3x3_wgrad(x, dy, ws),
           "bn_prelu_fwd": lambda: T.bn_train_fwd(x, gamma, beta, ws, slope=slope),
           "bn_prelu_bwd": lambda: T.bn_train_bwd(x, dy, stats, gamma, beta, ws, slope=slope),
           "conv_pack": lambda: T.conv_pack(w, 1)}
    out = {}
    for name, fn in ops.items():
        for _ in range(3):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        ev[0].record()
        for i in range(20):
            fn()
            ev[i + 1].record()
        ev[-1].synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
        out[name] = round(ms[10] * 1e3, 1)
    print(json.dumps({"batch": batch, "unit": "us", "ops": out}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", choices=["fp32", "f16"], default="fp32", help="3x3 conv products (training.conv_precision)")
    ap.add_argument("--ops", action="store_true", help="time the trunk operators one by one instead of the step")
    args = ap.parse_args()
    import torch
    from superresolution_amd import Generator, training as T, weights as W
    from superresolution_amd.functional import pixel_losses
    dev = torch.device("cuda", 0)
    if args.ops:
        return bench_ops(torch, T, dev, args.batch)
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.to(dev).train()
    step = T.PretrainStep(g, lr=1e-4, precision=args.precision)
    gen = torch.Generator(device="cpu").manual_seed(7)
    lr = torch.rand((args.batch, 3, 64, 64), generator=gen).to(dev)
    hr = torch.rand((args.batch, 3, 125, 125), generator=gen).to(dev)
    for _ in range(args.warmup):
        step.step(lr, hr)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ev[0].record()
    for i in range(args.steps):
        loss = step.step(lr, hr)
        ev[i + 1].record()
    ev[-1].synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    med = ms[len(ms) // 2]
    # phases of one step
    e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    step.optimizer.zero_grad()
    e[0].record()
    with T.conv_precision(args.precision), T.packed_weights(step.pack_jobs()):
        sr, S = step.forward(lr)
        e[1].record()
        vals = pixel_losses(sr, hr)
        dsr = T.pixel_loss_grad(sr, hr, step.ws, 1.0, 0.1)
        e[2].record()
        step.backward(dsr, S)
        e[3].record()
    step.optimizer.step()
    e[4].record()
    e[4].synchronize()
    ph = [e[i].elapsed_time(e[i + 1]) for i in range(4)]
    # FLOPs: forward 28.814 GFLOP / image; backward = data gradient + weight gradient of every conv except the
    # stem's data gradient: 2 x forward - 2 x 95.55 MMAC
    flop = args.batch * (3 * 2 * 14_407_045_632 - 2 * 95_551_488)
    print(json.dumps({"metric": "pre-train step images/s (configs[3], one GPU)", "value": round(args.batch / (med * 1e-3), 1),
                      "unit": "images/s", "batch": args.batch, "dtype": "f32", "conv_products": args.precision, "median_ms_per_step": round(med, 3),
                      "min_ms_per_step": round(ms[0], 3), "steps": args.steps, "loss_last": float(loss),
                      "phases_ms": {"forward": round(ph[0], 3), "loss+grad": round(ph[1], 3), "backward": round(ph[2], 3),
                                    "adamw": round(ph[3], 3)},
                      "algorithmic_tflops": round(flop / (med * 1e-3) / 1e12, 1),
                      "note": "fp32 tensors; 3x3 convs (forward, data and weight gradient) on the split kernels: fp32 = three f16 MFMAs per product, f16 = one (operands rounded to f16; the reference runs this step under bf16 autocast)"}))


if __name__ == "__main__":
    main()
