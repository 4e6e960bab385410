#!/usr/bin/env python3
"""Train-mode generator forward values at the reference's batch 256 (train.py:92 without autograd):
HIP events on torch's current stream around K steps (each = forward + running-statistics update).
Prints one JSON line; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    from superresolution_amd import Generator, weights as W
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda().train()
    x = torch.rand((a.batch, 3, 64, 64), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    with torch.no_grad():
        for _ in range(a.warmup):
            g(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            y = g(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        g.eval()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.steps):
            g(x)
        e1.record()
        torch.cuda.synchronize()
        ms_eval = e0.elapsed_time(e1) / a.steps
    print(json.dumps({"metric": "generator train-mode forward images/s", "value": a.batch / (ms * 1e-3),
                      "unit": "images/s", "ms_per_step": ms, "eval_ms_per_step": ms_eval, "batch": a.batch,
                      "steps": a.steps, "dtype": "f32", "finite": bool(torch.isfinite(y).all()),
                      "config": {"workload": "Generator.train()(x) under no_grad, batch 256 of 3x64x64, "
                                             "incl. the BatchNorm running-statistics update"}}))


if __name__ == "__main__":
    main()
