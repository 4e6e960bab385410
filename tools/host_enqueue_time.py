#!/usr/bin/env python3
"""tools/host_enqueue_time.py -- is the pre-train step bound by the host?  Times, per step, (a) the host time step() takes
to return (launches queued, nothing waited for) and (b) the device time between the steps' end events.

    python tools/host_enqueue_time.py [--batch 64] [--precision bf16] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--precision", default="bf16")
    args = ap.parse_args()
    import torch
    from superresolution_amd import Generator, training as T, weights as W
    dev = torch.device("cuda", 0)
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.to(dev).train()
    step = T.PretrainStep(g, lr=1e-4, precision=args.precision)
    gen = torch.Generator(device="cpu").manual_seed(7)
    lr = torch.rand((args.batch, 3, 64, 64), generator=gen).to(dev)
    hr = torch.rand((args.batch, 3, 125, 125), generator=gen).to(dev)
    for _ in range(5):
        step.step(lr, hr)
    torch.cuda.synchronize()
    # (1) back to back: the device never idles unless the host is slower
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host = []
    ev[0].record()
    for i in range(args.steps):
        t0 = time.perf_counter()
        step.step(lr, hr)
        host.append((time.perf_counter() - t0) * 1e3)
        ev[i + 1].record()
    ev[-1].synchronize()
    devms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    # (2) each step queued behind a long device backlog: the device time of a step with the host out of the way
    x = torch.empty((1 << 28,), device=dev)
    backlog = []
    for i in range(5):
        for _ in range(40):
            x.add_(1.0)                               # ~ 0.35 ms each: the queue is never empty while step() runs
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step.step(lr, hr)
        e1.record()
        e1.synchronize()
        backlog.append(e0.elapsed_time(e1))
    host.sort()
    backlog.sort()
    print(json.dumps({"batch": args.batch, "precision": args.precision,
                      "host_ms_per_step_median": round(host[len(host) // 2], 3), "host_ms_min": round(host[0], 3),
                      "device_ms_per_step_median_back_to_back": round(devms[len(devms) // 2], 3),
                      "device_ms_per_step_behind_a_backlog": round(backlog[len(backlog) // 2], 3)}))


if __name__ == "__main__":
    main()
