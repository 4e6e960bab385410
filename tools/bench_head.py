#!/usr/bin/env python3
"""tools/bench_head.py -- the head (Generator.py:71,80) through the C ABI, bf16 or fp32 (split-f16 products): the
row-streaming kernel at every band count against the tile kernel, HIP-event timed.
    python tools/bench_head.py [--batch 256] [--reps 20] [--dtype bf16|f32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    args = ap.parse_args()
    import torch
    from superresolution_amd import _lib, training as T
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    n = args.batch
    g = torch.Generator(device="cpu").manual_seed(0)
    bf = args.dtype == "bf16"
    v = torch.randn((n, 128, 128, 96), generator=g).to(torch.bfloat16 if bf else torch.float32).to(dev)
    w = (torch.randn((3, 96, 9, 9), generator=g) * 0.05).to(dev)
    b = torch.zeros(3, device=dev)
    pack = T.head_pack_bf16(w) if bf else T.head_pack_split(w)
    op = L.srgen_head_nhwc_bf16 if bf else L.srgen_head_nhwc_split
    env = "SRGEN_HEAD_BF16" if bf else "SRGEN_HEAD_F32S"
    y = torch.empty((n, 3, 125, 125), dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def timed():
        fn = lambda: _lib.check(op(v.data_ptr(), pack.data_ptr(), b.data_ptr(), y.data_ptr(), n, s))
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.reps

    flop = 2.0 * n * 128 * 128 * 864 * 32 * (1 if bf else 3)
    os.environ[env] = "tile"
    t = timed()
    print(f"batch {n} {args.dtype}: tile kernel {t:8.1f} us ({flop / t / 1e6:5.0f} TFLOP/s executed, {v.numel() * v.element_size() / t / 1e3:5.0f} GB/s of input)")
    del os.environ[env]
    for bands in (0, 1, 2, 3, 4, 6, 8):
        if bands:
            os.environ["SRGEN_HEAD_BANDS"] = str(bands)
        t = timed()
        print(f"  streaming, bands {bands or 'default'}: {t:8.1f} us ({v.numel() * v.element_size() / t / 1e3:5.0f} GB/s of input)")


if __name__ == "__main__":
    main()
