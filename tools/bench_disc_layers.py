#!/usr/bin/env python3
"""tools/bench_disc_layers.py -- per-layer timings of the discriminator's 64..512-channel 3x3 convs (Discriminator.py:25-35)
through the C ABI: forward, data gradient and weight gradient of each layer on random data, HIP-event timed.

    python tools/bench_disc_layers.py [--batch 128] [--reps 20] [--products bf16|fp32]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LAYERS = [(125, 64, 64, 2), (63, 64, 128, 1), (63, 128, 128, 2), (32, 128, 256, 1), (32, 256, 256, 2), (16, 256, 512, 1),
          (16, 512, 512, 2), (8, 512, 256, 1)]           # input size, cin, cout, stride


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--products", choices=["bf16", "fp32"], default="bf16")
    args = ap.parse_args()
    import torch
    from superresolution_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    n, b16 = args.batch, args.products == "bf16"
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(0)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.reps

    def pack(w, mode):
        out = torch.empty(w.numel(), dtype=torch.bfloat16 if b16 else torch.float32, device=dev)
        fn = L.srdisc_pack_bf16 if b16 else L.srdisc_pack
        _lib.check(fn(w.data_ptr(), w.shape[1], w.shape[0], mode, out.data_ptr(), s))
        return out

    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"batch {n}, {args.products} products; us per call (TFLOP/s)")
    for (h, cin, cout, st) in LAYERS:
        oh = (h - 1) // st + 1
        x = torch.rand((n, h, h, cin), generator=g).to(dev)
        w = (torch.randn((cout, cin, 3, 3), generator=g) * 0.05).to(dev)
        bias = torch.zeros(cout, device=dev)
        y = torch.empty((n, oh, oh, cout), device=dev)
        dpre = torch.randn((n, oh, oh, cout), generator=g).to(dev)
        dx = torch.empty_like(x)
        dw, db = torch.empty_like(w), torch.empty_like(bias)
        flop = 2.0 * n * oh * oh * 9 * cin * cout
        pf = pack(w, 0)
        conv = L.srdisc_conv_nhwc_bf16p if b16 else L.srdisc_conv_nhwc
        t_f = timed(lambda: _lib.check(conv(x.data_ptr(), pf.data_ptr(), bias.data_ptr(), None, y.data_ptr(), n, h, h, cin, cout, st,
                                            0.2, s)))
        if st == 1:
            pd = pack(w, 1)
            t_d = timed(lambda: _lib.check(conv(dpre.data_ptr(), pd.data_ptr(), None, x.data_ptr(), dx.data_ptr(), n, h, h, cout, cin,
                                                1, 0.2, s)))
        else:
            pd = pack(w, 4)
            d2 = L.srdisc_conv_dgrad2_bf16p if b16 else L.srdisc_conv_dgrad2
            t_d = timed(lambda: _lib.check(d2(dpre.data_ptr(), pd.data_ptr(), x.data_ptr(), dx.data_ptr(), n, h, h, cin, cout, 0.2, s)))
        import ctypes
        nb = ctypes.c_size_t()
        _lib.check(L.srdisc_conv_wgrad_scratch_bytes(n, h, cin, cout, st, ctypes.byref(nb)))
        scratch = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        wg = L.srdisc_conv_wgrad_bf16p if b16 else L.srdisc_conv_wgrad
        t_w = timed(lambda: _lib.check(wg(x.data_ptr(), dpre.data_ptr(), n, h, h, cin, cout, st, dw.data_ptr(),
                                          db.data_ptr(), scratch.data_ptr(), nb.value, s)))
        tot["fwd"] += t_f
        tot["dgrad"] += t_d
        tot["wgrad"] += t_w
        print(f"  {h:3d}^2 {cin:3d}->{cout:3d} s{st}:  fwd {t_f:7.1f} ({flop / t_f / 1e6:6.0f})   dgrad {t_d:7.1f} ({flop / t_d / 1e6:6.0f})"
              f"   wgrad {t_w:7.1f} ({flop / t_w / 1e6:6.0f})")
    print(f"  totals: fwd {tot['fwd']:.0f} us, dgrad {tot['dgrad']:.0f} us, wgrad {tot['wgrad']:.0f} us")


if __name__ == "__main__":
    main()
