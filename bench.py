#!/usr/bin/env python3
"""bench.py -- generator images/s (64x64 -> 125x125) on N MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|bf16] [--workload cfg1|cfg2|cfg3|cfg4]

Launching.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU, starts
N children of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous), waits for them and
exits with their worst code.  Under torch.distributed.run (WORLD_SIZE already set) it is a rank.  A rank fails
loudly when the ranks RCCL saw differ from --gpus; `n_gpus` on the line is dist.get_world_size().

Workloads (BASELINE.json `configs`), both measured by every run and printed by rank 0 on ONE JSON line:
  cfg1  (the headline, `value`)  a "step" is one eval forward of the drop-in Generator over one batch of 256
        synthetic LR images already resident in HBM, fp32 (configs[1]); every rank runs its own 256 images
        (weak scaling: images are independent, no data-path collective).  Exactly K steps are timed between
        barrier + synchronize pairs (max over ranks) -> `value`; a second leg of >= 50 steps is timed step by
        step with HIP events on the launch stream -> `median_ms_per_step` (BASELINE.md "GPU timing").
  cfg2  (object `cfg2`)  configs[2] as specified: bf16, every rank owns 4 whole batches of 256 (8192 images at
        8 GPUs), runs them through superresolution_amd.evaluate.validate -- generator forward, the PSNR/SSIM/L1
        kernel per batch, sums kept on the device -- and ONE RCCL SUM all-reduce of {sum psnr, sum ssim, sum l1,
        batches} ends it (train.py:57-72 semantics).  `--workload cfg2` makes this leg the headline instead.
  cfg3 / cfg4  (objects `cfg3` / `cfg4` of the default line, each with an "fp32" and an "f16" entry = the two conv
        arithmetics, <= 10 steps; `--workload cfg3|cfg4 [--train-precision f16]`: a line of their own)  configs[3] / [4]:
        the pre-train step and the GAN iteration at 64 images per GPU, fp32 tensors, data parallel: every rank steps on
        its own batch and one all-reduce per optimizer step averages the flat gradient buffer
        (superresolution_amd/training.py, gan_training.py).  Each carries `time_by_operator_class` (one single-stream
        step with a HIP event pair around every operator call), a `roofline` for the class with the largest share, and a
        `cpu_baseline`: the same step with stock torch CPU ops + autograd (oracle/torch_ref.py) at batch 4.

Further objects on the line:
  roofline      dominant kernel = the 33 trunk 3x3 convs (75 % of the FLOPs).  `achieved` = FLOPs the matrix
                pipe EXECUTES per launch / average launch duration measured live with HIP events on the launch
                stream (srgen_forward_profile); `frac` = achieved / peak of the instruction the kernel issues
                (2500 TFLOP/s f16 / bf16 dense, 157.3 TFLOP/s fp32 matrix: MI355X_MICROARCH.md).  The default fp32
                kernel keeps fp32 tensors and fp32 accumulation and forms every product from two-term f16 splits
                of both operands: THREE f16 MFMAs per algorithmic multiply-add, so executed = 3 x algorithmic and
                the peak is the f16 one.  SRGEN_F32_CONV3X3=wino (Winograd F(2x2,3x3), 16/36 of the multiplies) and
                =direct run on the exact-fp32 MFMA and are priced against 157.3.  The algorithmic rate (SURVEY.md
                8(d): 2*4096*96*864 FLOP per image and launch) is reported beside it as `algorithmic_tflops`.
                traffic = HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/traffic_*.json).
  f32_exact_mfma  the same cfg1 leg with the 3x3 convs on the exact-fp32 MFMA (Winograd kernel), a few steps: the
                number to quote if split products are not accepted as fp32 (errors of both against the reference's
                fp32 output are in `psnr`: max_rel_err_f32 and max_rel_err_f32_exact_mfma).
  bf16          the north-star dtype on the same line: batch-256 forward images/s, trunk conv ms per launch,
                frac of the 2.5 PF peak, traffic, PSNR delta.
  psnr          the "PSNR delta vs reference" half of the metric: this library's output on the two golden inputs
                against the REFERENCE's output for them (tests/golden/generator_fp32.npz, written by running the
                real reference), both scored with utils.psnr semantics against the same synthetic target.
  cpu_baseline  the reference's forward restated with stock torch CPU ops (oracle/torch_ref.py, kind "port") on
                the box's host cores, batch 16, bounded to ~10-20 s.  Runs BEFORE the GPU legs.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMAGE = 2 * 14_407_045_632
TRUNK_CONV_FLOP_PER_IMAGE = 2 * 4096 * 96 * 864  # one 96->96 3x3 conv at 64x64
PEAK = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}   # dense TFLOP/s, MI355X_MICROARCH.md
WINO_EXECUTED = 16.0 / 36.0                    # F(2x2,3x3): multiplies executed / algorithmic
SPLIT_EXECUTED = 3.0                           # split path: hi*hi + hi*lo + lo*hi on the f16 MFMA


def f32_path():
    """Which fp32 3x3 kernel a handle created now would run (srgen_api.cpp: srgen_create)."""
    return os.environ.get("SRGEN_F32_CONV3X3") or "split"


def f32_pricing(dt_name):
    """(path label, executed / algorithmic, peak TFLOP/s of the MFMA the path issues, kernel name, what is counted)."""
    if dt_name == "bf16":
        return ("bf16", 1.0, PEAK["bf16"], "conv3x3_bf16_kernel<64,false,*,*>", "algorithmic = executed (direct implicit GEMM)")
    path = f32_path()
    if path == "split":
        return ("split", SPLIT_EXECUTED, PEAK["f16"], "conv3x3_f32s_kernel<64,false,*,*>",
                "FLOPs the matrix pipe executes: fp32 tensors and accumulation, each product = 3 v_mfma_f32_32x32x16_f16 "
                "terms (hi*hi + hi*lo + lo*hi of two-term f16 splits), priced against the f16 dense peak")
    if path == "wino":
        return ("wino", WINO_EXECUTED, PEAK["f32"], "conv3x3_f32_wino_kernel<64,false,*,*,*>",
                "FLOPs the matrix pipe executes: Winograd F(2x2,3x3) runs 16/36 of the direct conv's multiplies")
    return ("direct", 1.0, PEAK["f32"], "conv3x3_f32_kernel<64,false,*,*>", "algorithmic = executed (direct implicit GEMM)")
BATCH = 256                                    # config.py:8, the reference batch
CFG2_BATCHES_PER_GPU = 4                       # 8192 images / 256 / 8 GPUs


def cpu_baseline(batch=16, iters=64, threads=None):
    import torch
    from oracle import torch_ref
    from superresolution_amd import weights as W
    threads = threads or min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    sd = W.synthetic_state_dict(0)
    x = torch.from_numpy(W.synthetic_lr_batch(batch, seed=0))
    torch_ref.generator_forward(sd, x[:2])            # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        torch_ref.generator_forward(sd, x)
    dt = time.perf_counter() - t0
    return {"value": round(batch * iters / dt, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{iters} fp32 eval forwards of batch {batch} (64x64->125x125) through oracle/torch_ref.py "
                      f"(stock torch CPU conv path), {dt:.1f} s"}


def spawn_ranks(n, argv):
    """One child process per GPU, started before this process has made any GPU call."""
    import torch
    have = torch.cuda.device_count()                  # counts devices without initialising the GPU
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible; refusing to report an {n}-GPU number")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    deadline = time.time() + 1500
    codes = [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        if failed or time.time() > deadline:
            for i, p in enumerate(procs):             # a rank died or hung: end exactly the children we started
                if codes[i] is None:
                    p.kill()
                    codes[i] = p.wait()
            break
        time.sleep(0.2)
    bad = [c for c in codes if c != 0]
    raise SystemExit(bad[0] if bad else 0)


def cpu_training_baseline(workload, batch=4, threads=None):
    """One optimizer step (cfg3) / one GAN iteration with a discriminator update (cfg4) of the reference's training loop
    restated with stock torch CPU ops and torch autograd (oracle/torch_ref.py; fp32), on the box's host cores."""
    import torch
    from oracle import torch_ref
    from superresolution_amd import Discriminator, weights as W
    threads = threads or min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    sd = W.synthetic_state_dict(0)
    P = {k: v.clone().requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    opt = torch.optim.AdamW(list(P.values()), lr=1e-4)
    lr = torch.from_numpy(W.synthetic_lr_batch(batch, 2, "dense"))
    hr = torch.from_numpy(W.synthetic_hr_batch(batch, seed=1))
    if workload == "cfg3":
        run = lambda: torch_ref.pretrain_step(P, opt, lr, hr)
        what = "pre-train steps (train.py:90-97)"
    else:
        torch.manual_seed(3)
        D = {k: v.detach().clone().requires_grad_() for k, v in Discriminator().named_parameters()}
        opt_d = torch.optim.AdamW(list(D.values()), lr=1e-6)
        it = [0]

        def run():
            it[0] += 1
            return torch_ref.gan_iteration(P, D, opt, opt_d, lr, hr, update_d=(it[0] - 1) % 3 == 0)
        what = "GAN iterations, discriminator updated every 3rd (train.py:145-176, config.py:21)"
    run()
    iters = 3
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    dt = time.perf_counter() - t0
    return {"value": round(batch * iters / dt, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{iters} fp32 {what} at batch {batch} through oracle/torch_ref.py (stock torch CPU ops + autograd), {dt:.1f} s"}


def training_object(workload, precision, dev, rank, n_ranks, use_dist, n=64, steps=10, warmup=3, reuse_forward=False,
                    cpu=None):
    """BASELINE.json configs[3] (pre-train step) / configs[4] (GAN iteration): `n` images per GPU, data parallel -- every
    rank steps on its own batch and ONE all-reduce per optimizer averages the flat gradient buffer (RCCL).  Returns the
    object bench.py prints (a line of its own with --workload cfg3|cfg4, objects `cfg3` / `cfg4` of the default line).
    Its `roofline` is that of the operator class with the largest share of one step's device time, measured live: every
    operator call of one single-stream step bracketed by HIP events (training.profile_ops)."""
    import torch
    import torch.distributed as dist
    from superresolution_amd import Discriminator, Generator, training as T, weights as W
    from superresolution_amd.gan_training import GanStep
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.to(dev).train()
    gen = torch.Generator(device="cpu").manual_seed(4321 + rank)
    lr = torch.rand((n, 3, 64, 64), generator=gen).to(dev)
    hr = torch.rand((n, 3, 125, 125), generator=gen).to(dev)
    if workload == "cfg3":
        step = T.PretrainStep(g, lr=1e-4, precision=precision)
        owner = step
        run = lambda: step.step(lr, hr)
        name = (f"BASELINE.json configs[3]: pre-train step (forward, loss, backward, AdamW), {n} images per GPU, "
                f"{'bf16' if precision == 'bf16' else 'fp32'} tensors")
        metric = "pre-train step images/s"
    else:
        torch.manual_seed(3)
        d = Discriminator().to(dev).train()
        gan = GanStep(g, d, precision=precision, reuse_forward=reuse_forward)
        owner = gan.gen
        it = [0]

        def run():
            it[0] += 1
            return gan.step(lr, hr, update_d=(it[0] - 1) % 3 == 0)["g_loss"]      # config.py:21 discriminator_update_freq = 3
        name = (f"BASELINE.json configs[4]: GAN iteration (discriminator half-step every 3rd iteration + generator "
                f"half-step), {n} images per GPU, {'bf16 generator' if precision == 'bf16' else 'fp32'} tensors"
                + (", generator forward of the discriminator half-step reused by the generator half-step" if reuse_forward else ""))
        metric = "GAN iteration images/s"

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(warmup, 3)):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = run()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # where the step's device time goes: three steps (one discriminator update among them for cfg4) on ONE stream, every
    # operator call between two HIP events; done twice, each class taking the smaller of its two sums (a one-off stall of
    # the host inside one call -- an allocator growth, the event pool -- otherwise lands on whichever class it hit)
    side, owner.side = owner.side, None
    try:
        rounds = []
        for _ in range(2):
            with T.profile_ops() as pr:
                for _ in range(3):
                    run()
            rounds.append(pr)
    finally:
        owner.side = side
    prof = {k: min((r[k] for r in rounds if k in r), key=lambda v: v[0]) for k in rounds[0]}
    total = sum(v[0] for v in prof.values()) or 1.0
    classes = {k: {"ms_per_step": round(v[0] / 3, 3), "launches_per_step": round(v[1] / 3, 1), "share": round(v[0] / total, 3)}
               for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    top, (top_ms, top_calls, top_work) = max(prof.items(), key=lambda kv: kv[1][0])
    f16, b16 = precision == "f16", precision == "bf16"
    wpath = T.wgrad_path()
    if top.startswith("conv3x3"):
        split = (T.conv_path() if "forward" in top else wpath) == "split"
        ratio, peak = ((1.0 if (f16 or b16) else SPLIT_EXECUTED), PEAK["f16"]) if split else (WINO_EXECUTED, PEAK["f32"])
        if b16:
            kname = ("c3::conv3x3_bf16_kernel<*> (the eval forward's kernel, identity epilogue)" if "forward" in top
                     else "wg1::conv3x3_wgrad_split_kernel<1, bf16 operands> + conv3x3_wgrad_reduce_kernel")
        else:
            kname = ("c3s::conv3x3_f32s_kernel<*, TERMS=%d>" % (1 if f16 else 3) if "forward" in top
                     else "wg1::conv3x3_wgrad_split_kernel<%d> + conv3x3_wgrad_reduce_kernel" % (1 if f16 else 3)) if split else top
        ach = top_work * ratio / (top_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": f"{kname} ({top}: {round(top_calls / 3)} launches per step, the class with the "
                                           f"largest share of the step's device time: {classes[top]['share']:.0%})",
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "ms_per_launch": round(top_ms / top_calls, 4), "flop_per_launch_executed": top_work * ratio / top_calls,
                "flop_per_launch_algorithmic": top_work / top_calls,
                "algorithmic_tflops": round(top_work / (top_ms * 1e-3) / 1e12, 2)}
    else:
        ach = top_work / (top_ms * 1e-3) / 1e9 if top_work else None
        roof = {"bound": "hbm" if top_work else "mfma", "kernel": f"{top} ({round(top_calls / 3)} calls per step, largest share: "
                                                                  f"{classes[top]['share']:.0%})",
                "achieved": round(ach, 1) if ach else None, "peak": 8000.0, "unit": "GB/s",
                "frac": round(ach / 8000.0, 4) if ach else None, "traffic": None, "ms_per_launch": round(top_ms / top_calls, 4)}
    out = {
        "metric": metric, "value": round(n * n_ranks * steps / dt, 1), "unit": "images/s", "n_gpus": n_ranks,
        "steps": steps, "warmup": max(warmup, 3), "ms_per_step": round(dt / steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("bf16 tensors and products, f32 accumulation" if b16 else
                  "f16 products, f32 accumulation and tensors" if f16 else "f32"), "data": "synthetic",
        "config": {"workload": name, "batch_per_gpu": n, "global_batch": n * n_ranks, "parallelism": f"dp{n_ranks}",
                   "conv_products": ("f16: the generator's 3x3 convs (forward, data and weight gradient) multiply operands "
                                     "rounded to f16 after power-of-two range scaling, ONE f16 MFMA per product -- the reference "
                                     "runs this step under bf16 autocast (train.py:91); tests/test_gpu_training.py shows the "
                                     "gradients closer to the reference's fp32 step than its own autocast step is") if f16
                                    else ("bf16: activation and gradient tensors are bf16 in HBM, every conv multiplies bf16 "
                                          "operands with fp32 accumulation, BatchNorm / PReLU / attention compute in fp32 and "
                                          "round once when they store -- the storage and product types of the reference's own "
                                          "step under torch.autocast(bfloat16) (train.py:91); tests/test_gpu_training_bf16.py: "
                                          "as close to the reference's fp32 step as its own autocast step is") if b16
                                    else "fp32 accuracy: two-term f16 splits of both operands, THREE f16 MFMAs per product",
                   "precision": ("bf16 tensors, fp32 accumulation, parameters / statistics / parameter gradients / images fp32"
                                 if b16 else
                                 f"fp32 tensors and accumulation; 3x3 conv kernels: {T.conv_path()} (forward, data gradient), "
                                 f"{wpath} (weight gradient); operands range-scaled by powers of two"),
                   "collective": "one all-reduce of the flat gradient buffer per optimizer step (12.86 MB generator, "
                                 "23.5 MB discriminator)"},
        "loss_last": float(loss), "roofline": roof, "time_by_operator_class": classes, "cpu_baseline": cpu}
    if workload == "cfg4":
        out["config"]["discriminator"] = (
            "fp32 tensors; 64..512-channel convs on bf16 products with fp32 accumulation -- forward, data gradient and weight "
            "gradient (LDS-tiled kernels srdisc_conv_bf16.hip / srdisc_wgrad_bf16.hip), what bf16 autocast gives them at "
            "train.py:148-152,166-174; stem and tail on fp32 products" if b16 else
            "fp32 tensors and products (v_mfma_f32_16x16x4_f32)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default=None, help="dtype of the cfg1 leg (default f32)")
    ap.add_argument("--workload", choices=["cfg1", "cfg2", "cfg3", "cfg4"], default="cfg1",
                    help="which leg is the headline `value` (cfg3 / cfg4: the training steps, their own line)")
    ap.add_argument("--train-batch", type=int, default=64, help="images per GPU per training step (cfg3 / cfg4)")
    ap.add_argument("--train-precision", choices=["fp32", "f16", "bf16"], default="fp32",
                    help="cfg3 / cfg4: products of the generator's 3x3 convs -- fp32 (three f16 MFMAs per product) or f16 "
                         "(operands rounded to f16, fp32 accumulation: what the reference's bf16 autocast does to its convs)")
    ap.add_argument("--gan-reuse-forward", action="store_true",
                    help="cfg4: GanStep(reuse_forward=True) -- one generator forward per discriminator-updating iteration "
                         "instead of the reference's two (same tensors bit for bit); off: the reference's schedule")
    ap.add_argument("--batch", type=int, default=BATCH, help="images per GPU per step (cfg1)")
    ap.add_argument("--cfg2-reps", type=int, default=5)
    ap.add_argument("--median-steps", type=int, default=50, help="HIP-event-timed steps of the median leg (>= 50)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-training-legs", action="store_true", help="default line without the cfg3 / cfg4 objects")
    ap.add_argument("--lean", action="store_true", help="profiling runs (rocprofv3): only the K timed steps of --dtype and "
                    "the per-kernel HIP-event leg; the line then carries no bf16 / cfg2 / psnr / hbm objects")
    ap.add_argument("--dry-run", action="store_true", help="ranks report their rendezvous environment and exit "
                    "(launcher rehearsal on a box without GPUs)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])
        return
    dtype = args.dtype or "f32"
    if args.dry_run:
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}),
              flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for the wrong rank count")

    cpu = None
    cpu_train = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload in ("cfg1", "cfg2"):
        cpu = cpu_baseline()                          # host-only legs first: the GPU legs then run back to back
        if not args.lean and not args.no_training_legs:
            cpu_train = {w: cpu_training_baseline(w) for w in ("cfg3", "cfg4")}

    import numpy as np
    import torch
    import torch.distributed as dist
    from superresolution_amd import Generator, evaluate as E, weights as W
    from superresolution_amd.functional import batch_metrics

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or bool(os.environ.get("SRGEN_BENCH_FORCE_DIST"))   # the latter: 1-rank RCCL rehearsal
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"RCCL sees {dist.get_world_size()} rank(s), --gpus {args.gpus}")
    n_ranks = dist.get_world_size() if use_dist else 1

    if args.workload in ("cfg3", "cfg4"):
        cpu_t = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu_t = cpu_training_baseline(args.workload)
        obj = training_object(args.workload, args.train_precision, dev, rank, n_ranks, use_dist, n=args.train_batch,
                              steps=args.steps, warmup=args.warmup, reuse_forward=args.gan_reuse_forward, cpu=cpu_t)
        if rank == 0:
            print(json.dumps(obj), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.to(dev).eval()
    # per-rank synthetic LR batch (different images per rank), resident in HBM before timing
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x32 = torch.rand((args.batch, 3, 64, 64), generator=gen).to(dev)
    xs = {"f32": x32, "bf16": x32.to(torch.bfloat16)}

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(v):
        if not use_dist:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_forward(dt_name, steps, warmup):
        """The contract's region: W untimed steps, then exactly K steps between barrier + synchronize pairs."""
        x = xs[dt_name]
        for _ in range(warmup):
            y = g(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = g(x)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        assert torch.isfinite(y.float()).all()
        return dt

    def median_forward(dt_name, steps, min_seconds=0.0, cap=400):
        """Per-step HIP events on the stream the forward launches on (torch's current stream); median."""
        x = xs[dt_name]
        evs = [torch.cuda.Event(enable_timing=True)]
        evs[0].record()
        t0 = time.perf_counter()
        done = 0
        while done < steps or (time.perf_counter() - t0 < min_seconds and done < cap):
            g(x)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
            done += 1
            if done % 16 == 0:
                e.synchronize()                       # keep the launch queue short: honest per-step intervals
        evs[-1].synchronize()
        ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(done))
        return {"median_steps": done, "median_ms_per_step": round(ms[done // 2], 4), "min_ms_per_step": round(ms[0], 4),
                "p90_ms_per_step": round(ms[int(done * 0.9)], 4)}

    def kernel_profile(dt_name, reps=3):
        acc = {}
        for _ in range(reps):
            for k, (ms, cnt) in g.profile_forward(xs[dt_name]).items():
                a = acc.setdefault(k, [0.0, 0])
                a[0] += ms
                a[1] += cnt
        return {k: {"ms_per_launch": round(v[0] / max(v[1], 1), 4), "launches_per_forward": v[1] // reps}
                for k, v in acc.items()}

    def roofline(dt_name, kernels):
        ms_launch = kernels["conv3x3_trunk"]["ms_per_launch"]
        algorithmic = TRUNK_CONV_FLOP_PER_IMAGE * args.batch / (ms_launch * 1e-3) / 1e12
        path, ratio, peak, kname, counts = f32_pricing(dt_name)
        executed = algorithmic * ratio
        traffic = mfma_busy = None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{dt_name}.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("path", "wino" if dt_name == "f32" else path) == path:      # a summary measured on another kernel says nothing here
                    traffic = tj.get("conv3x3_trunk_hbm_bytes_per_launch")
                    mfma_busy = tj.get("mfma_busy_frac")
            except Exception:
                traffic = mfma_busy = None
        return {"bound": "mfma", "kernel": f"{kname} (33 trunk launches)", "achieved": round(executed, 2),
                "peak": peak, "unit": "TFLOP/s", "frac": round(executed / peak, 4),
                "traffic": traffic, "ms_per_launch": ms_launch,
                # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) from the same committed PMC summary as traffic
                "mfma_busy_by_counter": mfma_busy,
                "flop_per_launch_executed": int(TRUNK_CONV_FLOP_PER_IMAGE * args.batch * ratio),
                "flop_per_launch_algorithmic": TRUNK_CONV_FLOP_PER_IMAGE * args.batch,
                "algorithmic_tflops": round(algorithmic, 2), "counts": counts}

    def psnr_vs_reference():
        """PSNR delta vs the reference: this library's output on the two golden inputs against the reference's
        own output for them (fixture written by the real reference), same target, utils.psnr semantics."""
        gold = np.load(os.path.join(ROOT, "tests", "golden", "generator_fp32.npz"))
        ref = torch.from_numpy(gold["y_dense_sparse"]).to(dev)
        xin = torch.from_numpy(np.concatenate([W.synthetic_lr_batch(1, 0, "dense"),
                                               W.synthetic_lr_batch(1, 0, "sparse")], 0)).to(dev)
        hr = torch.from_numpy(W.synthetic_hr_batch(2, seed=1)).to(dev)
        p_ref = float(batch_metrics(ref, hr)[0])
        out = {"target": "synthetic hr seed 1, golden inputs (dense, sparse), weights seed 0",
               "psnr_reference_output_db": round(p_ref, 5), "bar_bf16": "|delta| <= 0.05 dB (BASELINE.json north_star)"}
        for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            y = g(xin.to(dt)).float()
            p = float(batch_metrics(y, hr)[0])
            out[f"psnr_{name}_db"] = round(p, 5)
            out[f"psnr_delta_{name}_db"] = round(p - p_ref, 6)
            out[f"psnr_{name}_vs_reference_output_db"] = round(float(batch_metrics(y, ref)[0]), 2)
            out[f"max_rel_err_{name}"] = float((y - ref).abs().max() / ref.abs().max())
        if g_exact is not None:
            y = g_exact(xin).float()
            out["max_rel_err_f32_exact_mfma"] = float((y - ref).abs().max() / ref.abs().max())
        out["f32_conv3x3_path"] = f32_path()
        return out

    def cfg2_leg(reps):
        """configs[2]: bf16, this rank's 4 batches of 256 through validate() + the RCCL metric all-reduce."""
        nb = CFG2_BATCHES_PER_GPU
        gl = torch.Generator(device="cpu").manual_seed(4321 + rank)
        lr = torch.rand((nb * BATCH, 3, 64, 64), generator=gl).to(dev)
        gh = torch.Generator(device=dev).manual_seed(1 + rank)                  # hr target generated on the device
        hr = torch.rand((nb * BATCH, 3, 125, 125), generator=gh, device=dev)
        mine = E.shard_batches(nb * n_ranks, rank, n_ranks)
        assert len(mine) == nb

        def run():
            return E.validate(g, E.batched(lr, hr, BATCH, range(nb)), device=dev, forward_dtype=torch.bfloat16)
        m = run()                                                               # warm-up (workspace, RCCL channel)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            m = run()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        imgs = nb * BATCH * n_ranks * reps
        return {"workload": f"BASELINE.json configs[2]: bf16, {nb * BATCH * n_ranks} images = {nb} batches of {BATCH} per "
                            f"GPU x {n_ranks} GPU(s), evaluate.validate (forward + PSNR/SSIM/L1 kernel per batch) + one "
                            "RCCL SUM all-reduce of 4 scalars" + ("" if use_dist else " (single process: no process group)"),
                "images_per_s": round(imgs / dt, 2), "ms_per_validate": round(dt / reps * 1e3, 3), "reps": reps,
                "n_gpus": n_ranks, "collective": "nccl all_reduce(SUM, 4 x f64)" if use_dist else None,
                "metrics_mean_over_batches": {"psnr": round(m[0], 5), "ssim": round(m[1], 6), "l1": round(m[2], 6)},
                "_dt": dt, "_imgs": imgs}

    # ---- legs ---------------------------------------------------------------------------------------------
    dt_main = timed_forward(dtype, args.steps, args.warmup)
    if args.lean:
        if rank == 0:
            kern = kernel_profile(dtype)
            v = args.batch * n_ranks * args.steps / dt_main
            print(json.dumps({"metric": "generator images/s (64x64->125x125)", "value": round(v, 2), "unit": "images/s",
                              "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(dt_main / args.steps * 1e3, 3), "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
                              "config": {"workload": f"lean profiling run of the cfg1 leg, batch {args.batch}, {dtype}"},
                              "roofline": roofline(dtype, kern), "kernels": kern}), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    med = median_forward(dtype, max(50, args.median_steps), min_seconds=4.0 if world == 1 else 0.0)
    # the same leg on the exact-fp32 MFMA (Winograd kernel): a second module whose handle is created under the switch
    g_exact, exact_obj = None, None
    if f32_path() == "split":
        saved = os.environ.get("SRGEN_F32_CONV3X3")
        os.environ["SRGEN_F32_CONV3X3"] = "wino"
        try:
            g_exact = Generator()
            g_exact.load_state_dict(W.synthetic_state_dict(0))
            g_exact = g_exact.to(dev).eval()
            g_main, g = g, g_exact
            steps_exact = max(10, args.steps // 5)
            dt_exact = timed_forward("f32", steps_exact, 2)
            kern_exact = kernel_profile("f32", reps=1) if rank == 0 else None
            g = g_main
        finally:
            if saved is None:
                del os.environ["SRGEN_F32_CONV3X3"]
            else:
                os.environ["SRGEN_F32_CONV3X3"] = saved
        if rank == 0:
            v = args.batch * n_ranks * steps_exact / dt_exact
            tr = kern_exact["conv3x3_trunk"]["ms_per_launch"]
            alg = TRUNK_CONV_FLOP_PER_IMAGE * args.batch / (tr * 1e-3) / 1e12
            exact_obj = {"path": "SRGEN_F32_CONV3X3=wino: Winograd F(2x2,3x3) on v_mfma_f32_16x16x4_f32",
                         "images_per_s": round(v, 2), "ms_per_step": round(dt_exact / steps_exact * 1e3, 3),
                         "steps": steps_exact, "trunk_ms_per_launch": tr,
                         "trunk_frac_of_fp32_mfma_peak": round(alg * WINO_EXECUTED / PEAK["f32"], 4),
                         "trunk_algorithmic_tflops": round(alg, 2)}
    other = "bf16" if dtype == "f32" else "f32"
    dt_other = timed_forward(other, max(20, args.steps // 2), 3)
    steps_other = max(20, args.steps // 2)
    med_other = median_forward(other, 50)
    c2 = cfg2_leg(args.cfg2_reps)
    # configs[3] / [4] on the same line: both conv arithmetics, a few steps each (their own lines: --workload cfg3|cfg4)
    train_objs = {}
    if not args.no_training_legs:
        for wl in ("cfg3", "cfg4"):
            train_objs[wl] = {}
            for prec in ("fp32", "f16", "bf16"):
                o = training_object(wl, prec, dev, rank, n_ranks, use_dist, n=args.train_batch,
                                    steps=max(3, min(10, args.steps)), warmup=3, cpu=cpu_train.get(wl))
                for k in ("higher_is_better", "scaling", "vs_baseline", "data", "n_gpus", "unit"):
                    o.pop(k, None)
                train_objs[wl][prec] = o
            torch.cuda.empty_cache()

    out = None
    if rank == 0:
        kern = {d: kernel_profile(d) for d in ("f32", "bf16")}
        roofs = {d: roofline(d, kern[d]) for d in ("f32", "bf16")}
        psnr = psnr_vs_reference()
        hbm = {}
        for d, elt in (("f32", 4), ("bf16", 2)):
            ca_bytes = 2 * args.batch * 64 * 64 * 96 * elt          # one read + one write (pool fused upstream)
            ca_ms = kern[d]["chan_attn"]["ms_per_launch"]
            hbm[f"chan_attn_{d}"] = {"algorithmic_bytes": ca_bytes, "ms_per_launch": ca_ms,
                                     "GBps": round(ca_bytes / (ca_ms * 1e-3) / 1e9, 1),
                                     "frac_of_8TBps": round(ca_bytes / (ca_ms * 1e-3) / 8e12, 3)}
        from superresolution_amd import bilinear_resize
        z = torch.rand((args.batch, 3, 128, 128), device=dev)
        for _ in range(3):
            bilinear_resize(z, (125, 125))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                                              # the op launches on torch's current stream
        for _ in range(20):
            bilinear_resize(z, (125, 125))
        e1.record()
        e1.synchronize()
        rs_ms = e0.elapsed_time(e1) / 20
        rs_bytes = args.batch * 3 * (128 * 128 + 125 * 125) * 4
        hbm["bilinear_resize_128_to_125"] = {"algorithmic_bytes": rs_bytes, "ms_per_launch": round(rs_ms, 4),
                                             "GBps": round(rs_bytes / (rs_ms * 1e-3) / 1e9, 1),
                                             "frac_of_8TBps": round(rs_bytes / (rs_ms * 1e-3) / 8e12, 3),
                                             "note": "optional op, not on the forward path (the reference crops)"}

        def fwd_obj(d, dt, steps, md):
            v = args.batch * n_ranks * steps / dt
            path, ratio, peak, _, _ = f32_pricing(d)
            e2e_ratio = 0.9668 * ratio + 0.0332 * (peak / PEAK[d])   # 9x9 stem / head (3.3 %) run direct in dtype d
            return {"images_per_s": round(v, 2), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, **md,
                    "conv3x3_path": path,
                    "end_to_end_algorithmic_tflops": round(v * FLOP_PER_IMAGE / 1e12, 2),
                    "end_to_end_frac_of_peak": round(v / n_ranks * FLOP_PER_IMAGE / 1e12 / peak * e2e_ratio, 4),
                    "trunk_ms_per_launch": kern[d]["conv3x3_trunk"]["ms_per_launch"],
                    "trunk_frac_of_peak": roofs[d]["frac"], "traffic": roofs[d]["traffic"],
                    "psnr_delta_vs_reference_db": psnr[f"psnr_delta_{d}_db"]}

        main_obj = fwd_obj(dtype, dt_main, args.steps, med)
        other_obj = fwd_obj(other, dt_other, steps_other, med_other)
        cfg1_name = (f"BASELINE.json configs[1]: Generator eval inference (BN folded), batch {args.batch} of 3x64x64 per "
                     f"GPU -> 3x125x125, {dtype}" + ("" if dtype == "f32" else " (configs[1]'s shape at configs[2]'s dtype)"))
        headline_cfg2 = args.workload == "cfg2"
        value = c2["images_per_s"] if headline_cfg2 else main_obj["images_per_s"]
        ms_step = c2["ms_per_validate"] if headline_cfg2 else main_obj["ms_per_step"]
        out = {
            "metric": "generator images/s (64x64->125x125)", "value": value, "unit": "images/s",
            "n_gpus": n_ranks, "steps": c2["reps"] if headline_cfg2 else args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "median_ms_per_step": None if headline_cfg2 else med["median_ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if headline_cfg2 else dtype, "data": "synthetic",
            "config": {"workload": c2["workload"] if headline_cfg2 else cfg1_name,
                       "batch_per_gpu": BATCH * CFG2_BATCHES_PER_GPU if headline_cfg2 else args.batch,
                       "global_batch": (BATCH * CFG2_BATCHES_PER_GPU if headline_cfg2 else args.batch) * n_ranks,
                       "parallelism": f"batch-sharded x{n_ranks}, weights replicated, one process per GPU"},
            "roofline": roofs["bf16" if headline_cfg2 else dtype],
            dtype: main_obj, other: other_obj,
            "cfg2": {k: v for k, v in c2.items() if not k.startswith("_")},
            "kernels": kern[dtype], f"kernels_{other}": kern[other], "roofline_" + other: roofs[other],
            "hbm_kernels": hbm, "psnr": psnr,
        }
        if exact_obj is not None:
            out["f32_exact_mfma"] = exact_obj
        out.update(train_objs)
        out["config"]["precision"] = {
            "f32": ("fp32 tensors in HBM, fp32 accumulation; 3x3 conv products via two-term f16 splits of both operands "
                    "(3 f16 MFMAs per product, error of an fp32 product: see psnr.max_rel_err_f32 beside "
                    "max_rel_err_f32_exact_mfma)") if f32_path() == "split" else f"exact-fp32 MFMA ({f32_path()})",
            "bf16": "bf16 tensors and MFMA operands, fp32 accumulation"}["bf16" if headline_cfg2 else dtype]
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
