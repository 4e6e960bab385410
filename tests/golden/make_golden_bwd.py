#!/usr/bin/env python3
"""Generates tests/golden/train_ops.npz by running the REAL reference's modules backward (autograd) in the build
container -- the pin of the pre-train step operators (train.py:90-97,270):

    python tests/golden/make_golden_bwd.py

Same conventions as make_golden.py (reference loaded by file path; only data is stored; inputs and weights come from the
integer hash of superresolution_amd/weights.py so the GPU box regenerates them).  Contents:
  resblock_*   Generator.ResBlock(96) (Generator.py:30-42) in TRAIN mode, weights = synthetic seed 0 'blocks.0.block.*',
               x = (3, 96, 64, 64) from the hash, upstream gradient Gr from the hash; loss = sum(out * Gr):
               out (sample), dx (sample + sums), every parameter gradient (conv weights: stride-5 sample + sums; the
               rest in full), both BatchNorm batch statistics
  loss_*       weighted_l1 (train.py:18-20, restated: two lines) + 0.1 * (1 - utils.ssim) of sr = reference generator
               output on the golden inputs vs hr = synthetic seed 1: the loss value and d loss / d sr (stride-3 sample
               + sums); the same with the GAN-phase weights 1.0 / 0.3 (train.py:170-172)
  adamw_*      torch.optim.AdamW(lr=1e-4) (train.py:270 defaults) on a 4,099-element vector for three steps

`python tests/golden/make_golden_bwd.py step` writes train_step.npz instead: ONE pre-train step of the real reference
Generator (train.py:90-97 in fp32, i.e. without the bf16 autocast context: the stricter arithmetic): synthetic weights
seed 0, lr batch = 3 dense + 1 sparse images (seed 2, as generator_train.npz), hr = synthetic seed 1,
loss = weighted_l1 + 0.1 * (1 - ssim), loss.backward():
  loss, sr_sample; keys = parameter names in named_parameters() order; grad_sums (n_params, 2) = {sum, sum of squares} of
  every gradient (fp64); grad_absmax; grad_sample_<key> for a handful of tensors (stride-7 samples).

`python tests/golden/make_golden_bwd.py step_autocast` writes train_step_autocast.npz: the SAME step the way the reference
really runs it (train.py:91-94): forward and loss inside torch.autocast(dtype=bfloat16) (device type 'cpu' here: no GPU in
the build container), backward outside.  Stored: loss; grad_sample_<key> for the keys of train_step.npz; grad_rel_err
(n_params) = |g_autocast - g_fp32| / |g_fp32| (2-norms) of every parameter gradient and flat_cosine of the whole gradient
against the fp32 step: the reference's OWN mixed-precision deviation, the bar for this library's f16-product mode.
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


def hash_tensor(seed, key, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy((lo + (hi - lo) * W.hash_uniform(seed, key, n)).astype(np.float32).reshape(shape))


def sums(t):
    t = t.double()
    return np.array([t.sum().item(), (t * t).sum().item()])


def weighted_l1(sr, hr):                                   # train.py:18-20
    weight = (hr > 0.01).float() * 9 + 1
    return (weight * (sr - hr).abs()).mean()


def main():
    torch.set_num_threads(8)
    G = load_ref("ref_generator", "Generator.py")
    U = load_ref("ref_utils", "utils.py")
    out = {}

    # ---- ResBlock, train mode ------------------------------------------------------------------------------
    sd = W.synthetic_state_dict(0)
    rb = G.ResBlock(96)
    rb.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items() if k.startswith("blocks.0.")})
    rb.train()
    x = hash_tensor(11, "bwd.x", (3, 96, 64, 64)).requires_grad_(True)
    gr = hash_tensor(12, "bwd.g", (3, 96, 64, 64))
    y = rb(x)
    (y * gr).sum().backward()
    out["resblock_out_sample"] = y.detach().flatten()[::257].numpy()
    out["resblock_dx_sample"] = x.grad.flatten()[::257].numpy()
    out["resblock_dx_sums"] = sums(x.grad)
    for k, p in rb.named_parameters():
        g = p.grad
        if g.dim() == 4:
            out[f"resblock_grad_{k}_sample"] = g.flatten()[::5].numpy()
            out[f"resblock_grad_{k}_sums"] = sums(g)
        else:
            out[f"resblock_grad_{k}"] = g.numpy().copy()
    for i in (1, 4):                                         # batch statistics, recovered from the running-stat update
        bn = rb.block[i]
        m = 3 * 64 * 64
        mean = (bn.running_mean - 0.9 * sd[f"blocks.0.block.{i}.running_mean"]) / 0.1
        var_unbiased = (bn.running_var - 0.9 * sd[f"blocks.0.block.{i}.running_var"]) / 0.1
        out[f"resblock_bn{i}_mean"] = mean.numpy().copy()
        out[f"resblock_bn{i}_var_biased"] = (var_unbiased * (m - 1) / m).numpy().copy()

    # ---- pixel losses ----------------------------------------------------------------------------------------
    gold = np.load(os.path.join(OUT, "generator_fp32.npz"))
    hr = torch.from_numpy(W.synthetic_hr_batch(2, seed=1))
    for tag, w_l1, w_ssim in (("pre", 1.0, 0.1), ("gan", 1.0, 0.3)):
        sr = torch.from_numpy(gold["y_dense_sparse"]).clone().requires_grad_(True)
        loss = w_l1 * weighted_l1(sr, hr) + w_ssim * (1 - U.ssim(sr, hr))
        loss.backward()
        out[f"loss_{tag}_value"] = np.array(loss.item())
        out[f"loss_{tag}_dsr_sample"] = sr.grad.flatten()[::3].numpy()
        out[f"loss_{tag}_dsr_sums"] = sums(sr.grad)

    # ---- AdamW -------------------------------------------------------------------------------------------------
    p = torch.nn.Parameter(hash_tensor(13, "adamw.p", (4099,)))
    opt = torch.optim.AdamW([p], lr=1e-4)
    for step in range(1, 4):
        p.grad = hash_tensor(13 + step, "adamw.g", (4099,), -0.05, 0.05)
        opt.step()
        out[f"adamw_p{step}"] = p.detach().numpy().copy()

    np.savez_compressed(os.path.join(OUT, "train_ops.npz"), **out)
    print("wrote train_ops.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


def main_step():
    torch.set_num_threads(8)
    G = load_ref("ref_generator", "Generator.py")
    U = load_ref("ref_utils", "utils.py")
    g = G.Generator()
    g.load_state_dict(W.synthetic_state_dict(0), strict=True)
    g.train()
    lr = torch.from_numpy(np.concatenate([W.synthetic_lr_batch(3, 2, "dense"), W.synthetic_lr_batch(1, 2, "sparse")], 0))
    hr = torch.from_numpy(W.synthetic_hr_batch(4, seed=1))
    sr = g(lr)
    loss = weighted_l1(sr, hr) + 0.1 * (1 - U.ssim(sr, hr))
    loss.backward()
    names = [k for k, _ in g.named_parameters()]
    out = {"loss": np.array(loss.item()), "sr_sample": sr.detach().flatten()[::97].numpy(), "keys": np.array(names),
           "grad_sums": np.stack([sums(p.grad) for _, p in g.named_parameters()]),
           "grad_absmax": np.array([p.grad.abs().max().item() for _, p in g.named_parameters()])}
    for k in ("input.0.weight", "input.1.weight", "blocks.0.block.0.weight", "blocks.8.fc.0.weight", "blocks.8.fc.2.bias",
              "blocks.16.block.3.weight", "blocks.17.fc.2.weight", "mid.0.weight", "mid.1.weight", "upsample.conv.weight",
              "upsample.conv.bias", "upsample.act.weight", "output.0.weight", "output.1.weight", "output.2.weight",
              "output.2.bias"):
        out["grad_sample_" + k] = dict(g.named_parameters())[k].grad.flatten()[::7].numpy()
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **out)
    print("wrote train_step.npz: loss", loss.item(), "params", len(names))


SAMPLE_KEYS = ("input.0.weight", "input.1.weight", "blocks.0.block.0.weight", "blocks.8.fc.0.weight", "blocks.8.fc.2.bias",
               "blocks.16.block.3.weight", "blocks.17.fc.2.weight", "mid.0.weight", "mid.1.weight", "upsample.conv.weight",
               "upsample.conv.bias", "upsample.act.weight", "output.0.weight", "output.1.weight", "output.2.weight",
               "output.2.bias")


def main_step_autocast():
    torch.set_num_threads(8)
    G = load_ref("ref_generator", "Generator.py")
    U = load_ref("ref_utils", "utils.py")
    lr = torch.from_numpy(np.concatenate([W.synthetic_lr_batch(3, 2, "dense"), W.synthetic_lr_batch(1, 2, "sparse")], 0))
    hr = torch.from_numpy(W.synthetic_hr_batch(4, seed=1))
    grads, losses = {}, {}
    for tag in ("fp32", "autocast"):
        g = G.Generator()
        g.load_state_dict(W.synthetic_state_dict(0), strict=True)
        g.train()
        with torch.autocast(device_type="cpu", dtype=torch.bfloat16, enabled=tag == "autocast"):
            sr = g(lr)
            loss = weighted_l1(sr, hr) + 0.1 * (1 - U.ssim(sr, hr))
        loss.backward()
        grads[tag] = {k: p.grad.detach().double() for k, p in g.named_parameters()}
        losses[tag] = float(loss.item())
    names = list(grads["fp32"].keys())
    rel = np.array([((grads["autocast"][k] - grads["fp32"][k]).norm() / grads["fp32"][k].norm()).item() for k in names])
    fa = torch.cat([grads["autocast"][k].flatten() for k in names])
    f3 = torch.cat([grads["fp32"][k].flatten() for k in names])
    out = {"loss": np.array(losses["autocast"]), "loss_fp32": np.array(losses["fp32"]), "keys": np.array(names),
           "grad_rel_err": rel, "flat_cosine": np.array((fa @ f3 / (fa.norm() * f3.norm())).item()),
           "flat_rel_err": np.array(((fa - f3).norm() / f3.norm()).item())}
    for k in SAMPLE_KEYS:
        out["grad_sample_" + k] = grads["autocast"][k].float().flatten()[::7].numpy()
    np.savez_compressed(os.path.join(OUT, "train_step_autocast.npz"), **out)
    print("wrote train_step_autocast.npz: loss", losses, "flat cosine", out["flat_cosine"], "flat rel err", out["flat_rel_err"],
          "median / max per-parameter rel err", np.median(rel), rel.max())


if __name__ == "__main__":
    if sys.argv[1:] == ["step"]:
        main_step()
    elif sys.argv[1:] == ["step_autocast"]:
        main_step_autocast()
    else:
        main()
