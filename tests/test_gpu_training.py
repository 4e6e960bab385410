"""GPU: pre-train step operators (train.py:90-97,270) on the HIP path against
(a) tests/golden/train_ops.npz -- the REAL reference's modules run backward with autograd (make_golden_bwd.py) and
(b) torch autograd of the same stock ops on the box's CPU for other shapes / seeds.
Tolerance: gradients <= 1e-4 of max|reference gradient| (the fp32 bar of BASELINE.json's north_star, applied per tensor)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, max_rel

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "train_ops.npz"))


def _hash_tensor(seed, key, shape, lo=-1.0, hi=1.0):
    from superresolution_amd import weights as W
    n = int(np.prod(shape))
    return torch.from_numpy((lo + (hi - lo) * W.hash_uniform(seed, key, n)).astype(np.float32).reshape(shape))


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _resblock(seed=0):
    from superresolution_amd import weights as W
    from superresolution_amd.generator import ResBlock
    sd = W.synthetic_state_dict(seed)
    rb = ResBlock(96)
    rb.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items() if k.startswith("blocks.0.")})
    return rb


def test_resblock_backward_vs_reference_golden(gold):
    """Generator.ResBlock in train mode, forward + backward, against the real reference's autograd."""
    from superresolution_amd import training as T
    rb = _resblock().cuda()
    ws = T.Workspace(torch.device("cuda"))
    x = _nhwc(_hash_tensor(11, "bwd.x", (3, 96, 64, 64))).cuda()
    gr = _nhwc(_hash_tensor(12, "bwd.g", (3, 96, 64, 64))).cuda()
    out, saved = T.resblock_forward(x, rb, ws)
    assert max_rel(_nchw(out).cpu().flatten()[::257].numpy(), gold["resblock_out_sample"]) <= TOL
    for i, st in ((1, saved["st1"]), (4, saved["st2"])):
        assert np.abs(st[0].cpu().numpy() - gold[f"resblock_bn{i}_mean"]).max() <= 1e-5
        assert max_rel(st[1].cpu().numpy(), gold[f"resblock_bn{i}_var_biased"]) <= 1e-4
    dx, grads = T.resblock_backward(gr, saved, rb, ws)
    dxc = _nchw(dx).cpu()
    err = max_rel(dxc.flatten()[::257].numpy(), gold["resblock_dx_sample"])
    print("resblock dx err", err)
    assert err <= TOL
    s = np.array([dxc.double().sum().item(), (dxc.double() ** 2).sum().item()])
    assert abs(s[1] - gold["resblock_dx_sums"][1]) <= 1e-4 * gold["resblock_dx_sums"][1]
    for k, g in grads.items():
        gc = g.cpu()
        if gc.dim() == 4:
            ref, ref_s = gold[f"resblock_grad_{k}_sample"], gold[f"resblock_grad_{k}_sums"]
            e = max_rel(gc.flatten()[::5].numpy(), ref)
            assert abs((gc.double() ** 2).sum().item() - ref_s[1]) <= 1e-4 * ref_s[1], k
        else:
            ref = gold[f"resblock_grad_{k}"]
            # a conv bias in front of a train-mode BatchNorm has gradient sum(dc) = 0 exactly: what both sides hold
            # is the rounding residue of a cancellation whose terms are of the size of dbeta -> scale by that
            scale = max(np.abs(ref).max(), float(np.abs(gold["resblock_grad_block.4.bias"]).max()))
            per_ch = np.abs(gc.numpy() - ref) / scale
            # PReLU's derivative jumps at 0 (see the next test): a pre-activation within rounding of 0 falls on either side
            # in two correct fp32 computations and moves the per-channel sums behind it (dgamma, dbeta, dslope of that one
            # channel) by one term.  Up to two such channels may miss the bar, by no more than one term's worth.
            off = int((per_ch > TOL).sum())
            print("resblock grad", k, "channels over the bar:", off, "worst", float(per_ch.max()))
            assert off <= 2 and float(per_ch.max()) <= 2e-3, k
            e = float(np.sort(per_ch)[-3])
        print("resblock grad", k, e)
        assert e <= TOL, k


def test_resblock_backward_vs_torch_autograd_other_seed():
    """Another weight seed, batch 1 and a ragged batch, against torch autograd of the stock ops (CPU).
    (PReLU's derivative jumps at 0: a pre-activation within rounding of 0 may fall on different sides in two correct
    fp32 computations and flips one gradient element by O(1) -- the (n, seed) pairs here have no such element; n = 5
    with seed 4 has exactly one.)"""
    from superresolution_amd import training as T
    for n, seed in ((1, 3), (6, 4)):
        rb = _resblock(seed)
        x = _hash_tensor(20 + n, "x", (n, 96, 64, 64))
        gr = _hash_tensor(30 + n, "g", (n, 96, 64, 64))
        xr = x.clone().requires_grad_(True)
        seq = rb.block
        t = F.conv2d(xr, seq[0].weight, seq[0].bias, padding=1)
        t = F.prelu(F.batch_norm(t, None, None, seq[1].weight, seq[1].bias, training=True, eps=1e-5), seq[2].weight)
        t = F.batch_norm(F.conv2d(t, seq[3].weight, seq[3].bias, padding=1), None, None, seq[4].weight, seq[4].bias,
                         training=True, eps=1e-5)
        ((xr + 0.1 * t) * gr).sum().backward()
        ref = {k: p.grad.clone() for k, p in rb.named_parameters()}
        rbc = _resblock(seed).cuda()
        ws = T.Workspace(torch.device("cuda"))
        out, saved = T.resblock_forward(_nhwc(x).cuda(), rbc, ws)
        dx, grads = T.resblock_backward(_nhwc(gr).cuda(), saved, rbc, ws)
        assert max_rel(_nchw(dx).cpu().numpy(), xr.grad.numpy()) <= TOL, n
        big = max(float(v.abs().max()) for v in ref.values())
        for k, g in grads.items():
            scale = big if k.endswith((".0.bias", ".3.bias")) else float(ref[k].abs().max())   # (see the golden test)
            assert float((g.cpu() - ref[k]).abs().max()) / scale <= TOL, (n, k)


@pytest.mark.parametrize("pack", ["split_pack", "wino_pack"])
def test_conv3x3_dgrad_and_wgrad_at_128(gold, pack):
    """output[0] of the generator runs at 128 x 128 (Generator.py:69): the same operators at that size, with the weight
    image of either 3x3 kernel (split-f16: the default of the training step; Winograd: SRGEN_F32_CONV3X3=wino)."""
    from superresolution_amd import training as T
    packer = getattr(T, pack)
    torch.manual_seed(5)
    w = (torch.rand(96, 96, 3, 3) - 0.5) * 0.1
    b = (torch.rand(96) - 0.5) * 0.1
    x = _hash_tensor(41, "x", (2, 96, 128, 128))
    gr = _hash_tensor(42, "g", (2, 96, 128, 128))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (F.conv2d(xr, wr, br, padding=1) * gr).sum().backward()
    ws = T.Workspace(torch.device("cuda"))
    wc = w.cuda()
    y = T.conv3x3(_nhwc(x).cuda(), packer(wc), b.cuda())
    assert max_rel(_nchw(y).cpu().numpy(), F.conv2d(x, w, b, padding=1).numpy()) <= TOL
    dx = T.conv3x3(_nhwc(gr).cuda(), packer(wc, mode=1))
    assert max_rel(_nchw(dx).cpu().numpy(), xr.grad.numpy()) <= TOL
    dw, db = T.conv3x3_wgrad(_nhwc(x).cuda(), _nhwc(gr).cuda(), ws)
    assert max_rel(dw.cpu().numpy(), wr.grad.numpy()) <= TOL
    assert max_rel(db.cpu().numpy(), br.grad.numpy()) <= TOL


@pytest.mark.parametrize("wgrad", ["split", "wino", "direct"])
@pytest.mark.parametrize("shape,scale", [((3, 96, 64, 64), 1.0), ((2, 96, 128, 128), 2.0 ** -22), ((5, 96, 64, 64), 2.0 ** 9)])
def test_conv3x3_wgrad_kernels_vs_torch(wgrad, shape, scale, monkeypatch):
    """The three weight-gradient kernels (SRGEN_WGRAD) against torch's own on the CPU in fp64: ragged batch sizes (slices
    that end inside an image, empty slices) and gradients far outside f16's range, which the split kernel has to scale."""
    from superresolution_amd import training as T
    monkeypatch.setenv("SRGEN_WGRAD", wgrad)
    x = _hash_tensor(51, "wg.x", shape)
    gr = _hash_tensor(52, "wg.g", shape) * scale
    w = torch.zeros(96, 96, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(96, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w, b, padding=1) * gr.double()).sum().backward()
    ws = T.Workspace(torch.device("cuda"))
    dw, db = T.conv3x3_wgrad(_nhwc(x).cuda(), _nhwc(gr).cuda(), ws)
    ew, eb = max_rel(dw.cpu().double().numpy(), w.grad.numpy()), max_rel(db.cpu().double().numpy(), b.grad.numpy())
    print(wgrad, shape, scale, "dw", ew, "db", eb)
    assert ew <= 2e-5 and eb <= 2e-5


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("f16", 2e-3)])
def test_conv3x3_wgrad_split_at_64_images_vs_torch_fp64_and_linearity(precision, tol):
    """BASELINE configs[3] / [4] run 64 images per GPU: the split weight-gradient kernels' slice walk (80 slices with three
    terms, 168 with one; slices that end inside an image) at that batch against torch's conv2d weight gradient on the CPU in
    fp64, and against the size-independent property of the op: the gradient of the 64-image batch equals the sum of the
    gradients of its eight 8-image sub-batches (a different slice walk for the same sums)."""
    from superresolution_amd import training as T
    n = 64
    x = _hash_tensor(71, "wg64.x", (n, 96, 64, 64))
    gr = _hash_tensor(72, "wg64.g", (n, 96, 64, 64)) * (2.0 ** -12)
    w = torch.zeros(96, 96, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(96, dtype=torch.float64, requires_grad=True)
    for k in range(0, n, 16):                                     # (fp64 CPU conv in four pieces: memory)
        (F.conv2d(x[k:k + 16].double(), w, b, padding=1) * gr[k:k + 16].double()).sum().backward()
    ws = T.Workspace(torch.device("cuda"))
    xc, gc = _nhwc(x).cuda(), _nhwc(gr).cuda()
    with T.conv_precision(precision):
        dw, db = T.conv3x3_wgrad(xc, gc, ws)
        parts = [T.conv3x3_wgrad(xc[k:k + 8].contiguous(), gc[k:k + 8].contiguous(), ws) for k in range(0, n, 8)]
        dw, db = dw.clone(), db.clone()
    ew, eb = max_rel(dw.cpu().double().numpy(), w.grad.numpy()), max_rel(db.cpu().double().numpy(), b.grad.numpy())
    sw = torch.stack([p[0].double() for p in parts]).sum(0)
    sb = torch.stack([p[1].double() for p in parts]).sum(0)
    lw, lb = max_rel(dw.cpu().double().numpy(), sw.cpu().numpy()), max_rel(db.cpu().double().numpy(), sb.cpu().numpy())
    print("wgrad at 64 images", precision, "vs fp64: dw", ew, "db", eb, "| vs the sum of eight sub-batches: dw", lw, "db", lb)
    assert ew <= tol and eb <= 2e-5
    # sub-batches see the same dy scale bound only up to their own maximum: fp32 products agree to rounding, f16 products
    # round the same operands the same way (the scale is a power of two) -- both inside the fp32 bar
    assert lw <= 2e-5 and lb <= 2e-5


def test_pretrain_step_at_64_images_split_vs_winograd_weight_gradients(monkeypatch):
    """One forward_backward at BASELINE configs[3]'s 64 images per GPU: the default step (split convs, split weight
    gradient) against the same step with every weight gradient on the exact-fp32 Winograd-domain kernel (SRGEN_WGRAD=wino)
    -- two independent kernels for 35 of the 39 weight gradients at the batch size the benchmark runs."""
    from superresolution_amd import Generator, training as T, weights as W
    lr = torch.from_numpy(W.synthetic_lr_batch(64, 5, "dense")).cuda()
    hr = torch.from_numpy(W.synthetic_hr_batch(64, seed=6)).cuda()
    out = {}
    for mode in ("split", "wino"):
        monkeypatch.setenv("SRGEN_WGRAD", mode)
        g = Generator()
        g.load_state_dict(W.synthetic_state_dict(0))
        g = g.cuda().train()
        step = T.PretrainStep(g, lr=1e-4, precision="fp32")
        step.optimizer.zero_grad()
        loss = step.forward_backward(lr, hr)
        out[mode] = (float(loss), {k: p.grad.detach().clone() for k, p in g.named_parameters()})
        del step, g
    assert out["split"][0] == out["wino"][0]                     # same forward
    worst = 0.0
    for k, a in out["split"][1].items():
        b = out["wino"][1][k]
        if _conv_bias_before_bn(k):
            continue
        e = max_rel(a.cpu().numpy(), b.cpu().numpy())
        worst = max(worst, e)
        assert e <= 1e-4, (k, e)
    print("64-image step: split vs Winograd weight gradients, worst tensor", worst)


def test_pixel_loss_grad_vs_reference_golden(gold, golden):
    from superresolution_amd import pixel_losses, training as T, weights as W
    ws = T.Workspace(torch.device("cuda"))
    sr = torch.from_numpy(golden["y_dense_sparse"]).cuda()
    hr = torch.from_numpy(W.synthetic_hr_batch(2, seed=1)).cuda()
    vals = pixel_losses(sr, hr).cpu().numpy()                      # [psnr, ssim, l1, weighted_l1]
    for tag, w_l1, w_ssim in (("pre", 1.0, 0.1), ("gan", 1.0, 0.3)):
        assert abs(w_l1 * vals[3] + w_ssim * (1 - vals[1]) - float(gold[f"loss_{tag}_value"])) <= 1e-5
        d = T.pixel_loss_grad(sr, hr, ws, w_l1, w_ssim).cpu()
        e = max_rel(d.flatten()[::3].numpy(), gold[f"loss_{tag}_dsr_sample"])
        print("loss grad", tag, e)
        assert e <= TOL
        ref_s = gold[f"loss_{tag}_dsr_sums"]
        assert abs((d.double() ** 2).sum().item() - ref_s[1]) <= 1e-4 * ref_s[1]


def test_pixel_loss_grad_ragged_shape_vs_torch_autograd():
    from oracle import torch_ref as TR
    from superresolution_amd import training as T
    ws = T.Workspace(torch.device("cuda"))
    rng = np.random.default_rng(6)
    sr = torch.from_numpy(rng.random((3, 2, 37, 70), dtype=np.float32))
    hr = torch.from_numpy((rng.random((3, 2, 37, 70), dtype=np.float32) * 0.03))         # straddles the 0.01 weight step
    srr = sr.clone().requires_grad_(True)
    w = (hr > 0.01).float() * 9 + 1
    (0.7 * (w * (srr - hr).abs()).mean() + 0.3 * (1 - TR.ssim(srr, hr))).backward()
    d = T.pixel_loss_grad(sr.cuda(), hr.cuda(), ws, 0.7, 0.3).cpu()
    assert max_rel(d.numpy(), srr.grad.numpy()) <= TOL


def test_flat_adamw_vs_reference_golden(gold):
    from superresolution_amd import training as T
    p = torch.nn.Parameter(_hash_tensor(13, "adamw.p", (4099,)).cuda())
    q = torch.nn.Parameter(torch.zeros(7, 3).cuda())               # a second tensor in the same flat buffer
    opt = T.FlatAdamW([p, q], lr=1e-4)
    assert p.data_ptr() == opt.flat.data_ptr() and p.grad.data_ptr() == opt.flat_grad.data_ptr()
    for step in range(1, 4):
        opt.zero_grad()
        p.grad.copy_(_hash_tensor(13 + step, "adamw.g", (4099,), -0.05, 0.05).cuda())
        opt.step()
        err = float(np.abs(p.detach().cpu().numpy() - gold[f"adamw_p{step}"]).max())
        print("adamw step", step, err)
        assert err <= 2e-7
    assert torch.equal(q.detach().cpu(), torch.zeros(7, 3))        # zero gradient, zero value: stays zero
    # a torch Optimizer: the reference's scheduler (train.py:76-79) and param_groups[0]['lr'] (train.py:106) drive it
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", factor=0.5, patience=0, min_lr=1e-7)
    sched.step(1.0)
    sched.step(0.5)                                                # no improvement: lr halves
    assert abs(opt.param_groups[0]["lr"] - 5e-5) < 1e-12
    sd = opt.state_dict()
    p2 = torch.nn.Parameter(p.detach().clone())
    q2 = torch.nn.Parameter(q.detach().clone())
    opt2 = T.FlatAdamW([p2, q2], lr=1e-4)
    opt2.load_state_dict(sd)
    for o, pp in ((opt, p), (opt2, p2)):
        o.zero_grad()
        pp.grad.copy_(_hash_tensor(99, "adamw.g", (4099,), -0.05, 0.05).cuda())
        o.step()
    assert torch.equal(p.detach(), p2.detach())                    # resumed optimizer continues bit for bit


def test_flat_adamw_exchanges_state_with_torch_adamw():
    """The reference checkpoints `opt_g.state_dict()` of a torch.optim.AdamW (train.py:219-225): FlatAdamW reads that layout
    and writes it, and a module-level zero_grad() (gradients set to None) does not break the flat gradient views."""
    from superresolution_amd import training as T
    shapes = [(33, 7), (5,), (2, 3, 3, 3)]
    base = [_hash_tensor(70 + i, "xch.p", s) for i, s in enumerate(shapes)]
    grads = [[_hash_tensor(80 + 10 * k + i, "xch.g", s, -0.05, 0.05) for i, s in enumerate(shapes)] for k in range(3)]
    tp = [torch.nn.Parameter(b.clone().cuda()) for b in base]
    topt = torch.optim.AdamW(tp, lr=1e-3)
    fp = [torch.nn.Parameter(b.clone().cuda()) for b in base]
    fopt = T.FlatAdamW(fp, lr=1e-3)

    def step(opt, ps, k):
        for p, g in zip(ps, grads[k]):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            p.grad.copy_(g.cuda())
        opt.step()

    step(topt, tp, 0)
    fopt.load_state_dict(topt.state_dict())                  # torch -> flat, after one torch step
    with torch.no_grad():
        for p, q in zip(fp, tp):
            p.copy_(q)
    for p in fp:
        p.grad = None                                        # what nn.Module.zero_grad() leaves
    fopt.restore_grad_views()
    assert fp[0].grad.data_ptr() == fopt.flat_grad.data_ptr()
    step(topt, tp, 1)
    step(fopt, fp, 1)
    for p, q in zip(fp, tp):
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-7
    tp2 = [torch.nn.Parameter(p.detach().clone()) for p in fp]
    topt2 = torch.optim.AdamW(tp2, lr=1e-3)
    topt2.load_state_dict(fopt.state_dict())                 # flat -> torch
    step(topt2, tp2, 2)
    step(fopt, fp, 2)
    for p, q in zip(fp, tp2):
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-7
    assert int(float(topt2.state_dict()["state"][0]["step"])) == 3 and fopt.steps == 3


# ---- the whole pre-train step (train.py:90-97) ----------------------------------------------------------------------
# Gradients at the far end of the 39-layer backward chain carry the rounding of both fp32 computations (and the odd
# PReLU pre-activation that rounds to opposite sides of 0): measured worst 2.5e-4 of max|gradient| at the stem weight,
# <= 1e-5 near the loss.  Bar: 1e-3 per tensor.
GRAD_TOL = 1e-3
def _step_batch():
    from superresolution_amd import weights as W
    lr = np.concatenate([W.synthetic_lr_batch(3, 2, "dense"), W.synthetic_lr_batch(1, 2, "sparse")], 0)
    return torch.from_numpy(lr).cuda(), torch.from_numpy(W.synthetic_hr_batch(4, seed=1)).cuda()


def _conv_bias_before_bn(key):
    return key.endswith((".block.0.bias", ".block.3.bias")) or key == "mid.0.bias"


def _check_generator_grads_vs_golden(grads, gs, what):
    """``grads``: {parameter name -> gradient} against train_step.npz (checksums for every tensor, samples for sixteen)."""
    names = list(grads)
    assert names == [str(k) for k in gs["keys"]]
    scale_all = float(gs["grad_absmax"].max())
    worst = 0.0
    for i, k in enumerate(names):
        gr = grads[k].detach().double().cpu()
        ref_sum, ref_sq = gs["grad_sums"][i]
        if _conv_bias_before_bn(k):
            # true gradient 0 (the BatchNorm behind it removes the mean): both sides hold rounding residue
            assert float(gr.abs().max()) <= 1e-4 * scale_all, k
            continue
        n = gr.numel()
        e_sq = abs(float((gr * gr).sum()) - ref_sq) / max(ref_sq, 1e-30)
        e_sum = abs(float(gr.sum()) - ref_sum) / max(np.sqrt(n * ref_sq), 1e-30)
        worst = max(worst, e_sq, e_sum)
        assert e_sq <= 2 * GRAD_TOL and e_sum <= GRAD_TOL, (k, e_sq, e_sum)
        sk = "grad_sample_" + k
        if sk in gs.files:
            e = max_rel(gr.flatten()[::7].numpy(), gs[sk])
            worst = max(worst, e)
            print("  grad sample", k, e)
            assert e <= GRAD_TOL, (k, e)
    print(f"{what}: worst gradient error vs the reference", worst)


def test_pretrain_step_gradients_vs_reference_golden():
    """One pre-train step of the REAL reference generator (fp32; make_golden_bwd.py step): the loss, the output and the
    gradient of all 167 parameter tensors (checksums for every tensor, samples for sixteen of them)."""
    from superresolution_amd import Generator, training as T, weights as W
    gs = np.load(os.path.join(GOLDEN, "train_step.npz"))
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda().train()
    step = T.PretrainStep(g, lr=1e-4)
    lr, hr = _step_batch()
    loss = step.forward_backward(lr, hr)
    assert abs(float(loss) - float(gs["loss"])) <= 1e-5 * abs(float(gs["loss"]))
    _check_generator_grads_vs_golden({k: p.grad for k, p in g.named_parameters()}, gs, "pre-train step")


# ---- the reference's OWN loop body on the drop-in module (train.py:90-97 with only the import changed) ------------------
def _ref_weighted_l1(sr, hr):                                     # train.py:18-20, as train.py defines it (plain torch)
    weight = (hr > 0.01).float() * 9 + 1
    return (weight * (sr - hr).abs()).mean()


def _ref_grad_norm(model):                                        # train.py:23-28
    total = 0.0
    for p in model.parameters():
        if p.grad is not None:
            total += p.grad.data.norm(2).item() ** 2
    return total ** 0.5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_prelu_backward_with_the_pixel_shuffle_backward_folded_into_its_store(dtype):
    """prelu_scalar_bwd(unshuffle=True) (the upsample block, Generator.py:8-9): the same numbers as the plain operator,
    stored as the (2, 2, n, h/2, w/2, 96) planes the PixelShuffle's backward makes -- bit for bit what the permute copy
    made from the plain result; dslope identical."""
    from superresolution_amd import training as T
    ws = T.Workspace(torch.device("cuda"))
    x = _nhwc(_hash_tensor(61, "px", (3, 96, 128, 128), -1.0, 1.0)).to(dtype).cuda().contiguous()
    dy = (_nhwc(_hash_tensor(62, "pd", (3, 96, 128, 128), -1.0, 1.0)) * 2.0 ** -10).to(dtype).cuda().contiguous()
    slope = torch.tensor([0.25], device="cuda")
    dx, ds = T.prelu_scalar_bwd(x, dy, slope, ws)
    planes, ds2 = T.prelu_scalar_bwd(x, dy, slope, ws, unshuffle=True)
    ref = dx.view(3, 64, 2, 64, 2, 96).permute(2, 4, 0, 1, 3, 5).contiguous()
    assert planes.shape == (2, 2, 3, 64, 64, 96) and planes.dtype == dtype
    assert torch.equal(planes, ref)
    assert torch.equal(ds, ds2)


def test_reference_pretrain_loop_body_runs_on_the_dropin_generator():
    """train.py:90-97 verbatim -- ``optimizer.zero_grad(); sr = generator(lr); loss = weighted_l1(sr, hr) + 0.1 * (1 -
    ssim(sr, hr)); loss.backward(); grad_norm(generator); optimizer.step()`` -- with a STOCK ``torch.optim.AdamW`` on
    ``superresolution_amd.Generator``: ``loss.backward()`` reaches the HIP backward pass through autograd.GeneratorTrainFn.
    Without autocast the step is the reference's fp32 step: loss, every p.grad and the running statistics against
    train_step.npz / generator_train.npz; the AdamW update against torch's own on the same gradients (it IS torch's).
    ``ssim`` is the stock-torch restatement of utils.py:21-41 (oracle/torch_ref.py), differentiated by torch."""
    from oracle import torch_ref as TR
    from superresolution_amd import Generator, weights as W
    gs = np.load(os.path.join(GOLDEN, "train_step.npz"))
    gt = np.load(os.path.join(GOLDEN, "generator_train.npz"))
    generator = Generator()
    generator.load_state_dict(W.synthetic_state_dict(0))
    generator = generator.cuda()
    optimizer = torch.optim.AdamW(generator.parameters(), lr=1e-4)            # train.py:270
    generator.train()                                                          # train.py:80
    lr, hr = _step_batch()
    before = [p.detach().clone() for p in generator.parameters()]
    # ---- train.py:90-97 (autocast left out: the fp32 fixture) ----
    optimizer.zero_grad()
    sr = generator(lr)
    loss = _ref_weighted_l1(sr, hr) + 0.1 * (1 - TR.ssim(sr, hr))
    loss.backward()
    g = _ref_grad_norm(generator)
    optimizer.step()
    # ----
    assert sr.dtype == torch.float32 and sr.grad_fn is not None
    assert abs(loss.item() - float(gs["loss"])) <= 1e-5 * abs(float(gs["loss"]))
    grads = {k: p.grad for k, p in generator.named_parameters()}
    assert all(v is not None for v in grads.values())
    _check_generator_grads_vs_golden(grads, gs, "reference loop body on the drop-in")
    flat_sq = sum(float(gs["grad_sums"][i][1]) for i in range(len(gs["keys"])))
    assert abs(g - flat_sq ** 0.5) <= 1e-3 * flat_sq ** 0.5
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, generator.parameters()))
    sd = generator.state_dict()
    keys = [str(k) for k in gt["bn_keys"]]
    em = max(np.abs(sd[k + ".running_mean"].cpu().numpy() - gt["mean1"][i]).max() for i, k in enumerate(keys))
    assert em <= 1e-5 and int(sd[keys[0] + ".num_batches_tracked"]) == 1001
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()
    # a second iteration, now exactly as train.py:90-94 has it (inside bf16 autocast; zero_grad sets p.grad to None)
    optimizer.zero_grad()
    assert all(p.grad is None for p in generator.parameters())
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        sr = generator(lr)
        loss2 = _ref_weighted_l1(sr, hr) + 0.1 * (1 - TR.ssim(sr, hr))
    loss2.backward()
    optimizer.step()
    assert sr.dtype == torch.bfloat16
    assert loss2.item() < loss.item()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in generator.parameters())
    # the eval forward picks the stepped weights up
    with torch.no_grad():
        y = generator.eval()(lr)
    assert torch.isfinite(y).all()


def test_pixel_loss_function_is_differentiable_and_matches_the_torch_expression():
    """superresolution_amd.pixel_loss (autograd.PixelLossFn): value and gradient of w1 * weighted_l1 + w2 * (1 - ssim)
    against the torch expression of train.py:18-20,93 differentiated by torch."""
    from oracle import torch_ref as TR
    from superresolution_amd import pixel_loss, weights as W
    sr0 = torch.from_numpy(W.synthetic_hr_batch(3, seed=5)).cuda()
    hr = torch.from_numpy(W.synthetic_hr_batch(3, seed=1)).cuda()
    a, b = sr0.clone().requires_grad_(), sr0.clone().requires_grad_()
    la = 2.0 * pixel_loss(a, hr, 1.0, 0.3)
    lb = 2.0 * (_ref_weighted_l1(b, hr) + 0.3 * (1 - TR.ssim(b, hr)))
    la.backward()
    lb.backward()
    assert abs(la.item() - lb.item()) <= 1e-5 * abs(lb.item())
    assert max_rel(a.grad.cpu().numpy(), b.grad.cpu().numpy()) <= TOL


def test_pretrain_step_updates_parameters_and_running_stats():
    """step(): AdamW applied to the gradients just computed (checked against torch.optim.AdamW fed the same gradients),
    BatchNorm running statistics as the train-mode forward fixture has them, loss decreases over a few steps."""
    from superresolution_amd import Generator, training as T, weights as W
    gt = np.load(os.path.join(GOLDEN, "generator_train.npz"))
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda().train()
    step = T.PretrainStep(g, lr=1e-4)
    lr, hr = _step_batch()
    before = {k: p.detach().clone() for k, p in g.named_parameters()}
    loss0 = float(step.step(lr, hr))
    grads = {k: p.grad.detach().clone() for k, p in g.named_parameters()}
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in before.items()}
    opt = torch.optim.AdamW(list(ref.values()), lr=1e-4)
    for k, p in ref.items():
        p.grad = grads[k]
    opt.step()
    for k, p in g.named_parameters():
        assert float((p.detach() - ref[k].detach()).abs().max()) <= 2e-7, k
    sd = g.state_dict()
    keys = [str(k) for k in gt["bn_keys"]]
    em = max(np.abs(sd[k + ".running_mean"].cpu().numpy() - gt["mean1"][i]).max() for i, k in enumerate(keys))
    ev = max(max_rel(sd[k + ".running_var"].cpu().numpy(), gt["var1"][i]) for i, k in enumerate(keys))
    assert em <= 1e-5 and ev <= 1e-5                      # same batch as generator_train.npz's first forward
    assert int(sd[keys[0] + ".num_batches_tracked"]) == 1001
    losses = [loss0] + [float(step.step(lr, hr)) for _ in range(4)]
    print("pre-train losses over 5 steps on one batch:", losses)
    assert losses[-1] < losses[0]
    # the eval forward picks the updated weights up (re-fold from the stepped parameters and running statistics): it
    # equals a fresh Generator loaded with the stepped state_dict, and differs from the un-stepped one
    with torch.no_grad():
        y = g.eval()(lr)
    g2 = Generator()
    g2.load_state_dict({k: v.detach().cpu().clone() for k, v in g.state_dict().items()})
    g0 = Generator()
    g0.load_state_dict(W.synthetic_state_dict(0))
    with torch.no_grad():
        y2 = g2.cuda().eval()(lr)
        y0 = g0.cuda().eval()(lr)
    assert torch.isfinite(y).all()
    assert max_rel(y.cpu().numpy(), y2.cpu().numpy()) <= 1e-6
    assert max_rel(y.cpu().numpy(), y0.cpu().numpy()) >= 1e-3


# ---- mixed precision: f16 conv products (the counterpart of the reference's bf16 autocast, train.py:91) -----------------
F16_TOL = 2e-3            # operands rounded to 11 significant bits, fp32 accumulation: ~1e-4 of max on these sums; a conv
                          # on bf16 operands (what the reference's autocast runs) is at ~1e-3


@pytest.mark.parametrize("shape,scale", [((3, 96, 64, 64), 1.0), ((2, 96, 128, 128), 2.0 ** -22), ((5, 96, 64, 64), 2.0 ** 9)])
def test_conv3x3_f16_products_vs_torch_fp64(shape, scale):
    """srgen_conv3x3_nhwc_f16 / srgen_conv3x3_wgrad_f16 (forward, data gradient, weight gradient) against torch in fp64:
    inside F16_TOL, and NOT inside the fp32 bar (the switch really changes the arithmetic); gradients far outside f16's
    range are scaled like on the split path; the fp32 mode is untouched after the block."""
    from superresolution_amd import training as T
    torch.manual_seed(7)
    w = (torch.rand(96, 96, 3, 3) - 0.5) * 0.1
    b = (torch.rand(96) - 0.5) * 0.1
    x = _hash_tensor(61, "f16.x", shape)
    gr = _hash_tensor(62, "f16.g", shape) * scale
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=1)
    (yr * gr.double()).sum().backward()
    ws = T.Workspace(torch.device("cuda"))
    wc, xc, gc = w.cuda(), _nhwc(x).cuda(), _nhwc(gr).cuda()
    with T.conv_precision("f16"):
        y = T.conv3x3(xc, T.split_pack(wc), b.cuda())
        dx = T.conv3x3(gc, T.split_pack(wc, mode=1), scale_input=True)
        dw, db = T.conv3x3_wgrad(xc, gc, ws)
    errs = {"y": max_rel(_nchw(y).cpu().double().numpy(), yr.detach().numpy()),
            "dx": max_rel(_nchw(dx).cpu().double().numpy(), xr.grad.numpy()),
            "dw": max_rel(dw.cpu().double().numpy(), wr.grad.numpy()),
            "db": max_rel(db.cpu().double().numpy(), br.grad.numpy())}
    print("f16 products", shape, scale, errs)
    assert all(e <= F16_TOL for e in errs.values()), errs
    assert errs["y"] > 1e-5 and errs["dx"] > 1e-5 and errs["dw"] > 1e-6, errs      # one-term arithmetic, not the split one
    assert errs["db"] <= 2e-5                                     # the bias gradient is summed from the fp32 values
    y32 = T.conv3x3(xc, T.split_pack(wc), b.cuda())               # back on the default: fp32 products
    assert max_rel(_nchw(y32).cpu().double().numpy(), yr.detach().numpy()) <= 2e-6


def _sample_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_pretrain_step_f16_products_vs_reference_fp32_and_autocast_golden():
    """The pre-train step with f16 conv products (precision="f16", and "auto" inside torch.autocast(float16)) against ONE step of the real reference run both ways (make_golden_bwd.py step / step_autocast): every sampled
    gradient is closer to the reference's fp32 gradient than the reference's own bf16-autocast gradient is (every one of the
    sixteen sampled tensors; by 2x at the median), the loss likewise."""
    from superresolution_amd import Generator, training as T, weights as W
    g32 = np.load(os.path.join(GOLDEN, "train_step.npz"))
    gac = np.load(os.path.join(GOLDEN, "train_step_autocast.npz"))
    lr, hr = _step_batch()
    results = {}
    for mode in ("f16", "auto"):
        g = Generator()
        g.load_state_dict(W.synthetic_state_dict(0))
        g = g.cuda().train()
        step = T.PretrainStep(g, lr=1e-4, precision=mode)
        if mode == "auto":
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                assert T.resolve_precision("auto") == "bf16"             # (tests/test_gpu_training_bf16.py)
            with torch.autocast(device_type="cuda", dtype=torch.float16):
                assert T.resolve_precision("auto") == "f16"
                loss = step.forward_backward(lr, hr)
            assert T.resolve_precision("auto") == "fp32"
        else:
            loss = step.forward_backward(lr, hr)
        P = dict(g.named_parameters())
        results[mode] = (float(loss), {k: P[k[len("grad_sample_"):]].grad.detach().flatten()[::7].cpu().numpy()
                                       for k in g32.files if k.startswith("grad_sample_")})
    loss, grads = results["f16"]
    ref_loss, ac_loss = float(g32["loss"]), float(gac["loss"])
    print("loss: fp32 reference", ref_loss, "reference autocast", ac_loss, "this f16 step", loss)
    assert abs(loss - ref_loss) <= 0.5 * abs(ac_loss - ref_loss)
    ratios = []
    for k, v in grads.items():
        ours, theirs = _sample_rel(v, g32[k]), _sample_rel(gac[k], g32[k])
        print(f"  {k}: this step {ours:.2e}, reference autocast {theirs:.2e}")
        assert ours <= theirs and ours <= 2e-2, (k, ours, theirs)
        ratios.append(ours / theirs)
    print("  this step's deviation / the reference autocast's: median", float(np.median(ratios)), "max", max(ratios))
    assert float(np.median(ratios)) <= 0.5
    assert _sample_rel(grads["grad_sample_mid.0.weight"], g32["grad_sample_mid.0.weight"]) >= 1e-4    # (not the fp32 arithmetic)
    # "auto" inside autocast ran the same arithmetic
    assert results["auto"][0] == loss
    for k, v in results["auto"][1].items():
        assert np.array_equal(v, grads[k]), k


def test_two_stream_backward_is_bit_identical_to_one_stream(monkeypatch):
    """training.wgrad_stream: the weight gradients of the backward pass run on a second HIP stream beside the data-gradient
    chain (same kernels, same order per buffer).  The flat gradient of two steps in a row must equal the single-stream
    run's (SRGEN_WGRAD_STREAM=0) bit for bit -- a race on the shared weight-gradient scratch slab, or an operand handed back
    to the allocator too early, shows up here."""
    from superresolution_amd import Generator, training as T, weights as W
    lr, hr = _step_batch()
    flats = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SRGEN_WGRAD_STREAM", mode)
        g = Generator()
        g.load_state_dict(W.synthetic_state_dict(0))
        g = g.cuda().train()
        step = T.PretrainStep(g, lr=1e-4)
        assert (step.side is None) == (mode == "0")
        out = []
        for _ in range(2):
            step.step(lr, hr)
            out.append(step.optimizer.flat_grad.detach().clone())
        torch.cuda.synchronize()
        flats[mode] = out
    for a, b in zip(flats["0"], flats["1"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,bands,scale", [(1, 0, 1.0), (3, 1, 1.0), (3, 2, 1.0), (2, 3, 2.0 ** -20), (2, 5, 1.0), (3, 8, 300.0)])
def test_head_fp32_row_streaming_kernel_vs_fp64_and_tile_kernel(n, bands, scale, monkeypatch):
    """The fp32 head on split-f16 products (Generator.py:71,80) in its row-streaming form (srgen_head_f32s_stream.hip: pixels
    split in registers a row ahead, all weight fragments in LDS, ring of nine partial output rows) against conv2d in fp64
    and against the tile kernel it replaces, at every band count the launcher can pick.  Inputs of order `scale`: the
    training-op entry point passes no input bound, so only order-1 inputs keep fp32 accuracy (bar 3e-6 of max, the tile
    kernel's); the scaled cases check that both kernels degrade alike (same split, same products)."""
    from superresolution_amd import training as T
    v = (_nhwc(_hash_tensor(71, "hv", (n, 96, 128, 128))) * scale).cuda()
    w = (_hash_tensor(72, "hw", (3, 96, 9, 9)) * 0.05).cuda()
    b = _hash_tensor(73, "hb", (3,)).cuda()
    if bands:
        monkeypatch.setenv("SRGEN_HEAD_BANDS", str(bands))
    with T.conv_precision("fp32"):
        y = T.head_forward(v, w, b)
        monkeypatch.setenv("SRGEN_HEAD_F32S", "tile")
        yt = T.head_forward(v, w, b)
    torch.cuda.synchronize()
    ref = F.conv2d(v.cpu().double().permute(0, 3, 1, 2), w.cpu().double(), None, padding=4)[:, :, :125, :125]
    sc = float(ref.abs().max())
    bb = b.cpu().double().view(1, 3, 1, 1)
    e = float((y.cpu().double() - bb - ref).abs().max()) / sc
    et = float((yt.cpu().double() - bb - ref).abs().max()) / sc
    print(f"  head fp32 split (n {n}, bands {bands}, scale {scale:g}): streaming {e:.2e}, tile {et:.2e} of max vs fp64")
    if scale == 1.0:
        assert e <= 3e-6 and et <= 3e-6, (e, et)
    else:
        assert e <= 2.0 * et + 1e-6, (e, et)


@pytest.mark.parametrize("n,ih,out_hw,flip", [(3, 64, 64, False), (2, 125, 128, True)])
def test_conv9x9_split_kernel_with_weights_on_an_lds_ring_is_bit_identical(n, ih, out_hw, flip, monkeypatch):
    """The three-term 9x9 conv against a 3-channel image (eval stem Generator.py:50-51, head data gradient of the fp32 /
    f16 steps) with its weight slabs copied to an LDS ring (stem_f32s_ring_kernel) multiplies the same products in the same
    order per accumulator as the kernel that reads them from L2 (SRGEN_STEM_F32S=direct): equal bit for bit."""
    from superresolution_amd import training as T
    img = (_hash_tensor(81, "i", (n, 3, ih, ih)) * (2.0 ** -14 if flip else 1.0)).cuda()
    wm = (_hash_tensor(82, "w", (243, 96)) * 0.1).cuda()
    bias = None if flip else _hash_tensor(83, "b", (96,)).cuda()
    y = T.conv9x9_c3(img, wm, bias, out_hw, flip)
    monkeypatch.setenv("SRGEN_STEM_F32S", "direct")
    yd = T.conv9x9_c3(img, wm, bias, out_hw, flip)
    torch.cuda.synchronize()
    assert float(y.abs().max()) > 0
    assert torch.equal(y, yd)
