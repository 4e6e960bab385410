"""GAN step (train.py:145-176) on the HIP path: discriminator backward, RaGAN loss gradient, clip_grad_norm_, and one
whole iteration against the real reference (tests/golden/gan_step.npz, made by make_golden_gan.py with torch autograd in
fp32).  Operator-level checks compare with torch autograd evaluated here on the CPU (fp32) at ragged sizes."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5            # operator level, max-normalised
STEP_TOL = 1e-3       # whole-step gradients (same bar as the pre-train step: PReLU / LeakyReLU sign flips at rounding)


def max_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _torch_disc(d, x):
    """Discriminator.py:49-50 with stock torch ops on the mirror's parameters (CPU copies that require grad)."""
    from superresolution_amd.gan_training import disc_convs
    ps = [(c.weight.detach().cpu().clone().requires_grad_(), c.bias.detach().cpu().clone().requires_grad_()) for c in disc_convs(d)]
    strides = [1, 2, 1, 2, 1, 2, 1, 2, 1, 1]
    h = x
    for i, ((w, b), s) in enumerate(zip(ps, strides)):
        h = F.conv2d(h, w, b, stride=s, padding=1)
        if i < 9:
            h = F.leaky_relu(h, 0.2)
    return h, ps


@pytest.mark.parametrize("shape", [(2, 37, 53), (3, 16, 9), (1, 125, 125)])
def test_discriminator_backward_vs_torch_autograd(shape):
    from superresolution_amd import Discriminator, training as T
    from superresolution_amd.gan_training import DiscriminatorGrad, disc_convs
    torch.manual_seed(11)
    d = Discriminator().cuda()
    with torch.no_grad():
        for c in disc_convs(d):
            c.bias.uniform_(-0.05, 0.05)               # the default init has zero biases: make them count
    n, h, w = shape
    g = torch.Generator().manual_seed(5)
    x = torch.rand((n, 3, h, w), generator=g)
    xr = x.clone().requires_grad_()
    ref_logits, ps = _torch_disc(d, xr)
    dl = torch.randn(ref_logits.shape, generator=g) / ref_logits.numel()
    ref_logits.backward(dl)
    dg = DiscriminatorGrad(d, T.Workspace(torch.device("cuda", 0)))
    logits, saved = dg.forward(x.cuda())
    assert max_rel(logits.cpu().numpy(), ref_logits.detach().numpy()) <= TOL
    dimg = dg.backward(dl.cuda(), saved, want_wgrad=True, want_dimg=True)
    worst = max_rel(dimg.cpu().numpy(), xr.grad.numpy())
    assert worst <= TOL, ("dimg", worst)
    for i, (c, (w_, b_)) in enumerate(zip(disc_convs(d), ps)):
        ew = max_rel(c.weight.grad.cpu().numpy(), w_.grad.numpy())
        eb = max_rel(c.bias.grad.cpu().numpy(), b_.grad.numpy())
        worst = max(worst, ew, eb)
        assert ew <= TOL and eb <= TOL, (i, ew, eb)
    print("discriminator backward", shape, "worst", worst)
    # a second backward accumulates (autograd semantics of .grad)
    w0 = disc_convs(d)[3].weight.grad.clone()
    dg.backward(dl.cuda(), saved, want_wgrad=True, want_dimg=False)
    assert max_rel(disc_convs(d)[3].weight.grad.cpu().numpy(), (2 * w0).cpu().numpy()) <= 1e-6


def _rb(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("shape", [(2, 37, 53), (1, 125, 125)])
def test_discriminator_bf16_products_vs_torch_emulation(shape):
    """DiscriminatorGrad.products = "bf16" (what autocast gives the 64..512-channel convs at train.py:150,168): tensors
    stay fp32, a conv's activation (forward) or gradient (data gradient) operand and its weights are rounded to bf16 where
    they are multiplied, accumulation is fp32 (forward, data gradient, weight gradient); the stem and the tail keep fp32 products.
    (a) every layer's forward and data gradient against exactly that arithmetic written with stock torch ops on the CPU, fed
        the SAME operands (products of bf16 numbers are exact in fp32: what is left is summation order; bar 2e-5 of max --
        chained through the network instead, two correct implementations drift apart to the bf16 rounding noise itself,
        because a 1e-6 difference decides which way an operand rounds);
    (b) the whole forward / backward against the plain fp32 torch reference: a bf16-sized deviation, and not zero."""
    from superresolution_amd import Discriminator, training as T
    from superresolution_amd.gan_training import DiscriminatorGrad, disc_convs
    torch.manual_seed(11)
    d = Discriminator().cuda()
    with torch.no_grad():
        for c in disc_convs(d):
            c.bias.uniform_(-0.05, 0.05)
    n, h, w = shape
    g = torch.Generator().manual_seed(5)
    x = torch.rand((n, 3, h, w), generator=g)
    strides = [1, 2, 1, 2, 1, 2, 1, 2, 1, 1]
    ws = [(c.weight.detach().cpu(), c.bias.detach().cpu()) for c in disc_convs(d)]
    nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
    lr_d = lambda a: torch.where(a > 0, torch.ones_like(a), torch.full_like(a, 0.2))
    dg = DiscriminatorGrad(d, T.Workspace(torch.device("cuda", 0)))
    dg.products = "bf16"
    logits, saved = dg.forward(x.cuda())
    assert saved["bf16"]
    acts = [nchw(a) for a in saved["acts"]]
    # (a) forward, layer by layer on the GPU's own inputs
    for i in range(1, 9):
        ref = F.leaky_relu(F.conv2d(_rb(acts[i - 1]), _rb(ws[i][0]), ws[i][1], stride=strides[i], padding=1), 0.2)
        e = max_rel(acts[i].numpy(), ref.numpy())
        print(f"  layer {i} forward (bf16 products) vs emulation on the same input: {e:.2e}")
        assert e <= 2e-5, (i, e)
    # (a) data gradient of every layer: the operator on a random gradient and the saved activation below
    import ctypes
    from superresolution_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    for i in range(1, 9):
        conv, st = disc_convs(d)[i], strides[i]
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        below, here = acts[i - 1], acts[i]
        dpre = torch.randn(here.shape, generator=g) * 1e-3
        op = (below.shape[2] - ((dpre.shape[2] - 1) * st + 1), below.shape[3] - ((dpre.shape[3] - 1) * st + 1))
        ref = F.conv_transpose2d(_rb(dpre), _rb(ws[i][0]), stride=st, padding=1, output_padding=op) * lr_d(below)
        dp = dpre.permute(0, 2, 3, 1).contiguous().cuda()
        out = torch.empty_like(saved["acts"][i - 1])
        ih, iw = below.shape[2], below.shape[3]
        if st == 2:
            _lib.check(L.srdisc_conv_dgrad2_bf16p(dp.data_ptr(), dg._pack(conv, 4, True).data_ptr(), saved["acts"][i - 1].data_ptr(),
                                                  out.data_ptr(), n, ih, iw, cin, cout, 0.2, s))
        else:
            _lib.check(L.srdisc_conv_nhwc_bf16p(dp.data_ptr(), dg._pack(conv, 1, True).data_ptr(), None, saved["acts"][i - 1].data_ptr(),
                                                out.data_ptr(), n, ih, iw, cout, cin, 1, 0.2, s))
        e = max_rel(nchw(out).numpy(), ref.numpy())
        print(f"  layer {i} data gradient (stride {st}, bf16 products) vs emulation: {e:.2e}")
        assert e <= 2e-5, (i, e)
    # (b) the whole thing against plain fp32 autograd
    xr = x.clone().requires_grad_()
    ref_logits, ps = _torch_disc(d, xr)
    dl = torch.randn(ref_logits.shape, generator=g) / ref_logits.numel()
    ref_logits.backward(dl)
    dev = max_rel(logits.cpu().numpy(), ref_logits.detach().numpy())
    print("bf16-product logits vs the fp32 reference", dev)
    assert 1e-5 <= dev <= 3e-2
    dimg = dg.backward(dl.cuda(), saved, want_wgrad=True, want_dimg=True)

    def rel_cos(a, b_):
        a, b_ = a.double().flatten(), b_.double().flatten()
        return float((a - b_).norm() / b_.norm()), float(a @ b_ / (a.norm() * b_.norm()))
    # A freshly initialised 10-layer LeakyReLU chain fed noise: the gradient of a noise functional of its logits is the
    # most rounding-sensitive quantity there is -- every sign flip of a pre-activation near zero switches a path.  The
    # yardstick is therefore what torch's OWN bf16 autocast does to the same gradients (stock ops, CPU): this path has to
    # stay within 2x of that deviation per tensor (measured: 0.9x - 1.7x, the same size: 6 - 14 % for both).
    xa = x.clone().requires_grad_()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ac_logits, pa = _torch_disc(d, xa)
    ac_logits.float().backward(dl)
    r, c = rel_cos(dimg.cpu(), xr.grad)
    ra, ca = rel_cos(xa.grad, xr.grad)
    print(f"  image gradient vs fp32: rel {r:.2e}, cosine {c:.6f}   (torch autocast: rel {ra:.2e}, cosine {ca:.6f})")
    assert 1e-5 <= r <= 2.0 * ra
    for i, (cv, (w_, b_), (wa, _)) in enumerate(zip(disc_convs(d), ps, pa)):
        r, c = rel_cos(cv.weight.grad.cpu(), w_.grad)
        ra, ca = rel_cos(wa.grad, w_.grad)
        print(f"  layer {i} weight gradient vs fp32: rel {r:.2e}, cosine {c:.6f}   (torch autocast: rel {ra:.2e}, cosine {ca:.6f})")
        assert r <= 2.0 * ra, (i, r, ra)


@pytest.mark.parametrize("cfg", [(2, 37, 53, 64, 64, 2), (3, 19, 27, 64, 128, 1), (2, 19, 27, 128, 128, 2), (2, 16, 16, 256, 512, 1),
                                 (3, 16, 16, 512, 512, 2), (5, 8, 8, 512, 256, 1), (1, 5, 7, 64, 64, 1), (4, 125, 125, 64, 64, 2)])
def test_discriminator_wgrad_bf16_products_vs_torch_emulation(cfg):
    """srdisc_conv_wgrad_bf16p (the weight gradient inside train.py:148-152's autocast): both operands rounded to bf16 where
    they are multiplied, fp32 accumulation -- against exactly that arithmetic in fp64 with stock torch ops on the CPU, on
    the SAME operands (products of bf16 numbers are exact: what is left is fp32 summation over up to 16k pixels; bar 2e-5 of
    max); the bias gradient is the fp32 sum of dpre.  Shapes: both strides, ragged grids, an 8-wide grid, tiles that hang
    over the image, slices that end inside an image."""
    import ctypes
    from superresolution_amd import _lib
    n, h, w, cin, cout, st = cfg
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    oh, ow = (h - 1) // st + 1, (w - 1) // st + 1
    x = torch.rand((n, h, w, cin), generator=g) - 0.3
    dpre = torch.randn((n, oh, ow, cout), generator=g) * 1e-3
    xd, dd = x.cuda(), dpre.cuda()
    dw = torch.empty((cout, cin, 3, 3), device="cuda")
    db = torch.empty(cout, device="cuda")
    need = ctypes.c_size_t()
    _lib.check(L.srdisc_conv_wgrad_scratch_bytes(n, h, cin, cout, st, ctypes.byref(need)))
    scratch = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.srdisc_conv_wgrad_bf16p(xd.data_ptr(), dd.data_ptr(), n, h, w, cin, cout, st, dw.data_ptr(), db.data_ptr(),
                                         scratch.data_ptr(), scratch.numel(), s))
    xr = _rb(x).permute(0, 3, 1, 2).double()
    dr = _rb(dpre).permute(0, 3, 1, 2).double()
    ref = torch.nn.grad.conv2d_weight(xr, (cout, cin, 3, 3), dr, stride=st, padding=1)
    e = max_rel(dw.cpu().numpy(), ref.numpy())
    eb = max_rel(db.cpu().numpy(), dpre.double().sum(dim=(0, 1, 2)).numpy())
    print(f"  wgrad bf16 products {cfg}: {e:.2e}, bias {eb:.2e}")
    assert e <= 2e-5 and eb <= 2e-5, (cfg, e, eb)
    # the fp32-product operator on the same operands: a bf16-sized distance, and not zero (the kernels are different ones)
    dw32 = torch.empty_like(dw)
    _lib.check(L.srdisc_conv_wgrad(xd.data_ptr(), dd.data_ptr(), n, h, w, cin, cout, st, dw32.data_ptr(), db.data_ptr(),
                                   scratch.data_ptr(), scratch.numel(), s))
    d = max_rel(dw.cpu().numpy(), dw32.cpu().numpy())
    assert 1e-6 <= d <= 3e-2, d


def test_discriminator_bf16_products_full_batch_properties():
    """BASELINE configs[4] sizes (64 images per GPU, 125 x 125): size-independent properties of the bf16-product kernels.
    (1) Images are independent: logits and every saved activation of a 64-image forward equal, bit for bit, those of the
        same images run as batches of 37, 26 and 1 (tiles never mix images in one accumulator, and the 8-wide layers pair
        images without changing any image's summation order).
    (2) The weight gradient is additive over the batch: the 64-image gradient equals the sum of the three sub-batch
        gradients to fp32 summation order (2e-5 of max) -- slices that end inside an image, units that hang over it."""
    from superresolution_amd import Discriminator, training as T
    from superresolution_amd.gan_training import DiscriminatorGrad, disc_convs
    torch.manual_seed(5)
    d = Discriminator().cuda()
    dg = DiscriminatorGrad(d, T.Workspace(torch.device("cuda", 0)))
    dg.products = "bf16"
    g = torch.Generator().manual_seed(9)
    x = torch.rand((64, 3, 125, 125), generator=g).cuda()
    logits, saved = dg.forward(x)
    dl = (torch.randn(logits.shape, generator=g) / logits.numel()).cuda()
    parts = [(0, 37), (37, 63), (63, 64)]
    for c in disc_convs(d):
        c.weight.grad = None
        c.bias.grad = None
    grads = None
    for a, b in parts:
        lg, sv = dg.forward(x[a:b].contiguous())
        assert torch.equal(lg, logits[a:b]), (a, b)
        for i, (u, v) in enumerate(zip(sv["acts"], saved["acts"])):
            assert torch.equal(u, v[a:b]), (a, b, i)
        dg.backward(dl[a:b].contiguous(), sv, want_wgrad=True, want_dimg=False)          # accumulates into .grad
    grads = [c.weight.grad.clone() for c in disc_convs(d)]
    for c in disc_convs(d):
        c.weight.grad = None
        c.bias.grad = None
    dg.backward(dl, saved, want_wgrad=True, want_dimg=False)
    for i, (c, gsum) in enumerate(zip(disc_convs(d), grads)):
        e = max_rel(c.weight.grad.cpu().numpy(), gsum.cpu().numpy())
        assert e <= 2e-5, (i, e)


def test_ragan_loss_grad_vs_torch_autograd():
    from superresolution_amd.gan_training import ragan_loss_grad
    g = torch.Generator().manual_seed(2)
    real = (torch.randn((3, 1, 8, 8), generator=g) * 2).requires_grad_()
    fake = (torch.randn((5, 1, 8, 8), generator=g) * 2 + 0.5).requires_grad_()
    bce = torch.nn.BCEWithLogitsLoss()
    for tr, tf, scale in ((0.9, 0.1, 1.0), (0.0, 1.0, 0.005)):
        real.grad = fake.grad = None
        rvf, fvr = real - fake.mean(), fake - real.mean()
        loss = scale * (bce(rvf, tr * torch.ones_like(rvf)) + bce(fvr, tf * torch.ones_like(fvr))) / 2
        loss.backward()
        dr, df = ragan_loss_grad(real.detach().cuda(), fake.detach().cuda(), tr, tf, scale)
        assert max_rel(dr.cpu().numpy(), real.grad.numpy()) <= 1e-5
        assert max_rel(df.cpu().numpy(), fake.grad.numpy()) <= 1e-5


def test_clip_grad_norm_vs_torch():
    from superresolution_amd import training as T
    from superresolution_amd.gan_training import clip_grad_norm_
    ws = T.Workspace(torch.device("cuda", 0))
    g = torch.Generator().manual_seed(4)
    for n, max_norm in ((1_000_003, 10.0), (4099, 1e6), (17, 0.5)):
        v = torch.randn(n, generator=g)
        p = torch.nn.Parameter(torch.zeros(n))
        p.grad = v.clone()
        want = torch.nn.utils.clip_grad_norm_([p], max_norm)
        buf = v.cuda()
        norm = clip_grad_norm_(buf, max_norm, ws)
        assert abs(float(norm) - float(want)) <= 5e-5 * float(want)      # torch sums in fp32, the kernel in fp64
        assert max_rel(buf.cpu().numpy(), p.grad.numpy()) <= 5e-5


def _gan_setup():
    from superresolution_amd import Discriminator, Generator, weights as W
    from superresolution_amd.gan_training import GanStep
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda().train()
    torch.manual_seed(3)
    d = Discriminator().cuda().train()
    lr = torch.from_numpy(W.synthetic_lr_batch(2, 2, "dense")).cuda()
    hr = torch.from_numpy(W.synthetic_hr_batch(2, seed=1)).cuda()
    return g, d, GanStep(g, d), lr, hr


def _check_disc_grads_vs_golden(P, gs):
    """``P``: {name -> gradient} of the discriminator after the discriminator half-step, against gan_step.npz."""
    assert list(P) == [str(k) for k in gs["d_keys"]]
    # One LeakyReLU sign flip between two fp32 computations (see the adversarial-gradient test below) moves the bias /
    # weight gradient of its channel -- a sum of 256 (last layers) to 31,000 signed terms -- by up to a few percent of
    # itself.  So: every tensor's sum of squares at 2e-3 (measured <= 1e-5), 99 % of ALL sampled entries within 1e-3 of
    # their tensor's max (measured: worst single entry 6.3e-3, in the 512-channel layer whose sums have 256 terms), none
    # beyond 5e-2.  The same backward on identical inputs is held to 2e-5 in test_discriminator_backward_vs_torch_autograd.
    errs = []
    for i, k in enumerate(P):
        gr = P[k].detach().double().cpu()
        ref_sum, ref_sq, ref_max = gs["d_grad_sums"][i]
        if k == "head.2.bias":
            # = sum of dL/dlogits over real and fake, which is exactly 0 for the relativistic-average loss (each mean
            # term cancels the other side's sum): both sides hold rounding residue only
            assert float(gr.abs().max()) <= 1e-6 and ref_max <= 1e-6
            continue
        e_sq = abs(float((gr * gr).sum()) - ref_sq) / max(ref_sq, 1e-30)
        assert e_sq <= 2 * STEP_TOL, (k, e_sq)
        stride = 13 if gr.numel() < 100000 else 509
        errs.append(np.abs(gr.flatten()[::stride].numpy() - gs["d_grad_sample_" + k]) / ref_max)
    errs = np.concatenate(errs)
    print("GAN step: discriminator gradient samples: max", errs.max(), "99th percentile", np.quantile(errs, 0.99), "of", errs.size)
    assert np.quantile(errs, 0.99) <= STEP_TOL and errs.max() <= 5e-2


def _check_gen_grads_vs_golden(G, gs):
    """``G``: {name -> clipped gradient} of the generator after the generator half-step, against gan_step.npz."""
    assert list(G) == [str(k) for k in gs["g_keys"]]
    scale_all = float(gs["g_grad_sums"][:, 2].max())
    worst = 0.0
    for i, k in enumerate(G):
        gr = G[k].detach().double().cpu()
        ref_sum, ref_sq, ref_max = gs["g_grad_sums"][i]
        if k.endswith((".block.0.bias", ".block.3.bias")) or k == "mid.0.bias":     # true gradient 0 (BatchNorm behind it)
            assert float(gr.abs().max()) <= 1e-4 * scale_all, k
            continue
        e_sq = abs(float((gr * gr).sum()) - ref_sq) / max(ref_sq, 1e-30)
        worst = max(worst, e_sq)
        assert e_sq <= 2 * STEP_TOL, (k, e_sq)
        sk = "g_grad_sample_" + k
        if sk in gs.files:
            e = float(np.abs(gr.flatten()[::7].numpy() - gs[sk]).max()) / ref_max
            worst = max(worst, e)
            assert e <= STEP_TOL, (k, e)
    print("GAN step: worst generator gradient error (after clipping)", worst)


def test_gan_step_vs_reference_golden():
    """One GAN iteration of the REAL reference (make_golden_gan.py): discriminator loss / logits / gradients, the
    adversarial gradient at the fake image, the generator's loss terms, its total gradient norm and clipped gradients."""
    gs = np.load(os.path.join(GOLDEN, "gan_step.npz"))
    g, d, step, lr, hr = _gan_setup()
    # -- discriminator step
    d_loss, real_logits = step.d_step(lr, hr)
    assert abs(float(d_loss) - float(gs["d_loss"])) <= 1e-5 * abs(float(gs["d_loss"]))
    assert max_rel(real_logits.cpu().numpy(), gs["real_logits"]) <= 1e-4
    _check_disc_grads_vs_golden({k: p.grad for k, p in d.named_parameters()}, gs)
    # -- generator step (the discriminator has just been updated, as in the reference)
    stats = step.g_step(lr, hr, real_logits)
    for k in ("g_adv", "g_pix", "g_ssim", "g_loss"):
        assert abs(float(stats[k]) - float(gs[k])) <= 2e-5 * abs(float(gs[k])), (k, float(stats[k]), float(gs[k]))
    assert abs(float(step.last_grad_norm) - float(gs["g_grad_norm"])) <= 1e-3 * float(gs["g_grad_norm"])
    _check_gen_grads_vs_golden({k: p.grad for k, p in g.named_parameters()}, gs)
    # the discriminator after its AdamW update
    assert max_rel(step.real_logits(hr).cpu().numpy(), gs["real_logits_after"]) <= 1e-4


def test_reference_gan_loop_body_runs_on_the_dropin_modules():
    """train.py:145-176 verbatim (one iteration that updates the discriminator) on ``superresolution_amd.Generator`` /
    ``Discriminator`` / ``RelativisticAverageLoss`` with STOCK ``torch.optim.AdamW`` and ``torch.nn.utils.clip_grad_norm_``:
    ``d_loss.backward()`` and ``g_loss.backward()`` reach the HIP backward operators through superresolution_amd/autograd.py.
    Run without autocast = the fp32 iteration of the real reference in gan_step.npz: same bars as GanStep's own test.
    ``weighted_l1`` as train.py:18-20 defines it; ``ssim`` = the stock-torch restatement of utils.py:21-41."""
    from oracle import torch_ref as TR
    from superresolution_amd import Discriminator, Generator, RelativisticAverageLoss, weights as W
    gs = np.load(os.path.join(GOLDEN, "gan_step.npz"))
    generator = Generator()
    generator.load_state_dict(W.synthetic_state_dict(0))
    generator = generator.cuda()
    torch.manual_seed(3)
    discriminator = Discriminator().cuda()
    opt_g = torch.optim.AdamW(generator.parameters(), lr=1e-4)                 # train.py:274-275, config.py:11-12
    opt_d = torch.optim.AdamW(discriminator.parameters(), lr=1e-6)
    criterion = RelativisticAverageLoss()                                       # train.py:128
    generator.train()
    discriminator.train()
    lr = torch.from_numpy(W.synthetic_lr_batch(2, 2, "dense")).cuda()
    hr = torch.from_numpy(W.synthetic_hr_batch(2, seed=1)).cuda()

    def weighted_l1(sr, hr):                                                    # train.py:18-20
        weight = (hr > 0.01).float() * 9 + 1
        return (weight * (sr - hr).abs()).mean()

    # ---- train.py:145-159 ----
    opt_d.zero_grad()
    with torch.no_grad():
        fake = generator(lr)
    real_logits = discriminator(hr)
    fake_logits = discriminator(fake)
    d_loss = criterion.discriminator_loss(real_logits, fake_logits)
    d_loss.backward()
    assert abs(d_loss.item() - float(gs["d_loss"])) <= 1e-5 * abs(float(gs["d_loss"]))
    assert max_rel(real_logits.detach().cpu().numpy(), gs["real_logits"]) <= 1e-4
    assert max_rel(fake_logits.detach().cpu().numpy(), gs["fake_logits"]) <= 2e-4
    _check_disc_grads_vs_golden({k: p.grad for k, p in discriminator.named_parameters()}, gs)
    opt_d.step()
    # ---- train.py:165-176 ----
    opt_g.zero_grad()
    fake = generator(lr)
    fake_logits = discriminator(fake)
    g_adv = criterion.generator_loss(real_logits.detach(), fake_logits)
    g_pix = weighted_l1(fake, hr)
    g_ssim = 1 - TR.ssim(fake, hr)
    g_loss = 0.005 * g_adv + 1.0 * g_pix + 0.3 * g_ssim
    g_loss.backward()
    norm = torch.nn.utils.clip_grad_norm_(generator.parameters(), max_norm=10.0)
    opt_g.step()
    # ----
    for k, v in (("g_adv", g_adv), ("g_pix", g_pix), ("g_ssim", g_ssim), ("g_loss", g_loss)):
        assert abs(v.item() - float(gs[k])) <= 2e-5 * abs(float(gs[k])), (k, v.item(), float(gs[k]))
    assert max_rel(fake_logits.detach().cpu().numpy(), gs["fake_logits2"]) <= 2e-4
    assert abs(norm.item() - float(gs["g_grad_norm"])) <= 1e-3 * float(gs["g_grad_norm"])
    _check_gen_grads_vs_golden({k: p.grad for k, p in generator.named_parameters()}, gs)
    with torch.no_grad():
        assert max_rel(discriminator(hr).cpu().numpy(), gs["real_logits_after"]) <= 1e-4
    # the next iteration as train.py has it: inside bf16 autocast, discriminator not updated (train.py:160-176)
    with torch.no_grad():
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            real_logits = discriminator(hr)
    opt_g.zero_grad()
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        fake = generator(lr)
        fake_logits = discriminator(fake)
        g_adv = criterion.generator_loss(real_logits.detach(), fake_logits)
        g_loss = 0.005 * g_adv + 1.0 * weighted_l1(fake, hr) + 0.3 * (1 - TR.ssim(fake, hr))
    g_loss.backward()
    torch.nn.utils.clip_grad_norm_(generator.parameters(), max_norm=10.0)
    opt_g.step()
    assert fake.dtype == torch.bfloat16 and fake_logits.dtype == torch.bfloat16
    assert torch.isfinite(g_loss) and all(torch.isfinite(p.grad).all() for p in generator.parameters())


def test_adversarial_gradient_at_the_fake_image_vs_reference_golden():
    """d generator_loss / d fake through the discriminator (its whole data-gradient chain), reference autograd."""
    from superresolution_amd.gan_training import ragan_loss_grad
    gs = np.load(os.path.join(GOLDEN, "gan_step.npz"))
    g, d, step, lr, hr = _gan_setup()
    d_loss, real_logits = step.d_step(lr, hr)          # updates D exactly as the fixture did before its generator step
    with torch.no_grad():
        fake = g(lr)
    fake_logits, saved = step.dgrad.forward(fake)
    assert max_rel(fake_logits.cpu().numpy(), gs["fake_logits2"]) <= 2e-4
    _, df = ragan_loss_grad(real_logits, fake_logits, 0.0, 1.0, want_real=False)
    dimg = step.dgrad.backward(df, saved, want_wgrad=False, want_dimg=True)
    # Two fp32 computations of a 10-layer LeakyReLU chain disagree on the sign of a few pre-activations that lie within
    # rounding of zero; each such unit changes the derivative 1 <-> 0.2 in its 3x3 footprint (measured: 336 of 93,750
    # elements above 1e-3 of max, in isolated 3x3 patches, worst 2e-2; mean error 1.1e-5).  So: the bulk tight, the
    # flips bounded.  The same chain on identical inputs is checked at 2e-5 in test_discriminator_backward_vs_torch_autograd.
    err = np.abs(dimg.cpu().numpy() - gs["dadv_dfake"]) / np.abs(gs["dadv_dfake"]).max()
    print("adversarial gradient at the fake image: max", err.max(), "mean", err.mean(), "above 1e-3:", int((err > 1e-3).sum()))
    assert np.quantile(err, 0.99) <= 1e-4 and err.mean() <= 1e-4
    assert (err > 1e-3).mean() <= 0.01 and err.max() <= 0.1


def test_gan_steps_run_and_skip_discriminator_update():
    g, d, step, lr, hr = _gan_setup()
    before = [p.detach().clone() for p in d.parameters()]
    out = step.step(lr, hr, update_d=False)              # train.py:160-163
    assert out["d_loss"] is None and all(torch.equal(a, b.detach()) for a, b in zip(before, d.parameters()))
    out = step.step(lr, hr, update_d=True)
    assert torch.isfinite(out["d_loss"]) and torch.isfinite(out["g_loss"])
    assert not all(torch.equal(a, b.detach()) for a, b in zip(before, d.parameters()))


def test_gan_step_16_bit_arithmetics_stay_close_to_the_fp32_step():
    """GanStep(precision="f16" | "bf16") -- and "auto" inside torch.autocast, the way train.py:148-149,166-167 run the
    generator (bfloat16 -> the bf16 step, float16 -> f16 products) -- against the fp32 iteration on the same inputs: losses
    within 1e-3 (f16) / 5e-3 (bf16), the flat generator gradient within 2 % / 8 % in the 2-norm (cosine > 0.9998 / 0.997: what
    11 and 8 significant bits leave), and not identical (the switch reaches the generator's kernels through GanStep)."""
    ref = None
    bars = {"f16": (1e-3, 2e-2, 0.9998), "bf16": (5e-3, 8e-2, 0.997)}
    for mode, ac in (("fp32", None), ("f16", None), ("auto", torch.float16), ("bf16", None), ("auto", torch.bfloat16)):
        g, d, _, lr, hr = _gan_setup()
        from superresolution_amd.gan_training import GanStep
        step = GanStep(g, d, precision=mode)
        if ac is not None:
            with torch.autocast(device_type="cuda", dtype=ac):
                out = step.step(lr, hr, update_d=True)
        else:
            out = step.step(lr, hr, update_d=True)
        flat = step.opt_g.flat_grad.detach().double().cpu()
        res = ({k: float(v) for k, v in out.items() if v is not None}, flat)
        if mode == "fp32":
            ref = res
            continue
        arith = mode if ac is None else ("bf16" if ac == torch.bfloat16 else "f16")
        loss_tol, rel_tol, cos_min = bars[arith]
        for k, v in res[0].items():
            assert abs(v - ref[0][k]) <= loss_tol * max(abs(ref[0][k]), 1e-3), (mode, k, v, ref[0][k])
        rel = float((res[1] - ref[1]).norm() / ref[1].norm())
        cos = float(res[1] @ ref[1] / (res[1].norm() * ref[1].norm()))
        print(f"GAN iteration, {mode} ({arith}): flat generator gradient vs fp32: rel {rel:.2e}, cosine {cos:.6f}")
        assert 1e-5 <= rel <= rel_tol and cos >= cos_min, (mode, rel, cos)


def test_gan_step_reusing_the_forward_is_bit_identical():
    """GanStep(reuse_forward=True): in an iteration that updates the discriminator the generator half-step takes the output
    and saved tensors of the discriminator half-step's generator forward (same weights, same batch) instead of running it
    again.  Losses, both flat gradients, every parameter and every BatchNorm running statistic after two iterations equal
    the two-forward schedule bit for bit; a half-step called on its own never sees a kept forward."""
    from superresolution_amd.gan_training import GanStep
    res = {}
    for reuse in (False, True):
        g, d, _, lr, hr = _gan_setup()
        step = GanStep(g, d, reuse_forward=reuse)
        outs = [step.step(lr, hr, update_d=True), step.step(lr, hr, update_d=False)]
        assert step._kept_forward is None
        step.d_step(lr, hr)
        assert step._kept_forward is None
        res[reuse] = ([{k: (None if v is None else v.clone()) for k, v in o.items()} for o in outs],
                      step.opt_g.flat_grad.clone(), step.opt_d.flat_grad.clone(),
                      {k: v.clone() for k, v in g.state_dict().items()}, [p.detach().clone() for p in d.parameters()])
    a, b = res[False], res[True]
    for oa, ob in zip(a[0], b[0]):
        for k in oa:
            assert (oa[k] is None and ob[k] is None) or torch.equal(oa[k], ob[k]), k
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert all(torch.equal(a[3][k], b[3][k]) for k in a[3])
    assert all(torch.equal(x, y) for x, y in zip(a[4], b[4]))
