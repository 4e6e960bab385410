"""GPU parity tests of the discriminator row: the HIP forward and the RaGAN loss
kernel, through the C ABI, against the fixture made from the real reference and the on-box oracle.
fp32 bar as for the generator: max error normalised by max|ref| <= 1e-4."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, max_rel

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4


@pytest.fixture(scope="module")
def gd():
    return np.load(os.path.join(GOLDEN, "discriminator.npz"))


@pytest.fixture(scope="module")
def disc():
    from superresolution_amd import Discriminator
    torch.manual_seed(0)                      # make_golden.py: torch.manual_seed(0); Discriminator()
    return Discriminator().cuda()


def test_logits_and_losses_vs_reference_golden(disc, gd, golden):
    from superresolution_amd import RelativisticAverageLoss, weights as W
    hr = torch.from_numpy(W.synthetic_hr_batch(2, seed=1)).cuda()
    sr = torch.from_numpy(golden["y_dense_sparse"]).cuda()
    with torch.no_grad():
        dr, df = disc(hr), disc(sr)
        crit = RelativisticAverageLoss()
        ld, lg = crit.discriminator_loss(dr, df), crit.generator_loss(dr, df)
    assert dr.shape == (2, 1, 8, 8) and dr.dtype == torch.float32 and dr.is_contiguous()
    e1, e2 = max_rel(dr.cpu().numpy(), gd["d_real"]), max_rel(df.cpu().numpy(), gd["d_fake"])
    print("discriminator max rel err vs reference:", e1, e2)
    assert e1 <= FP32_TOL and e2 <= FP32_TOL
    assert ld.dim() == 0 and lg.dim() == 0
    print("losses:", ld.item(), float(gd["loss_d"]), lg.item(), float(gd["loss_g"]))
    assert abs(ld.item() - float(gd["loss_d"])) <= 1e-5
    assert abs(lg.item() - float(gd["loss_g"])) <= 1e-5


@pytest.mark.parametrize("hw", [(125, 125), (37, 53), (16, 9), (1, 1)])
def test_every_layer_vs_cpu_oracle(hw):
    """Other weights than the fixture, sizes that leave ragged pixel tiles and odd strided rows;
    every conv output is compared (taps), not only the logits."""
    from oracle import oracle as O
    from superresolution_amd import Discriminator
    torch.manual_seed(5)
    d = Discriminator()
    with torch.no_grad():
        for p in d.parameters():
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)          # the default init zeroes every bias
    sd = {k: v.numpy().copy() for k, v in d.state_dict().items()}
    d = d.cuda()
    n = 3 if hw[0] < 100 else 1
    x = np.random.default_rng(2).random((n, 3) + hw, dtype=np.float32)
    ref_taps = []
    ref = O.discriminator_forward(sd, x, ref_taps)
    with torch.no_grad():
        y, taps = d.forward_taps(torch.from_numpy(x).cuda())
    assert len(taps) == len(ref_taps) == 10
    for i, (t, r) in enumerate(zip(taps, ref_taps)):
        assert tuple(t.shape) == r.shape, i
        err = max_rel(t.cpu().numpy(), r)
        print("layer", i, tuple(t.shape), err)
        assert err <= FP32_TOL, i
    assert max_rel(y.cpu().numpy(), ref) <= FP32_TOL


@pytest.mark.parametrize("n", [1, 5, 64])
def test_batches_are_image_independent(disc, n):
    x = torch.rand((n, 3, 125, 125), generator=torch.Generator().manual_seed(n)).cuda()
    with torch.no_grad():
        y = disc(x)
        assert y.shape == (n, 1, 8, 8) and torch.isfinite(y).all()
        for i in sorted({0, n // 2, n - 1}):
            assert torch.equal(y[i:i + 1], disc(x[i:i + 1])), (n, i)
        assert disc(x[:0]).shape == (0, 1, 8, 8)


@pytest.mark.parametrize("nr,nf", [(128, 128), (64, 192), (1, 1), (100000, 77777)])
def test_ragan_loss_vs_oracle(nr, nf):
    from oracle import oracle as O
    from superresolution_amd import RelativisticAverageLoss
    rng = np.random.default_rng(nr + nf)
    real = (rng.standard_normal(nr) * 3 + 0.5).astype(np.float32)
    fake = (rng.standard_normal(nf) * 3 - 0.5).astype(np.float32)
    real[0], fake[0] = 40.0, -40.0             # saturated logits: log1p(exp(-|x|)) must not overflow
    for labels in ((0.9, 0.1), (1.0, 0.0)):
        crit = RelativisticAverageLoss(*labels)
        with torch.no_grad():
            ld, lg = crit.both_losses(torch.from_numpy(real).cuda(), torch.from_numpy(fake).cuda())
        rd, rg = O.ragan_losses(real, fake, *labels)
        assert abs(ld.item() - rd) <= 1e-5 * max(1.0, abs(rd)), (ld.item(), rd)
        assert abs(lg.item() - rg) <= 1e-5 * max(1.0, abs(rg)), (lg.item(), rg)


def test_autograd_calls_carry_a_graph_and_weight_reload(disc):
    from superresolution_amd import RelativisticAverageLoss
    x = torch.rand(1, 3, 125, 125, device="cuda")
    y = disc(x)                                 # parameters require grad and grad mode is on: a graph (autograd.py)
    assert y.requires_grad and y.grad_fn is not None
    with torch.no_grad():
        assert max_rel(y.detach().cpu().numpy(), disc(x).cpu().numpy()) <= 1e-5     # the inference kernels agree
    with torch.no_grad():
        y0 = disc(x).clone()
        sd0 = {k: v.clone() for k, v in disc.state_dict().items()}
        sd1 = {k: v * 0.5 for k, v in sd0.items()}
        disc.load_state_dict(sd1)
        y1 = disc(x).clone()
        disc.load_state_dict(sd0)
        y2 = disc(x)
    assert not torch.equal(y0, y1) and torch.equal(y0, y2)
    r = torch.zeros(4, device="cuda", requires_grad=True)
    loss = RelativisticAverageLoss().generator_loss(r, r)
    assert loss.requires_grad and abs(loss.item() - float(np.log(2.0))) <= 1e-6
    with pytest.raises(RuntimeError, match="Error\\(s\\) in loading state_dict for Discriminator|size mismatch"):
        bad = dict(sd0)
        bad["head.2.weight"] = torch.zeros(1, 128, 3, 3, device="cuda")
        disc.load_state_dict(bad)


def test_full_batch_256_every_slot_matches_golden(disc, gd, golden):
    """The reference's GAN-phase batch (config.py:8): the two golden sr images repeated 128 times make the
    persistent workgroups of every conv layer walk many tiles; every copy must reproduce the reference's
    logits, and all copies must be bit-identical to each other."""
    sr = torch.from_numpy(np.tile(golden["y_dense_sparse"], (128, 1, 1, 1))).cuda()
    with torch.no_grad():
        y = disc(sr)
    assert y.shape == (256, 1, 8, 8)
    assert torch.equal(y[0::2], y[0:1].expand(128, -1, -1, -1))
    assert torch.equal(y[1::2], y[1:2].expand(128, -1, -1, -1))
    assert max_rel(y[:2].cpu().numpy(), gd["d_fake"]) <= FP32_TOL
    assert max_rel(y[254:].cpu().numpy(), gd["d_fake"]) <= FP32_TOL


def test_gan_phase_forward_values_end_to_end(disc, golden_inputs):
    """The forward values of one GAN-phase batch (train.py:149-172) composed from the HIP pieces -- generator,
    discriminator on hr and on the generator's output, RaGAN losses, weighted_l1 and SSIM -- against the same
    chain composed from the CPU oracle."""
    from oracle import oracle as O
    from superresolution_amd import Generator, RelativisticAverageLoss, pixel_losses, weights as W
    sd = W.synthetic_state_dict(0)
    g = Generator()
    g.load_state_dict(sd)
    g = g.cuda().eval()
    lr = torch.from_numpy(golden_inputs).cuda()
    hr_np = W.synthetic_hr_batch(2, seed=1)
    hr = torch.from_numpy(hr_np).cuda()
    crit = RelativisticAverageLoss()
    with torch.no_grad():
        fake = g(lr)
        real_logits, fake_logits = disc(hr), disc(fake)
        d_loss, g_adv = crit.both_losses(real_logits, fake_logits)
        pl = pixel_losses(fake, hr)
    dsd = {k: v.cpu().numpy() for k, v in disc.state_dict().items()}
    fake_ref = O.generator_forward(W.synthetic_state_dict_numpy(0), golden_inputs)
    rl_ref, fl_ref = O.discriminator_forward(dsd, hr_np), O.discriminator_forward(dsd, fake_ref)
    d_ref, g_ref = O.ragan_losses(rl_ref, fl_ref)
    _, ssim_ref, _ = O.metrics(fake_ref, hr_np)
    assert max_rel(fake_logits.cpu().numpy(), fl_ref) <= FP32_TOL
    assert abs(d_loss.item() - d_ref) <= 1e-5 and abs(g_adv.item() - g_ref) <= 1e-5
    assert abs(pl[3].item() - O.weighted_l1(fake_ref, hr_np)) <= 1e-5
    assert abs(pl[1].item() - ssim_ref) <= 1e-4


def test_bf16_autocast_surface(disc, gd, golden):
    """train.py:150,166 call D and the criterion under torch.autocast(bfloat16): logits come back bf16 (fp32
    arithmetic, rounded once), the loss fp32; both within bf16 rounding of the fp32 values."""
    from superresolution_amd import RelativisticAverageLoss
    sr = torch.from_numpy(golden["y_dense_sparse"]).cuda()
    with torch.no_grad():
        y32 = disc(sr)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = disc(sr)
            loss = RelativisticAverageLoss().generator_loss(yb, yb)
        assert disc(sr.to(torch.bfloat16)).dtype == torch.bfloat16
    assert yb.dtype == torch.bfloat16 and loss.dtype == torch.float32
    assert torch.equal(yb, y32.to(torch.bfloat16))
    assert np.isfinite(loss.item())
