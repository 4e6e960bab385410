"""CPU suite: pins the oracle (C restatement and torch functional restatement) against the golden
vectors generated from the real reference (tests/golden/make_golden.py).  The reference has no
tests of its own so these fixtures are the pin."""
import numpy as np
import pytest
import torch

from conftest import max_rel
from oracle import oracle as O
from oracle import torch_ref as TR
from superresolution_amd import weights as W


@pytest.fixture(scope="module")
def oracle_taps(synth_sd_np, golden_inputs):
    taps = {}
    y = O.generator_forward(synth_sd_np, golden_inputs[:1], taps)
    return y, taps


def test_c_oracle_output_matches_reference(oracle_taps, golden):
    y, _ = oracle_taps
    assert y.shape == (1, 3, 125, 125)
    assert max_rel(y, golden["y_dense_sparse"][:1]) <= 1e-5


def test_c_oracle_taps_match_reference(oracle_taps, golden):
    _, taps = oracle_taps
    for nm in O.TAP_NAMES:
        t = taps[nm]
        assert list(t.shape) == list(golden[f"tap_{nm}_shape"]), nm
        assert max_rel(t.reshape(-1)[::64], golden[f"tap_{nm}_sample"]) <= 1e-5, nm
        ss = (t.astype(np.float64) ** 2).sum()
        assert abs(ss - float(golden[f"tap_{nm}_sumsq"])) <= 1e-5 * ss, nm
    for nm in ("ca8_scale", "ca17_scale"):
        assert np.abs(taps[nm] - golden[nm]).max() <= 1e-6


def test_c_oracle_sparse_image(synth_sd_np, golden, golden_inputs):
    y = O.generator_forward(synth_sd_np, golden_inputs[1:2])
    assert max_rel(y, golden["y_dense_sparse"][1:2]) <= 1e-5


def test_torch_ref_matches_reference(golden, golden_inputs):
    sd = W.synthetic_state_dict(0)
    y = TR.generator_forward(sd, torch.from_numpy(golden_inputs)).numpy()
    assert max_rel(y, golden["y_dense_sparse"]) <= 1e-5


def test_torch_ref_default_init_config1(golden_default):
    """BASELINE.json configs[0]: Generator() default init, batch 16 fp32 eval on CPU.  Our module
    tree constructs parameters in the reference's order, so the same torch seed gives the
    reference's parameters; the functional restatement then reproduces its output."""
    from superresolution_amd import Generator
    torch.manual_seed(0)
    g = Generator()
    sd = g.state_dict()
    assert list(sd.keys()) == list(golden_default["keys"])
    assert sum(p.numel() for p in g.parameters()) == int(golden_default["n_params"]) == 3_214_974
    sums = np.array([v.double().sum().item() for v in sd.values()])
    assert np.abs(sums - golden_default["key_sums"]).max() <= 1e-9
    x = torch.from_numpy(W.synthetic_lr_batch(16, seed=0))
    y = TR.generator_forward(sd, x)
    assert list(y.shape) == list(golden_default["y_shape"])
    assert max_rel(y[:2].numpy(), golden_default["y_first2"]) <= 1e-5
    assert max_rel(y.contiguous().flatten()[::97].numpy(), golden_default["y_sample"]) <= 1e-5


def test_pixel_shuffle_exact():
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "pixel_shuffle.npz"))
    assert np.array_equal(O.pixel_shuffle(g["x"], 2), g["y"])


def test_metrics_oracle_matches_reference(golden, golden_metrics):
    hr = W.synthetic_hr_batch(2, seed=1)
    ps, ss, l1 = O.metrics(golden["y_dense_sparse"], hr)
    assert abs(ps - float(golden_metrics["psnr"])) <= 1e-4
    assert abs(ss - float(golden_metrics["ssim"])) <= 1e-5
    assert abs(l1 - float(golden_metrics["l1"])) <= 1e-6
    ps, ss, l1 = O.metrics(hr, hr)
    assert ps == 100.0 and abs(ss - 1.0) <= 1e-6 and l1 == 0.0      # utils.py:7-8
    noise = W.hash_normal(3, "noise", hr.size).astype(np.float32).reshape(hr.shape)
    srn = np.clip(hr + 0.05 * noise, 0, 1).astype(np.float32)
    ps, ss, l1 = O.metrics(srn, hr)
    assert abs(ps - float(golden_metrics["psnr_noisy"])) <= 1e-3
    assert abs(ss - float(golden_metrics["ssim_noisy"])) <= 1e-4
    assert abs(l1 - float(golden_metrics["l1_noisy"])) <= 1e-6


def test_torch_ref_metrics_match_reference(golden, golden_metrics):
    hr = torch.from_numpy(W.synthetic_hr_batch(2, seed=1))
    sr = torch.from_numpy(golden["y_dense_sparse"])
    assert abs(TR.psnr(sr, hr).item() - float(golden_metrics["psnr"])) <= 1e-5
    assert abs(TR.ssim(sr, hr).item() - float(golden_metrics["ssim"])) <= 1e-6


def test_bilinear_oracle_matches_torch():
    x = torch.from_numpy(W.hash_uniform(2, "bil", 2 * 3 * 128 * 128).astype(np.float32).reshape(2, 3, 128, 128))
    ref = torch.nn.functional.interpolate(x, (125, 125), mode="bilinear", align_corners=False).numpy()
    assert np.abs(O.bilinear_resize(x.numpy(), 125, 125) - ref).max() <= 2e-6


def test_oracle_ops_against_torch_small():
    """Each C op against the stock torch op on a small ragged case (odd sizes, k=9 border)."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 5, 7, 9)).astype(np.float32)
    w = rng.standard_normal((4, 5, 9, 9)).astype(np.float32) * 0.1
    b = rng.standard_normal(4).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=4)
    assert max_rel(O.conv2d(x, w, b), ref.numpy()) <= 1e-5
    g, bt, mu, var = (rng.standard_normal(5).astype(np.float32) for _ in range(4))
    var = np.abs(var) + 0.5
    ref = torch.nn.functional.batch_norm(torch.from_numpy(x), torch.from_numpy(mu), torch.from_numpy(var),
                                         torch.from_numpy(g), torch.from_numpy(bt), False, 0.1, 1e-5)
    assert max_rel(O.batchnorm_eval(x, g, bt, mu, var), ref.numpy()) <= 1e-6
    sl = np.abs(rng.standard_normal(5)).astype(np.float32)
    ref = torch.nn.functional.prelu(torch.from_numpy(x), torch.from_numpy(sl))
    assert np.array_equal(O.prelu(x, sl), ref.numpy())
    assert np.array_equal(O.crop(x, 3, 4), x[:, :, :3, :4])
    # empty batch
    assert O.conv2d(x[:0], w, b).shape == (0, 4, 7, 9)


def test_oracle_train_mode_forward_matches_reference_golden():
    """oracle.generator_forward_train (batch-statistics BatchNorm) against the fixture made by running the
    real reference in train mode (tests/golden/make_golden_train.py): output and the running statistics of
    all 33 BatchNorm layers after one forward."""
    import os
    from conftest import GOLDEN
    from oracle import oracle as O
    from superresolution_amd import weights as W
    gt = np.load(os.path.join(GOLDEN, "generator_train.npz"))
    x = np.concatenate([W.synthetic_lr_batch(3, 2, "dense"), W.synthetic_lr_batch(1, 2, "sparse")], 0)
    y, stats = O.generator_forward_train(W.synthetic_state_dict_numpy(0), x)
    assert max_rel(y, gt["y1"]) <= 1e-5
    for i, k in enumerate(str(k) for k in gt["bn_keys"]):
        assert np.abs(stats[k][0] - gt["mean1"][i]).max() <= 1e-6, k
        assert max_rel(stats[k][1], gt["var1"][i]) <= 1e-6, k
    assert len(stats) == 33


def test_torch_restatement_of_the_training_steps_matches_reference_golden():
    """oracle/torch_ref.py's pre-train step and GAN iteration (bench.py's cpu_baseline for configs[3] / [4]) against ONE step
    of the real reference (tests/golden/train_step.npz, gan_step.npz: made by make_golden_bwd.py / make_golden_gan.py)."""
    import os
    from conftest import GOLDEN
    from superresolution_amd import Discriminator
    gs = np.load(os.path.join(GOLDEN, "train_step.npz"))
    sd = W.synthetic_state_dict(0)
    P = {k: v.clone().requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    opt = torch.optim.AdamW(list(P.values()), lr=1e-4)
    lr = torch.from_numpy(np.concatenate([W.synthetic_lr_batch(3, 2, "dense"), W.synthetic_lr_batch(1, 2, "sparse")], 0))
    hr = torch.from_numpy(W.synthetic_hr_batch(4, seed=1))
    loss = TR.pretrain_step(P, opt, lr, hr)
    assert abs(loss - float(gs["loss"])) <= 1e-5 * abs(float(gs["loss"]))
    gg = np.load(os.path.join(GOLDEN, "gan_step.npz"))
    P = {k: v.clone().requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    torch.manual_seed(3)
    D = {k: v.detach().clone().requires_grad_() for k, v in Discriminator().named_parameters()}
    opt_g, opt_d = torch.optim.AdamW(list(P.values()), lr=1e-4), torch.optim.AdamW(list(D.values()), lr=1e-6)
    g_loss = TR.gan_iteration(P, D, opt_g, opt_d, torch.from_numpy(W.synthetic_lr_batch(2, 2, "dense")),
                              torch.from_numpy(W.synthetic_hr_batch(2, seed=1)))
    assert abs(g_loss - float(gg["g_loss"])) <= 2e-5 * abs(float(gg["g_loss"]))
