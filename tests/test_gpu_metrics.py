"""GPU: metrics kernel (utils.psnr / utils.ssim / L1), bilinear resize, validate() on the HIP path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_metrics_kernel_vs_reference_golden(golden, golden_metrics):
    from superresolution_amd import batch_metrics, weights as W
    hr = torch.from_numpy(W.synthetic_hr_batch(2, seed=1)).cuda()
    sr = torch.from_numpy(golden["y_dense_sparse"]).cuda()
    m = batch_metrics(sr, hr).cpu().numpy()
    assert abs(m[0] - float(golden_metrics["psnr"])) <= 1e-3
    assert abs(m[1] - float(golden_metrics["ssim"])) <= 1e-5
    assert abs(m[2] - float(golden_metrics["l1"])) <= 1e-6
    m = batch_metrics(hr, hr).cpu().numpy()
    assert m[0] == 100.0 and abs(m[1] - 1.0) <= 1e-5 and m[2] == 0.0       # utils.py:7-8
    noise = torch.from_numpy(W.hash_normal(3, "noise", hr.numel()).astype(np.float32)).view_as(hr).cuda()
    srn = (hr + 0.05 * noise).clamp(0, 1)
    m = batch_metrics(srn, hr).cpu().numpy()
    assert abs(m[0] - float(golden_metrics["psnr_noisy"])) <= 1e-3
    assert abs(m[1] - float(golden_metrics["ssim_noisy"])) <= 1e-4
    assert abs(m[2] - float(golden_metrics["l1_noisy"])) <= 1e-6


def test_metrics_kernel_vs_oracle_ragged_shape():
    from oracle import oracle as O
    from superresolution_amd import batch_metrics
    rng = np.random.default_rng(3)
    a = rng.random((3, 2, 37, 70), dtype=np.float32)
    b = rng.random((3, 2, 37, 70), dtype=np.float32)
    m = batch_metrics(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    ps, ss, l1 = O.metrics(a, b)
    assert abs(m[0] - ps) <= 1e-3 and abs(m[1] - ss) <= 1e-5 and abs(m[2] - l1) <= 1e-6


def test_pixel_losses_weighted_l1(golden, golden_metrics):
    """weighted_l1 (train.py:18-20) from the metrics pass: against the oracle's fp64 restatement and,
    as the expression is two lines of torch, against torch itself; the first three outputs must be
    batch_metrics' bit for bit (same pass, same reduction order).  The fixtures hold no weighted_l1
    value from the reference run: that number is pinned by the torch expression only."""
    from oracle import oracle as O
    from superresolution_amd import batch_metrics, pixel_losses, weighted_l1, weights as W
    hr = torch.from_numpy(W.synthetic_hr_batch(2, seed=1)).cuda()
    sr = torch.from_numpy(golden["y_dense_sparse"]).cuda()
    rng = np.random.default_rng(4)
    a = torch.from_numpy(rng.random((3, 2, 37, 70), dtype=np.float32)).cuda()
    b = torch.from_numpy((rng.random((3, 2, 37, 70), dtype=np.float32) * 0.03)).cuda()   # straddles 0.01
    for s, h in ((sr, hr), (a, b)):
        out = pixel_losses(s, h)
        assert out.shape == (4,) and torch.equal(out[:3], batch_metrics(s, h))
        ref = O.weighted_l1(s.cpu().numpy(), h.cpu().numpy())
        tref = (((h > 0.01).float() * 9 + 1) * (s - h).abs()).mean().item()
        assert abs(out[3].item() - ref) <= 1e-6 * max(1.0, abs(ref)), (out[3].item(), ref)
        assert abs(out[3].item() - tref) <= 2e-6 * max(1.0, abs(tref))
        assert weighted_l1(s, h).item() == out[3].item()


def test_bilinear_resize_matches_interpolate():
    """Optional op: same numbers as F.interpolate(bilinear, align_corners=False)."""
    from superresolution_amd import bilinear_resize
    x = torch.rand(4, 3, 128, 128, device="cuda")
    y = bilinear_resize(x, (125, 125))
    ref = torch.nn.functional.interpolate(x.cpu(), (125, 125), mode="bilinear", align_corners=False)
    assert y.shape == (4, 3, 125, 125)
    assert (y.cpu() - ref).abs().max().item() <= 2e-6
    # identity size and an upscale
    assert torch.equal(bilinear_resize(x, (128, 128)), x)
    y2 = bilinear_resize(x[:1], (200, 131))
    ref2 = torch.nn.functional.interpolate(x[:1].cpu(), (200, 131), mode="bilinear", align_corners=False)
    assert (y2.cpu() - ref2).abs().max().item() <= 2e-6


def test_validate_on_hip_path_matches_oracle(golden_inputs):
    from oracle import oracle as O
    from superresolution_amd import Generator, evaluate as E, weights as W
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda().eval()
    lr = torch.from_numpy(golden_inputs)
    hr = torch.from_numpy(W.synthetic_hr_batch(2, seed=1))
    got = E.validate(g, E.batched(lr, hr, 1, range(2)), device=torch.device("cuda"))
    sd = W.synthetic_state_dict_numpy(0)
    per = [O.metrics(O.generator_forward(sd, golden_inputs[i:i + 1]), hr[i:i + 1].numpy()) for i in range(2)]
    want = np.mean(np.array(per), axis=0)
    assert abs(got[0] - want[0]) <= 1e-3 and abs(got[1] - want[1]) <= 1e-5 and abs(got[2] - want[2]) <= 1e-6


def _big_eval_set(nb, bs):
    from superresolution_amd import weights as W
    lr = torch.from_numpy(W.synthetic_lr_batch(nb * bs, seed=31, kind="dense"))
    hr = torch.from_numpy(W.synthetic_hr_batch(nb * bs, seed=32))
    return lr, hr


def test_validate_bf16_four_batches_of_256():
    """BASELINE.json configs[2], one rank's share: bf16, 4 batches of 256 through evaluate.validate
    (train.py:57-72 semantics) == the hand mean of per-batch batch_metrics, and its PSNR within 0.05 dB of the
    fp32 path on the same data (north_star's bf16 bar)."""
    from superresolution_amd import Generator, batch_metrics, evaluate as E, weights as W
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda().eval()
    nb, bs = 4, 256
    lr, hr = _big_eval_set(nb, bs)
    dev = torch.device("cuda")
    got = E.validate(g, E.batched(lr, hr, bs, range(nb)), device=dev, forward_dtype=torch.bfloat16)
    per = []
    for b in range(nb):
        sl = slice(b * bs, (b + 1) * bs)
        sr = g(lr[sl].cuda().to(torch.bfloat16))
        assert sr.dtype == torch.bfloat16
        per.append(batch_metrics(sr.float(), hr[sl].cuda()).double().cpu())
    hand = torch.stack(per).mean(0).numpy()
    assert np.allclose(np.array(got), hand, rtol=0, atol=1e-9)
    f32 = E.validate(g, E.batched(lr, hr, bs, range(nb)), device=dev)
    print("validate 4x256: bf16", got, "fp32", f32)
    assert abs(got[0] - f32[0]) <= 0.05
    assert abs(got[1] - f32[1]) <= 1e-3 and abs(got[2] - f32[2]) <= 1e-3


def test_validate_on_train_mode_generator_uses_eval_and_leaves_buffers():
    """train.py:59,71,104: validate() is called on a train-mode generator; it must evaluate with running
    statistics, leave every BatchNorm buffer untouched and hand the module back in train mode."""
    from superresolution_amd import Generator, evaluate as E, weights as W
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda()
    lr, hr = _big_eval_set(2, 3)
    want = E.validate(g.eval(), E.batched(lr, hr, 3, range(2)), device=torch.device("cuda"))
    before = {k: v.clone() for k, v in g.state_dict().items()}
    g.train()
    got = E.validate(g, E.batched(lr, hr, 3, range(2)), device=torch.device("cuda"))
    assert g.training
    assert got == want
    after = g.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)
