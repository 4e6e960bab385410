"""GPU: the 16-bit pre-train step (bf16 activation / gradient tensors, `precision="bf16"`; the storage and product types of
the reference's own step under torch.autocast(bfloat16), train.py:91).

(a) Every bf16 operator against its fp32 namesake ON THE SAME VALUES (inputs that are exactly representable in bf16): products
    of bf16 numbers are exact in fp32 and both sides accumulate in fp32, so tensor outputs may differ by the final rounding to
    bf16 only (2^-8 relative per element, stated as 2^-7 of the tensor's maximum to admit cancellation) and parameter
    gradients (fp32 on both sides) by summation order (1e-5 of max).
(b) One whole step against ONE step of the real reference run both ways (tests/golden/train_step.npz: fp32;
    train_step_autocast.npz: inside torch.autocast(bfloat16) -- make_golden_bwd.py): the bf16 step has to be as close to the
    fp32 step as the reference's own autocast step is (same arithmetic class): loss and per-tensor gradient deviation within
    1.5x the reference autocast's (measured: see the printed table)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from test_gpu_training import _hash_tensor, _nhwc, _ref_weighted_l1, _resblock, _sample_rel, _step_batch

pytestmark = pytest.mark.gpu
ROUND = 2.0 ** -7       # of the tensor's max: one bf16 rounding of the stored result
SUMS = 1e-5             # fp32 outputs: summation order only


def _b(t):
    """Round to bf16: (the bf16 tensor, the same values as fp32)."""
    tb = t.to(torch.bfloat16).contiguous()
    return tb, tb.float().contiguous()


def _close_rounded(got_bf16, ref_f32, what, tol=ROUND):
    assert got_bf16.dtype == torch.bfloat16, what
    err = float((got_bf16.float() - ref_f32).abs().max() / ref_f32.abs().max())
    print(f"  {what}: {err:.2e} of max")
    assert err <= tol, (what, err)


def _close_sums(got, ref, what, tol=SUMS):
    assert got.dtype == torch.float32, what
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    print(f"  {what}: {err:.2e} of max")
    assert err <= tol, (what, err)


@pytest.mark.parametrize("hw,n", [(64, 3), (128, 2)])
def test_conv3x3_bf16_forward_dgrad_wgrad_vs_the_fp32_operators(hw, n):
    from superresolution_amd import training as T
    dev = torch.device("cuda")
    ws = T.Workspace(dev)
    w = _hash_tensor(3, "w", (96, 96, 3, 3), -0.06, 0.06).to(torch.bfloat16).float().cuda()       # bf16-representable weights
    bias = _hash_tensor(4, "b", (96,)).cuda()
    xb, xf = _b(_nhwc(_hash_tensor(5, "x", (n, 96, hw, hw))).cuda())
    gb, gf = _b((_nhwc(_hash_tensor(6, "g", (n, 96, hw, hw))) * 2.0 ** -20).cuda())               # a gradient-sized tensor
    rb_, rf = _b((_nhwc(_hash_tensor(7, "r", (n, 96, hw, hw))) * 2.0 ** -20).cuda())
    with T.conv_precision("bf16"):
        yb = T.conv3x3(xb, T.conv_pack(w), bias)
        dxb = T.conv3x3(gb, T.conv_pack(w, mode=1), res=rb_ if hw == 64 else None)
        dwb, dbb = T.conv3x3_wgrad(xb, gb, ws)
    with T.conv_precision("fp32"):
        yf = T.conv3x3(xf, T.conv_pack(w), bias)
        dxf = T.conv3x3(gf, T.conv_pack(w, mode=1), res=rf if hw == 64 else None, scale_input=True)
        dwf, dbf = T.conv3x3_wgrad(xf, gf, ws)
    _close_rounded(yb, yf, "conv forward")
    _close_rounded(dxb, dxf, "conv data gradient (+ residual)")
    _close_sums(dwb, dwf, "weight gradient")
    _close_sums(dbb, dbf, "bias gradient")


def test_upsample_conv_bf16_with_pixel_shuffle_vs_fp32():
    from superresolution_amd import training as T
    w = _hash_tensor(8, "wu", (384, 96, 3, 3), -0.06, 0.06).to(torch.bfloat16).float().cuda()
    bias = _hash_tensor(9, "bu", (384,)).cuda()
    xb, xf = _b(_nhwc(_hash_tensor(10, "xu", (2, 96, 64, 64))).cuda())
    with T.conv_precision("bf16"):
        yb = T.conv3x3(xb, T.conv_pack(w), bias, cout_blocks=4, shuffle=True)
        gb, gf = _b(_nhwc(_hash_tensor(11, "gu", (2, 96, 64, 64))).cuda())
        db = T.conv3x3(gb, T.conv_pack(w, mode=1, block=2))
    with T.conv_precision("fp32"):
        yf = T.conv3x3(xf, T.conv_pack(w), bias, cout_blocks=4, shuffle=True)
        df = T.conv3x3(gf, T.conv_pack(w, mode=1, block=2), scale_input=True)
    assert yb.shape == (2, 128, 128, 96)
    _close_rounded(yb, yf, "upsample conv + PixelShuffle")
    _close_rounded(db, df, "data gradient of cout block 2")


def test_batchnorm_prelu_bf16_forward_backward_vs_fp32():
    from superresolution_amd import training as T
    dev = torch.device("cuda")
    ws = T.Workspace(dev)
    rb = _resblock().cuda()
    gamma, beta, slope = rb.block[1].weight.detach(), rb.block[1].bias.detach(), rb.block[2].weight.detach()
    cb, cf = _b(_nhwc(_hash_tensor(21, "c", (3, 96, 64, 64), -2.0, 3.0)).cuda())
    rb_, rf = _b(_nhwc(_hash_tensor(22, "res", (3, 96, 64, 64))).cuda())
    db_, df = _b((_nhwc(_hash_tensor(23, "d", (3, 96, 64, 64))) * 2.0 ** -18).cuda())
    for kw_b, kw_f, what in (({"slope": slope}, {"slope": slope}, "BatchNorm + PReLU"),
                             ({"res": rb_, "res_factor": 0.1}, {"res": rf, "res_factor": 0.1}, "BatchNorm + residual")):
        ob, sb = T.bn_train_fwd(cb, gamma, beta, ws, **kw_b)
        of, sf = T.bn_train_fwd(cf, gamma, beta, ws, **kw_f)
        _close_rounded(ob, of, what + " forward")
        _close_sums(sb, sf, what + " batch statistics", 1e-6)
    for sl, scale in ((slope, 1.0), (None, 0.1)):
        dcb, dgb, dbb, dsb = T.bn_train_bwd(cb, db_, sb, gamma, beta, ws, slope=sl, dout_scale=scale)
        dcf, dgf, dbf, dsf = T.bn_train_bwd(cf, df, sf, gamma, beta, ws, slope=sl, dout_scale=scale)
        _close_rounded(dcb, dcf, "BatchNorm backward dc")
        _close_sums(dgb, dgf, "dgamma")
        _close_sums(dbb, dbf, "dbeta")
        if sl is not None:
            _close_sums(dsb, dsf, "dslope")


def test_scalar_prelu_and_channel_attention_bf16_vs_fp32():
    from superresolution_amd import Generator, training as T, weights as W
    dev = torch.device("cuda")
    ws = T.Workspace(dev)
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda()
    xb, xf = _b(_nhwc(_hash_tensor(31, "x", (3, 96, 64, 64), -1.0, 2.0)).cuda())
    db_, df = _b((_nhwc(_hash_tensor(32, "d", (3, 96, 64, 64))) * 2.0 ** -19).cuda())
    d2b, d2f = _b((_nhwc(_hash_tensor(33, "d2", (3, 96, 64, 64))) * 2.0 ** -19).cuda())
    slope = g.input[1].weight.detach()
    _close_rounded(T.prelu_scalar(xb, slope), T.prelu_scalar(xf, slope), "scalar PReLU forward")
    dxb, dsb = T.prelu_scalar_bwd(xb, db_, slope, ws, dy2=d2b)
    dxf, dsf = T.prelu_scalar_bwd(xf, df, slope, ws, dy2=d2f)
    _close_rounded(dxb, dxf, "scalar PReLU backward")
    _close_sums(dsb, dsf, "dslope")
    ca = g.blocks[8]
    yb, Sb = T.chan_attn_forward(xb, ca)
    yf, Sf = T.chan_attn_forward(xf, ca)
    _close_rounded(yb, yf, "channel attention forward")
    _close_sums(Sb["scale"], Sf["scale"], "attention scales", 1e-6)
    dxb, gb = T.chan_attn_backward(db_, Sb, ca, ws)
    dxf, gf = T.chan_attn_backward(df, Sf, ca, ws)
    _close_rounded(dxb, dxf, "channel attention backward")
    for k in gb:
        _close_sums(gb[k], gf[k], "attention " + k)


def test_stem_and_head_bf16_vs_fp32():
    from superresolution_amd import Generator, training as T, weights as W
    dev = torch.device("cuda")
    ws = T.Workspace(dev)
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda()
    lr, _ = _step_batch()
    w = g.input[0].weight.detach()
    wm = w.permute(1, 2, 3, 0).reshape(243, 96).contiguous()
    x1f = T.conv9x9_c3(lr, wm, g.input[0].bias.detach(), 64, False)
    x1b = T.conv9x9_c3(lr, wm, g.input[0].bias.detach(), 64, False, out_dtype=torch.bfloat16)
    _close_rounded(x1b, x1f, "stem forward")
    # stem weight gradient: bf16 activation gradient against the fp32 image
    db_, df = _b((_nhwc(_hash_tensor(41, "d", (4, 96, 64, 64))) * 2.0 ** -18).cuda())
    # (the 16-bit operator multiplies bf16 x bf16 -- inside autocast the conv's image operand is bf16 too -- so both sides get
    # an image that is representable in bf16)
    lrr = lr.to(torch.bfloat16).float().contiguous()
    rb_, bb = T.conv9x9_c3_wgrad(db_, lrr, ws, flip=False, want_bias=True)
    rf, bf = T.conv9x9_c3_wgrad(df, lrr, ws, flip=False, want_bias=True)
    _close_sums(rb_, rf, "stem weight gradient")
    _close_sums(bb, bf, "stem bias gradient")
    # head: bf16 (n, 128, 128, 96) -> fp32 image; its weights are rounded to bf16 by the 16-bit step, so compare on those
    wh = g.output[2].weight.detach().to(torch.bfloat16).float()
    vb, vf = _b(_nhwc(_hash_tensor(42, "v", (2, 96, 128, 128))).cuda())
    yb = T.head_forward(vb, wh, g.output[2].bias)
    yf = T.head_forward(vf, wh, g.output[2].bias)
    _close_sums(yb, yf, "head forward")
    # head data gradient: fp32 image gradient -> bf16 (n, 128, 128, 96)
    dsr = (_hash_tensor(43, "dsr", (2, 3, 125, 125)) * 2.0 ** -16).cuda()
    whm = g.output[2].weight.detach().permute(0, 2, 3, 1).reshape(243, 96).contiguous()
    _close_rounded(T.conv9x9_c3(dsr, whm, None, 128, True, out_dtype=torch.bfloat16), T.conv9x9_c3(dsr, whm, None, 128, True),
                   "head data gradient")


@pytest.mark.parametrize("flip,n", [(False, 3), (True, 3), (True, 1)])
def test_conv9x9_weight_gradient_bf16_mfma_vs_torch_emulation(flip, n):
    """srgen_conv9x9_c3_wgrad_bf16 on rows of 64 / 128 pixels (srgen_wgrad9_bf16.hip: bf16 MFMA, eight shifted copies of
    every image row): against the gradient GEMM written with stock torch ops in fp64 on the CPU, both operands rounded to bf16
    (products exact, fp32 accumulation over up to 49k pixels: bar 2e-5 of max).  flip = False: the stem (Generator.py:50; act
    = d x1 at 64 x 64, img = the lr input); flip = True: the head (:71; act = its 128 x 128 input, img = d out at 125 x 125,
    the crop's zero rows / columns outside).  n = 3: slices that start in the middle of an image."""
    from superresolution_amd import training as T
    ws = T.Workspace(torch.device("cuda"))
    hw, ih = (128, 125) if flip else (64, 64)
    act_b, act_f = _b(_nhwc(_hash_tensor(51, "a9", (n, 96, hw, hw))).cuda())
    img = (_hash_tensor(52, "i9", (n, 3, ih, ih)) * (2.0 ** -12 if flip else 1.0)).cuda()
    r, db = T.conv9x9_c3_wgrad(act_b, img, ws, flip=flip, want_bias=True)
    a64 = act_f.cpu().double().permute(0, 3, 1, 2)                          # (n, 96, hw, hw)
    i64 = img.cpu().to(torch.bfloat16).double()
    if flip:       # head: dw[co][ci][ky][kx] of conv2d(v, w, padding=4)[..., :125, :125]; r[ci][(co, ky, kx)]
        dout = torch.zeros((n, 3, hw, hw), dtype=torch.float64)
        dout[:, :, :ih, :ih] = i64
        dw = torch.nn.grad.conv2d_weight(a64, (3, 96, 9, 9), dout, padding=4)
        ref = dw.permute(1, 0, 2, 3).reshape(96, 243)
    else:          # stem: dw[co][c3][ky][kx] of conv2d(lr, w, padding=4); r[co][(c3, ky, kx)]
        ref = torch.nn.grad.conv2d_weight(i64, (96, 3, 9, 9), a64, padding=4).reshape(96, 243)
    e = float((r.cpu().double() - ref).abs().max() / ref.abs().max())
    eb = float((db.cpu().double() - a64.sum(dim=(0, 2, 3))).abs().max() / a64.sum(dim=(0, 2, 3)).abs().max())
    print(f"  9x9 weight gradient (flip {flip}, n {n}) vs fp64 emulation: {e:.2e}, bias {eb:.2e}")
    assert e <= 2e-5 and eb <= 2e-5, (e, eb)


def test_pretrain_step_bf16_vs_reference_fp32_and_autocast_golden():
    """(b) of the module docstring."""
    from superresolution_amd import Generator, training as T, weights as W
    g32 = np.load(os.path.join(GOLDEN, "train_step.npz"))
    gac = np.load(os.path.join(GOLDEN, "train_step_autocast.npz"))
    lr, hr = _step_batch()
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda().train()
    step = T.PretrainStep(g, lr=1e-4, precision="bf16")
    loss = float(step.forward_backward(lr, hr))
    # "auto" inside bf16 autocast -- the way train.py:91 calls the generator -- runs the same arithmetic
    g2 = Generator()
    g2.load_state_dict(W.synthetic_state_dict(0))
    g2 = g2.cuda().train()
    step2 = T.PretrainStep(g2, lr=1e-4)
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        assert T.resolve_precision("auto") == "bf16"
        assert float(step2.forward_backward(lr, hr)) == loss
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(g.parameters(), g2.parameters()))
    ref_loss, ac_loss = float(g32["loss"]), float(gac["loss"])
    print("loss: fp32 reference", ref_loss, "reference autocast", ac_loss, "this bf16 step", loss)
    assert abs(loss - ref_loss) <= 1.5 * abs(ac_loss - ref_loss) + 1e-5 * abs(ref_loss)
    P = dict(g.named_parameters())
    ratios = []
    for k in g32.files:
        if not k.startswith("grad_sample_"):
            continue
        v = P[k[len("grad_sample_"):]].grad.detach().flatten()[::7].cpu().numpy()
        ours, theirs = _sample_rel(v, g32[k]), _sample_rel(gac[k], g32[k])
        print(f"  {k}: this step {ours:.2e}, reference autocast {theirs:.2e}")
        # (a tensor of one or two values -- a scalar PReLU slope -- is one draw of the rounding noise, not a norm over many)
        assert ours <= (1.5 if v.size > 4 else 3.0) * theirs and ours <= 6e-2, (k, ours, theirs)
        ratios.append(ours / theirs)
    print("  this step's deviation / the reference autocast's: median", float(np.median(ratios)), "max", max(ratios))
    assert float(np.median(ratios)) <= 1.0
    for p in g.parameters():
        assert p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all())


def test_reference_loop_body_under_bf16_autocast_vs_reference_autocast_golden():
    """train.py:90-95 EXACTLY as the reference has it -- inside ``torch.autocast(device_type='cuda', dtype=torch.bfloat16)``
    -- on the drop-in Generator with a stock AdamW: the forward runs the 16-bit step's operators, ``sr`` comes back bf16,
    torch differentiates the reference's own loss expressions, ``loss.backward()`` reaches the HIP backward through
    autograd.GeneratorTrainFn.  Against ONE step of the real reference run the same way (train_step_autocast.npz) and in
    fp32 (train_step.npz): every sampled gradient deviates from the reference's fp32 gradient by no more than 1.5x what
    the reference's own autocast gradient does."""
    from oracle import torch_ref as TR
    from superresolution_amd import Generator, weights as W
    g32 = np.load(os.path.join(GOLDEN, "train_step.npz"))
    gac = np.load(os.path.join(GOLDEN, "train_step_autocast.npz"))
    generator = Generator()
    generator.load_state_dict(W.synthetic_state_dict(0))
    generator = generator.cuda()
    optimizer = torch.optim.AdamW(generator.parameters(), lr=1e-4)
    generator.train()
    lr, hr = _step_batch()
    optimizer.zero_grad()
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        sr = generator(lr)
        loss = _ref_weighted_l1(sr, hr) + 0.1 * (1 - TR.ssim(sr, hr))
    loss.backward()
    optimizer.step()
    assert sr.dtype == torch.bfloat16 and sr.grad_fn is not None
    ref_loss, ac_loss = float(g32["loss"]), float(gac["loss"])
    print("loss: fp32 reference", ref_loss, "reference autocast", ac_loss, "the loop body on the drop-in", loss.item())
    assert abs(loss.item() - ref_loss) <= 2.0 * abs(ac_loss - ref_loss) + 1e-4 * abs(ref_loss)
    P = dict(generator.named_parameters())
    ratios = []
    for k in g32.files:
        if not k.startswith("grad_sample_"):
            continue
        v = P[k[len("grad_sample_"):]].grad.detach().float().flatten()[::7].cpu().numpy()
        ours, theirs = _sample_rel(v, g32[k]), _sample_rel(gac[k], g32[k])
        print(f"  {k}: drop-in {ours:.2e}, reference autocast {theirs:.2e}")
        assert ours <= (1.5 if v.size > 4 else 3.0) * theirs + 2e-3, (k, ours, theirs)
        ratios.append(ours / theirs)
    print("  deviation / the reference autocast's: median", float(np.median(ratios)), "max", max(ratios))
    assert float(np.median(ratios)) <= 1.1


def test_pretrain_steps_bf16_reduce_the_loss_like_the_fp32_step():
    """Twelve optimizer steps on one batch: the bf16 step's loss follows the fp32 step's (same data, same initial weights)."""
    from superresolution_amd import Generator, training as T, weights as W
    lr, hr = _step_batch()
    finals = {}
    for prec in ("fp32", "bf16"):
        g = Generator()
        g.load_state_dict(W.synthetic_state_dict(0))
        g = g.cuda().train()
        step = T.PretrainStep(g, lr=1e-4, precision=prec)
        losses = [float(step.step(lr, hr)) for _ in range(12)]
        finals[prec] = losses
        assert losses[-1] < losses[0]
    print("loss after 12 steps: fp32", finals["fp32"][-1], "bf16", finals["bf16"][-1])
    assert abs(finals["bf16"][-1] - finals["fp32"][-1]) <= 2e-3 * abs(finals["fp32"][-1])


@pytest.mark.parametrize("n,bands", [(1, 0), (3, 1), (3, 2), (2, 3), (2, 5), (3, 8)])
def test_head_bf16_row_streaming_kernel_vs_fp64_and_tile_kernel(n, bands, monkeypatch):
    """The bf16 head (Generator.py:71,80) in its row-streaming form (srgen_head_bf16_stream.hip: bands of rows walked with a
    ring of nine partial output rows) against conv2d in fp64 on the same bf16-rounded operands (fp32 accumulation over
    7,776 products: bar 2e-6 of max) and against the tile kernel it replaces; every band count the launcher can pick
    (band edges in the middle of the image, the last band's rows below the image, one band = no halo)."""
    from superresolution_amd import training as T
    vb, vf = _b(_nhwc(_hash_tensor(61, "hv", (n, 96, 128, 128))).cuda())
    w = (_hash_tensor(62, "hw", (3, 96, 9, 9)) * 0.05).to(torch.bfloat16).float().cuda()
    b = _hash_tensor(63, "hb", (3,)).cuda()
    if bands:
        monkeypatch.setenv("SRGEN_HEAD_BANDS", str(bands))
    y = T.head_forward(vb, w, b)
    monkeypatch.setenv("SRGEN_HEAD_BF16", "tile")
    yt = T.head_forward(vb, w, b)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(vf.cpu().double().permute(0, 3, 1, 2), w.cpu().double(), b.cpu().double(), padding=4)[:, :, :125, :125]
    scale = float(ref.abs().max())
    e = float((y.cpu().double() - ref).abs().max()) / scale
    et = float((yt.cpu().double() - ref).abs().max()) / scale
    print(f"  head bf16 (n {n}, bands {bands}): streaming {e:.2e}, tile {et:.2e} of max vs fp64")
    assert e <= 2e-6 and et <= 2e-6, (e, et)
