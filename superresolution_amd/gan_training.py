"""The GAN step of the reference (``/root/reference/train.py:145-176``) on the HIP path: discriminator forward with saved
activations, its backward pass (data and weight gradients of all ten convs of ``Discriminator.py:17-50``), the gradient of
``RelativisticAverageLoss`` (``Discriminator.py:53-72``), the adversarial gradient's way through the discriminator into the
generator, ``clip_grad_norm_`` and the two AdamW updates.

One GAN iteration of the reference, and what runs here for it (``GanStep.step``):

    fake = generator(lr)                         # no_grad, TRAIN mode (running stats move)   train.py:147-149
    d_loss = D-loss(D(hr), D(fake)); backward; opt_d.step()                                    train.py:151-157
    fake = generator(lr); fake_logits = D(fake)                                                train.py:167-168
    g_loss = w_adv * G-loss(real_logits.detach(), fake_logits) + w_pix * weighted_l1 + w_ssim * (1 - ssim)
    g_loss.backward(); clip_grad_norm_(generator, 10); opt_g.step()                            train.py:169-176

The discriminator has no BatchNorm, so D(hr) and D(fake) of the discriminator step run as ONE batch of 2n images (bit
for bit the same logits, one weight-gradient pass over both).  fp32 arithmetic; the reference runs under bf16 autocast
(train.py:148-166), which fp32 over-satisfies.  Everything here needs ROCm tensors and ``libsrgen.so``: there is no CPU or
autograd fallback.
"""
import ctypes
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from .training import FlatAdamW, PretrainStep, Workspace, _timed, allreduce_mean, conv_precision, packed_weights, pixel_loss_grad, resolve_precision

SLOPE = 0.2                                  # nn.LeakyReLU(0.2), Discriminator.py:10,24,36


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _down(v: int) -> int:
    return (v - 1) // 2 + 1


def disc_convs(d: torch.nn.Module) -> List[torch.nn.Conv2d]:
    """The ten convs of the reference's module tree in forward order (Discriminator.py:22-38)."""
    return [d.backbone[0]] + [d.backbone[k].block[0] for k in range(2, 9)] + [d.head[0], d.head[2]]


def ragan_loss_grad(real: torch.Tensor, fake: torch.Tensor, target_real: float, target_fake: float, scale: float = 1.0,
                    want_real: bool = True) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """(dreal | None, dfake): gradient of ``scale * (bce(real - fake.mean(), t_r) + bce(fake - real.mean(), t_f)) / 2``."""
    if not (real.is_cuda and fake.is_cuda and real.dtype == torch.float32 and fake.dtype == torch.float32):
        raise RuntimeError("ragan_loss_grad: fp32 logits on a ROCm device; there is no CPU fallback")
    r, f = real.detach().contiguous(), fake.detach().contiguous()
    dr = torch.empty_like(r) if want_real else None
    df = torch.empty_like(f)
    with torch.cuda.device(r.device):
        _lib.check(_lib.lib().srgen_ragan_loss_grad(r.data_ptr(), f.data_ptr(), r.numel(), f.numel(), float(target_real),
                                                    float(target_fake), float(scale), dr.data_ptr() if want_real else None,
                                                    df.data_ptr(), _stream(r)))
    return dr, df


def clip_grad_norm_(flat_grad: torch.Tensor, max_norm: float, ws: Workspace) -> torch.Tensor:
    """``torch.nn.utils.clip_grad_norm_`` over one flat fp32 gradient buffer, in place; returns the 0-dim total norm."""
    if not flat_grad.is_cuda or flat_grad.dtype != torch.float32 or not flat_grad.is_contiguous():
        raise RuntimeError("clip_grad_norm_: a contiguous fp32 ROCm buffer; there is no CPU fallback")
    need = ctypes.c_size_t()
    _lib.check(_lib.lib().srgen_clip_grad_norm_scratch_bytes(ctypes.byref(need)))
    scratch = ws.get("clip", need.value)
    out = torch.empty(2, dtype=torch.float32, device=flat_grad.device)
    with torch.cuda.device(flat_grad.device):
        _lib.check(_lib.lib().srgen_clip_grad_norm(flat_grad.data_ptr(), flat_grad.numel(), float(max_norm), out.data_ptr(),
                                                   scratch.data_ptr(), scratch.numel(), _stream(flat_grad)))
    return out[0]


class DiscriminatorGrad:
    """Forward with saved activations and backward of the drop-in ``Discriminator`` (base_ch 64, in_ch 3) from its live
    parameters.  ``forward`` returns (logits (n,1,oh,ow), saved); ``backward`` accumulates into ``p.grad`` of the ten
    convs (``want_wgrad``) and / or returns the gradient w.r.t. the input image (``want_dimg``)."""

    def __init__(self, disc: torch.nn.Module, ws: Workspace):
        if disc.base_ch != 64 or disc.in_ch != 3:
            raise NotImplementedError("the GAN operators are built for Discriminator(3, 64)")
        self.d, self.ws = disc, ws
        self.convs = disc_convs(disc)
        p0 = self.convs[0].weight
        if not p0.is_cuda or p0.dtype != torch.float32:
            raise RuntimeError("DiscriminatorGrad needs the discriminator's fp32 parameters on a ROCm device; there is no CPU fallback")
        self.dev = p0.device
        self.strides = [1, 2, 1, 2, 1, 2, 1, 2, 1, 1]
        # products of the eight 64..512-channel convs (forward and data gradient): "fp32" (exact-fp32 MFMA) or "bf16" (what
        # torch.autocast(bfloat16) gives them at train.py:150,168: operands rounded to bf16 where they are multiplied, fp32
        # accumulation, in the forward, the data gradient and the weight gradient; tensors stay fp32, the stem and the tail
        # keep fp32 products).  A forward remembers its setting for its backward
        self.products = "fp32"
        self._packs = {}

    def _pack(self, conv: torch.nn.Conv2d, mode: int, bf16: bool = False) -> torch.Tensor:
        """Kernel-side image of a conv's live weights (made on the device).  Kept until the parameter changes: the
        discriminator is updated every third iteration only (config.py:21), and a generator half-step packs the same
        weights its discriminator half-step just used.  A change is an in-place update (the version counter every torch
        optimizer and FlatAdamW bump) or a new storage.  Writes that bypass the version counter (``p.data.copy_(...)``,
        raw kernels) need ``invalidate_packs()``."""
        p = conv.weight
        bf16 = bf16 and mode in (0, 1, 4)
        key, tag = (id(conv), mode, bf16), (p._version, p.data_ptr())
        hit = self._packs.get(key)
        if hit is not None and hit[0] == tag:
            return hit[1]
        w = p.detach().contiguous()
        cout, cin = w.shape[0], w.shape[1]
        numel = {0: w.numel(), 1: w.numel(), 2: 7 * 4 * 64, 3: 9 * cin, 4: w.numel()}[mode]
        out = torch.empty(numel, dtype=torch.bfloat16 if bf16 else torch.float32, device=w.device)
        fn = _lib.lib().srdisc_pack_bf16 if bf16 else _lib.lib().srdisc_pack
        with torch.cuda.device(w.device):
            _lib.check(fn(w.data_ptr(), cin, cout, mode, out.data_ptr(), _stream(w)))
        self._packs[key] = (tag, out)
        return out

    def invalidate_packs(self) -> None:
        """Forget every kept weight image (after a weight write torch's version counters do not see)."""
        self._packs.clear()

    # -- forward --------------------------------------------------------------------------------------------------------
    @_timed("discriminator forward")
    def forward(self, x: torch.Tensor):
        if not x.is_cuda or x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"expected an (n, 3, h, w) ROCm tensor, got {tuple(x.shape)} on {x.device}; there is no CPU fallback")
        L = _lib.lib()
        if self.products not in ("fp32", "bf16"):
            raise RuntimeError(f"discriminator products '{self.products}': expected fp32 or bf16")
        b16 = self.products == "bf16"
        conv_fn = L.srdisc_conv_nhwc_bf16p if b16 else L.srdisc_conv_nhwc
        xc = x.detach().to(torch.float32).contiguous()
        n, _, h, w = xc.shape
        acts: List[torch.Tensor] = []
        sizes: List[Tuple[int, int]] = []
        with torch.cuda.device(self.dev):
            s = _stream(xc)
            c0 = self.convs[0]
            a = torch.empty((n, h, w, 64), dtype=torch.float32, device=self.dev)
            _lib.check(L.srdisc_stem_nhwc(xc.data_ptr(), self._pack(c0, 2).data_ptr(), c0.bias.detach().data_ptr(), a.data_ptr(),
                                          n, h, w, SLOPE, s))
            acts.append(a)
            sizes.append((h, w))
            ch, cw = h, w
            for i in range(1, 9):
                conv, st = self.convs[i], self.strides[i]
                cout, cin = conv.weight.shape[0], conv.weight.shape[1]
                oh, ow = (_down(ch), _down(cw)) if st == 2 else (ch, cw)
                o = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=self.dev)
                _lib.check(conv_fn(acts[-1].data_ptr(), self._pack(conv, 0, b16).data_ptr(), conv.bias.detach().data_ptr(),
                                   None, o.data_ptr(), n, ch, cw, cin, cout, st, SLOPE, s))
                acts.append(o)
                ch, cw = oh, ow
                sizes.append((ch, cw))
            c9 = self.convs[9]
            logits = torch.empty((n, 1, ch, cw), dtype=torch.float32, device=self.dev)
            _lib.check(L.srdisc_tail(acts[-1].data_ptr(), self._pack(c9, 3).data_ptr(), c9.bias.detach().data_ptr(),
                                     logits.data_ptr(), n, ch, cw, c9.weight.shape[1], s))
        return logits, {"x": xc, "acts": acts, "sizes": sizes, "bf16": b16}

    # -- backward -------------------------------------------------------------------------------------------------------
    def _grad_views(self, conv) -> Tuple[torch.Tensor, torch.Tensor]:
        for p in (conv.weight, conv.bias):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        return conv.weight.grad, conv.bias.grad

    @_timed("discriminator backward")
    def backward(self, dlogits: torch.Tensor, saved: dict, want_wgrad: bool = True, want_dimg: bool = False,
                 into: Optional[Dict[str, torch.Tensor]] = None) -> Optional[torch.Tensor]:
        """``into``: {state_dict key -> contiguous fp32 tensor} the weight / bias gradients are WRITTEN to (the autograd
        wrapper hands them to torch, which accumulates); default: ADDED to ``p.grad``, as ``loss.backward()`` would."""
        L, ws = _lib.lib(), self.ws
        acts, sizes, xc, b16 = saved["acts"], saved["sizes"], saved["x"], saved.get("bf16", False)
        conv_fn = L.srdisc_conv_nhwc_bf16p if b16 else L.srdisc_conv_nhwc
        dgrad2_fn = L.srdisc_conv_dgrad2_bf16p if b16 else L.srdisc_conv_dgrad2
        n = xc.shape[0]
        dl = dlogits.detach().to(torch.float32).contiguous()
        need = ctypes.c_size_t()
        key_of = {id(p): k for k, p in self.d.named_parameters()} if into is not None else None

        def put(conv, dw, db):                     # accumulate, as autograd does into .grad
            if into is not None:
                into[key_of[id(conv.weight)]].copy_(dw.view_as(conv.weight))
                into[key_of[id(conv.bias)]].copy_(db.view_as(conv.bias))
                return
            gw, gb = self._grad_views(conv)
            gw.add_(dw)
            gb.add_(db)

        with torch.cuda.device(self.dev):
            s = _stream(xc)
            c9 = self.convs[9]
            h8, w8 = sizes[8]
            cin9 = c9.weight.shape[1]
            dpre = torch.empty_like(acts[8])
            dw9 = torch.empty_like(c9.weight) if want_wgrad else None
            db9 = torch.empty(1, dtype=torch.float32, device=self.dev) if want_wgrad else None
            _lib.check(L.srdisc_tail_bwd(dl.data_ptr(), c9.weight.detach().contiguous().data_ptr(), acts[8].data_ptr(), n, h8, w8, cin9,
                                         SLOPE, dpre.data_ptr(), dw9.data_ptr() if want_wgrad else None,
                                         db9.data_ptr() if want_wgrad else None, s))
            if want_wgrad:
                put(c9, dw9, db9)
            for i in range(8, 0, -1):              # dpre = gradient at layer i's pre-activation, (n, oh, ow, cout_i)
                conv, st = self.convs[i], self.strides[i]
                cout, cin = conv.weight.shape[0], conv.weight.shape[1]
                ih, iw = sizes[i - 1]
                oh, ow = sizes[i]
                if want_wgrad:
                    _lib.check(L.srdisc_conv_wgrad_scratch_bytes(n, ih, cin, cout, st, ctypes.byref(need)))
                    scratch = ws.get("disc_wgrad", need.value)
                    dw, db = torch.empty_like(conv.weight), torch.empty_like(conv.bias)
                    wgrad_fn = L.srdisc_conv_wgrad_bf16p if b16 else L.srdisc_conv_wgrad
                    _lib.check(wgrad_fn(acts[i - 1].data_ptr(), dpre.data_ptr(), n, ih, iw, cin, cout, st, dw.data_ptr(),
                                        db.data_ptr(), scratch.data_ptr(), scratch.numel(), s))
                    put(conv, dw, db)
                if i == 1 and not want_wgrad and not want_dimg:
                    break
                nxt = torch.empty((n, ih, iw, cin), dtype=torch.float32, device=self.dev)
                if st == 2:                        # one launch per output parity: only the taps that reach it
                    _lib.check(dgrad2_fn(dpre.data_ptr(), self._pack(conv, 4, b16).data_ptr(), acts[i - 1].data_ptr(),
                                         nxt.data_ptr(), n, ih, iw, cin, cout, SLOPE, s))
                else:
                    _lib.check(conv_fn(dpre.data_ptr(), self._pack(conv, 1, b16).data_ptr(), None, acts[i - 1].data_ptr(),
                                       nxt.data_ptr(), n, ih, iw, cout, cin, 1, SLOPE, s))
                dpre = nxt
            c0 = self.convs[0]
            h0, w0 = sizes[0]
            if want_wgrad:
                _lib.check(L.srdisc_stem_wgrad_scratch_bytes(ctypes.byref(need)))
                scratch = ws.get("disc_stem_wgrad", need.value)
                dw, db = torch.empty_like(c0.weight), torch.empty_like(c0.bias)
                _lib.check(L.srdisc_stem_wgrad(xc.data_ptr(), dpre.data_ptr(), n, h0, w0, dw.data_ptr(), db.data_ptr(),
                                               scratch.data_ptr(), scratch.numel(), s))
                put(c0, dw, db)
            if not want_dimg:
                return None
            dimg = torch.empty_like(xc)
            _lib.check(L.srdisc_stem_dgrad(dpre.data_ptr(), c0.weight.detach().contiguous().data_ptr(), n, h0, w0, dimg.data_ptr(), s))
            return dimg


class GanStep:
    """One GAN iteration of the reference (``train.py:142-182``) for the drop-in ``Generator`` / ``Discriminator`` on one
    GPU; see the module docstring.  ``step(lr, hr, update_d)`` returns the reference's per-iteration statistics as 0-dim
    device tensors: ``d_loss`` (None when the discriminator is not updated), ``g_loss``, ``g_adv``, ``g_pix``, ``g_ssim``.
    Both optimizers are flat AdamW (``opt_g``, ``opt_d``: ``train.py:274-275``); with ``torch.distributed`` initialised and
    world size > 1 each flat gradient buffer is averaged over the ranks with one all-reduce before its update."""

    def __init__(self, generator, discriminator, lr_g: float = 1e-4, lr_d: float = 1e-6, weight_adversarial: float = 0.005,
                 weight_pixel: float = 1.0, weight_ssim: float = 0.3, grad_clip_norm: float = 10.0, real_label: float = 0.9,
                 fake_label: float = 0.1, group=None, precision: str = "auto", reuse_forward: bool = False):
        # precision: the arithmetic of the generator's step, as PretrainStep ("auto": that of the autocast dtype, the way
        # train.py:148-149,166-167 call the generator); with "bf16" the discriminator's 64..512-channel convs take bf16
        # products too (DiscriminatorGrad.products), otherwise they stay on the exact-fp32 MFMA
        self.gen = PretrainStep(generator, lr=lr_g, ssim_weight=weight_ssim, l1_weight=weight_pixel, group=group,
                                precision=precision)
        # reuse_forward: in an iteration that updates the discriminator the reference runs the generator twice on the same
        # batch (train.py:148-149 without a graph, :166-167 with one) and the generator's weights do not change in
        # between, so both calls return the same tensors bit for bit; ``step`` then keeps the first call's output and saved
        # tensors for the generator half-step and only repeats its side effect (the second running-statistics update).
        # Off by default: the two half-steps stay callable on their own and the benchmark times the reference's schedule
        self.reuse_forward = reuse_forward
        self._keep_next, self._kept_forward = False, None           # (only ``step`` pairs the two half-steps)
        self.opt_g = self.gen.optimizer
        self.ws = self.gen.ws
        self.d = discriminator
        self.opt_d = FlatAdamW(discriminator.parameters(), lr=lr_d)
        self.dgrad = DiscriminatorGrad(discriminator, self.ws)
        self.w_adv, self.w_pix, self.w_ssim = weight_adversarial, weight_pixel, weight_ssim
        self.clip, self.real_label, self.fake_label, self.group = grad_clip_norm, real_label, fake_label, group
        self.last_grad_norm = None

    def _allreduce(self, buf: torch.Tensor) -> None:
        allreduce_mean(buf, self.group)

    def _ragan(self, real: torch.Tensor, fake: torch.Tensor) -> torch.Tensor:
        out = torch.empty(2, dtype=torch.float32, device=real.device)
        with torch.cuda.device(real.device):
            _lib.check(_lib.lib().srgen_ragan_loss(real.data_ptr(), fake.data_ptr(), real.numel(), fake.numel(),
                                                   float(self.real_label), float(self.fake_label), out.data_ptr(), _stream(real)))
        return out                                                   # {discriminator_loss, generator_loss}

    # train.py:145-159
    def _set_products(self) -> str:
        prec = resolve_precision(self.gen.precision)
        self.dgrad.products = "bf16" if prec == "bf16" else "fp32"
        return prec

    def d_step(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Updates the discriminator; returns (d_loss, real_logits) -- the logits the generator step of the same iteration
        uses (computed BEFORE the update, as in the reference)."""
        with conv_precision(self._set_products()), packed_weights(self.gen.pack_jobs(backward=False)):
            return self._d_step(lr_img, hr_img)

    def _d_step(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if not self.gen.g.training:
            raise RuntimeError("GanStep runs the generator in train mode (train.py:132): call generator.train()")
        # train-mode forward without a graph (train.py:147-149): batch statistics, running stats move.  Through the step's
        # own operators (device-made weight images) rather than generator(lr), which would re-upload and re-pack every
        # weight on the host after each optimizer step
        fake, S = self.gen.forward(lr_img)
        self.gen._update_running_stats(S, fake.shape[0])
        self._kept_forward = (fake, S) if self._keep_next else None
        self._keep_next = False
        del S
        hr = hr_img.detach().to(torch.float32)
        n = hr.shape[0]
        self.opt_d.restore_grad_views()
        self.opt_d.zero_grad()
        logits, saved = self.dgrad.forward(torch.cat([hr, fake.to(torch.float32)], 0))
        real_logits, fake_logits = logits[:n].contiguous(), logits[n:].contiguous()
        d_loss = self._ragan(real_logits, fake_logits)[0]
        dr, df = ragan_loss_grad(real_logits, fake_logits, self.real_label, self.fake_label)
        self.dgrad.backward(torch.cat([dr, df], 0), saved, want_wgrad=True, want_dimg=False)
        self._allreduce(self.opt_d.flat_grad)
        self.opt_d.step()
        return d_loss, real_logits

    def real_logits(self, hr_img: torch.Tensor) -> torch.Tensor:     # train.py:161-163
        self._set_products()
        logits, _ = self.dgrad.forward(hr_img)
        return logits

    # train.py:165-176
    def g_step(self, lr_img: torch.Tensor, hr_img: torch.Tensor, real_logits: torch.Tensor) -> Dict[str, torch.Tensor]:
        with conv_precision(self._set_products()), packed_weights(self.gen.pack_jobs()):
            return self._g_step(lr_img, hr_img, real_logits)

    def _g_step(self, lr_img: torch.Tensor, hr_img: torch.Tensor, real_logits: torch.Tensor) -> Dict[str, torch.Tensor]:
        from .functional import pixel_losses
        if not self.gen.g.training:
            raise RuntimeError("GanStep runs the generator in train mode (train.py:132): call generator.train()")
        hr = hr_img.detach().to(torch.float32).contiguous()
        self.opt_g.restore_grad_views()
        self.opt_g.zero_grad()
        kept, self._kept_forward = self._kept_forward, None
        sr, S = kept if kept is not None else self.gen.forward(lr_img)
        fake_logits, saved = self.dgrad.forward(sr)
        g_adv = self._ragan(real_logits, fake_logits)[1]
        _, df = ragan_loss_grad(real_logits, fake_logits, 0.0, 1.0, scale=self.w_adv, want_real=False)
        dsr = self.dgrad.backward(df, saved, want_wgrad=False, want_dimg=True)
        vals = pixel_losses(sr, hr)                                  # [psnr, ssim, l1, weighted_l1]
        dsr.add_(pixel_loss_grad(sr, hr, self.ws, self.w_pix, self.w_ssim))
        self.gen.backward(dsr, S)
        self.gen._update_running_stats(S, sr.shape[0])
        self._allreduce(self.opt_g.flat_grad)
        self.last_grad_norm = clip_grad_norm_(self.opt_g.flat_grad, self.clip, self.ws)
        self.opt_g.step()
        g_pix, g_ssim = vals[3], 1.0 - vals[1]
        return {"g_loss": self.w_adv * g_adv + self.w_pix * g_pix + self.w_ssim * g_ssim, "g_adv": g_adv, "g_pix": g_pix,
                "g_ssim": g_ssim}

    def step(self, lr_img: torch.Tensor, hr_img: torch.Tensor, update_d: bool = True) -> Dict[str, Optional[torch.Tensor]]:
        if update_d:
            self._keep_next = self.reuse_forward
            d_loss, real_logits = self.d_step(lr_img, hr_img)
        else:
            d_loss, real_logits = None, self.real_logits(hr_img)
        try:
            out = self.g_step(lr_img, hr_img, real_logits)
        finally:
            self._kept_forward = None
        out["d_loss"] = d_loss
        return out
