"""Drop-in ``Discriminator`` and ``RelativisticAverageLoss`` (forward values) on gfx950 kernels.

Mirrors ``/root/reference/Discriminator.py:5-72`` and its call sites in the GAN phase
(``train.py:152-154,168-170``): same constructors, same module tree and therefore the same 20
``state_dict`` keys, parameter shapes and initialisation (default ``nn.Conv2d`` init followed by
``kaiming_normal_(a=0.2)`` in ``modules()`` order, so ``torch.manual_seed(s); Discriminator()``
yields the reference's weights), ``discriminator(x)`` returning ``(N, 1, 8, 8)`` logits for
``125 x 125`` inputs, and ``loss.discriminator_loss(real, fake)`` / ``loss.generator_loss(real, fake)``
returning 0-dim tensors.

Under ``torch.no_grad()`` the calls run the inference kernels (``srdisc_forward``).  With gradient recording on and
parameters or inputs that require grad -- how the reference calls them (``train.py:152-156,168-174``) -- they return
tensors whose graph is a ``torch.autograd.Function`` over the HIP backward operators (``superresolution_amd.autograd``:
``DiscriminatorFn`` = ``gan_training.DiscriminatorGrad``, ``RaganLossFn`` = ``ragan_loss_grad``), so ``d_loss.backward()``
and ``g_loss.backward()`` fill ``p.grad`` and carry the adversarial gradient into the generator.  Built for the reference's
``Discriminator(in_ch=3, base_ch=64)``; other widths raise under autograd.  No eager-PyTorch or CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib


class DiscBlock(nn.Module):
    # keys: block.0.{weight,bias}                                        (Discriminator.py:5-14)
    def __init__(self, in_ch: int, out_ch: int, stride: int = 1):
        super().__init__()
        self.block = nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, stride=stride, padding=1),
                                   nn.LeakyReLU(0.2, inplace=True))

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("DiscBlock is a parameter container of the fused HIP Discriminator; call the "
                           "Discriminator itself")


def _wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t.requires_grad for t in tensors)


def ragan_both(r: torch.Tensor, f: torch.Tensor, real_label: float, fake_label: float) -> torch.Tensor:
    """[discriminator_loss, generator_loss] of contiguous fp32 ROCm logits, one launch, result left on the device."""
    out = torch.empty(2, dtype=torch.float32, device=r.device)
    if r.numel() == 0 or f.numel() == 0:
        return out.fill_(float("nan"))                      # mean of an empty tensor, as torch gives
    idx = r.device.index
    with torch.cuda.device(idx):
        _lib.check(_lib.lib().srgen_ragan_loss(r.data_ptr(), f.data_ptr(), r.numel(), f.numel(), float(real_label),
                                               float(fake_label), out.data_ptr(), torch.cuda.current_stream(idx).cuda_stream))
    return out


class Discriminator(nn.Module):
    """PatchGAN critic 3xHxW -> 1x(H/16)x(W/16) logits (``Discriminator.py:17-50``), HIP-backed."""

    def __init__(self, in_ch: int = 3, base_ch: int = 64):
        super().__init__()
        self.in_ch, self.base_ch = in_ch, base_ch
        f = base_ch
        self.backbone = nn.Sequential(
            nn.Conv2d(in_ch, f, 3, padding=1), nn.LeakyReLU(0.2, inplace=True),
            DiscBlock(f, f, stride=2), DiscBlock(f, f * 2),
            DiscBlock(f * 2, f * 2, stride=2), DiscBlock(f * 2, f * 4),
            DiscBlock(f * 4, f * 4, stride=2), DiscBlock(f * 4, f * 8),
            DiscBlock(f * 8, f * 8, stride=2))
        self.head = nn.Sequential(nn.Conv2d(f * 8, f * 4, 3, padding=1), nn.LeakyReLU(0.2, inplace=True),
                                  nn.Conv2d(f * 4, 1, 3, padding=1))
        self._init_weights()
        self._handles: Dict[int, int] = {}
        self._weights_sig: Dict[int, Tuple] = {}
        self._workspace: Dict[int, torch.Tensor] = {}
        self._dgrads: Dict[int, object] = {}            # device index -> gan_training.DiscriminatorGrad
        # products of the 64..512-channel convs under autograd: "auto" = bf16 products inside torch.autocast(bfloat16) (or
        # with a bf16 input), the way train.py:150,168 call the discriminator, fp32 otherwise; or "fp32" / "bf16"
        self.train_products = "auto"

    def _init_weights(self) -> None:                                      # Discriminator.py:42-47
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0.2, nonlinearity="leaky_relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    # ---- HIP handle / weights ----------------------------------------------------------
    def _signature(self) -> Tuple:
        return tuple((t.data_ptr(), t._version) for t in self.state_dict(keep_vars=True).values())

    def _handle(self, idx: int) -> int:
        L = _lib.lib()
        if idx not in self._handles:
            cfg = _lib.srdisc_config(self.in_ch, self.base_ch)
            h = ctypes.c_void_p()
            with torch.cuda.device(idx):
                _lib.check(L.srdisc_create(ctypes.byref(cfg), ctypes.byref(h)), "Discriminator")
            self._handles[idx] = h.value
        h = self._handles[idx]
        sig = self._signature()
        if self._weights_sig.get(idx) != sig:
            keep = []
            for key, t in self.state_dict().items():
                c = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
                keep.append(c)
                _lib.check(L.srdisc_set_tensor(h, key.encode(), c.data_ptr(), c.numel()), "Discriminator")
            with torch.cuda.device(idx):
                _lib.check(L.srdisc_commit_weights(h, torch.cuda.current_stream(idx).cuda_stream), "Discriminator")
            self._weights_sig[idx] = sig
        return h

    @staticmethod
    def out_size(v: int) -> int:
        """Side after the four stride-2 convs (3x3, pad 1): 125 -> 63 -> 32 -> 16 -> 8."""
        for _ in range(4):
            v = (v - 1) // 2 + 1
        return v

    # ---- forward -------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, None)

    def forward_taps(self, x: torch.Tensor):
        """(logits, [NCHW fp32 output of each of the ten convs after its LeakyReLU]) for parity tests."""
        taps: list = []
        y = self._run(x, taps)
        return y, taps

    def _run(self, x: torch.Tensor, taps: Optional[list]) -> torch.Tensor:
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise RuntimeError(f"Expected 4D (N, {self.in_ch}, H, W) input, got {tuple(getattr(x, 'shape', ()))}")
        if x.shape[1] != self.in_ch:
            raise RuntimeError(f"Given groups=1, weight of size [{self.base_ch}, {self.in_ch}, 3, 3], expected "
                               f"input{list(x.shape)} to have {self.in_ch} channels, but got {x.shape[1]} "
                               "channels instead")
        if not x.is_cuda:
            raise RuntimeError("the HIP discriminator runs on a ROCm device only (input is on "
                               f"{x.device}); there is no CPU fallback. Move the module and input to 'cuda'.")
        p = self.backbone[0].weight
        if p.device != x.device:
            raise RuntimeError(f"Input type ({x.device}) and weight type ({p.device}) should be the same")
        # the reference runs D under torch.autocast(bfloat16) (train.py:150,166): bf16 inputs / bf16 autocast are
        # accepted; tensors stay fp32 and the returned logits are rounded.  Forward values (no graph) are computed with
        # fp32 products; under autograd ``train_products`` decides (bf16 products in this case by default)
        bf16_out = x.dtype == torch.bfloat16 or (
            torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError(f"unsupported input dtype {x.dtype}: fp32, or bf16 / bf16 autocast (computed in fp32)")
        if taps is None and _wants_grad(x, *self.parameters()):
            return self._run_autograd(x, bf16_out)
        xc = x.detach().float().contiguous()
        n, _, hgt, wid = xc.shape
        idx = xc.device.index
        oh, ow = self.out_size(hgt), self.out_size(wid)
        y = torch.empty((n, 1, oh, ow), dtype=torch.float32, device=xc.device)
        if n == 0:
            return y.to(torch.bfloat16) if bf16_out else y
        h = self._handle(idx)
        L = _lib.lib()
        need = ctypes.c_size_t()
        _lib.check(L.srdisc_workspace_bytes(h, n, hgt, wid, ctypes.byref(need)), "Discriminator")
        ws = self._workspace.get(idx)
        if ws is None or ws.numel() < need.value:
            self._workspace.pop(idx, None)
            ws = None
            ws = torch.empty(need.value, dtype=torch.uint8, device=xc.device)
            self._workspace[idx] = ws
        arr = None
        if taps is not None:
            f, sh, sw = self.base_ch, [hgt], [wid]
            for _ in range(4):
                sh.append((sh[-1] - 1) // 2 + 1)
                sw.append((sw[-1] - 1) // 2 + 1)
            chans = [f, f, 2 * f, 2 * f, 4 * f, 4 * f, 8 * f, 8 * f, 4 * f, 1]
            level = [0, 1, 1, 2, 2, 3, 3, 4, 4, 4]
            taps.extend(torch.empty((n, c, sh[l], sw[l]), dtype=torch.float32, device=xc.device)
                        for c, l in zip(chans, level))
            arr = (ctypes.c_void_p * _lib.DISC_NUM_TAPS)(*[t.data_ptr() for t in taps])
        with torch.cuda.device(idx):
            stream = torch.cuda.current_stream(idx).cuda_stream
            _lib.check(L.srdisc_forward(h, xc.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), n, hgt, wid,
                                        arr, stream), "Discriminator")
        return y.to(torch.bfloat16) if bf16_out else y

    def _run_autograd(self, x: torch.Tensor, bf16_out: bool) -> torch.Tensor:
        """Logits with a graph (train.py:152-153,168): forward with saved activations from the live parameters,
        ``autograd.DiscriminatorFn`` behind it."""
        from . import autograd as A
        from .gan_training import DiscriminatorGrad
        from .training import Workspace
        if self.base_ch != 64 or self.in_ch != 3:
            raise NotImplementedError("Discriminator.forward under autograd: the backward operators are built for the "
                                      "reference's Discriminator(in_ch=3, base_ch=64); other widths run forward values "
                                      "under torch.no_grad(). There is no eager fallback.")
        idx = x.device.index
        dg = self._dgrads.get(idx)
        if dg is None:
            dg = self._dgrads[idx] = DiscriminatorGrad(self, Workspace(x.device))
        names = [k for k, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        if self.train_products not in ("auto", "fp32", "bf16"):
            raise RuntimeError(f"train_products '{self.train_products}': expected auto, fp32 or bf16")
        dg.products = ("bf16" if bf16_out else "fp32") if self.train_products == "auto" else self.train_products
        with torch.cuda.device(idx):
            y = A.DiscriminatorFn.apply(dg, names, x if x.dtype == torch.float32 else x.float(), *params)
        return y.to(torch.bfloat16) if bf16_out else y

    # ---- module protocol -------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._weights_sig.clear()
        self._workspace.clear()
        self._dgrads.clear()
        return out

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_handles"], state["_weights_sig"], state["_workspace"], state["_dgrads"] = {}, {}, {}, {}
        return state

    def __del__(self):
        try:
            L = _lib.lib()
            for h in self.__dict__.get("_handles", {}).values():
                L.srdisc_destroy(h)
        except Exception:
            pass


class RelativisticAverageLoss(nn.Module):
    """``Discriminator.py:53-72``: BCE-with-logits on ``real - fake.mean()`` / ``fake - real.mean()``.

    Both losses come out of one single-launch kernel (``srgen_ragan_loss``); each method returns its
    0-dim fp32 tensor, left on the device (no host sync)."""

    def __init__(self, real_label: float = 0.9, fake_label: float = 0.1):
        super().__init__()
        self.real_label = real_label
        self.fake_label = fake_label

    @staticmethod
    def _check(real: torch.Tensor, fake: torch.Tensor) -> None:
        for nm, t in (("real", real), ("fake", fake)):
            if not isinstance(t, torch.Tensor) or not t.is_cuda:
                raise RuntimeError(f"RelativisticAverageLoss: `{nm}` must be a tensor on a ROCm device; there is "
                                   "no CPU fallback")
            if t.dtype not in (torch.float32, torch.bfloat16):
                raise RuntimeError(f"RelativisticAverageLoss: `{nm}` is {t.dtype}; fp32 or bf16 logits (computed in fp32)")
        if real.device != fake.device:
            raise RuntimeError(f"real is on {real.device}, fake on {fake.device}")

    def _both(self, real: torch.Tensor, fake: torch.Tensor) -> torch.Tensor:
        self._check(real, fake)
        r, f = real.detach().float().contiguous(), fake.detach().float().contiguous()   # BCEWithLogits autocasts to fp32
        return ragan_both(r, f, self.real_label, self.fake_label)

    def _one(self, real: torch.Tensor, fake: torch.Tensor, which: int) -> torch.Tensor:
        if _wants_grad(real, fake):                         # train.py:154-156,169-174: the loss is differentiated
            from . import autograd as A
            self._check(real, fake)
            if real.numel() == 0 or fake.numel() == 0:
                raise RuntimeError("RelativisticAverageLoss under autograd needs non-empty logits")
            return A.RaganLossFn.apply(real, fake, float(self.real_label), float(self.fake_label), which)
        return self._both(real, fake)[which]

    def discriminator_loss(self, real: torch.Tensor, fake: torch.Tensor) -> torch.Tensor:
        return self._one(real, fake, 0)

    def generator_loss(self, real: torch.Tensor, fake: torch.Tensor) -> torch.Tensor:
        return self._one(real, fake, 1)

    def both_losses(self, real: torch.Tensor, fake: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(discriminator_loss, generator_loss) from one launch (no counterpart in the reference); forward values."""
        o = self._both(real, fake)
        return o[0], o[1]
