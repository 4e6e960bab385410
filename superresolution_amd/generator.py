"""Drop-in ``Generator`` whose eval forward runs on hand-written gfx950 kernels.

Mirrors the reference's public surface for this path (``/root/reference/Generator.py:45-80``
and its call sites ``train.py:37,65,247,250,268``): same constructor, same module tree and
therefore the same 266 ``state_dict`` keys / parameter shapes / default initialisation
order, ``.to() / .eval() / .train() / .parameters()``, and ``generator(x)`` returning
``(N, 3, 125, 125)``.

The sub-modules are stock ``torch.nn`` layers used purely as parameter containers; nothing
here ever calls their ``forward``.  ``Generator.forward`` hands the input, an output tensor
and a workspace to ``libsrgen.so`` (``include/srgen.h``) on the current HIP stream.  There
is no eager-PyTorch or CPU fallback: a CPU tensor or a missing extension raises.

Train mode (``train.py:92,149,167``) runs the HIP training operators of ``superresolution_amd.training``: batch-statistics
BatchNorm, running-statistics update, and -- with gradient recording on -- a ``torch.autograd.Function``
(``superresolution_amd.autograd``) whose backward fills ``p.grad`` through the HIP backward operators, so the reference's
``loss.backward(); optimizer.step()`` runs unchanged.  What is not built raises instead of silently yielding no gradient:
the gradient w.r.t. the INPUT image (train or eval mode; the reference never asks for it), and parameter gradients of
the eval-mode forward (folded BatchNorm; the reference never trains in eval mode).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib


class _Container(nn.Module):
    """Parameter container; the fused HIP forward never dispatches through sub-modules."""

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(f"{type(self).__name__} is a parameter container of the fused HIP Generator; "
                           "call the Generator itself")


class UpsampleBlock(_Container):
    # keys: conv.{weight,bias}, act.weight                       (Generator.py:4-12)
    def __init__(self, in_ch: int, out_ch: int, scale: int = 2):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, out_ch * scale * scale, kernel_size=3, padding=1)
        self.shuffle = nn.PixelShuffle(scale)
        self.act = nn.PReLU()


class ChannelAttention(_Container):
    # keys: fc.0.{weight,bias}, fc.2.{weight,bias}                 (Generator.py:15-27)
    def __init__(self, ch: int, reduction: int = 8):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Conv2d(ch, ch // reduction, 1), nn.ReLU(inplace=True),
                                nn.Conv2d(ch // reduction, ch, 1), nn.Sigmoid())


class ResBlock(_Container):
    # keys: block.{0,3}.{weight,bias}, block.{1,4}.<BN>, block.2.weight  (Generator.py:30-42)
    def __init__(self, ch: int):
        super().__init__()
        self.block = nn.Sequential(nn.Conv2d(ch, ch, 3, padding=1), nn.BatchNorm2d(ch), nn.PReLU(ch),
                                   nn.Conv2d(ch, ch, 3, padding=1), nn.BatchNorm2d(ch))


class Generator(nn.Module):
    """SRGAN generator, 3x64x64 -> 3x125x125 (``Generator.py:45-80``), HIP-backed in eval mode."""

    LR, OUT = 64, 125

    def __init__(self, num_blocks: int = 16, in_ch: int = 3, base_ch: int = 96):
        super().__init__()
        self.num_blocks, self.in_ch, self.base_ch = num_blocks, in_ch, base_ch
        # construction order == the reference's, so torch.manual_seed(s); Generator() yields the
        # same default-initialised parameters as the reference does.
        self.input = nn.Sequential(nn.Conv2d(in_ch, base_ch, 9, padding=4), nn.PReLU())
        blocks = []
        for i in range(num_blocks):
            blocks.append(ResBlock(base_ch))
            if i in (7, 15):
                blocks.append(ChannelAttention(base_ch))
        self.blocks = nn.Sequential(*blocks)
        self.mid = nn.Sequential(nn.Conv2d(base_ch, base_ch, 3, padding=1), nn.BatchNorm2d(base_ch))
        self.upsample = UpsampleBlock(base_ch, base_ch)
        self.output = nn.Sequential(nn.Conv2d(base_ch, base_ch, 3, padding=1), nn.PReLU(),
                                    nn.Conv2d(base_ch, in_ch, 9, padding=4))
        # HIP-side state (not part of state_dict)
        self._handles: Dict[int, int] = {}            # device index -> srgen_handle*
        self._weights_sig: Dict[int, Tuple] = {}       # device index -> signature of uploaded weights
        self._workspace: Dict[Tuple[int, int], torch.Tensor] = {}
        self._engines: Dict[int, object] = {}          # device index -> training.GeneratorGrad (train-mode operators)
        # arithmetic of the train-mode forward / backward: "auto" = that of the autocast dtype inside torch.autocast ("bf16":
        # bf16 tensors and products, how train.py:91,148,166 call the generator; "f16" for float16), fp32 otherwise; or one
        # of "fp32" / "f16" (fp32 tensors, 16-bit conv products) / "bf16" (training.conv_precision)
        self.train_precision = "auto"

    # ---- HIP handle / weights ----------------------------------------------------------
    def _signature(self) -> Tuple:
        """(parameters, buffers): the train-mode forward reads no BatchNorm running statistics, so its own
        updates of those buffers must not force a re-upload before the next train-mode call."""
        return (tuple((t.data_ptr(), t._version) for t in self.parameters()),
                tuple((t.data_ptr(), t._version) for t in self.buffers()))

    def _handle(self, device: torch.device, train: bool = False) -> int:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        L = _lib.lib()
        if idx not in self._handles:
            cfg = _lib.srgen_config(self.num_blocks, self.in_ch, self.base_ch)
            h = ctypes.c_void_p()
            with torch.cuda.device(idx):
                _lib.check(L.srgen_create(ctypes.byref(cfg), ctypes.byref(h)))
            self._handles[idx] = h.value
        h = self._handles[idx]
        sig = self._signature()
        have = self._weights_sig.get(idx)
        if have != sig and not (train and have is not None and have[0] == sig[0]):
            self._upload(h, idx)
            self._weights_sig[idx] = sig
        return h

    def _upload(self, h: int, idx: int) -> None:
        """load_state_dict -> library: every floating-point entry by its reference key."""
        L = _lib.lib()
        keep = []
        for key, t in self.state_dict().items():
            if not t.is_floating_point():
                continue                                   # num_batches_tracked
            c = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
            keep.append(c)
            _lib.check(L.srgen_set_tensor(h, key.encode(), c.data_ptr(), c.numel()))
        with torch.cuda.device(idx):
            stream = torch.cuda.current_stream(idx).cuda_stream
            _lib.check(L.srgen_commit_weights(h, stream))

    def _ws(self, h: int, idx: int, n: int, dtype_code: int) -> torch.Tensor:
        need = ctypes.c_size_t()
        _lib.check(_lib.lib().srgen_workspace_bytes(h, n, dtype_code, ctypes.byref(need)))
        key = (idx, dtype_code)
        ws = self._workspace.get(key)
        if ws is None or ws.numel() < need.value:
            ws = None
            self._workspace.pop(key, None)
            ws = torch.empty(need.value, dtype=torch.uint8, device=torch.device("cuda", idx))
            self._workspace[key] = ws
        return ws

    # ---- forward -------------------------------------------------------------------------
    def _prepare(self, x: torch.Tensor):
        if torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.requires_grad:
            raise NotImplementedError(
                "Generator.forward with an input that requires grad: the gradient w.r.t. the input image is not built "
                "(the reference never asks for it: lr comes from the data loader, train.py:88,143), and silently giving "
                "none would be wrong. Parameter gradients come from .train() mode (superresolution_amd/autograd.py); "
                "otherwise detach the input or call under torch.no_grad().")
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise RuntimeError(f"Expected 4D (N, {self.in_ch}, 64, 64) input, got {tuple(getattr(x, 'shape', ()))}")
        if x.shape[1] != self.in_ch:
            raise RuntimeError(f"Given groups=1, weight of size [{self.base_ch}, {self.in_ch}, 9, 9], expected "
                               f"input{list(x.shape)} to have {self.in_ch} channels, but got {x.shape[1]} "
                               "channels instead")
        if x.shape[2] != self.LR or x.shape[3] != self.LR:
            raise RuntimeError(f"the HIP generator is built for {self.LR}x{self.LR} inputs, got "
                               f"{x.shape[2]}x{x.shape[3]}")
        if not x.is_cuda:
            raise RuntimeError("the HIP generator runs on a ROCm device only (input is on "
                               f"{x.device}); there is no CPU fallback. Move the module and input to 'cuda'.")
        p = self.input[0].weight
        if p.device != x.device:
            raise RuntimeError(f"Input type ({x.device}) and weight type ({p.device}) should be the same")
        bf16 = x.dtype == torch.bfloat16 or (
            torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        if not bf16 and x.dtype != torch.float32:
            raise RuntimeError(f"unsupported input dtype {x.dtype}: fp32, or bf16 / bf16 autocast")
        dt = torch.bfloat16 if bf16 else torch.float32
        return x.detach().to(dt).contiguous(), (_lib.SRGEN_BF16 if bf16 else _lib.SRGEN_F32), dt

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            if (self.num_blocks, self.in_ch, self.base_ch) == (16, 3, 96):
                return self._run_train_ops(x)
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                raise NotImplementedError(
                    "Generator.forward in train mode under autograd: the backward operators are built for the reference's "
                    "Generator(num_blocks=16, in_ch=3, base_ch=96) only; other shapes run train-mode forward VALUES under "
                    "torch.no_grad(). There is no eager fallback.")
            return self._run_train(x)
        return self._run(x, None)

    def _engine(self, device: torch.device):
        from . import training as T
        idx = device.index if device.index is not None else torch.cuda.current_device()
        e = self._engines.get(idx)
        if e is None:
            e = self._engines[idx] = T.GeneratorGrad(self)
        return e

    def _run_train_ops(self, x: torch.Tensor) -> torch.Tensor:
        """Train-mode forward on the training operators (``training.GeneratorGrad``; the reference's call sites
        train.py:92,149,167): batch-statistics BatchNorm from the LIVE parameters (weight images are made on the device, so
        an optimizer step costs no host re-upload), running statistics updated as ``nn.BatchNorm2d`` does.  With gradient
        recording on, the result carries a graph whose backward is the HIP backward pass (``autograd.GeneratorTrainFn``).
        Under ``torch.autocast(bfloat16)`` (or with a bf16 input) the step runs on bf16 tensors with bf16 products
        (``train_precision``; the output image is computed in fp32 and rounded to bf16 once, as the reference's autocast
        returns it); otherwise tensors are fp32."""
        from . import autograd as A, training as T
        xc, code, dt = self._prepare(x)
        if xc.shape[0] == 0:
            raise ValueError("Expected more than 1 value per channel when training, got an empty batch")
        if self.train_precision not in ("auto", "fp32", "f16", "bf16"):
            raise RuntimeError(f"train_precision '{self.train_precision}': expected auto, fp32, f16 or bf16")
        precision = T.resolve_precision(self.train_precision)
        if self.train_precision == "auto" and code != _lib.SRGEN_F32 and T.conv_path() == "split":
            precision = "bf16"                                 # a bf16 input outside autocast: same 16-bit arithmetic
        engine = self._engine(xc.device)
        with torch.cuda.device(xc.device):
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                y = A.generator_train_forward(self, engine, xc.float(), precision)
            else:
                with T.conv_precision(precision), T.packed_weights(engine.pack_jobs(backward=False)):
                    y, S = engine.forward(xc.float())
                engine._update_running_stats(S, y.shape[0])
                del S
        return y if dt == torch.float32 else y.to(dt)

    def _bn_modules(self):
        """The BatchNorm2d layers in module order (= the order of srgen_forward_train's bn_batch_stats)."""
        return [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]

    def _run_train(self, x: torch.Tensor) -> torch.Tensor:
        """Train-mode forward values (``Generator.train()(x)`` without autograd): batch-statistics BatchNorm
        in the kernels, then the running-statistics update of ``nn.BatchNorm2d`` (momentum 0.1, unbiased
        variance, ``num_batches_tracked += 1``) on this module's buffers.

        The kernels are fp32.  Under ``torch.autocast(bfloat16)`` -- the only way the reference calls the
        train-mode forward (``train.py:91-92,148-149,166-167``) -- or with a bf16 input, the arithmetic still runs
        in fp32 on ``x.float()`` and the result is rounded to bf16 once (the convention of ``Discriminator._run``):
        stricter than autocast, whose convs round every layer; the statistics update stays fp32, as
        ``nn.BatchNorm2d`` keeps it under autocast."""
        xc, code, dt = self._prepare(x)
        out_dt = dt
        if code != _lib.SRGEN_F32:
            xc, code, dt = xc.float(), _lib.SRGEN_F32, torch.float32
        n, idx = xc.shape[0], xc.device.index
        y = torch.empty((n, self.in_ch, self.OUT, self.OUT), dtype=dt, device=xc.device)
        if n == 0:
            raise ValueError("Expected more than 1 value per channel when training, got an empty batch")
        h = self._handle(xc.device, train=True)
        ws = self._ws(h, idx, n, code)
        bns = self._bn_modules()
        stats = torch.empty((len(bns), 2, self.base_ch), dtype=torch.float32, device=xc.device)
        with torch.cuda.device(idx):
            _lib.check(_lib.lib().srgen_forward_train(h, xc.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), n,
                                                      stats.data_ptr(), torch.cuda.current_stream(idx).cuda_stream))
        m = n * self.LR * self.LR
        with torch.no_grad():                              # nn.BatchNorm2d's update, batched over the layers
            live = [(k, bn) for k, bn in enumerate(bns) if bn.track_running_stats and bn.running_mean is not None]
            mom = 0.1                                       # Generator.py builds every BatchNorm2d with the default
            for k, bn in live:
                if bn.momentum != mom:
                    raise NotImplementedError(f"BatchNorm2d momentum {bn.momentum}: only the default 0.1 is built")
            means, varis = [bn.running_mean for _, bn in live], [bn.running_var for _, bn in live]
            torch._foreach_mul_(means, 1 - mom)
            torch._foreach_add_(means, [stats[k, 0] for k, _ in live], alpha=mom)
            torch._foreach_mul_(varis, 1 - mom)
            torch._foreach_add_(varis, [stats[k, 1] for k, _ in live], alpha=mom * m / max(m - 1, 1))
            torch._foreach_add_([bn.num_batches_tracked for _, bn in live], 1)
        return y if out_dt == dt else y.to(out_dt)

    def forward_taps(self, x: torch.Tensor):
        """(y, {tap name: NCHW fp32 tensor}) -- layer taps for parity tests (srgen_forward_taps)."""
        taps: Dict[str, torch.Tensor] = {}
        y = self._run(x, taps)
        return y, taps

    def _run(self, x: torch.Tensor, taps: Optional[dict]) -> torch.Tensor:
        xc, code, dt = self._prepare(x)
        n = xc.shape[0]
        idx = xc.device.index
        y = torch.empty((n, self.in_ch, self.OUT, self.OUT), dtype=dt, device=xc.device)
        if n == 0:
            return y
        h = self._handle(xc.device)
        ws = self._ws(h, idx, n, code)
        L = _lib.lib()
        with torch.cuda.device(idx):
            stream = torch.cuda.current_stream(idx).cuda_stream
            if taps is None:
                _lib.check(L.srgen_forward(h, xc.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), n, code,
                                           stream))
            else:
                C = self.base_ch
                shapes = [(n, C, 64, 64)] * 7 + [(n, C, 128, 128)] * 2 + [(n, C)] * 2
                bufs = [torch.empty(s, dtype=torch.float32, device=xc.device) for s in shapes]
                arr = (ctypes.c_void_p * _lib.NUM_TAPS)(*[b.data_ptr() for b in bufs])
                _lib.check(L.srgen_forward_taps(h, xc.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), n,
                                                code, arr, stream))
                taps.update(dict(zip(_lib.TAP_NAMES, bufs)))
        return y

    def profile_forward(self, x: torch.Tensor):
        """One forward with a HIP event pair around every launch (srgen_forward_profile).
        Returns {kernel class: (summed device ms, launches)}.  Synchronises the stream."""
        xc, code, dt = self._prepare(x)
        n, idx = xc.shape[0], xc.device.index
        y = torch.empty((n, self.in_ch, self.OUT, self.OUT), dtype=dt, device=xc.device)
        h = self._handle(xc.device)
        ws = self._ws(h, idx, n, code)
        ms = (ctypes.c_float * len(_lib.KCLASS_NAMES))()
        cnt = (ctypes.c_int * len(_lib.KCLASS_NAMES))()
        with torch.cuda.device(idx):
            stream = torch.cuda.current_stream(idx).cuda_stream
            _lib.check(_lib.lib().srgen_forward_profile(h, xc.data_ptr(), y.data_ptr(), ws.data_ptr(),
                                                        ws.numel(), n, code, stream, ms, cnt))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(_lib.KCLASS_NAMES)}

    # ---- module protocol -------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._weights_sig.clear()          # parameters moved / cast: re-upload on next forward
        self._workspace.clear()
        self._engines.clear()
        return out

    def __getstate__(self):
        # pickling / deepcopy must not share (and later double-free) the library handles
        state = self.__dict__.copy()
        state["_handles"], state["_weights_sig"], state["_workspace"], state["_engines"] = {}, {}, {}, {}
        return state

    def __del__(self):
        try:
            L = _lib.lib()
            for h in self.__dict__.get("_handles", {}).values():
                L.srgen_destroy(h)
        except Exception:
            pass
