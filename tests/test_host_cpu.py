"""CPU suite: host-side logic -- the C-ABI library loads and exports every symbol the header
declares (no compute without a GPU), the drop-in module protocol, loud failure off-GPU, and the
determinism of the synthetic weight generator."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from superresolution_amd import Generator, _lib, weights as W


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "srgen.h")).read()
    return sorted(set(re.findall(r"SRGEN_API\s+[\w\s\*]+?\b(srgen_\w+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    syms = _declared_symbols()
    assert len(syms) >= 13, syms
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), f"libsrgen.so does not export {s}"
    assert _lib.lib().srgen_abi_version() == 1


def test_argument_errors_without_gpu():
    """Error paths that return before any HIP call."""
    L = _lib.lib()
    need = ctypes.c_size_t()
    assert L.srgen_workspace_bytes(None, 1, 0, ctypes.byref(need)) == -1
    assert b"NULL" in L.srgen_last_error()
    assert L.srgen_set_tensor(None, b"input.0.weight", None, 0) == -1
    assert L.srgen_forward(None, None, None, None, 0, 1, 0, None) == -1
    assert L.srgen_metrics_scratch_bytes(0, 3, 125, 125, ctypes.byref(need)) == -1
    assert L.srgen_metrics_scratch_bytes(2, 3, 125, 125, ctypes.byref(need)) == 0 and need.value > 0
    cfg = _lib.srgen_config(16, 3, 64)
    h = ctypes.c_void_p()
    assert L.srgen_create(ctypes.byref(cfg), ctypes.byref(h)) == -6 and h.value is None
    assert b"base_ch=96" in L.srgen_last_error()
    # training operators: argument checks come before any HIP call too
    assert L.srgen_split_pack_many(0, None, None, None, None, None, None) == -1
    assert b"count" in L.srgen_last_error()
    assert L.srgen_split_pack_many(97, None, None, None, None, None, None) == -1
    assert L.srgen_split_pack_many(1, None, None, None, None, None, None) == -1 and b"NULL" in L.srgen_last_error()
    assert L.srgen_conv3x3_nhwc_f16(None, None, None, None, None, 1, 64, 1, 0, None, None) == -1
    assert L.srgen_conv3x3_wgrad_f16(None, None, 1, 64, None, None, None, None, 0, None) == -1
    assert L.srgen_bf16_pack_many(0, None, None, None, None, None, None) == -1 and b"count" in L.srgen_last_error()
    assert L.srgen_conv3x3_nhwc_bf16(None, None, None, None, None, 1, 64, 1, 0, None) == -1
    assert L.srgen_conv3x3_wgrad_bf16(None, None, 1, 64, None, None, None, 0, None) == -1
    assert L.srgen_bn_train_bwd_bf16(None, None, None, None, None, None, 1.0, None, None, None, None, 0, None, 0, None) == -1


def test_conv_precision_switch_on_the_host():
    """training.conv_precision / resolve_precision: names are checked, the state is restored on exit and on error; off
    the GPU there is no autocast for ROCm devices, so "auto" is fp32."""
    from superresolution_amd import training as T
    assert T.resolve_precision("auto") == "fp32" and T.resolve_precision("f16") == "f16"
    with T.conv_precision("f16"):
        assert T._CONV_PRECISION[0] == "f16"
        with pytest.raises(RuntimeError, match="expected fp32, f16 or bf16"):
            with T.conv_precision("fp8"):
                pass
        assert T._CONV_PRECISION[0] == "f16"
        with T.conv_precision("bf16"):
            assert T._CONV_PRECISION[0] == "bf16"
        assert T._CONV_PRECISION[0] == "f16"
    assert T._CONV_PRECISION[0] == "fp32"
    with pytest.raises(RuntimeError, match="precision"):
        T.PretrainStep.__init__(object.__new__(T.PretrainStep), None, precision="half")


def test_state_dict_surface_matches_reference_contract():
    g = Generator()
    sd = g.state_dict()
    assert len(sd) == 266
    assert list(sd.keys()) == list(W.generator_state_shapes().keys())
    for k, shp in W.generator_state_shapes().items():
        assert tuple(sd[k].shape) == tuple(shp), k
    assert sum(p.numel() for p in g.parameters()) == 3_214_974
    g.load_state_dict(W.synthetic_state_dict(0), strict=True)
    with pytest.raises(RuntimeError):
        bad = W.synthetic_state_dict(0)
        bad.pop("mid.0.bias")
        g.load_state_dict(bad, strict=True)
    # num_blocks other than 16 keeps ChannelAttention after blocks 7 and 15 (Generator.py:57)
    assert "blocks.8.fc.0.weight" in Generator(num_blocks=9).state_dict()
    assert not any(".fc." in k for k in Generator(num_blocks=4).state_dict())


def test_forward_fails_loudly_off_gpu_and_in_train_mode():
    g = Generator()
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.train()(x)                                     # train mode has no eager fallback either
    with pytest.raises(NotImplementedError, match="input image"):
        g.train()(x.clone().requires_grad_())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.eval()(x)
    with pytest.raises(RuntimeError, match="channels"):
        g.eval()(torch.zeros(1, 4, 64, 64))
    with pytest.raises(RuntimeError):
        g.eval()(torch.zeros(3, 64, 64))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "superresolution_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_synthetic_weights_are_deterministic_and_nontrivial():
    a, b = W.synthetic_state_dict_numpy(0), W.synthetic_state_dict_numpy(0)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = W.synthetic_state_dict_numpy(1)
    assert not np.array_equal(a["mid.0.weight"], c["mid.0.weight"])
    assert a["blocks.0.block.1.running_var"].min() >= 0.5 and a["blocks.0.block.1.running_var"].max() <= 1.5
    assert a["blocks.3.block.2.weight"].std() > 0.05            # per-channel slopes really differ
    # pinned values: a change of the hash silently invalidates every fixture
    pinned = [0.04277176409959793, 0.11164812743663788, 0.2863072156906128]
    got = [float(a["input.0.weight"].reshape(-1)[0]), float(a["blocks.5.block.4.running_mean"][3]),
           float(W.synthetic_lr_batch(1, 0)[0, 1, 2, 3])]
    assert got == pinned
    x = W.synthetic_lr_batch(2, 0, "sparse")
    assert x.shape == (2, 3, 64, 64) and 0.95 < (x == 0).mean() < 0.99 and x.max() <= 1.0


def test_bench_launcher_starts_one_rank_per_gpu_without_touching_the_gpu(monkeypatch, capfd):
    """bench.py --gpus N (N > 1, no WORLD_SIZE): the parent starts N children with the torchrun environment
    (127.0.0.1 rendezvous) and exits with their worst code; with fewer visible GPUs than N it refuses."""
    import importlib
    import json
    import torch
    bench = importlib.import_module("bench")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 3)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(3, ["--gpus", "3", "--dry-run"])
    assert e.value.code == 0
    lines = [json.loads(l) for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert sorted(int(l["RANK"]) for l in lines) == [0, 1, 2]
    assert all(l["WORLD_SIZE"] == "3" and l["MASTER_ADDR"] == "127.0.0.1" and l["LOCAL_RANK"] == l["RANK"] for l in lines)
    assert len({l["MASTER_PORT"] for l in lines}) == 1
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench.spawn_ranks(2, ["--gpus", "2", "--dry-run"])
    # a failing rank fails the launch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(2, ["--gpus", "2", "--no-such-flag"])
    assert e.value.code not in (0, None)
