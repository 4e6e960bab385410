"""Data formats on either side of the generator forward.

Mirrors, for the files the reference reads and writes around the hot path:

* ``.pt`` chunk files ``{'lr': (n,C,64,64) f32, 'hr': (n,C,125,125) f32, 'num_samples': n}``
  written by ``scripts/convert_to_pt.py:39`` and streamed by ``dataset.py:7-61``;
* ``normalization_stats.pt`` (``dataset.py:104-113``; ``train.py:234`` reads ``['hr_p995']``);
* checkpoints ``{'generator': state_dict, ...}`` (``train.py:119,124,213-225``; read at ``:268``).

``JetImageDataset`` yields what the reference's dataset yields, in the same order (file split,
round-robin worker sharding, per-chunk shuffle; pinned by ``tests/golden/dataset.npz``).  What changes
is where the normalisation runs: with ``normalize=False`` the dataset yields RAW tensors and
``normalize_on_device`` applies ``clamp(x / hr_max, 0, 1)`` (``dataset.py:52-54``) with a HIP kernel
after the H2D copy, bit-identically, instead of in the DataLoader workers.
Files are read with ``torch.load(..., weights_only=True)``: nothing in them is executed.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import torch
from torch.utils.data import IterableDataset, get_worker_info

from . import _lib

STATS_FILE = "normalization_stats.pt"


def list_chunk_files(tensor_dir: str):
    files = sorted(glob.glob(os.path.join(tensor_dir, "*.pt")))          # dataset.py:10-11
    return [f for f in files if not f.endswith(STATS_FILE)]


def load_chunk(path: str) -> Tuple[torch.Tensor, torch.Tensor]:
    data = torch.load(path, map_location="cpu", weights_only=True)          # dataset.py:46-47
    return data["lr"].float(), data["hr"].float()


def load_normalization_stats(tensor_dir: str) -> Dict[str, float]:
    """``torch.load(f'{tensor_dir}/normalization_stats.pt')`` (train.py:234): plain floats."""
    st = torch.load(os.path.join(tensor_dir, STATS_FILE), map_location="cpu", weights_only=True)
    return {k: float(v) for k, v in st.items()}


def load_generator_checkpoint(path: str, map_location="cpu"):
    """``torch.load(path, map_location)['generator']`` (train.py:268) -> OrderedDict for
    ``Generator.load_state_dict``; works for pretrain_*.pt, best_model.pt and epoch_*.pt."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    if "generator" not in ck:
        raise KeyError(f"{path}: no 'generator' entry (keys: {sorted(ck)})")
    return ck["generator"]


_SPLITS = ("train", "val", "test")


def split_chunk_files(files: Sequence[str], split: str, train_ratio: float, val_ratio: float) -> List[str]:
    """The slice of the sorted chunk list a split owns: the first ``int(n * train_ratio)`` files train,
    the next ``int(n * val_ratio)`` validate, the remainder tests (behaviour of ``dataset.py:14-24``)."""
    if split not in _SPLITS:
        raise ValueError(f"Invalid split: {split}")
    cut_a = int(len(files) * train_ratio)
    cut_b = cut_a + int(len(files) * val_ratio)
    bounds = {"train": (0, cut_a), "val": (cut_a, cut_b), "test": (cut_b, len(files))}[split]
    return list(files[bounds[0]:bounds[1]])


def _loader_slot() -> Tuple[int, int]:
    """(worker index, worker count) of the DataLoader process we are in; (0, 1) outside one."""
    info = get_worker_info()
    return (info.id, info.num_workers) if info is not None else (0, 1)


def iter_chunk_samples(path: str, order_rng: torch.Generator, scale: Optional[float]):
    """Samples of one chunk in a ``torch.randperm`` order drawn from ``order_rng``.  ``scale`` (the p99.5
    statistic) applies ``clamp(x / scale, 0, 1)`` on the host, as the reference's workers do
    (``dataset.py:52-54``); ``None`` yields raw values for :func:`normalize_on_device`.  The permutation is
    drawn AFTER the load, so an unreadable chunk consumes no random numbers."""
    lr, hr = load_chunk(path)
    if scale:
        lr, hr = (lr / scale).clamp_(0, 1), (hr / scale).clamp_(0, 1)
    order = torch.randperm(lr.size(0), generator=order_rng)
    yield from zip(lr[order].unbind(0), hr[order].unbind(0))


class JetImageDataset(IterableDataset):
    """Streams ``(lr, hr)`` pairs out of ``.pt`` chunk files with the reference's constructor and yield
    order (``dataset.py:7-61``; ``tests/golden/dataset.npz`` was produced by iterating the reference's
    class over the same files): files are dealt round-robin over DataLoader workers, each worker shuffles
    inside a chunk with a generator seeded ``seed + worker index``, and a chunk that fails to load is
    reported and skipped.  ``normalize=False`` keeps the host out of the normalisation."""

    def __init__(self, tensor_dir, split="train", train_ratio=0.8, val_ratio=0.1, hr_max=None, seed=42,
                 normalize=True):
        super().__init__()
        found = list_chunk_files(tensor_dir)
        if not found:
            raise ValueError(f"No .pt files found in {tensor_dir}")
        self.files = split_chunk_files(found, split, train_ratio, val_ratio)
        self.hr_max, self.seed, self.normalize = hr_max, seed, normalize

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        slot, slots = _loader_slot()
        order_rng = torch.Generator().manual_seed(self.seed + slot)
        scale = self.hr_max if self.normalize else None
        for path in self.files[slot::slots]:
            try:
                yield from iter_chunk_samples(path, order_rng, scale)
            except Exception as exc:                      # the reference keeps streaming (dataset.py:48-50)
                print(f"Error loading {path}: {exc}")


def normalize_on_device(x: torch.Tensor, hr_max: float, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """``torch.clamp(x / hr_max, 0, 1)`` (dataset.py:52-54) on the HIP path; x: ROCm fp32 tensor.
    Returns a new fp32 or bf16 tensor (bf16 feeds the bf16 generator directly)."""
    if not x.is_cuda:
        raise RuntimeError(f"normalize_on_device needs a ROCm device tensor (got {x.device}); there is no CPU fallback")
    if x.dtype != torch.float32:
        raise RuntimeError(f"raw chunks are fp32 (got {x.dtype})")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"unsupported output dtype {out_dtype}")
    if not hr_max:
        return x.to(out_dtype)                       # dataset.py:52: `if self.hr_max:` -> no-op
    xc = x.contiguous()
    y = torch.empty(xc.shape, dtype=out_dtype, device=xc.device)
    code = _lib.SRGEN_BF16 if out_dtype == torch.bfloat16 else _lib.SRGEN_F32
    with torch.cuda.device(xc.device):
        stream = torch.cuda.current_stream(xc.device).cuda_stream
        _lib.check(_lib.lib().srgen_normalize(xc.data_ptr(), y.data_ptr(), xc.numel(), float(hr_max), code, stream))
    return y
