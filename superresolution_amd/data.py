"""Data formats on either side of the generator forward.

Mirrors, for the files the reference reads and writes around the hot path:

* ``.pt`` chunk files ``{'lr': (n,C,64,64) f32, 'hr': (n,C,125,125) f32, 'num_samples': n}``
  written by ``scripts/convert_to_pt.py:39`` and streamed by ``dataset.py:7-61``;
* ``normalization_stats.pt`` (``dataset.py:104-113``; ``train.py:234`` reads ``['hr_p995']``);
* checkpoints ``{'generator': state_dict, ...}`` (``train.py:119,124,213-225``; read at ``:268``).

``JetImageDataset`` reproduces the reference's file split, worker sharding and per-file shuffle
(same ``torch.Generator`` seed discipline => same sample order; fixture-tested).  What changes
is where the normalisation runs: with ``normalize=False`` the dataset yields RAW tensors and
``normalize_on_device`` applies ``clamp(x / hr_max, 0, 1)`` (``dataset.py:52-54``) with a HIP kernel
after the H2D copy, bit-identically, instead of in the DataLoader workers.
Files are read with ``torch.load(..., weights_only=True)``: nothing in them is executed.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, Iterator, Tuple

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


class JetImageDataset(IterableDataset):
    """Same constructor, split arithmetic, worker sharding and shuffle as ``dataset.py:7-61``."""

    def __init__(self, tensor_dir, split="train", train_ratio=0.8, val_ratio=0.1, hr_max=None, seed=42,
                 normalize=True):
        super().__init__()
        all_files = list_chunk_files(tensor_dir)
        if not all_files:
            raise ValueError(f"No .pt files found in {tensor_dir}")
        n = len(all_files)
        n_train = int(n * train_ratio)
        n_val = int(n * val_ratio)
        if split == "train":
            self.files = all_files[:n_train]
        elif split == "val":
            self.files = all_files[n_train:n_train + n_val]
        elif split == "test":
            self.files = all_files[n_train + n_val:]
        else:
            raise ValueError(f"Invalid split: {split}")
        self.hr_max = hr_max
        self.seed = seed
        self.normalize = normalize

    def _get_worker_files(self):
        worker = get_worker_info()
        if worker is None:
            return self.files
        return [f for i, f in enumerate(self.files) if i % worker.num_workers == worker.id]

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        worker_info = get_worker_info()
        worker_id = worker_info.id if worker_info else 0
        rng = torch.Generator().manual_seed(self.seed + worker_id)
        for path in self._get_worker_files():
            try:
                lr, hr = load_chunk(path)
            except Exception as e:                                         # dataset.py:48-50
                print(f"Error loading {path}: {e}")
                continue
            if self.hr_max and self.normalize:                             # dataset.py:52-54
                lr = torch.clamp(lr / self.hr_max, 0, 1)
                hr = torch.clamp(hr / self.hr_max, 0, 1)
            n = lr.size(0)
            perm = torch.randperm(n, generator=rng)
            lr, hr = lr[perm], hr[perm]
            for i in range(n):
                yield lr[i], hr[i]


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
