"""Batch-sharded validation: the multi-GPU form of ``validate()`` (``train.py:57-72``).

The reference evaluates on one GPU: for every 256-image batch it runs the generator, takes
PSNR / SSIM / L1 of that batch (three ``.item()`` host syncs) and finally averages the
per-batch values.  Images are independent in eval mode, so here each rank (one process per
GPU) takes a contiguous run of WHOLE batches, keeps its running sums on the device, and the
only collective is one SUM all-reduce of four scalars {sum psnr, sum ssim, sum l1, batches}
over RCCL (``torch.distributed`` backend ``"nccl"``; ``"gloo"`` in the CPU tests).  Sharding
on batch boundaries keeps the reference's "mean of per-batch values" semantics, so N ranks
give the 1-rank answer up to floating-point summation order.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_batches(num_batches: int, rank: int, world_size: int) -> range:
    """Contiguous, near-equal run of whole batches for ``rank`` (first ranks take the remainder)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(num_batches, world_size)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def reduce_metric_sums(sums: torch.Tensor, group=None) -> torch.Tensor:
    """SUM all-reduce of the 4-vector {sum psnr, sum ssim, sum l1, n_batches} (fp64)."""
    if sums.shape != (4,):
        raise ValueError("expected a 4-vector {psnr, ssim, l1, count}")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


@torch.no_grad()
def validate(generator: Callable[[torch.Tensor], torch.Tensor],
             batches: Iterable[Tuple[torch.Tensor, torch.Tensor]],
             metrics_fn: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
             device: Optional[torch.device] = None, group=None,
             forward_dtype: torch.dtype = torch.float32) -> Tuple[float, float, float]:
    """(mean psnr, mean ssim, mean l1) over every batch of every rank.

    ``batches`` yields this rank's ``(lr, hr)`` pairs (see :func:`shard_batches`).  ``metrics_fn``
    maps ``(sr, hr)`` to a 3-vector [psnr, ssim, l1] on ``sr``'s device; it defaults to the HIP
    kernel behind ``superresolution_amd.batch_metrics``.  The generator call is
    ``generator(lr.float())`` as at ``train.py:65`` (``forward_dtype=torch.bfloat16`` runs
    BASELINE.json's bf16 configuration instead); there is one host sync, after the all-reduce.
    """
    if metrics_fn is None:
        from .functional import batch_metrics as metrics_fn
    # train.py:59,71: validate() switches the generator to eval for its own duration and back afterwards
    # (it is called on a train-mode generator at train.py:104,193); without it a train-mode module would
    # silently use batch statistics and overwrite its BatchNorm running statistics here.
    module = generator if isinstance(generator, torch.nn.Module) else None
    was_training = module.training if module is not None else False
    if module is not None:
        module.eval()
    acc = None
    try:
        for lr, hr in batches:
            if device is not None:
                lr, hr = lr.to(device), hr.to(device)
            sr = generator(lr.to(forward_dtype))
            m = metrics_fn(sr.float(), hr.float()).to(torch.float64)
            one = torch.ones(1, dtype=torch.float64, device=m.device)
            v = torch.cat([m.reshape(3), one])
            acc = v if acc is None else acc + v
    finally:
        if module is not None:
            module.train(was_training)
    if acc is None:                                   # a rank may own no batch at all
        acc = torch.zeros(4, dtype=torch.float64, device=_empty_shard_device(device, group))
    acc = reduce_metric_sums(acc, group)
    s = acc.cpu()
    n = float(s[3])
    if n == 0:
        raise ValueError("validate(): no batches on any rank")
    return float(s[0]) / n, float(s[1]) / n, float(s[2]) / n


def _empty_shard_device(device: Optional[torch.device], group) -> torch.device:
    """Where a rank with no batch builds its zero accumulator: RCCL ("nccl") reduces device tensors only, so
    such a rank must not fall back to the CPU while its peers are already inside the collective."""
    if device is not None:
        return torch.device(device)
    if dist.is_available() and dist.is_initialized() and "nccl" in str(dist.get_backend(group)):
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def batched(lr: torch.Tensor, hr: torch.Tensor, batch_size: int, which: Sequence[int]):
    """Yield the whole batches ``which`` (indices) of an in-memory (lr, hr) set."""
    for b in which:
        sl = slice(b * batch_size, (b + 1) * batch_size)
        yield lr[sl], hr[sl]
