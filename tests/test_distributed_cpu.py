"""world_size-2 gloo test of the batch-sharded validate(): N ranks == 1 rank.
The generator/metrics are stood in by the CPU oracle here (test infrastructure); the GPU test
of the same function with the HIP kernels is tests/test_gpu_metrics.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from superresolution_amd import evaluate as E
from superresolution_amd import weights as W


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_generator(lr):
    # cheap deterministic stand-in with the right shapes: 64x64 -> 125x125
    up = torch.nn.functional.interpolate(lr, size=(125, 125), mode="nearest")
    return 0.9 * up + 0.01


def _oracle_metrics(sr, hr):
    from oracle import oracle as O
    return torch.tensor(O.metrics(sr.numpy(), hr.numpy()), dtype=torch.float32)


def _data(nb, bs):
    lr = torch.from_numpy(W.synthetic_lr_batch(nb * bs, seed=4, kind="dense"))
    hr = torch.from_numpy(W.synthetic_hr_batch(nb * bs, seed=5))
    return lr, hr


def _worker(rank, world, port, nb, bs, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lr, hr = _data(nb, bs)
        mine = E.shard_batches(nb, rank, world)
        res = E.validate(_fake_generator, E.batched(lr, hr, bs, mine), metrics_fn=_oracle_metrics)
        if rank == 0:
            np.save(out, np.array(res))
    finally:
        dist.destroy_process_group()


def test_shard_batches_partition():
    for nb in (0, 1, 5, 32):
        for world in (1, 2, 3, 8):
            got = [b for r in range(world) for b in E.shard_batches(nb, r, world)]
            assert got == list(range(nb))
    assert list(E.shard_batches(32, 3, 8)) == [12, 13, 14, 15]      # cfg3: 4 batches per GPU
    with pytest.raises(ValueError):
        E.shard_batches(4, 2, 2)


@pytest.mark.parametrize("nb", [5, 1])       # 1 batch on 2 ranks: rank 1 owns nothing (ragged)
def test_validate_two_ranks_equals_one_rank(tmp_path, nb):
    bs = 2
    lr, hr = _data(nb, bs)
    single = E.validate(_fake_generator, E.batched(lr, hr, bs, range(nb)), metrics_fn=_oracle_metrics)
    # by hand, as train.py:57-72 does it: mean of per-batch values
    per = [_oracle_metrics(_fake_generator(lr[b * bs:(b + 1) * bs]), hr[b * bs:(b + 1) * bs]).double()
           for b in range(nb)]
    hand = torch.stack(per).mean(0)
    assert np.allclose(single, hand.numpy(), rtol=0, atol=1e-9)
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), nb, bs, out), nprocs=2, join=True)
    two = np.load(out)
    assert np.allclose(two, np.array(single), rtol=0, atol=1e-9)


def test_validate_raises_without_batches():
    with pytest.raises(ValueError):
        E.validate(_fake_generator, [], metrics_fn=_oracle_metrics)


class _ModeProbe(torch.nn.Module):
    """Stand-in generator that records the mode it is called in (train.py:59,71 semantics of validate)."""

    def __init__(self):
        super().__init__()
        self.seen = []

    def forward(self, lr):
        self.seen.append(self.training)
        return _fake_generator(lr)


def test_validate_runs_in_eval_mode_and_restores_training_flag():
    lr, hr = _data(2, 2)
    g = _ModeProbe().train()
    E.validate(g, E.batched(lr, hr, 2, range(2)), metrics_fn=_oracle_metrics)
    assert g.seen == [False, False] and g.training           # eval inside, train restored (train.py:71)
    g = _ModeProbe().eval()
    E.validate(g, E.batched(lr, hr, 2, range(2)), metrics_fn=_oracle_metrics)
    assert g.seen == [False, False] and not g.training

    def boom(sr, hr):
        raise RuntimeError("metrics failed")
    g = _ModeProbe().train()
    with pytest.raises(RuntimeError, match="metrics failed"):
        E.validate(g, E.batched(lr, hr, 2, range(1)), metrics_fn=boom)
    assert g.training                                         # restored on the error path too


def test_empty_shard_accumulator_device():
    assert E._empty_shard_device(None, None) == torch.device("cpu")          # no process group: CPU
    assert E._empty_shard_device(torch.device("cpu"), None) == torch.device("cpu")
    assert E._empty_shard_device("cuda:0", None) == torch.device("cuda", 0)  # explicit device wins
