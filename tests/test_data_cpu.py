"""Data formats around the hot path: the JetImageDataset mirror against a
fixture produced by iterating the REAL reference dataset (dataset.py) over the same synthetic chunk
files (tests/golden/make_golden.py dataset), normalization_stats.pt, and the checkpoint container."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def chunk_dir(tmp_path_factory):
    from superresolution_amd import weights as W
    d = tmp_path_factory.mktemp("chunks")
    W.write_synthetic_chunks(str(d))
    return str(d)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "dataset.npz"))


def _rows(ds):
    return np.array([(float(lr.double().sum()), float(hr.double().sum()), float(lr.max()), float(hr.max()))
                     for lr, hr in ds], np.float64).reshape(-1, 4)


def test_stats_file(chunk_dir, gold):
    from superresolution_amd import data
    st = data.load_normalization_stats(chunk_dir)
    assert st["hr_p995"] == float(gold["hr_p995"])
    assert set(st) == {"lr_max", "hr_max", "lr_p995", "hr_p995"}          # dataset.py:104-109


@pytest.mark.parametrize("split", ["train", "val", "test"])
def test_split_order_and_values_match_reference(chunk_dir, gold, split):
    """Same files per split (80/10/10 of the sorted list), same per-file shuffle, same normalised
    values as the reference dataset: per-sample checksums in yield order are identical."""
    from superresolution_amd import data
    ds = data.JetImageDataset(chunk_dir, split=split, hr_max=float(gold["hr_p995"]))
    assert len(ds.files) == int(gold[f"{split}_nfiles"])
    rows = _rows(ds)
    ref = gold[f"{split}_rows"]
    assert rows.shape == ref.shape
    assert np.array_equal(rows, ref)
    assert rows[:, 2:].max() <= 1.0 and rows[:, 2].max() == 1.0           # the clamp at 1 is exercised


def test_raw_mode_and_no_hr_max(chunk_dir, gold):
    from superresolution_amd import data
    raw = data.JetImageDataset(chunk_dir, split="train", hr_max=None)
    rows = np.array([(float(lr.double().sum()), float(hr.double().sum())) for lr, hr in raw], np.float64)
    assert np.array_equal(rows, gold["train_raw_rows"])
    # normalize=False with an hr_max: raw values in the reference's order (device normalisation later)
    raw2 = data.JetImageDataset(chunk_dir, split="train", hr_max=float(gold["hr_p995"]), normalize=False)
    rows2 = np.array([(float(lr.double().sum()), float(hr.double().sum())) for lr, hr in raw2], np.float64)
    assert np.array_equal(rows2, gold["train_raw_rows"])


def test_dataloader_worker_sharding_matches_reference(chunk_dir, gold):
    """DataLoader(batch_size=4, num_workers=2): files are sharded by index over the workers and each
    worker seeds its own generator (dataset.py:33-42): batch checksums equal the reference's."""
    from torch.utils.data import DataLoader
    from superresolution_amd import data
    ds = data.JetImageDataset(chunk_dir, split="train", hr_max=float(gold["hr_p995"]))
    rows = np.array([(float(lr.double().sum()), float(hr.double().sum()), lr.shape[0])
                     for lr, hr in DataLoader(ds, batch_size=4, num_workers=2)], np.float64)
    assert np.array_equal(rows, gold["train_loader_w2_b4"])


def test_errors_like_reference(tmp_path, chunk_dir):
    from superresolution_amd import data
    with pytest.raises(ValueError, match="No .pt files"):
        data.JetImageDataset(str(tmp_path))
    with pytest.raises(ValueError, match="Invalid split"):
        data.JetImageDataset(chunk_dir, split="dev")


def test_corrupt_chunk_is_skipped(tmp_path, capsys):
    """dataset.py:45-50: a chunk that fails to load is reported and skipped, not fatal."""
    from superresolution_amd import data, weights as W
    W.write_synthetic_chunks(str(tmp_path), num_files=3)
    files = data.list_chunk_files(str(tmp_path))
    with open(files[1], "wb") as f:
        f.write(b"not a torch file")
    ds = data.JetImageDataset(str(tmp_path), split="train", train_ratio=1.0, hr_max=1.37)
    n = sum(1 for _ in ds)
    assert n == sum(data.load_chunk(f)[0].shape[0] for f in (files[0], files[2]))
    assert "Error loading" in capsys.readouterr().out


def test_checkpoint_container_roundtrip(tmp_path):
    """train.py:119,213-225 write {'generator': sd[, 'discriminator', 'epoch', ...]}; :268 reads
    ['generator'] into load_state_dict.  Same container, read with weights_only=True."""
    from superresolution_amd import Generator, data, weights as W
    sd = W.synthetic_state_dict(2)
    p1, p2 = str(tmp_path / "pretrain_final.pt"), str(tmp_path / "epoch_010.pt")
    torch.save({"generator": sd}, p1)
    torch.save({"epoch": 10, "generator": sd, "discriminator": {"w": torch.zeros(1)}}, p2)
    g = Generator()
    for p in (p1, p2):
        out = g.load_state_dict(data.load_generator_checkpoint(p))
        assert not out.missing_keys and not out.unexpected_keys
    assert all(torch.equal(g.state_dict()[k].reshape(-1), v.reshape(-1)) for k, v in sd.items())
    torch.save({"discriminator": {}}, p1)
    with pytest.raises(KeyError, match="generator"):
        data.load_generator_checkpoint(p1)


def test_normalize_on_device_refuses_cpu():
    from superresolution_amd import data
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        data.normalize_on_device(torch.zeros(4), 1.37)
