"""GPU: the input-normalisation kernel (srgen_normalize) against the reference expression
torch.clamp(x / hr_max, 0, 1) of dataset.py:52-54, bit for bit, and the chunk -> device ->
normalise -> generator pipeline against the CPU-normalised path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("numel", [1, 3, 4, 5, 1023, 3 * 64 * 64 * 7 + 2])
@pytest.mark.parametrize("hr_max", [1.37, 0.0123, 250.0])
def test_normalize_bit_exact_fp32(numel, hr_max):
    from superresolution_amd import data
    g = torch.Generator().manual_seed(numel)
    x = (torch.rand(numel, generator=g) - 0.2) * 3.0 * hr_max         # negatives, > hr_max, in range
    if numel > 8:
        x[3], x[5], x[7] = 0.0, hr_max, float("inf")
    ref = torch.clamp(x / hr_max, 0, 1)
    y = data.normalize_on_device(x.cuda(), hr_max).cpu()
    assert torch.equal(y, ref)


def test_normalize_bf16_out_and_nan():
    from superresolution_amd import data
    x = torch.rand(4097) * 2.0
    x[11] = float("nan")
    ref = torch.clamp(x / 1.37, 0, 1)
    y = data.normalize_on_device(x.cuda(), 1.37, torch.bfloat16).cpu()
    assert y.dtype == torch.bfloat16
    assert torch.equal(y[~ref.isnan()], ref[~ref.isnan()].to(torch.bfloat16))
    assert torch.isnan(y[11])                                           # NaN propagates like ATen's clamp
    z = data.normalize_on_device(x.cuda(), 0.0)                         # dataset.py:52 `if self.hr_max:` -> raw
    assert torch.equal(z.cpu()[~x.isnan()], x[~x.isnan()])


def test_chunks_to_generator_device_normalisation_equals_cpu_normalisation(tmp_path):
    """raw chunk -> H2D -> HIP normalise -> generator == CPU-normalised (reference order) -> generator."""
    from superresolution_amd import Generator, data, weights as W
    W.write_synthetic_chunks(str(tmp_path), num_files=5)
    hr_max = data.load_normalization_stats(str(tmp_path))["hr_p995"]
    cpu_norm = data.JetImageDataset(str(tmp_path), split="train", hr_max=hr_max)
    raw = data.JetImageDataset(str(tmp_path), split="train", hr_max=hr_max, normalize=False)
    lr_ref = torch.stack([lr for lr, _ in cpu_norm])
    lr_raw = torch.stack([lr for lr, _ in raw])
    assert lr_ref.shape == lr_raw.shape and lr_ref.shape[0] >= 8
    g = Generator()
    g.load_state_dict(W.synthetic_state_dict(0))
    g = g.cuda().eval()
    x_dev = data.normalize_on_device(lr_raw.cuda(), hr_max)
    assert torch.equal(x_dev.cpu(), lr_ref)
    assert torch.equal(g(x_dev), g(lr_ref.cuda()))
