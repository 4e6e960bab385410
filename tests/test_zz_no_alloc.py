"""GPU: srgen_metrics_noalloc (the one entry point that allocates nothing: per-tile partials in a file-scope
__device__ array of the library) against srgen_metrics, through the C ABI, two calls in a row with the second
on a smaller input, bit for bit (same kernels, same reduction order)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(sr, hr):
    from superresolution_amd import _lib, batch_metrics
    n, c, h, w = sr.shape
    out = torch.full((3,), float("nan"), dtype=torch.float32, device=sr.device)
    stream = torch.cuda.current_stream(sr.device).cuda_stream
    _lib.check(_lib.lib().srgen_metrics_noalloc(sr.data_ptr(), hr.data_ptr(), n, c, h, w, out.data_ptr(), stream))
    return out, batch_metrics(sr, hr)


def test_metrics_noalloc_matches_metrics_two_calls_second_smaller():
    from superresolution_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(7)
    big = [torch.rand((64, 3, 125, 125), generator=g).cuda() for _ in range(2)]
    small = [torch.rand((3, 2, 37, 70), generator=g).cuda() for _ in range(2)]
    a, ra = _both(*big)              # fills 64*3*16 tiles of the static array
    b, rb = _both(*small)            # 3*2*2*3 tiles: stale partials of the first call lie behind them
    torch.cuda.synchronize()
    assert torch.equal(a, ra) and torch.equal(b, rb)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    c, rc = _both(*big)
    assert torch.equal(c, ra)
    # beyond the static maximum: refused with a clear error, nothing launched
    L = _lib.lib()
    n_big = 65536 // (3 * 16) + 1
    rc_ = L.srgen_metrics_noalloc(big[0].data_ptr(), big[1].data_ptr(), n_big, 3, 125, 125, a.data_ptr(), None)
    assert rc_ == -5 and b"static scratch" in L.srgen_last_error()
