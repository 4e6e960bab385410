"""superresolution_amd -- MI355X-native SRGAN generator forward behind the reference's Python API.

    from superresolution_amd import Generator
    g = Generator().to("cuda").eval(); g.load_state_dict(torch.load(p)["generator"]); sr = g(lr)
"""
from .generator import Generator, ResBlock, ChannelAttention, UpsampleBlock  # noqa: F401
from .functional import bilinear_resize, batch_metrics  # noqa: F401
from . import weights  # noqa: F401
from . import data  # noqa: F401

__all__ = ["Generator", "ResBlock", "ChannelAttention", "UpsampleBlock", "bilinear_resize", "batch_metrics",
           "weights", "data"]
