"""superresolution_amd -- MI355X-native SRGAN generator forward behind the reference's Python API
(plus, from SURVEY.md 8(f): batch metrics, data formats, discriminator forward + RaGAN loss values).

    from superresolution_amd import Generator
    g = Generator().to("cuda").eval(); g.load_state_dict(torch.load(p)["generator"]); sr = g(lr)
"""
from .generator import Generator, ResBlock, ChannelAttention, UpsampleBlock  # noqa: F401
from .discriminator import Discriminator, DiscBlock, RelativisticAverageLoss  # noqa: F401
from .functional import bilinear_resize, batch_metrics, pixel_losses, weighted_l1  # noqa: F401
from .autograd import pixel_loss  # noqa: F401
from . import weights  # noqa: F401
from . import data  # noqa: F401

__all__ = ["Generator", "ResBlock", "ChannelAttention", "UpsampleBlock", "Discriminator", "DiscBlock",
           "RelativisticAverageLoss", "bilinear_resize", "batch_metrics", "pixel_losses", "weighted_l1", "pixel_loss", "weights", "data"]
