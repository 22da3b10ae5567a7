"""`spade.models.networks` surface of the reference (its __init__.py:7-11)."""
from .base_network import BaseNetwork
from .discriminator import (AcCropDiscriminator, AcDiscriminator, MultiscaleDiscriminator,
                            MultiscaleMaskDiscriminator2, NLayerDiscriminator)
from .generator import SPADEGenerator
from .loss import GANLoss, KLDLoss, VGGLoss

__all__ = ["BaseNetwork", "SPADEGenerator", "MultiscaleDiscriminator", "NLayerDiscriminator", "AcCropDiscriminator",
           "AcDiscriminator", "MultiscaleMaskDiscriminator2", "GANLoss", "VGGLoss", "KLDLoss"]
