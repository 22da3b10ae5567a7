"""SPADE residual block (reference: spade/models/networks/architecture.py:21-68)."""
import torch.nn as nn
import torch.nn.utils.spectral_norm as spectral_norm

from ....sg2im.layers import Conv2d
from .normalization import SPADE


class SPADEResnetBlock(nn.Module):
    def __init__(self, fin, fout, opt):
        super().__init__()
        self.learned_shortcut = (fin != fout)
        fmiddle = min(fin, fout)
        self.conv_0 = Conv2d(fin, fmiddle, kernel_size=3, padding=1)
        self.conv_1 = Conv2d(fmiddle, fout, kernel_size=3, padding=1)
        if self.learned_shortcut:
            self.conv_s = Conv2d(fin, fout, kernel_size=1, bias=False)
        if 'spectral' in opt.norm_G:
            self.conv_0 = spectral_norm(self.conv_0)
            self.conv_1 = spectral_norm(self.conv_1)
            if self.learned_shortcut:
                self.conv_s = spectral_norm(self.conv_s)
        cfg = opt.norm_G.replace('spectral', '')
        self.norm_0 = SPADE(cfg, fin, opt.semantic_nc)
        self.norm_1 = SPADE(cfg, fmiddle, opt.semantic_nc)
        if self.learned_shortcut:
            self.norm_s = SPADE(cfg, fin, opt.semantic_nc)

    def forward(self, x, seg):
        # LeakyReLU(0.2) of `actvn` is fused into the SPADE apply pass; the skip add into conv_1's epilogue
        x_s = self.shortcut(x, seg)
        dx = self.conv_0(self.norm_0(x, seg, fused_slope=0.2))
        return self.conv_1(self.norm_1(dx, seg, fused_slope=0.2), residual=x_s)

    def shortcut(self, x, seg):
        return self.conv_s(self.norm_s(x, seg)) if self.learned_shortcut else x


class VGG19(nn.Module):
    def __init__(self, requires_grad=False):
        super().__init__()
        raise NotImplementedError("VGG19 perceptual features need pretrained torchvision weights; next-row "
                                  "component (SURVEY.md §8f rank 2) — train with --no_vgg_loss")
