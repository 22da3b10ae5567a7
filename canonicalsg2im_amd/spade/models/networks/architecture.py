"""SPADE residual block and the VGG19 feature stack (reference: spade/models/networks/architecture.py:21-68, 93-123)."""
import torch
import torch.nn as nn

from .... import ops
from ....spectral_norm import spectral_norm
from ....sg2im.layers import Conv2d
from .normalization import SPADE, spade_pair


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
        if self.learned_shortcut:
            # norm_s and norm_0 normalise the same x: one statistics pass, one backward pass over x (spade_pair)
            xs, h = spade_pair(self.norm_s, self.norm_0, x, seg, 1.0, 0.2)
            x_s = self.conv_s(xs)
        else:
            x_s, h = x, self.norm_0(x, seg, fused_slope=0.2)
        dx = self.conv_0(h)
        return self.conv_1(self.norm_1(dx, seg, fused_slope=0.2), residual=x_s)

    def shortcut(self, x, seg):
        return self.conv_s(self.norm_s(x, seg)) if self.learned_shortcut else x


# torchvision's vgg19 'E' configuration up to relu5_1 — the part VGG19 slices (reference
# architecture.py:96-110 takes features[0:30]); numbers are conv widths, 'M' is MaxPool2d(2, 2).
_VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512)
_VGG19_SLICES = ((0, 2), (2, 7), (7, 12), (12, 21), (21, 30))     # architecture.py:101-110


class _VGGConv(nn.Module):
    """Conv3x3(pad 1) + ReLU of the feature stack (ReLU fused into the conv epilogue).  Holds the
    parameters under torchvision's names (`weight`, `bias`)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_normal_(self.weight, mode='fan_out', nonlinearity='relu')   # torchvision's init
        nn.init.zeros_(self.bias)
        self._packs = None

    def forward(self, x):
        if self.weight.requires_grad:
            return ops.conv2d(x, self.weight, self.bias, 1, 1, act=ops.ACT_LEAKY, slope=0.0)
        if self._packs is None or self._packs[-1] != self.weight._version or self._packs[0].device != x.device:
            self._packs = ops.pack_conv_weight(self.weight) + (self.weight._version,)
        return ops.conv2d(x, self.weight, self.bias, 1, 1, act=ops.ACT_LEAKY, slope=0.0, packs=self._packs[:-1])


class _VGGPool(nn.Module):
    def forward(self, x):
        return ops.maxpool2(x)


class _Slice(nn.Module):
    """Sub-modules registered under their torchvision `features` index (the ReLU indices are
    parameter-free and fused away), so that state_dict keys equal the reference's VGG19:
    `slice1.0.weight`, `slice2.2.weight`, `slice2.5.weight`, ..."""

    def forward(self, x):
        for m in self.children():
            x = m(x)
        return x


def _find_vgg19_weights():
    """torchvision-format state dict of vgg19 ('features.N.weight' keys) from $CSG_VGG19_WEIGHTS, or
    from torchvision when it is installed.  Returns None if neither is available."""
    import os
    path = os.environ.get("CSG_VGG19_WEIGHTS")
    if path:
        sd = torch.load(path, map_location="cpu")
        return sd.get("state_dict", sd)
    try:
        import torchvision
        return torchvision.models.vgg19(pretrained=True).state_dict()
    except Exception:                                       # absent package or no network
        return None


class VGG19(nn.Module):
    """Reference: spade/models/networks/architecture.py:93-123.  Five slices of vgg19().features ending at
    relu1_1, relu2_1, relu3_1, relu4_1, relu5_1; frozen unless `requires_grad`.

    The reference downloads torchvision's ImageNet weights.  Here they are read from the file named
    by $CSG_VGG19_WEIGHTS (torchvision's `vgg19-dcbb9e9d.pth`) or from torchvision if installed;
    with neither the constructor raises unless `weights='random'` (or $CSG_VGG19_RANDOM=1) is given —
    a random-feature perceptual loss is a throughput stand-in, not the reference's objective."""

    def __init__(self, requires_grad=False, weights=None):
        super().__init__()
        import os
        idx, cin, layers = 0, 3, {}
        for v in _VGG19_CFG:
            if v == 'M':
                layers[idx] = _VGGPool()
                idx += 1
            else:
                layers[idx] = _VGGConv(cin, v)
                cin = v
                idx += 2                                    # conv + its (fused) ReLU
        for k, (lo, hi) in enumerate(_VGG19_SLICES):
            sl = _Slice()
            for i in range(lo, hi):
                if i in layers:
                    sl.add_module(str(i), layers[i])
            setattr(self, "slice%d" % (k + 1), sl)
        if weights is None and os.environ.get("CSG_VGG19_RANDOM") == "1":
            weights = "random"
        if weights != "random":
            sd = weights if isinstance(weights, dict) else _find_vgg19_weights()
            if sd is None:
                raise RuntimeError(
                    "VGG19: no pretrained weights. Point CSG_VGG19_WEIGHTS at torchvision's vgg19 state dict "
                    "(vgg19-dcbb9e9d.pth), pass weights='random' / set CSG_VGG19_RANDOM=1 for a random-feature "
                    "stand-in, or train with --no_vgg_loss")
            self.load_torchvision(sd)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def load_torchvision(self, sd):
        with torch.no_grad():
            for k, (lo, hi) in enumerate(_VGG19_SLICES):
                for name, m in getattr(self, "slice%d" % (k + 1)).named_children():
                    if isinstance(m, _VGGConv):
                        m.weight.copy_(sd["features.%s.weight" % name])
                        m.bias.copy_(sd["features.%s.bias" % name])

    def forward(self, X):
        h1 = self.slice1(X)
        h2 = self.slice2(h1)
        h3 = self.slice3(h2)
        h4 = self.slice4(h3)
        h5 = self.slice5(h4)
        return [h1, h2, h3, h4, h5]
