"""SPADE and the discriminator's norm wrapper (reference: spade/models/networks/normalization.py).

One SPADE call = three kernels: mlp_shared conv (+ReLU epilogue), ONE conv producing gamma||beta
(the two 128->C convs share their input, so their weights are concatenated along Cout), and the
fused BatchNorm-apply + (1+gamma)*x+beta + LeakyReLU pass (K9)."""
import re

import torch
import torch.nn as nn

from .... import ops
from ....spectral_norm import spectral_norm
from ....sg2im.layers import Conv2d, _FusedActivation
from .sync_batchnorm import LocalBatchNorm2d, SynchronizedBatchNorm2d


class SegPyramid:
    """Layouts of one batch at every resolution the generator needs, produced by one layout
    kernel per level (nearest sampling folded in) instead of F.interpolate per SPADE call."""

    def __init__(self, levels):
        self.levels = dict(levels)

    def at(self, h):
        return self.levels[int(h)]


class InstanceNormAct(nn.InstanceNorm2d):
    """nn.InstanceNorm2d(affine=False) with an optional fused LeakyReLU (`fused_slope`)."""

    def __init__(self, num_features, fused_slope=1.0):
        super().__init__(num_features, affine=False)
        self.fused_slope = fused_slope

    def forward(self, x, gb=None, fused_slope=None):
        slope = self.fused_slope if fused_slope is None else fused_slope
        return ops.norm_act(x, gb, None, None, instance=True, training=self.training, slope=slope, eps=self.eps)


class SyncBatchNormAct(SynchronizedBatchNorm2d):
    """SynchronizedBatchNorm2d(affine=True) with an optional fused LeakyReLU (`fused_slope`, set by the PatchGAN)."""

    def __init__(self, num_features, fused_slope=1.0):
        super().__init__(num_features, affine=True)
        self.fused_slope = fused_slope

    def forward(self, x):
        return super().forward(x, None, self.fused_slope)


def get_nonspade_norm_layer(opt, norm_type='instance'):
    """Returns `wrap(conv)` for the PatchGAN layers (reference normalization.py:16-50): an optional
    `spectral` prefix applies spectral normalisation, the remainder names the activation norm that
    follows the conv (whose bias is then dropped): 'instance' (the default `spectralinstance`), 'batch',
    'sync_batch' or 'none'; the LeakyReLU behind the norm is fused into its apply pass."""
    use_sn = norm_type.startswith('spectral')
    after = norm_type[len('spectral'):] if use_sn else norm_type

    def wrap(conv):
        if use_sn:
            conv = spectral_norm(conv)
        if after in ('', 'none'):
            return conv
        if after not in ('instance', 'batch', 'sync_batch'):
            raise ValueError('normalization layer %s is not recognized' % after)
        if getattr(conv, 'bias', None) is not None:          # a bias in front of a normalisation is a no-op
            del conv.bias
            conv.register_parameter('bias', None)
        width = getattr(conv, 'out_channels', None) or conv.weight.size(0)
        if after == 'batch':                                 # nn.BatchNorm2d(affine=True), normalization.py:41-42
            from ....sg2im.layers import BatchNormAct
            return nn.Sequential(conv, BatchNormAct(width))
        if after == 'sync_batch':                            # SynchronizedBatchNorm2d(affine=True), :43-44
            return nn.Sequential(conv, SyncBatchNormAct(width))
        return nn.Sequential(conv, InstanceNormAct(width))

    return wrap


class _JoinedPair(torch.autograd.Function):
    """`joined` IS the memory of the two parameters `a` (first half along dim 0) and `b` (second half): the forward
    hands it out as one tensor without a copy, the backward hands the halves of its gradient to the parameters."""

    @staticmethod
    def forward(ctx, a, b, joined):
        ctx.n = a.shape[0]
        return joined.view_as(joined)

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.n], g[ctx.n:], None


def _join_storage(pa, pb):
    """Make the parameters `pa`, `pb` (same shape) the two halves of ONE allocation and return it ((2n, ...), in the
    parameters' memory format).  `.data` is re-pointed, the Parameter objects (optimiser state, state_dict keys) stay;
    called again only when something (`module.to`, a fresh load) has separated them."""
    cl = pa.dim() == 4
    joined = torch.empty((2 * pa.shape[0],) + tuple(pa.shape[1:]), device=pa.device, dtype=pa.dtype)
    if cl:
        joined = joined.contiguous(memory_format=torch.channels_last)
    n = pa.shape[0]
    with torch.no_grad():
        joined[:n].copy_(pa)
        joined[n:].copy_(pb)
    pa.data = joined[:n]
    pb.data = joined[n:]
    return joined


def _joined(mod, name, pa, pb):
    j = getattr(mod, name, None)
    n = pa.shape[0]
    if (j is None or j.device != pa.device or pa.data_ptr() != j.data_ptr() or
            pb.data_ptr() != j.data_ptr() + n * j.stride(0) * j.element_size() or pa.stride() != j.stride()
            or pb.stride() != j.stride()):
        j = _join_storage(pa, pb)
        object.__setattr__(mod, name, j)               # a plain attribute: not a buffer, not in the state_dict
    return _JoinedPair.apply(pa, pb, j)


class SPADE(nn.Module):
    def __init__(self, config_text, norm_nc, label_nc):
        super().__init__()
        assert config_text.startswith('spade')
        parsed = re.search(r'spade(\D+)(\d)x\d', config_text)
        kind, ks = str(parsed.group(1)), int(parsed.group(2))
        if kind == 'instance':
            self.param_free_norm = InstanceNormAct(norm_nc)
        elif kind == 'syncbatch':
            self.param_free_norm = SynchronizedBatchNorm2d(norm_nc, affine=False)
        elif kind == 'batch':
            self.param_free_norm = LocalBatchNorm2d(norm_nc, affine=False)
        else:
            raise ValueError('%s is not a recognized param-free norm type in SPADE' % kind)
        nhidden = 128
        self.pw = ks // 2
        self.mlp_shared = nn.Sequential(Conv2d(label_nc, nhidden, kernel_size=ks, padding=self.pw,
                                               act=ops.ACT_LEAKY, slope=0.0), _FusedActivation())
        self.mlp_gamma = Conv2d(nhidden, norm_nc, kernel_size=ks, padding=self.pw)
        self.mlp_beta = Conv2d(nhidden, norm_nc, kernel_size=ks, padding=self.pw)

    def forward(self, x, segmap, fused_slope=1.0):
        if self.fusable(x):
            return ops.spade_fused(x, [self.fused_operands(x, segmap, fused_slope)], self.param_free_norm.eps,
                                   self.param_free_norm.momentum, getattr(self.param_free_norm, "sync", True))[0]
        pn = self.param_free_norm
        if self.joinable(x):
            actv, w, b, rm, rv, slope, in_slope = self.fused_operands(x, segmap, fused_slope)
            return ops.spade_joined(x, actv, w, b, rm, rv, self.pw, slope, in_slope, pn.eps, pn.momentum,
                                    getattr(pn, "sync", True))
        return pn(x, gb=self.modulation(x, segmap), fused_slope=fused_slope)

    def joinable(self, x):
        """Training-mode param-free BatchNorm on a map the fused epilogue does not serve (8 x 8, 16 x 16): statistics,
        convolution and modulation ordered inside one Function (ops._SpadeJoined) so that the SyncBN messages of N > 1 ranks
        travel under the gamma || beta convolution; the kernels are those of the plain path."""
        pn = self.param_free_norm
        return (ops.SPADE_JOINED and isinstance(pn, (SynchronizedBatchNorm2d, LocalBatchNorm2d)) and not pn.affine
                and pn.training and x.dim() == 4 and x.is_cuda and self.mlp_gamma.weight.shape[0] % 4 == 0)

    def fusable(self, x):
        """The modulation can ride in the epilogue of the gamma || beta convolution (ops._SpadeFused): a param-free
        BatchNorm in training mode, 3x3 convolutions on a map F(4x4,3x3) serves."""
        pn = self.param_free_norm
        return (isinstance(pn, (SynchronizedBatchNorm2d, LocalBatchNorm2d)) and not pn.affine and pn.training
                and ops.spade_fused_eligible(x, self.mlp_gamma.weight.shape[1], self.mlp_gamma.weight.shape[0],
                                             self.mlp_gamma.weight.shape[2], pn.training))

    def joined_weight(self):
        """The gamma || beta weight as ONE (2C, nhidden, ks, ks) tensor (the two parameters' own memory), joined now if it
        is not yet — ops.prepack_weights asks before the layer's first call."""
        _joined(self, "_joined_w", self.mlp_gamma.weight, self.mlp_beta.weight)
        return self.__dict__.get("_joined_w")

    def fused_operands(self, x, segmap, slope):
        if isinstance(segmap, SegPyramid):
            seg = segmap.at(x.size(2))
        else:
            seg = segmap if segmap.shape[2:] == x.shape[2:] else ops.nearest_resize(segmap, x.shape[2:])
        sh = self.mlp_shared[0]
        actv = ops.conv2d(seg, sh.weight, sh.bias, 1, sh.padding[0], sh.act, sh.slope, grad_is_pre=True)
        w = _joined(self, "_joined_w", self.mlp_gamma.weight, self.mlp_beta.weight)
        b = _joined(self, "_joined_b", self.mlp_gamma.bias, self.mlp_beta.bias)
        pn = self.param_free_norm
        return (actv, w, b, pn.running_mean, pn.running_var, slope, sh.slope)

    def modulation(self, x, segmap):
        """gamma || beta (B, 2C, h, w) of this SPADE for a feature map shaped like x."""
        if isinstance(segmap, SegPyramid):
            seg = segmap.at(x.size(2))
        else:
            seg = segmap if segmap.shape[2:] == x.shape[2:] else ops.nearest_resize(segmap, x.shape[2:])
        # actv = ReLU(mlp_shared(seg)) has ONE consumer, the gamma||beta convolution: the ReLU derivative is folded into
        # that convolution's backward-data epilogue (in_act), and mlp_shared's backward receives the gradient of its
        # pre-activation directly (grad_is_pre) — no separate pass over the 128-channel maps
        sh = self.mlp_shared[0]
        actv = ops.conv2d(seg, sh.weight, sh.bias, 1, sh.padding[0], sh.act, sh.slope, grad_is_pre=True)
        # gamma and beta convolutions share their input: ONE convolution with 2C outputs, whose weight is the two
        # parameters laid out back to back in one allocation (no torch.cat per call)
        w = _joined(self, "_joined_w", self.mlp_gamma.weight, self.mlp_beta.weight)
        b = _joined(self, "_joined_b", self.mlp_gamma.bias, self.mlp_beta.bias)
        return ops.conv2d(actv, w, b, 1, self.pw, in_act=(sh.act, sh.slope))    # (B, 2C, h, w): gamma || beta


def spade_pair(norm_a, norm_b, x, segmap, slope_a, slope_b):
    """norm_a(x, seg) and norm_b(x, seg) of the SAME x (a residual block's norm_0 and norm_s).  In training both
    param-free BatchNorms see the same batch: one statistics pass, and a backward that folds both gradients into one
    pass over x (ops.norm_act_pair).  Anything else (eval mode: each module has its own running statistics; instance
    norm; affine norms) takes the two ordinary calls."""
    pa, pb = norm_a.param_free_norm, norm_b.param_free_norm
    same = (type(pa) is type(pb) and isinstance(pa, (SynchronizedBatchNorm2d, LocalBatchNorm2d)) and not pa.affine
            and not pb.affine and pa.training and pb.training and pa.eps == pb.eps and pa.momentum == pb.momentum
            and getattr(pa, "sync", True) == getattr(pb, "sync", True) and x.dim() == 4)
    if not same:
        return norm_a(x, segmap, fused_slope=slope_a), norm_b(x, segmap, fused_slope=slope_b)
    if norm_a.fusable(x) and norm_b.fusable(x):
        ya, yb = ops.spade_fused(x, [norm_a.fused_operands(x, segmap, slope_a), norm_b.fused_operands(x, segmap, slope_b)],
                                 pa.eps, pa.momentum, getattr(pa, "sync", True))
        return ya, yb
    return ops.norm_act_pair(x, norm_a.modulation(x, segmap), norm_b.modulation(x, segmap), pa.running_mean, pa.running_var,
                             pb.running_mean, pb.running_var, slope_a, slope_b, pa.eps, pa.momentum,
                             getattr(pa, "sync", True))
