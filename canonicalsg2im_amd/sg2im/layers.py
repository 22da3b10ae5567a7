"""Layer builders of the hot path (reference: sg2im/layers.py).

`Linear` and `Conv2d` keep nn.Linear / nn.Conv2d's parameters, init and state_dict keys but run
on the fp32-MFMA implicit-GEMM kernel with bias and activation fused in the epilogue."""
import torch
import torch.nn as nn

from .. import ops


class Linear(nn.Linear):
    def __init__(self, din, dout, bias=True, fused_relu=False):
        super().__init__(din, dout, bias=bias)
        self.fused_relu = fused_relu

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias, ops.ACT_LEAKY if self.fused_relu else ops.ACT_NONE, 0.0)


class Conv2d(nn.Conv2d):
    """nn.Conv2d surface; `act`/`slope` fuse ReLU / LeakyReLU / tanh into the epilogue and
    `residual` adds a skip tensor there."""

    def __init__(self, *args, act=ops.ACT_NONE, slope=0.0, **kwargs):
        super().__init__(*args, **kwargs)
        self.act, self.slope = act, slope
        k, s, p = self.kernel_size, self.stride, self.padding
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or self.dilation != (1, 1) or self.groups != 1:
            raise NotImplementedError("Conv2d: only square, undilated, ungrouped convolutions are on the hot path")
        # channels-last parameter memory == the kernels' [Cout][KH][KW][Cin] operand AND the weight-gradient kernels'
        # output: no repack in the forward, and autograd adopts dW as .grad without a layout copy (same values, same
        # state_dict; `spectral_norm` puts its `weight_orig` back to row-major, the order of its u / v vectors)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x, residual=None):
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.act, self.slope, residual)


class _FusedActivation(nn.Identity):
    """Placeholder that keeps nn.Sequential indices (state_dict keys `net.0`, `net.2`) where the
    reference has a ReLU module; the activation itself runs in the preceding GEMM's epilogue."""


def get_activation(name):
    kwargs = {}
    if name.lower().startswith('leakyrelu'):
        if '-' in name:
            kwargs = {'negative_slope': float(name.split('-')[1])}
        name = 'leakyrelu'
    table = {'relu': nn.ReLU, 'leakyrelu': nn.LeakyReLU, 'sigmoid': nn.Sigmoid}
    if name.lower() not in table:
        raise ValueError('Invalid activation "%s"' % name)
    return table[name.lower()](**kwargs)


def build_mlp(dim_list, activation='relu', batch_norm='none', dropout=0, final_nonlinearity='relu'):
    """[Linear, (BatchNorm1d), ReLU]* Linear [ReLU] (reference sg2im/layers.py:6-25) with the same nn.Sequential
    indices (state_dict keys `net.0`, `net.2` — or `net.0`, `net.1`, `net.3` with batch_norm='batch').  Every ReLU is
    fused: into the GEMM epilogue, or into the BatchNorm apply pass when a BatchNorm1d sits in between."""
    if dropout > 0 or activation != 'relu' or final_nonlinearity not in (None, 'relu'):
        raise NotImplementedError("build_mlp: only relu activations without dropout (the trainer's settings) are "
                                  "on the hot path")
    if batch_norm not in ('none', 'batch'):
        raise ValueError('Invalid mlp normalization "%s"' % batch_norm)
    layers = []
    n = len(dim_list) - 1
    for i in range(n):
        last = i == n - 1
        bn = (not last) and batch_norm == 'batch'
        relu = (not last) or final_nonlinearity == 'relu'
        layers.append(Linear(dim_list[i], dim_list[i + 1], fused_relu=relu and not bn))
        if bn:
            layers.append(BatchNorm1dAct(dim_list[i + 1], fused_slope=0.0))
        if relu:
            layers.append(_FusedActivation())
    return nn.Sequential(*layers)


class GlobalAvgPool(nn.Module):
    def forward(self, x):
        return x.reshape(x.size(0), x.size(1), -1).mean(dim=2)


class Interpolate(nn.Module):
    def __init__(self, size=None, scale_factor=None, mode='nearest', align_corners=None):
        super().__init__()
        self.size, self.scale_factor, self.mode, self.align_corners = size, scale_factor, mode, align_corners

    def forward(self, x):
        if self.mode == 'nearest' and self.scale_factor == 2 and x.is_cuda:
            return ops.upsample2x(x)
        return nn.functional.interpolate(x, size=self.size, scale_factor=self.scale_factor, mode=self.mode,
                                         align_corners=self.align_corners)


def affine_batch_norm(x, weight, bias, running_mean, running_var, training, slope, eps, momentum, sync):
    """Batch normalisation with a per-channel affine on the HIP statistics + apply kernels.  x is (N,C), (N,C,L) or
    (N,C,H,W) — channels in dim 1, as torch's BatchNorm{1,2}d.  The affine is fed to the modulation kernel as a
    broadcast gamma||beta map (y = xhat * (1 + (w - 1)) + b): these layers only see small tensors (object crops,
    per-object vectors), so the extra map costs nothing measurable."""
    if not x.is_cuda:
        raise RuntimeError("canonicalsg2im_amd ops need HIP (cuda) tensors; there is no CPU path")
    shape = x.shape
    x4 = x if x.dim() == 4 else x.reshape(shape[0], shape[1], -1, 1)
    B, C, H, W = x4.shape
    gb = torch.cat([weight - 1.0, bias]).view(1, 2 * C, 1, 1).expand(B, 2 * C, H, W)
    y = ops.norm_act(x4, gb, running_mean, running_var, instance=False, training=training, slope=slope, eps=eps,
                     momentum=momentum, sync=sync)
    return y if x.dim() == 4 else y.reshape(shape)


class BatchNormAct(nn.BatchNorm2d):
    """nn.BatchNorm2d (affine, running stats) whose forward is the fused HIP statistics + apply pass;
    a following LeakyReLU can be folded in with `fused_slope`."""

    def __init__(self, num_features, fused_slope=1.0):
        super().__init__(num_features)
        self.fused_slope = fused_slope

    def forward(self, x):
        if self.training and self.track_running_stats:
            self.num_batches_tracked.add_(1)
        return affine_batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                 self.fused_slope, self.eps, self.momentum, sync=False)


class BatchNorm1dAct(nn.BatchNorm1d):
    """nn.BatchNorm1d of `build_mlp(batch_norm='batch')` (reference sg2im/layers.py:13-14) with the ReLU behind it
    folded in (`fused_slope=0`)."""

    def __init__(self, num_features, fused_slope=1.0):
        super().__init__(num_features)
        self.fused_slope = fused_slope

    def forward(self, x):
        self._check_input_dim(x)
        if x.size(1) != self.num_features:          # what F.batch_norm reports for the reference's module
            raise RuntimeError("running_mean should contain %d elements not %d" % (x.size(1), self.num_features))
        if self.training and self.track_running_stats:
            self.num_batches_tracked.add_(1)
        return affine_batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                 self.fused_slope, self.eps, self.momentum, sync=False)


def build_hot_cnn(arch, normalization='batch', activation='leakyrelu-0.2', padding='valid'):
    """build_cnn for the object discriminator (arch 'C4-64-2,C4-128-2,C4-256-2'): same nn.Sequential
    indices and state_dict keys as the reference builder, executed on the HIP kernels — Conv2d on the
    implicit GEMM, BatchNorm + LeakyReLU fused in one pass."""
    if isinstance(arch, str):
        arch = arch.split(',')
    if normalization != 'batch' or not activation.lower().startswith('leakyrelu'):
        raise NotImplementedError("object discriminator: only d_normalization=batch, d_activation=leakyrelu-* "
                                  "(the trainer defaults) are on the hot path")
    slope = float(activation.split('-')[1]) if '-' in activation else 0.01
    cur, first, layers = 3, True, []
    for s in arch:
        if s[0] != 'C':
            raise NotImplementedError('build_hot_cnn: layer "%s" is not on the hot path' % s)
        if not first:
            layers.append(BatchNormAct(cur, fused_slope=slope))
            layers.append(_FusedActivation())
        first = False
        vals = [int(v) for v in s[1:].split('-')]
        K, nxt = vals[0], vals[1]
        stride = vals[2] if len(vals) == 3 else 1
        pad = (K - 1) // 2 if padding == 'same' else 0
        layers.append(Conv2d(cur, nxt, kernel_size=K, padding=pad, stride=stride))
        cur = nxt
    return nn.Sequential(*layers), cur


def build_cnn(arch, normalization='batch', activation='relu', padding='same', pooling='max', init='default'):
    """Arch-string CNN ('C4-64-2,C4-128-2,...', reference sg2im/layers.py:28-112).  On the hot path it
    only creates the never-executed `image_encoder` parameters of G and D (generator.py:50-62), so
    plain nn layers are used; the C/P/U/FC subset of the grammar is supported."""
    if isinstance(arch, str):
        arch = arch.split(',')
    cur = 3
    if arch and arch[0][0] == 'I':
        cur = int(arch[0][1:])
        arch = arch[1:]
    first, layers = True, []
    for i, s in enumerate(arch):
        if s[0] == 'C':
            if not first:
                if normalization == 'batch':
                    layers.append(nn.BatchNorm2d(cur))
                elif normalization == 'instance':
                    layers.append(nn.InstanceNorm2d(cur))
                layers.append(get_activation(activation))
            first = False
            vals = [int(v) for v in s[1:].split('-')]
            K, nxt = vals[0], vals[1]
            stride = vals[2] if len(vals) == 3 else 1
            pad = (K - 1) // 2 if padding == 'same' else 0
            layers.append(nn.Conv2d(cur, nxt, kernel_size=K, padding=pad, stride=stride))
            cur = nxt
        elif s[0] == 'U':
            layers.append(Interpolate(scale_factor=int(s[1:]), mode='nearest'))
        elif s[0] == 'P':
            f = int(s[1:])
            layers.append(nn.MaxPool2d(f, f) if pooling == 'max' else nn.AvgPool2d(f, f))
        else:
            raise NotImplementedError('build_cnn: layer "%s" is not on the hot path' % s)
    return nn.Sequential(*layers), cur
