"""Layer builders of the hot path (reference: sg2im/layers.py).

`Linear` and `Conv2d` keep nn.Linear / nn.Conv2d's parameters, init and state_dict keys but run
on the fp32-MFMA implicit-GEMM kernel with bias and activation fused in the epilogue."""
import torch
import torch.nn as nn

from .. import ops


class Linear(nn.Linear):
    """nn.Linear surface; `fused_relu` / `fused_slope` fuse a ReLU / LeakyReLU(slope) into the GEMM epilogue."""

    def __init__(self, din, dout, bias=True, fused_relu=False, fused_slope=None):
        super().__init__(din, dout, bias=bias)
        self.fused_relu = fused_relu
        self.fused_slope = 0.0 if (fused_relu and fused_slope is None) else fused_slope
        # set by build_mlp for a Linear -> (Leaky)ReLU -> Linear chain with nothing in between: the second Linear's
        # backward-data pass applies the activation derivative in its epilogue (`in_slope`: the slope of the activation
        # that produced its input), the first receives the gradient of its pre-activation (`grad_is_pre`) — no separate
        # activation-derivative pass over the hidden tensor
        self.in_slope = None
        self.grad_is_pre = False

    def forward(self, x, grad_is_pre=None):
        """`grad_is_pre=True` (per call): the caller's consumer of this layer's fused activation returns the gradient of the
        pre-activation itself (GraphTripleConv: ops.segment_avg(h_is_relu=True))."""
        pre = self.grad_is_pre if grad_is_pre is None else bool(grad_is_pre)
        in_act = (ops.ACT_LEAKY, float(self.in_slope)) if (self.in_slope is not None and x.requires_grad) else None
        if self.fused_slope is None:
            return ops.linear(x, self.weight, self.bias, ops.ACT_NONE, 0.0, in_act=in_act)
        return ops.linear(x, self.weight, self.bias, ops.ACT_LEAKY, float(self.fused_slope), in_act=in_act, grad_is_pre=pre)


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

    def forward(self, x, residual=None, pre_slope=None):
        """`pre_slope`: the layer sees leaky_relu(x, pre_slope) (folded into the kernel's loaders where there is one)."""
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.act, self.slope, residual,
                          pre_slope=pre_slope)


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


def _fusable_slope(name):
    """Negative slope of an activation the kernels fuse (ReLU = 0, LeakyReLU-x = x), else None."""
    if name is None:
        return None
    low = name.lower()
    if low == 'relu':
        return 0.0
    if low.startswith('leakyrelu'):
        return float(name.split('-')[1]) if '-' in name else 0.01
    return None


def build_mlp(dim_list, activation='relu', batch_norm='none', dropout=0, final_nonlinearity='relu'):
    """[Linear, (BatchNorm1d), act, (Dropout)]* Linear (Dropout) [act] (reference sg2im/layers.py:6-25) with the same
    nn.Sequential indices (state_dict keys `net.0`, `net.2` — or `net.0`, `net.1`, `net.3` with batch_norm='batch').
    ReLU / LeakyReLU-x are fused: into the GEMM epilogue, or into the BatchNorm apply pass when a BatchNorm1d sits in
    between (a `_FusedActivation` placeholder keeps the index); sigmoid and dropout — not used by any recipe of the
    reference — are plain torch modules on the (per-object, tiny) outputs of the HIP GEMM."""
    if batch_norm not in ('none', 'batch'):
        raise ValueError('Invalid mlp normalization "%s"' % batch_norm)
    if activation is not None:
        get_activation(activation)                       # raises ValueError('Invalid activation ...') like the reference
    if final_nonlinearity is not None:
        get_activation(final_nonlinearity)
    layers = []
    n = len(dim_list) - 1
    for i in range(n):
        last = i == n - 1
        bn = (not last) and batch_norm == 'batch'
        inner = None if last else activation
        # the final non-linearity sits behind the last layer's dropout in the reference: fusable only without dropout
        tail = final_nonlinearity if (last and dropout == 0) else None
        slope = _fusable_slope(inner if not last else tail)
        layers.append(Linear(dim_list[i], dim_list[i + 1], fused_slope=None if bn else slope))
        if bn:
            layers.append(BatchNorm1dAct(dim_list[i + 1], fused_slope=1.0 if slope is None else slope))
        if inner is not None:
            layers.append(_FusedActivation() if slope is not None else get_activation(inner))
        if dropout > 0:
            layers.append(nn.Dropout(p=dropout))
        if last and final_nonlinearity is not None:
            fused = tail is not None and slope is not None
            layers.append(_FusedActivation() if fused else get_activation(final_nonlinearity))
    # Linear -> fused (Leaky)ReLU -> Linear with nothing in between: hand the activation derivative to the consumer
    linears = [m for m in layers if isinstance(m, Linear)]
    if batch_norm == 'none' and dropout == 0:
        for a, b in zip(linears[:-1], linears[1:]):
            if a.fused_slope is not None and a.out_features % 4 == 0:
                a.grad_is_pre, b.in_slope = True, a.fused_slope
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
        # the device kernel samples at floor(dst * in / out) — F.interpolate's rule when a SIZE is given; with a
        # scale_factor (and no recompute_scale_factor) torch samples at floor(dst / scale_factor), which is the same only
        # when in * scale_factor is an integer in both dimensions
        exact = self.size is not None or (float(x.size(2) * self.scale_factor).is_integer()
                                          and float(x.size(3) * self.scale_factor).is_integer())
        if self.mode == 'nearest' and x.is_cuda and x.dim() == 4 and x.size(1) % 4 == 0 and exact:
            size = self.size if self.size is not None else (int(x.size(2) * self.scale_factor), int(x.size(3) * self.scale_factor))
            size = (size, size) if isinstance(size, int) else size
            return ops.nearest_resize(x, size)
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


class InstanceNorm2dAct(nn.InstanceNorm2d):
    """nn.InstanceNorm2d (no affine, no running statistics: `get_normalization_2d(.., 'instance')`) on the HIP
    statistics + apply kernels, with an optional fused LeakyReLU (`fused_slope`)."""

    def __init__(self, num_features, fused_slope=1.0):
        super().__init__(num_features)
        self.fused_slope = fused_slope

    def forward(self, x):
        return ops.norm_act(x, None, None, None, instance=True, training=self.training, slope=self.fused_slope, eps=self.eps)


class MaxPool2(nn.Module):
    """nn.MaxPool2d(kernel_size=2, stride=2) on csrc/perceptual.hip (first-maximum routing like ATen)."""

    def forward(self, x):
        return ops.maxpool2(x)


class AvgPool2(nn.Module):
    """nn.AvgPool2d(kernel_size=2, stride=2) on csrc/perceptual.hip (`build_cnn(pooling='avg')`)."""

    def forward(self, x):
        return ops.avgpool2(x)


class Flatten(nn.Module):
    """`x.view(N, -1)` of the reference (sg2im/layers.py:160-165): the values in (C, H, W) order, whatever the memory
    format of the activation is here."""

    def forward(self, x):
        return x.reshape(x.size(0), -1)

    def __repr__(self):
        return 'Flatten()'


class ResidualBlock(nn.Module):
    """sg2im/layers.py:190-217: [norm, act, conv, norm, act, conv] (the norms absent with normalization='none') and
    `shortcut + self.net(x)`.  Two properties of the reference are kept on purpose: `self.net(x)` is evaluated TWICE per
    call (:214-215; the first result is dropped, but in training mode the BatchNorm running statistics advance twice), and
    padding='valid' slices an empty shortcut (`x[:, :, 0:-0, 0:-0]`) — it fails there, so it is refused here.  The skip
    addition rides in the last convolution's epilogue.  Same nn.Sequential indices, hence the same state_dict keys."""

    def __init__(self, channels, normalization='batch', activation='relu', padding='same', kernel_size=3, init='default'):
        super().__init__()
        if padding != 'same':
            raise NotImplementedError("ResidualBlock: padding='valid' slices an empty shortcut in the reference "
                                      "(sg2im/layers.py:211-213) and cannot run there either")
        K, C = kernel_size, channels
        assert K % 2 == 1, 'Invalid kernel size %d for "same" padding' % K
        self.padding = (K - 1) // 2
        slope = _fusable_slope(activation)
        layers = []
        for _ in range(2):
            if normalization == 'batch':
                layers.append(BatchNormAct(C, fused_slope=1.0 if slope is None else slope))
            elif normalization == 'instance':
                layers.append(InstanceNorm2dAct(C, fused_slope=1.0 if slope is None else slope))
            elif normalization != 'none':
                raise ValueError('Unrecognized normalization type "%s"' % normalization)
            fused = slope is not None and normalization != 'none'
            layers.append(_FusedActivation() if fused else get_activation(activation))
            conv = Conv2d(C, C, kernel_size=K, padding=self.padding)
            if init == 'kaiming-normal':
                nn.init.kaiming_normal_(conv.weight)
            elif init == 'kaiming-uniform':
                nn.init.kaiming_uniform_(conv.weight)
            layers.append(conv)
        # the reference filters absent norms out BEFORE building the Sequential: indices 0..3 then, 0..5 with norms
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        mods = list(self.net)
        for rep in range(2):                       # (:214-215) the block runs its net twice; the first result is dropped
            h = x
            for m in mods[:-1]:
                h = m(h)
            if rep == 1:
                return mods[-1](h, residual=x)
            mods[-1](h)


def build_cnn(arch, normalization='batch', activation='relu', padding='same', pooling='max', init='default'):
    """Arch-string CNN (reference sg2im/layers.py:28-112) with the same nn.Sequential indices and state_dict keys as the
    reference builder, executed on the HIP kernels: IX (input channels), CK-X[-S] (Conv2d on the implicit GEMM /
    Winograd kernels; every convolution except the first is preceded by normalisation and non-linearity — BatchNorm2d
    or InstanceNorm2d with ReLU / LeakyReLU-x fused into its apply pass, or, with normalization='none', the activation
    fused into the previous convolution's epilogue), R (ResidualBlock), UX (nearest-neighbour upsampling), P2 with max
    or average pooling, FC-X-Y (Flatten + Linear, activated unless it closes the network).  Pooling factors other than 2
    are used by no call site of the reference and raise."""
    if isinstance(arch, str):
        arch = arch.split(',')
    if normalization not in ('batch', 'instance', 'none'):
        raise ValueError('Unrecognized normalization type "%s"' % normalization)
    get_activation(activation)                                # ValueError on an unknown name, like the reference
    slope = _fusable_slope(activation)
    cur = 3
    if len(arch) > 0 and arch[0][0] == 'I':
        cur = int(arch[0][1:])
        arch = arch[1:]
    first, layers, prev_conv, flat = True, [], None, False
    for idx, s in enumerate(arch):
        if s[0] == 'C':
            if not first:
                if normalization == 'batch':
                    layers.append(BatchNormAct(cur, fused_slope=1.0 if slope is None else slope))
                elif normalization == 'instance':
                    layers.append(InstanceNorm2dAct(cur, fused_slope=1.0 if slope is None else slope))
                elif slope is not None and prev_conv is not None and prev_conv is layers[-1]:
                    prev_conv.act, prev_conv.slope = ops.ACT_LEAKY, slope      # no norm in between: the conv's own epilogue
                fused = slope is not None and (normalization != 'none' or (prev_conv is not None and prev_conv is layers[-1]))
                layers.append(_FusedActivation() if fused else get_activation(activation))
            first = False
            vals = [int(v) for v in s[1:].split('-')]
            K, nxt = vals[0], vals[1]
            stride = vals[2] if len(vals) == 3 else 1
            if padding == 'same':
                assert K % 2 == 1, 'Invalid kernel size %d for "same" padding' % K
            pad = (K - 1) // 2 if padding == 'same' else 0
            prev_conv = Conv2d(cur, nxt, kernel_size=K, padding=pad, stride=stride)
            if init == 'kaiming-normal':
                nn.init.kaiming_normal_(prev_conv.weight)
            elif init == 'kaiming-uniform':
                nn.init.kaiming_uniform_(prev_conv.weight)
            layers.append(prev_conv)
            cur = nxt
        elif s[0] == 'U':
            layers.append(Interpolate(scale_factor=int(s[1:]), mode='nearest'))
        elif s[0] == 'R':
            # (normalization is dropped for a block that opens the network, as in the reference)
            layers.append(ResidualBlock(cur, normalization='none' if first else normalization, activation=activation,
                                        padding=padding, init=init))
            first = False
            prev_conv = None
        elif s[0] == 'P' and int(s[1:]) == 2 and pooling in ('max', 'avg'):
            layers.append(MaxPool2() if pooling == 'max' else AvgPool2())
        elif s[:2] == 'FC':
            _, din, dout = s.split('-')
            din, dout = int(din), int(dout)
            if not flat:
                layers.append(Flatten())
            flat = True
            more = idx + 1 < len(arch)                       # every FC except the network's last layer is activated
            lin = Linear(din, dout, fused_slope=slope if more else None)
            layers.append(lin)
            if more:
                layers.append(_FusedActivation() if slope is not None else get_activation(activation))
            cur = dout
            prev_conv = None
        elif s[0] == 'P':
            raise NotImplementedError('build_cnn: pooling layer "%s" (pooling=%s): only factor 2, max or avg, is built on the '
                                      'HIP kernels (no call site of the reference uses another)' % (s, pooling))
        else:
            raise ValueError('Invalid layer "%s"' % s)
    return nn.Sequential(*layers), cur


def build_hot_cnn(arch, normalization='batch', activation='leakyrelu-0.2', padding='valid'):
    """The object discriminator's CNN (arch 'C4-64-2,C4-128-2,C4-256-2'): `build_cnn` under its round-1 name."""
    return build_cnn(arch, normalization=normalization, activation=activation, padding=padding)
