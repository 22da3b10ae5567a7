"""GAN objectives (reference: spade/models/networks/loss.py).  Elementwise + tiny reductions on
already-computed maps: plain tensor ops."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class GANLoss(nn.Module):
    """`GANLoss(gan_mode)(prediction, target_is_real, for_discriminator)` of the reference (loss.py:13-98).
    A multiscale prediction (list over scales of lists of feature maps) is reduced to the mean over scales
    of the loss on each scale's LAST map, kept as a 1-element tensor like the reference's."""

    MODES = ('ls', 'original', 'w', 'hinge')

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor, opt=None):
        super().__init__()
        if gan_mode not in self.MODES:
            raise ValueError('Unexpected gan_mode {}'.format(gan_mode))
        self.gan_mode, self.opt = gan_mode, opt
        self.real_label, self.fake_label = target_real_label, target_fake_label

    def loss(self, input, target_is_real, for_discriminator=True):
        mode = self.gan_mode
        if mode == 'hinge':
            if not for_discriminator:
                assert target_is_real, "The generator's hinge loss must be aiming for real"
                return -input.mean()
            margin = (input - 1) if target_is_real else (-input - 1)
            return -torch.clamp(margin, max=0.0).mean()
        if mode == 'w':
            return -input.mean() if target_is_real else input.mean()
        label = torch.full_like(input, self.real_label if target_is_real else self.fake_label)
        return F.binary_cross_entropy_with_logits(input, label) if mode == 'original' else F.mse_loss(input, label)

    def _kind(self, target_is_real, for_discriminator):
        """ops.hinge_mean's term for this call, or None (modes / calls it does not serve)."""
        if self.gan_mode == 'hinge':
            if not for_discriminator:
                return 0 if target_is_real else None
            return 1 if target_is_real else 2
        return None

    def __call__(self, input, target_is_real, for_discriminator=True):
        if not isinstance(input, list):
            return self.loss(input, target_is_real, for_discriminator)
        kind = self._kind(target_is_real, for_discriminator)
        if kind is not None:
            from .... import ops
            fused = ops.hinge_mean([p[-1] if isinstance(p, list) else p for p in input], kind)   # every scale in one launch
            if fused is not None:
                return fused
        per_scale = [self.loss(p[-1] if isinstance(p, list) else p, target_is_real, for_discriminator).reshape(1)
                     for p in input]
        total = 0
        for term in per_scale:
            total = total + term
        return total / len(input)


class VGGLoss(nn.Module):
    """Reference: spade/models/networks/loss.py:102-117 — sum_i w_i * L1(vgg(x)[i], vgg(y)[i].detach()),
    w = 1/32, 1/16, 1/8, 1/4, 1.  The target pass runs without autograd (its features are constants)."""

    def __init__(self, gpu_ids, weights=None):
        super().__init__()
        from .architecture import VGG19
        self.vgg = VGG19(weights=weights)
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def forward(self, x, y):
        from .... import ops
        x_vgg = self.vgg(x)
        with torch.no_grad():
            y_vgg = self.vgg(y)
        loss = 0
        for w, fx, fy in zip(self.weights, x_vgg, y_vgg):
            loss = loss + w * ops.l1_mean(fx, fy)
        return loss


class KLDLoss(nn.Module):
    def forward(self, mu, logvar):
        return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
