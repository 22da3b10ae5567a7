"""GAN objectives (reference: spade/models/networks/loss.py).  Elementwise + tiny reductions on
already-computed maps: plain tensor ops."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class GANLoss(nn.Module):
    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor, opt=None):
        super().__init__()
        if gan_mode not in ('ls', 'original', 'w', 'hinge'):
            raise ValueError('Unexpected gan_mode {}'.format(gan_mode))
        self.real_label, self.fake_label = target_real_label, target_fake_label
        self.gan_mode, self.opt = gan_mode, opt

    def loss(self, input, target_is_real, for_discriminator=True):
        if self.gan_mode == 'original':
            target = torch.full_like(input, self.real_label if target_is_real else self.fake_label)
            return F.binary_cross_entropy_with_logits(input, target)
        if self.gan_mode == 'ls':
            target = torch.full_like(input, self.real_label if target_is_real else self.fake_label)
            return F.mse_loss(input, target)
        if self.gan_mode == 'hinge':
            if for_discriminator:
                x = input - 1 if target_is_real else -input - 1
                return -torch.mean(torch.clamp(x, max=0.0))
            assert target_is_real, "The generator's hinge loss must be aiming for real"
            return -torch.mean(input)
        return -input.mean() if target_is_real else input.mean()

    def __call__(self, input, target_is_real, for_discriminator=True):
        if isinstance(input, list):
            loss = 0
            for pred_i in input:
                if isinstance(pred_i, list):
                    pred_i = pred_i[-1]
                loss = loss + self.loss(pred_i, target_is_real, for_discriminator).reshape(1)
            return loss / len(input)
        return self.loss(input, target_is_real, for_discriminator)


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
