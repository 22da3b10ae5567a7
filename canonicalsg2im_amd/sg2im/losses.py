"""Classic GAN losses used by the object discriminator (reference: sg2im/losses.py)."""
import torch


def bce_loss(input, target):
    neg_abs = -input.abs()
    return (input.clamp(min=0) - input * target + (1 + neg_abs.exp()).log()).mean()


def gan_g_loss(scores_fake):
    s = scores_fake.reshape(-1)
    return bce_loss(s, torch.ones_like(s))


def gan_d_loss(scores_real, scores_fake):
    assert scores_real.size() == scores_fake.size()
    r, f = scores_real.reshape(-1), scores_fake.reshape(-1)
    return bce_loss(r, torch.ones_like(r)) + bce_loss(f, torch.zeros_like(f))


def wgan_g_loss(scores_fake):
    return -scores_fake.mean()


def wgan_d_loss(scores_real, scores_fake):
    return scores_fake.mean() - scores_real.mean()


def lsgan_g_loss(scores_fake):
    s = scores_fake.reshape(-1)
    return torch.nn.functional.mse_loss(s.sigmoid(), torch.ones_like(s))


def lsgan_d_loss(scores_real, scores_fake):
    r, f = scores_real.reshape(-1), scores_fake.reshape(-1)
    return torch.nn.functional.mse_loss(r.sigmoid(), torch.ones_like(r)) + \
        torch.nn.functional.mse_loss(f.sigmoid(), torch.zeros_like(f))


def get_gan_losses(gan_type):
    table = {'gan': (gan_g_loss, gan_d_loss), 'wgan': (wgan_g_loss, wgan_d_loss), 'lsgan': (lsgan_g_loss, lsgan_d_loss)}
    if gan_type not in table:
        raise ValueError('Unrecognized GAN type "%s"' % gan_type)
    return table[gan_type]
