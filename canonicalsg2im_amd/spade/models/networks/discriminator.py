"""Multiscale PatchGAN discriminator (reference: spade/models/networks/discriminator.py:66-206)."""
import numpy as np
import torch
import torch.nn as nn

from .... import ops
from ....sg2im.attribute_embed import AttributeEmbeddings
from ....sg2im.layers import Conv2d, _FusedActivation, build_mlp
from ....sg2im.utils import real_object_mask
from .base_network import BaseNetwork
from .generator import AppearanceEncoder
from .normalization import get_nonspade_norm_layer


class NLayerDiscriminator(BaseNetwork):
    """conv4x4/2 + LReLU | (n_layers-1) x [SN-conv4x4 + InstanceNorm + LReLU] | conv4x4 -> 1.
    LeakyReLU is fused into the conv epilogue (model0) or the InstanceNorm apply pass."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        kw, padw, nf = 4, int(np.ceil((4 - 1.0) / 2)), opt.ndf
        norm_layer = get_nonspade_norm_layer(opt, opt.norm_D)
        seq = [[Conv2d(self.compute_D_input_nc(), nf, kernel_size=kw, stride=2, padding=padw, act=ops.ACT_LEAKY,
                       slope=0.2), _FusedActivation()]]
        for n in range(1, opt.n_layers_D):
            nf_prev, nf = nf, min(nf * 2, 512)
            stride = 1 if n == opt.n_layers_D - 1 else 2
            block = norm_layer(Conv2d(nf_prev, nf, kernel_size=kw, stride=stride, padding=padw))
            if isinstance(block, nn.Sequential):
                block[1].fused_slope = 0.2
                seq += [[block, _FusedActivation()]]
            else:
                seq += [[block, nn.LeakyReLU(0.2, False)]]
        seq += [[Conv2d(nf, 1, kernel_size=kw, stride=1, padding=padw)]]
        for n, layers in enumerate(seq):
            self.add_module('model' + str(n), nn.Sequential(*layers))

    def compute_D_input_nc(self):
        return self.opt.semantic_nc + 3

    def forward(self, input, seg_first=0):
        """`seg_first=S`: `input` is the packed buffer of ops.disc_input ([layout(S)|img(3)|pad]); the
        first conv's weight (trained on cat([img, layout])) is permuted to that channel order."""
        results = [input]
        for name, sub in self.named_children():
            x = results[-1]
            if name == 'model0' and seg_first:
                conv = sub[0]
                w = conv.weight
                pad = x.size(1) - w.size(1)
                parts = [w[:, 3:3 + seg_first], w[:, :3]]
                if pad:
                    parts.append(w.new_zeros(w.size(0), pad, w.size(2), w.size(3)))
                x = ops.conv2d(x, torch.cat(parts, dim=1), conv.bias, conv.stride[0], conv.padding[0], conv.act,
                               conv.slope)
            else:
                x = sub(x)
            results.append(x)
        return results[1:] if not self.opt.no_ganFeat_loss else results[-1]


class MultiscaleDiscriminator(BaseNetwork):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.attribute_embedding = AttributeEmbeddings(opt.vocab['attributes'], opt.embedding_dim,
                                                       use_attr_fc_gen=True)
        # never used in forward; kept for checkpoint compatibility (discriminator.py:74-87)
        self.repr_input = opt.g_mask_dim
        self.repr_net = build_mlp([self.repr_input, 64, opt.rep_size], batch_norm=opt.mlp_normalization)
        self.image_encoder = AppearanceEncoder(vocab=opt.vocab, arch='C4-64-2,C4-128-2,C4-256-2',
                                               normalization=opt.appearance_normalization,
                                               activation=opt.a_activation, padding='valid',
                                               vecs_size=opt.g_mask_dim)
        for i in range(opt.num_D):
            self.add_module('discriminator_%d' % i, NLayerDiscriminator(opt))

    def downsample(self, input):
        return ops.avgpool3s2(input)

    def forward(self, img, objs, layout_boxes, layout_masks=None, gt_train=True, fool=False):
        """`fool` is accepted and ignored, as in the reference."""
        obj_vecs = self.attribute_embedding(objs)
        valid = real_object_mask(objs, self.opt.vocab)
        S = obj_vecs.size(-1)
        x = ops.disc_input(img, obj_vecs, layout_boxes, valid, self.opt.image_size[0], masks=layout_masks)
        result = []
        for name, D in self.named_children():
            if name.startswith('discriminator'):
                out = D(x, seg_first=S)
                result.append(out if not self.opt.no_ganFeat_loss else [out])
                x = self.downsample(x)
        return result


class AcCropDiscriminator(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("the object-crop discriminator is the next-row component (SURVEY.md §8f rank 1); "
                                  "train with --use_img_disc 1")


class AcDiscriminator(AcCropDiscriminator):
    pass


class MultiscaleMaskDiscriminator2(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("the mask discriminator needs --mask_size > 0 (SURVEY.md §8f rank 4)")
