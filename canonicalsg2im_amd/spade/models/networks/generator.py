"""AttSPADE generator (reference: spade/models/networks/generator.py:13-147)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ....sg2im.attribute_embed import AttributeEmbeddings
from ....sg2im.layers import Conv2d, GlobalAvgPool, build_cnn, build_mlp
from ....sg2im.utils import real_object_mask
from .architecture import SPADEResnetBlock
from .base_network import BaseNetwork
from .normalization import SegPyramid


class AppearanceEncoder(nn.Module):
    """Constructed by G and D but never called in their forward (generator.py:50-62); it exists
    for state_dict compatibility only."""

    def __init__(self, vocab, arch, normalization='none', activation='relu', padding='same', vecs_size=1024,
                 pooling='avg'):
        super().__init__()
        self.vocab = vocab
        cnn, channels = build_cnn(arch=arch, normalization=normalization, activation=activation, pooling=pooling,
                                  padding=padding)
        self.cnn = nn.Sequential(cnn, GlobalAvgPool(), nn.Linear(channels, vecs_size))

    def forward(self, crops):
        return self.cnn(crops)


class SPADEGenerator(BaseNetwork):
    def __init__(self, opt):
        super().__init__()
        self.attribute_embedding = AttributeEmbeddings(opt.vocab['attributes'], opt.embedding_dim)
        self.opt = opt
        nf = opt.ngf
        self.sw, self.sh = self.compute_latent_vector_size(opt)
        if opt.use_vae:
            raise NotImplementedError("--use_vae is never wired in the reference trainer (SURVEY.md §2 row 22)")
        self.fc = Conv2d(opt.semantic_nc, 16 * nf, 3, padding=1)
        self.head_0 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_0 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_1 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.up_0 = SPADEResnetBlock(16 * nf, 8 * nf, opt)
        self.up_1 = SPADEResnetBlock(8 * nf, 4 * nf, opt)
        self.up_2 = SPADEResnetBlock(4 * nf, 2 * nf, opt)
        self.up_3 = SPADEResnetBlock(2 * nf, 1 * nf, opt)
        final_nc = nf
        if opt.num_upsampling_layers == 'most':
            self.up_4 = SPADEResnetBlock(1 * nf, nf // 2, opt)
            final_nc = nf // 2
        self.conv_img = Conv2d(final_nc, 3, 3, padding=1, act=ops.ACT_TANH)      # tanh fused (generator.py:124)
        self.up = nn.Upsample(scale_factor=2)
        # unused sub-modules of the reference, kept for checkpoint compatibility (generator.py:50-62)
        self.repr_input = opt.g_mask_dim
        self.repr_net = build_mlp([self.repr_input, 64, opt.rep_size], batch_norm=opt.mlp_normalization)
        self.image_encoder = AppearanceEncoder(vocab=opt.vocab, arch='C4-64-2,C4-128-2,C4-256-2',
                                               normalization=opt.appearance_normalization,
                                               activation=opt.a_activation, padding='valid',
                                               vecs_size=opt.g_mask_dim)

    def compute_latent_vector_size(self, opt):
        n = {'normal': 5, 'more': 6, 'most': 7}.get(opt.num_upsampling_layers)
        if n is None:
            raise ValueError('opt.num_upsampling_layers [%s] not recognized' % opt.num_upsampling_layers)
        sw = opt.image_size[0] // (2 ** n)
        return sw, round(sw / opt.aspect_ratio)

    def forward(self, objs, layout_boxes, layout_masks, test_mode=False):
        if layout_masks is not None and test_mode:
            raise NotImplementedError("masks_to_layout(test_mode=True) compositing is an inference-only path")
        if self.sw != self.sh:
            raise NotImplementedError("aspect_ratio != 1 is not on the hot path")
        H = self.opt.image_size[0]
        obj_vecs = self.attribute_embedding(objs)
        valid = real_object_mask(objs, self.opt.vocab)
        sizes, h = [], self.sw
        while h <= H:
            sizes.append(h)
            h *= 2
        seg = SegPyramid(zip(sizes, ops.layout_pyramid(obj_vecs, layout_boxes, valid, H, sizes, masks=layout_masks)))
        x = self.fc(seg.at(self.sw))          # F.interpolate(seg, (sh,sw)) == pyramid level sw
        x = self.head_0(x, seg)
        x = ops.upsample2x(x)
        x = self.G_middle_0(x, seg)
        if self.opt.num_upsampling_layers in ('more', 'most'):
            x = ops.upsample2x(x)
        x = self.G_middle_1(x, seg)
        for blk in (self.up_0, self.up_1, self.up_2, self.up_3):
            x = ops.upsample2x(x)
            x = blk(x, seg)
        if self.opt.num_upsampling_layers == 'most':
            x = ops.upsample2x(x)
            x = self.up_4(x, seg)
        return self.conv_img(F.leaky_relu(x, 2e-1))
