"""AttSPADE generator (reference: spade/models/networks/generator.py:13-147)."""
import torch.nn as nn

from .... import ops
from .... import spectral_norm as csg_spectral_norm
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


# (attribute name, input width, output width) in units of ngf, in execution order — the attribute
# names are the reference's (generator.py:30-44) and therefore the checkpoint keys
_BLOCKS = (("head_0", 16, 16), ("G_middle_0", 16, 16), ("G_middle_1", 16, 16),
           ("up_0", 16, 8), ("up_1", 8, 4), ("up_2", 4, 2), ("up_3", 2, 1))
_LATENT_HALVINGS = {'normal': 5, 'more': 6, 'most': 7}          # generator.py:64-77


class SPADEGenerator(BaseNetwork):
    """layout -> fc on the coarsest pyramid level -> 7 (8) SPADE residual blocks with nearest x2 upsampling in
    between -> LeakyReLU -> conv -> tanh (reference generator.py:79-127).  The layout is produced once, at every
    resolution the blocks need, by one kernel (`ops.layout_pyramid`)."""

    def __init__(self, opt):
        super().__init__()
        if opt.use_vae:
            raise NotImplementedError("--use_vae is never wired in the reference trainer (SURVEY.md §2 row 22)")
        self.attribute_embedding = AttributeEmbeddings(opt.vocab['attributes'], opt.embedding_dim)
        self.opt = opt
        nf = opt.ngf
        self.sw, self.sh = self.compute_latent_vector_size(opt)
        self.fc = Conv2d(opt.semantic_nc, 16 * nf, 3, padding=1)
        blocks = list(_BLOCKS)
        if opt.num_upsampling_layers == 'most':
            blocks.append(("up_4", 1, 0.5))
        for name, win, wout in blocks:
            setattr(self, name, SPADEResnetBlock(int(win * nf), int(wout * nf), opt))
        self._block_names = [b[0] for b in blocks]
        self.conv_img = Conv2d(int(blocks[-1][2] * nf), 3, 3, padding=1, act=ops.ACT_TANH)   # tanh fused (:124)
        self.up = nn.Upsample(scale_factor=2)       # parameter-free; kept because the reference registers it
        # never executed by the reference either (generator.py:50-62): present for checkpoint compatibility
        self.repr_input = opt.g_mask_dim
        self.repr_net = build_mlp([self.repr_input, 64, opt.rep_size], batch_norm=opt.mlp_normalization)
        self.image_encoder = AppearanceEncoder(vocab=opt.vocab, arch='C4-64-2,C4-128-2,C4-256-2', padding='valid',
                                               normalization=opt.appearance_normalization,
                                               activation=opt.a_activation, vecs_size=opt.g_mask_dim)

    def compute_latent_vector_size(self, opt):
        if opt.num_upsampling_layers not in _LATENT_HALVINGS:
            raise ValueError('opt.num_upsampling_layers [%s] not recognized' % opt.num_upsampling_layers)
        sw = opt.image_size[0] >> _LATENT_HALVINGS[opt.num_upsampling_layers]
        return sw, round(sw / opt.aspect_ratio)

    def forward(self, objs, layout_boxes, layout_masks, test_mode=False):
        if self.sw != self.sh:
            raise NotImplementedError("aspect_ratio != 1 is not on the hot path")
        csg_spectral_norm.prepare(self)              # all 18 spectrally normalised weights in one launch per stage
        ops.prepack_weights(self)                    # and the Winograd operands of every 3x3 weight in one launch per 24
        H = self.opt.image_size[0]
        levels = [self.sw << k for k in range(H.bit_length()) if (self.sw << k) <= H]
        valid = real_object_mask(objs, self.opt.vocab)
        if layout_masks is not None and test_mode:          # painter's compositing (layout.py:135-151), inference only
            maps = ops.layout_paint(self.attribute_embedding(objs), layout_boxes, valid, layout_masks, H, levels)
        else:
            maps = ops.layout_pyramid(self.attribute_embedding(objs), layout_boxes, valid, H, levels, masks=layout_masks)
        seg = SegPyramid(zip(levels, maps))
        x = self.fc(seg.at(self.sw))                 # nearest resize to (sh, sw) == the coarsest pyramid level
        # x2 upsampling precedes every block except head_0 and — unless 'more'/'most' — G_middle_1
        no_upsample = {"head_0"} | (set() if self.opt.num_upsampling_layers in ('more', 'most') else {"G_middle_1"})
        for name in self._block_names:
            if name not in no_upsample:
                x = ops.upsample2x(x)
            x = getattr(self, name)(x, seg)
        return self.conv_img(x, pre_slope=2e-1)          # LeakyReLU(0.2) of generator.py:123 rides in conv_img's loaders
