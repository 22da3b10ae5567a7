"""Multiscale PatchGAN discriminator (reference: spade/models/networks/discriminator.py:66-206)."""
import os

import numpy as np

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from .... import spectral_norm as csg_spectral_norm
from ....sg2im.attribute_embed import AttributeEmbeddings
from ....sg2im.layers import Conv2d, GlobalAvgPool, Linear, _FusedActivation, build_hot_cnn, build_mlp
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

    def forward(self, input, seg_first=0, grad_channels=None):
        """`seg_first=S`: `input` is the packed buffer of ops.disc_input ([layout(S)|img(3)|pad]); the
        first conv's weight (trained on cat([img, layout])) is permuted to that channel order.
        `grad_channels=(lo, hi)`: the only input channels whose gradient anybody reads."""
        results = [input]
        for name, sub in self.named_children():
            x = results[-1]
            if name == 'model0' and seg_first:
                conv = sub[0]
                w = conv.weight
                pad = x.size(1) - w.size(1)
                # the permuted, padded weight is the same for every pass between two optimiser steps (and, with the
                # parameters frozen during the generator's passes, a constant there): built once per weight version
                key = (ops.weight_epoch(), w._version, w.data_ptr(), w.requires_grad and torch.is_grad_enabled(), seg_first, pad)
                if getattr(self, "_w0_key", None) != key:
                    parts = [w[:, 3:3 + seg_first], w[:, :3]]
                    if pad:
                        parts.append(w.new_zeros(w.size(0), pad, w.size(2), w.size(3)))
                    self._w0 = torch.cat(parts, dim=1).contiguous(memory_format=torch.channels_last)
                    self._w0_key = key
                x = ops.conv2d(x, self._w0, conv.bias, conv.stride[0], conv.padding[0], conv.act,
                               conv.slope, dx_range=grad_channels)
            else:
                x = sub(x)
            results.append(x)
        return results[1:] if not self.opt.no_ganFeat_loss else results[-1]


SCALE_STREAMS = os.environ.get("CSG_D_SCALE_STREAMS", "1") != "0"
_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


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
        csg_spectral_norm.prepare(self)              # the spectrally normalised weights of every scale, one launch per stage
        obj_vecs = self.attribute_embedding(objs)
        valid = real_object_mask(objs, self.opt.vocab)
        S = obj_vecs.size(-1)
        x = ops.disc_input(img, obj_vecs, layout_boxes, valid, self.opt.image_size[0], masks=layout_masks)
        # Who reads d(input)?  In the generator pass only the image (the discriminator is frozen), in the
        # discriminator passes only the layout (through D's own embedding): the first conv's backward-data
        # is restricted to those channels.
        want_img, want_seg = img.requires_grad, obj_vecs.requires_grad
        grad_channels = None
        if want_img != want_seg and S % 4 == 0:
            grad_channels = (S, x.size(1)) if want_img else (0, S)
        scales = [D for name, D in self.named_children() if name.startswith('discriminator')]
        if len(scales) > 1 and SCALE_STREAMS and x.is_cuda:
            return self._forward_concurrent(scales, x, S, grad_channels)
        result = []
        for i, D in enumerate(scales):
            # the input of a scale feeds that scale AND, pooled, the next one: ops.pool_fanout sums its two gradients inside
            # the pooling backward (the last scale's input has one consumer)
            x, nxt = ops.pool_fanout(x) if (i + 1 < len(scales) and x.is_cuda) else (x, None)
            out = D(x, seg_first=S, grad_channels=grad_channels)
            result.append(out if not self.opt.no_ganFeat_loss else [out])
            x = nxt
        return result

    def _forward_concurrent(self, scales, x, S, grad_channels):
        """Scale 0 on the caller's stream, the lower-resolution scales (grids that cannot fill the chip) on a side
        stream, joined by events at both ends; autograd replays each scale's backward on the stream of its forward."""
        main = torch.cuda.current_stream(x.device)
        side = _side_stream(x.device)
        # the fan-out of the input (scale 0 | pooled copy for the lower scales) stays on the caller's stream: its backward
        # (ops._PoolFanout: d x = g_scale0 + avgpool_bwd(g_lower) in one pass) then runs there too, on gradients of both streams
        x, xs = ops.pool_fanout(x)
        side.wait_stream(main)
        xs.record_stream(side)
        lower = []
        with torch.cuda.stream(side):
            for i, D in enumerate(scales[1:]):
                if i > 0:
                    xs = self.downsample(xs)
                lower.append(D(xs, seg_first=S, grad_channels=grad_channels))
        first = scales[0](x, seg_first=S, grad_channels=grad_channels)
        main.wait_stream(side)
        for out in lower:
            for t in (out if isinstance(out, (list, tuple)) else [out]):
                t.record_stream(main)
        return [out if not self.opt.no_ganFeat_loss else [out] for out in [first] + lower]


class AcDiscriminator(nn.Module):
    """Auxiliary-classifier object discriminator (reference discriminator.py:209-237): crops ->
    CNN -> global average pool -> Linear(1024) -> {real/fake score, object class logits}."""

    def __init__(self, vocab, arch, normalization='none', activation='relu', padding='same', pooling='avg'):
        super().__init__()
        self.vocab = vocab
        cnn, D = build_hot_cnn(arch, normalization=normalization, activation=activation, padding=padding)
        self.cnn = nn.Sequential(cnn, GlobalAvgPool(), Linear(D, 1024))
        num_objects = max(vocab['object_name_to_idx'].values()) + 1
        self.real_classifier = Linear(1024, 1)
        self.obj_classifier = Linear(1024, num_objects)

    def forward(self, x, y):
        if x.dim() == 3:
            x = x[:, None]
        feats = self.cnn[0](x)
        vecs = self.cnn[2](self.cnn[1](feats))
        real_scores = self.real_classifier(vecs)
        obj_scores = self.obj_classifier(vecs)
        ac_loss = F.cross_entropy(obj_scores, y)          # (N, classes) logits: a tiny tensor op
        return real_scores, ac_loss


class AcCropDiscriminator(nn.Module):
    """reference discriminator.py:240-261: bilinear crops of every real object, then AcDiscriminator."""

    def __init__(self, vocab, arch, normalization='none', activation='relu', object_size=64, padding='same',
                 pooling='avg'):
        super().__init__()
        self.vocab = vocab
        self.discriminator = AcDiscriminator(vocab, arch, normalization, activation, padding, pooling)
        self.object_size = object_size

    def prefetch_index(self, objs):
        """Start fetching the real-object list of this batch WITHOUT draining the GPU queue: the mask
        kernel and a 1-byte-per-object copy to pinned memory are enqueued now; `forward` (called much
        later in the step) only waits on their event.  Calling `nonzero()` in forward instead would
        synchronise the whole stream three times per step."""
        valid = real_object_mask(objs, self.vocab)
        host = getattr(self, "_pinned", None)
        if host is None or host.shape != valid.shape:       # pinned allocations are slow: keep one per shape
            host = torch.empty(valid.shape, dtype=torch.uint8, pin_memory=True)
            self._pinned = host
        host.copy_(valid, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._index_cache = (objs.data_ptr(), objs._version, tuple(objs.shape), host, ev, None)

    def release_index(self):
        """Forget the prefetched list (called by Trainer.step once the step's last object-discriminator pass has
        consumed it): the cache is keyed by the tensor's address, and the caching allocator hands the next batch's
        `objs` the same address."""
        self._index_cache = None

    def _object_index(self, objs):
        c = getattr(self, "_index_cache", None)
        if c is not None and c[:3] == (objs.data_ptr(), objs._version, tuple(objs.shape)):
            if c[5] is None:
                c[4].synchronize()
                nz = c[3].nonzero()                          # on the host copy: image-major, like the reference
                dev = (nz[:, 0].contiguous().to(objs.device, non_blocking=True),
                       nz[:, 1].contiguous().to(objs.device, non_blocking=True))
                self._index_cache = c[:5] + (dev,)
                return dev
            return c[5]
        nz = real_object_mask(objs, self.vocab).nonzero()   # (N,2) [image, object]; synchronises
        return nz[:, 0].contiguous(), nz[:, 1].contiguous()

    def forward(self, imgs, objs, boxes):
        img_idx, obj_idx = self._object_index(objs)
        flat_boxes = boxes[img_idx, obj_idx]
        labels = objs[img_idx, obj_idx, 0]
        crops = ops.crop_objects(imgs, flat_boxes, img_idx, self.object_size)
        real_scores, ac_loss = self.discriminator(crops, labels)
        return real_scores, ac_loss, crops[:, :imgs.size(1)]


class NLayerMaskDiscriminator2(NLayerDiscriminator):
    def compute_D_input_nc(self):
        return max(self.opt.vocab['object_name_to_idx'].values()) + 2


class MultiscaleMaskDiscriminator2(BaseNetwork):
    """Mask discriminator (reference discriminator.py:264-308): one-hot(object) x mask -> PatchGAN.
    Only exercised with --mask_size > 0; always constructed when use_img_disc=0 (meta_models.py:83-90)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        for i in range(opt.num_D):
            self.add_module('discriminator_%d' % i, NLayerMaskDiscriminator2(opt))

    def downsample(self, input):
        return ops.avgpool3s2(input)

    def forward(self, objs, layout_masks, gt_train=True):
        """Input per REAL object (image-major order, as the reference's per-sample loop): one-hot(class)
        broadcast over the M x M grid, then the mask as the last channel; built directly with the channel
        count padded to a multiple of 4 (zeros) so that conv and avg-pool kernels take it as is."""
        valid = real_object_mask(objs, self.opt.vocab).bool()
        nz = valid.nonzero()
        labels = objs[nz[:, 0], nz[:, 1], 0]
        masks = layout_masks[nz[:, 0], nz[:, 1]].float()
        N, M = masks.size(0), masks.size(-1)
        ncls = max(self.opt.vocab['object_name_to_idx'].values()) + 1
        Cp = ncls + 1 + (-(ncls + 1)) % 4
        one_hot = F.one_hot(labels, Cp).to(masks.dtype).view(N, Cp, 1, 1)
        x = one_hot + F.pad(masks.unsqueeze(1), (0, 0, 0, 0, ncls, Cp - ncls - 1))      # (N, Cp, M, M)
        x = ops.nhwc(x)
        result = []
        for name, D in self.named_children():
            if name.startswith('discriminator'):
                out = D(x)
                result.append(out if not self.opt.no_ganFeat_loss else [out])
                x = self.downsample(x)
        return result
