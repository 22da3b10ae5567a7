"""Trainer-level containers with the reference's surface (sg2im/meta_models.py:9-90): attribute names
`sg_to_layout`, `layout_to_image_model`, `img_/obj_/mask_discriminator`, `optimizer_d_img/_obj/_mask`
and the state_dict prefixes `sg_to_layout.module.*`, `layout_to_image_model.module.*`.

One process drives one GPU here, so `DataParallelWithCallback` is a transparent wrapper that only
contributes the `.module` level of the checkpoint keys."""
import torch
import torch.nn as nn

from .. import streams
from ..spade.models import networks as spade_nets
from ..spade.models.networks.sync_batchnorm import DataParallelWithCallback
from .model import Sg2LayoutModel


def _wrapped(net, opt, device):
    return DataParallelWithCallback(net, device_ids=opt.gpu_ids).to(device)


class MetaGeneratorModel(nn.Module):
    """Scene graph -> (boxes, masks) -> image.  Either half can be switched off with
    `--skip_graph_model` / `--skip_generation` (reference :13-23)."""

    def __init__(self, opt, device):
        super().__init__()
        self.args = vars(opt)
        self.vocab = opt.vocab
        self.has_graph = not opt.skip_graph_model
        self.has_image = not opt.skip_generation
        if self.has_graph:
            self.sg_to_layout = _wrapped(Sg2LayoutModel(opt), opt, device)
        if self.has_image:
            self.layout_to_image_model = _wrapped(spade_nets.SPADEGenerator(opt), opt, device)

    def forward(self, objs, triplets, triplet_type, boxes_gt=None, masks_gt=None, test_mode=False):
        img = boxes_pred = masks_pred = None
        side = None
        if self.has_graph:
            # the generator below consumes the ground-truth boxes / masks (reference :47-49): when it does not need the
            # encoder's outputs the encoder runs beside it on a stream of its own (canonicalsg2im_amd/streams.py)
            independent = self.has_image and boxes_gt is not None and (masks_gt is not None or not self.args.get("mask_size"))
            if independent and torch.is_grad_enabled() and streams.usable(objs):
                with streams.beside("encoder", objs.device) as side:
                    boxes_pred, masks_pred = self.sg_to_layout(objs, triplets, triplet_type, boxes_gt)[1:]
            else:
                boxes_pred, masks_pred = self.sg_to_layout(objs, triplets, triplet_type, boxes_gt)[1:]
        if self.has_image:
            # ground truth wins over the prediction wherever it is given (reference :47-49)
            img = self.layout_to_image_model(objs, boxes_gt if boxes_gt is not None else boxes_pred,
                                             masks_gt if masks_gt is not None else masks_pred, test_mode=test_mode)
        if side is not None:
            streams.join(side, boxes_pred, masks_pred)
        return img, boxes_pred, masks_pred


class MetaDiscriminatorModel(nn.Module):
    """Image discriminator always; object-crop and mask discriminators unless `--use_img_disc 1`
    (reference :56-90).  Optimisers are created by `build_optimizers` once the module is on its device —
    the reference hard-codes `torch.cuda.FloatTensor` in the constructor instead."""

    def __init__(self, opt):
        super().__init__()
        self.args = vars(opt)
        self.img_discriminator = spade_nets.MultiscaleDiscriminator(opt)
        if not opt.use_img_disc:
            self.obj_discriminator = spade_nets.AcCropDiscriminator(
                vocab=opt.vocab, arch=opt.d_obj_arch, normalization=opt.d_normalization, activation=opt.d_activation,
                padding=opt.d_padding, object_size=opt.crop_size)
            self.mask_discriminator = spade_nets.MultiscaleMaskDiscriminator2(opt)
        self.train()

    def build_optimizers(self, opt, capturable_img=False):
        """Adam with betas (beta1, 0.999); learning rates `img_learning_rate`, `learning_rate`,
        `mask_learning_rate` for the image / object / mask discriminator (reference :67-69, :79-81, :88-90).
        `capturable_img`: the image discriminator's step counter lives on the device so that the step can be replayed
        from a HIP graph (canonicalsg2im_amd/graphs.py); the arithmetic is the same fused kernel."""
        extra = {'fused': True} if next(self.parameters()).is_cuda else {}        # one multi-tensor kernel per step
        plan = [("img", opt.img_learning_rate)]
        if not opt.use_img_disc:
            plan += [("obj", opt.learning_rate), ("mask", opt.mask_learning_rate)]
        for tag, lr in plan:
            net = getattr(self, tag + "_discriminator")
            cap = {'capturable': True} if (capturable_img and tag == "img" and extra) else {}
            setattr(self, "optimizer_d_" + tag,
                    torch.optim.Adam(list(net.parameters()), lr=lr, betas=(opt.beta1, 0.999), **extra, **cap))
        return self
