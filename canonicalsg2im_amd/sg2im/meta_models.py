"""Trainer-level containers (reference: sg2im/meta_models.py:9-90).  Same attribute names and
state_dict prefixes (`sg_to_layout.module.*`, `layout_to_image_model.module.*`)."""
import torch
import torch.nn as nn

from ..spade.models.networks import (AcCropDiscriminator, MultiscaleDiscriminator, MultiscaleMaskDiscriminator2,
                                     SPADEGenerator)
from ..spade.models.networks.sync_batchnorm import DataParallelWithCallback
from .model import Sg2LayoutModel


class MetaGeneratorModel(nn.Module):
    def __init__(self, opt, device):
        super().__init__()
        self.args = vars(opt)
        self.vocab = self.args["vocab"]
        if not self.args['skip_graph_model']:
            self.sg_to_layout = DataParallelWithCallback(Sg2LayoutModel(opt), device_ids=self.args['gpu_ids']).to(device)
        if not self.args['skip_generation']:
            self.layout_to_image_model = DataParallelWithCallback(SPADEGenerator(opt),
                                                                  device_ids=self.args['gpu_ids']).to(device)

    def forward(self, objs, triplets, triplet_type, boxes_gt=None, masks_gt=None, test_mode=False):
        boxes_pred = masks_pred = img = None
        if not self.args['skip_graph_model']:
            _, boxes_pred, masks_pred = self.sg_to_layout(objs, triplets, triplet_type, boxes_gt)
        if not self.args["skip_generation"]:
            layout_boxes = boxes_pred if boxes_gt is None else boxes_gt
            layout_masks = masks_pred if masks_gt is None else masks_gt
            img = self.layout_to_image_model(objs, layout_boxes, layout_masks, test_mode=test_mode)
        return img, boxes_pred, masks_pred


class MetaDiscriminatorModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.args = vars(opt)
        self.img_discriminator = MultiscaleDiscriminator(opt)
        self.img_discriminator.train()
        if not opt.use_img_disc:
            self.obj_discriminator = AcCropDiscriminator(vocab=opt.vocab, arch=opt.d_obj_arch,
                                                         normalization=opt.d_normalization,
                                                         activation=opt.d_activation, padding=opt.d_padding,
                                                         object_size=opt.crop_size)
            self.obj_discriminator.train()
            self.mask_discriminator = MultiscaleMaskDiscriminator2(opt)
            self.mask_discriminator.train()

    def build_optimizers(self, opt):
        """Adam(betas=(beta1, 0.999)) per discriminator (reference meta_models.py:67-69,79-81,88-90);
        called after the module sits on its device."""
        on_gpu = next(self.img_discriminator.parameters()).is_cuda
        fused = {'fused': True} if on_gpu else {}
        self.optimizer_d_img = torch.optim.Adam(list(self.img_discriminator.parameters()),
                                                lr=opt.img_learning_rate, betas=(opt.beta1, 0.999), **fused)
        if not opt.use_img_disc:
            self.optimizer_d_obj = torch.optim.Adam(list(self.obj_discriminator.parameters()),
                                                    lr=opt.learning_rate, betas=(opt.beta1, 0.999), **fused)
            self.optimizer_d_mask = torch.optim.Adam(list(self.mask_discriminator.parameters()),
                                                     lr=opt.mask_learning_rate, betas=(opt.beta1, 0.999), **fused)
        return self
