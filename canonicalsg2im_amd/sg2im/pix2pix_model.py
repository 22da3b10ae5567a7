"""Loss module of the trainer (reference: sg2im/pix2pix_model.py:12-223).

The discriminator passes run on the HIP kernels; the loss arithmetic on their outputs is
elementwise + small reductions and stays as tensor ops."""
import torch
import torch.nn.functional as F

from .. import ops
from ..spade.models import networks
from .losses import get_gan_losses


class _L1Loss(torch.nn.Module):
    """nn.L1Loss() against a constant target as one fused reduction kernel (+ one for the gradient)."""

    def forward(self, input, target):
        return ops.l1_mean(input, target)


class Pix2PixModel(torch.nn.Module):
    def __init__(self, opt, discriminator, netE=None):
        super().__init__()
        self.opt = opt
        self.discriminator = discriminator
        if hasattr(discriminator, 'img_discriminator'):
            self.netD_img = discriminator.img_discriminator
        if hasattr(opt, 'use_img_disc') and not opt.use_img_disc:
            if hasattr(discriminator, 'obj_discriminator'):
                self.netD_obj = discriminator.obj_discriminator
            if hasattr(discriminator, 'mask_discriminator'):
                self.netD_mask = discriminator.mask_discriminator
        if opt.isTrain:
            self.criterionGAN = networks.GANLoss(opt.gan_mode, opt=self.opt)
            self.criterionFeat = _L1Loss()
            self.gan_g_loss, self.gan_d_loss = get_gan_losses(opt.gan_loss_type)
            if not opt.no_vgg_loss:
                self.criterionVGG = networks.VGGLoss(self.opt.gpu_ids)
            if opt.use_vae:
                self.KLDLoss = networks.KLDLoss()

    def use_gpu(self):
        return len(self.opt.gpu_ids) > 0

    # ------------------------------------------------------------------ generator side (:65-143)
    def compute_generator_loss(self, batch, model_out):
        imgs, objs, boxes, triplets, _, _, masks, _ = batch
        imgs_pred, boxes_pred, masks_pred = model_out
        G = {}
        if not self.opt.skip_graph_model:
            l = F.smooth_l1_loss(boxes_pred.reshape(-1, 4), boxes.reshape(-1, 4), reduction='none') \
                * self.opt.bbox_pred_loss_weight
            flat = objs.reshape(-1, objs.size(-1))
            mask = (flat.sum(1, keepdim=True) != 0) if objs.size(-1) > 1 else (flat != 0)
            mask = mask.to(l.dtype)
            l = l * mask
            G["bbox_pred_all"] = l.view(boxes.shape).sum(dim=[1, 2]) / mask.view(boxes.shape[0], boxes.shape[1]).sum(dim=1)
            G["bbox_pred"] = G["bbox_pred_all"].mean()
            if masks is not None:                                   # :88-92 — BCE over the real objects' masks
                M = masks.size(-1)
                bce = F.binary_cross_entropy(masks_pred.reshape(-1, M, M), masks.reshape(-1, M, M).float(),
                                             reduction='none').mean(dim=(1, 2))
                # mean over real objects, as masks_loss[object_mask.nonzero()[:, 0]].mean() without the host sync
                G["masks_pred"] = (bce * mask.view(-1)).sum() / mask.sum() * self.opt.mask_pred_loss_weight
        if not self.opt.skip_generation:
            pred_fake = self.netD_img(imgs_pred, objs, boxes, layout_masks=masks, gt_train=True, fool=False)
            G['GAN_Img'] = self.criterionGAN(pred_fake, True, for_discriminator=False).squeeze(0) \
                * self.opt.discriminator_img_loss_weight
            if not self.opt.no_ganFeat_loss:
                pred_real = self.netD_img(imgs, objs, boxes, layout_masks=masks, gt_train=True, fool=False)
                num_D = len(pred_fake)
                feat = imgs.new_zeros(1)
                for i in range(num_D):
                    for j in range(len(pred_fake[i]) - 1):       # last output is the final prediction
                        feat = feat + self.criterionFeat(pred_fake[i][j], pred_real[i][j].detach()) \
                            * self.opt.lambda_feat / num_D
                G['GAN_Feat'] = feat.squeeze(0)
            if not self.opt.no_vgg_loss:
                G['VGG'] = self.criterionVGG(imgs_pred, imgs) * self.opt.lambda_vgg
            if not self.opt.use_img_disc:                      # object discriminator (:115-121)
                scores_fake, ac_loss, _ = self.netD_obj(imgs_pred, objs, boxes)
                G['GAN_Obj'] = self.criterionGAN(scores_fake, True, for_discriminator=False).squeeze(0) \
                    * self.opt.discriminator_obj_loss_weight
                G['GAN_Ac'] = ac_loss * self.opt.ac_loss_weight
                if getattr(self, 'netD_mask', None) is not None and self.opt.mask_size > 0 and masks_pred is not None:
                    scores_fake = self.netD_mask(objs, masks_pred)                                # :124-138
                    G['GAN_Mask'] = self.criterionGAN(scores_fake, True, for_discriminator=False).squeeze(0) \
                        * self.opt.discriminator_img_loss_weight
                    if not self.opt.no_ganFeat_loss:
                        scores_real = self.netD_mask(objs, masks)
                        num_D = len(scores_fake)
                        feat = imgs.new_zeros(1)
                        for i in range(num_D):
                            for j in range(len(scores_fake[i]) - 1):
                                feat = feat + self.criterionFeat(scores_fake[i][j], scores_real[i][j].detach()) \
                                    * self.opt.lambda_feat / num_D
                        G['GAN_Mask_Feat'] = feat.squeeze(0)
        scalars = [k for k in G if k != "bbox_pred_all"]
        G['total_loss'] = torch.stack([G[k] for k in scalars], dim=0).sum()
        return G

    # ------------------------------------------------------------------ discriminator side (:145-202)
    def compute_discriminator_loss(self, batch, model_out):
        imgs, objs, boxes, _, _, _, masks, _ = batch
        imgs_pred = model_out[0].detach()
        D = {}
        pred_fake = self.netD_img(imgs_pred, objs, boxes, layout_masks=masks, gt_train=True, fool=False)
        gt_real = self.netD_img(imgs, objs, boxes, layout_masks=masks, gt_train=True, fool=False)
        D["D_img_fake"] = self.criterionGAN(pred_fake, False, for_discriminator=True)
        D["D_img_real"] = self.criterionGAN(gt_real, True, for_discriminator=True)
        D["total_img_loss"] = torch.stack(list(D.values()), dim=0).sum()
        if not self.opt.use_img_disc:
            # "wrong layout" pass: `fool` is ignored by the discriminator, the value is logged but never
            # back-propagated; it still advances the spectral-norm state, so it is replayed (:168-172)
            with torch.no_grad():
                pred_wrong = self.netD_img(imgs, objs, boxes, layout_masks=masks, gt_train=True, fool=True)
                D["D_img_wrong"] = self.criterionGAN(pred_wrong, False, for_discriminator=True) * (1 / 2) * (.5)
            scores_real, ac_loss_real, self.d_real_crops = self.netD_obj(imgs, objs, boxes)       # :178-185
            scores_fake, ac_loss_fake, self.d_fake_crops = self.netD_obj(imgs_pred, objs, boxes)
            D["D_obj"] = self.gan_d_loss(scores_real, scores_fake) * 0.5
            D["D_ac_real"] = ac_loss_real
            D["D_ac_fake"] = ac_loss_fake
            D["total_obj_loss"] = torch.stack([D["D_obj"], D["D_ac_real"], D["D_ac_fake"]], dim=0).sum()
            if self.opt.mask_size > 0 and model_out[2] is not None:                               # :188-196
                scores_fake = self.netD_mask(objs, model_out[2].detach())
                scores_real = self.netD_mask(objs, masks)
                D["D_mask_fake"] = self.criterionGAN(scores_fake, False, for_discriminator=True) * 0.5
                D["D_mask_real"] = self.criterionGAN(scores_real, True, for_discriminator=True) * 0.5
                D["total_mask_loss"] = torch.stack([D["D_mask_fake"], D["D_mask_real"]], dim=0).sum()
        return D

    def forward(self, batch, model_out, mode):
        if mode == "compute_discriminator_loss":
            return self.compute_discriminator_loss(batch, model_out)
        if mode == "compute_generator_loss":
            return self.compute_generator_loss(batch, model_out)
        raise ValueError("unknown mode %r" % mode)
