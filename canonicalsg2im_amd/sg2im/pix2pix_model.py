"""Loss module of the trainer — the surface of the reference's `Pix2PixModel` (sg2im/pix2pix_model.py:12-223):
`forward(batch, model_out, mode)` with modes `compute_generator_loss` / `compute_discriminator_loss` returning
the same loss dictionaries (same keys, same weighting).

All discriminator passes, the VGG features and the L1 feature-matching distances run on the HIP kernels; what
is left for torch are scalar combinations and three tiny elementwise losses (smooth-L1 on (B,O,4) boxes, BCE on
the (B,O,M,M) masks, cross-entropy on the crop logits — SURVEY.md §2.2 K12)."""
import torch
import torch.nn.functional as F

from .. import ops, streams
from ..spade.models import networks
from .losses import get_gan_losses


class _L1Loss(torch.nn.Module):
    """nn.L1Loss() against a constant target as one fused reduction kernel (+ one for the gradient)."""

    def forward(self, input, target):
        return ops.l1_mean(input, target)


def _total(losses, skip=()):
    return torch.stack([v for k, v in losses.items() if k not in skip], dim=0).sum()


class Pix2PixModel(torch.nn.Module):
    def __init__(self, opt, discriminator, netE=None):
        super().__init__()
        self.opt, self.discriminator = opt, discriminator
        self.netD_img = getattr(discriminator, 'img_discriminator', None)
        self.netD_obj = self.netD_mask = None
        if not getattr(opt, 'use_img_disc', 1):
            self.netD_obj = getattr(discriminator, 'obj_discriminator', None)
            self.netD_mask = getattr(discriminator, 'mask_discriminator', None)
        if not opt.isTrain:
            return
        self.criterionGAN = networks.GANLoss(opt.gan_mode, opt=opt)
        self.criterionFeat = _L1Loss()
        self.gan_g_loss, self.gan_d_loss = get_gan_losses(opt.gan_loss_type)
        if not opt.no_vgg_loss:
            self.criterionVGG = networks.VGGLoss(opt.gpu_ids)
        if opt.use_vae:
            self.KLDLoss = networks.KLDLoss()

    def use_gpu(self):
        return len(self.opt.gpu_ids) > 0

    def forward(self, batch, model_out, mode):
        handlers = {"compute_generator_loss": self.compute_generator_loss,
                    "compute_discriminator_loss": self.compute_discriminator_loss}
        if mode not in handlers:
            raise ValueError("unknown mode %r" % mode)
        return handlers[mode](batch, model_out)

    # ---------------------------------------------------------------------------------------- pieces
    def _fool(self, scores, weight):
        """Generator-side GAN term: the discriminator should call `scores` real."""
        return self.criterionGAN(scores, True, for_discriminator=False).squeeze(0) * weight

    def _feature_matching(self, fake, real):
        """lambda_feat * mean over scales of the L1 distances of every intermediate map (reference :99-109)."""
        # the 2 x 4 distances are fused reductions (ops.l1_mean); their weighted sum is ONE stack + sum + scale instead of
        # a multiply, a divide and an add per map (24 scalar launches per pass, and as many again in the backward)
        terms = [self.criterionFeat(f, r.detach()) for f_scale, r_scale in zip(fake, real)
                 for f, r in zip(f_scale[:-1], r_scale[:-1])]        # the last entry of a scale is the prediction itself
        return torch.stack(terms).sum() * (self.opt.lambda_feat / len(fake))

    def _layout_terms(self, out, objs, boxes, boxes_pred, masks, masks_pred):
        """Box regression (:71-85) and mask BCE (:88-92), both averaged over the REAL objects only."""
        opt = self.opt
        per_coord = F.smooth_l1_loss(boxes_pred.reshape(-1, 4), boxes.reshape(-1, 4), reduction='none')
        ids = objs.reshape(-1, objs.size(-1))
        real = ((ids.sum(1, keepdim=True) != 0) if ids.size(1) > 1 else (ids != 0)).to(per_coord.dtype)     # (B*O, 1)
        per_coord = per_coord * opt.bbox_pred_loss_weight * real
        B, O = boxes.shape[0], boxes.shape[1]
        out["bbox_pred_all"] = per_coord.view(B, O, 4).sum(dim=[1, 2]) / real.view(B, O).sum(dim=1)
        out["bbox_pred"] = out["bbox_pred_all"].mean()
        if masks is not None:
            M = masks.size(-1)
            bce = F.binary_cross_entropy(masks_pred.reshape(-1, M, M), masks.reshape(-1, M, M).float(),
                                         reduction='none').mean(dim=(1, 2))
            # == masks_loss[object_mask.nonzero()[:, 0]].mean() of the reference, without the host round trip
            out["masks_pred"] = (bce * real.view(-1)).sum() / real.sum() * opt.mask_pred_loss_weight

    # ---------------------------------------------------------------------------------------- generator
    # The two loss dictionaries are assembled from four term groups.  Eagerly they run back to back (the reference's
    # order); `canonicalsg2im_amd/graphs.py` replays the image groups — whose tensor shapes depend only on the batch
    # and image size — as captured HIP graphs and runs the object groups (one crop per real object: a data-dependent
    # count) eagerly in between.
    def generator_image_terms(self, imgs, objs, boxes, masks, imgs_pred):
        """GAN_Img, GAN_Feat, VGG (reference :94-113): two passes of the (frozen) image discriminator."""
        opt, out = self.opt, {}
        d_args = dict(layout_masks=masks, gt_train=True, fool=False)
        fake = self.netD_img(imgs_pred, objs, boxes, **d_args)
        out['GAN_Img'] = self._fool(fake, opt.discriminator_img_loss_weight)
        if not opt.no_ganFeat_loss:
            out['GAN_Feat'] = self._feature_matching(fake, self.netD_img(imgs, objs, boxes, **d_args))
        if not opt.no_vgg_loss:
            out['VGG'] = self.criterionVGG(imgs_pred, imgs) * opt.lambda_vgg
        return out

    def generator_object_terms(self, imgs_pred, objs, boxes, masks, masks_pred):
        """GAN_Obj, GAN_Ac (:115-121) and the mask discriminator's terms (:124-138)."""
        opt, out = self.opt, {}
        crop_scores, ac_loss, _ = self.netD_obj(imgs_pred, objs, boxes)
        out['GAN_Obj'] = self._fool(crop_scores, opt.discriminator_obj_loss_weight)
        out['GAN_Ac'] = ac_loss * opt.ac_loss_weight
        if self.netD_mask is not None and opt.mask_size > 0 and masks_pred is not None:
            m_fake = self.netD_mask(objs, masks_pred)
            out['GAN_Mask'] = self._fool(m_fake, opt.discriminator_img_loss_weight)
            if not opt.no_ganFeat_loss:
                out['GAN_Mask_Feat'] = self._feature_matching(m_fake, self.netD_mask(objs, masks))
        return out

    def compute_generator_loss(self, batch, model_out):
        imgs, objs, boxes, masks = batch[0], batch[1], batch[2], batch[6]
        imgs_pred, boxes_pred, masks_pred = model_out
        opt, out = self.opt, {}
        if not opt.skip_graph_model:
            self._layout_terms(out, objs, boxes, boxes_pred, masks, masks_pred)
        if not opt.skip_generation:
            if not opt.use_img_disc and streams.usable(imgs_pred) and torch.is_grad_enabled():
                # the object discriminator's passes (crops of the fresh image: a handful of small launches) beside the
                # PatchGAN's (canonicalsg2im_amd/streams.py); dictionary order as in the sequential form
                with streams.beside("object_d", imgs_pred.device) as side:
                    obj_terms = self.generator_object_terms(imgs_pred, objs, boxes, masks, masks_pred)
                out.update(self.generator_image_terms(imgs, objs, boxes, masks, imgs_pred))
                streams.join(side, obj_terms)
                out.update(obj_terms)
            else:
                out.update(self.generator_image_terms(imgs, objs, boxes, masks, imgs_pred))
                if not opt.use_img_disc:
                    out.update(self.generator_object_terms(imgs_pred, objs, boxes, masks, masks_pred))
        out['total_loss'] = _total(out, skip=("bbox_pred_all",))
        return out

    # ---------------------------------------------------------------------------------------- discriminators
    def discriminator_image_terms(self, imgs, objs, boxes, masks, fake_img):
        """D_img_fake, D_img_real, total_img_loss (:150-166) and, in the default recipe, the logged "wrong layout" pass
        (:168-172): `fool` is ignored by the discriminator and the value is only logged, but the call advances the
        spectral-norm vectors, so it is replayed — without autograd."""
        opt, crit = self.opt, self.criterionGAN
        d_args = dict(layout_masks=masks, gt_train=True)
        out = {"D_img_fake": crit(self.netD_img(fake_img, objs, boxes, fool=False, **d_args), False, for_discriminator=True),
               "D_img_real": crit(self.netD_img(imgs, objs, boxes, fool=False, **d_args), True, for_discriminator=True)}
        out["total_img_loss"] = _total(out)
        if not opt.use_img_disc:
            with torch.no_grad():
                wrong = self.netD_img(imgs, objs, boxes, fool=True, **d_args)
                out["D_img_wrong"] = crit(wrong, False, for_discriminator=True) * (1 / 2) * (.5)
        return out

    def discriminator_object_terms(self, imgs, objs, boxes, masks, fake_img, masks_pred):
        """D_ac_real, D_ac_fake, D_obj, total_obj_loss (:178-185) and the mask discriminator's terms (:188-196)."""
        opt, crit, out = self.opt, self.criterionGAN, {}
        s_real, out["D_ac_real"], self.d_real_crops = self.netD_obj(imgs, objs, boxes)
        s_fake, out["D_ac_fake"], self.d_fake_crops = self.netD_obj(fake_img, objs, boxes)
        out["D_obj"] = self.gan_d_loss(s_real, s_fake) * 0.5
        out["total_obj_loss"] = out["D_obj"] + out["D_ac_real"] + out["D_ac_fake"]
        if opt.mask_size > 0 and masks_pred is not None:
            out["D_mask_fake"] = crit(self.netD_mask(objs, masks_pred.detach()), False, for_discriminator=True) * 0.5
            out["D_mask_real"] = crit(self.netD_mask(objs, masks), True, for_discriminator=True) * 0.5
            out["total_mask_loss"] = out["D_mask_fake"] + out["D_mask_real"]
        return out

    def compute_discriminator_loss(self, batch, model_out):
        imgs, objs, boxes, masks = batch[0], batch[1], batch[2], batch[6]
        fake_img = model_out[0].detach()
        if self.opt.use_img_disc:
            return self.discriminator_image_terms(imgs, objs, boxes, masks, fake_img)
        if streams.usable(fake_img) and torch.is_grad_enabled():
            with streams.beside("object_d", fake_img.device) as side:
                obj_terms = self.discriminator_object_terms(imgs, objs, boxes, masks, fake_img, model_out[2])
            out = self.discriminator_image_terms(imgs, objs, boxes, masks, fake_img)
            streams.join(side, obj_terms)
            out.update(obj_terms)
            return out
        out = self.discriminator_image_terms(imgs, objs, boxes, masks, fake_img)
        out.update(self.discriminator_object_terms(imgs, objs, boxes, masks, fake_img, model_out[2]))
        return out
