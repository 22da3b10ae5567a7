"""Object vectors + boxes -> dense layout (reference: sg2im/layout.py:12-45).

Single-image API kept for drop-in use; the generator and discriminator call the batched
kernels in `ops` directly (one launch per batch instead of the reference's per-sample loop)."""
import torch

from .. import ops


def boxes_to_layout(vecs, boxes, H, W=None, pooling='sum'):
    """vecs (O,D), boxes (O,4) as [x0,y0,w,h] in [0,1] -> (1,D,H,W)."""
    if pooling != 'sum':
        raise ValueError('Invalid pooling "%s"' % pooling)
    W = H if W is None else W
    if W != H:
        raise NotImplementedError("boxes_to_layout: non-square layouts are not on the hot path")
    valid = torch.ones((1, vecs.size(0)), dtype=torch.uint8, device=vecs.device)
    (out,) = ops.layout_pyramid(vecs.unsqueeze(0), boxes.unsqueeze(0), valid, H, (H,))
    return out


def masks_to_layout(vecs, boxes, masks, H, W=None, pooling='sum', test_mode=False):
    """vecs (O,D), boxes (O,4) xywh, masks (O,M,M) -> (1,D,H,W) (reference sg2im/layout.py:48-77):
    each object's vector is modulated by the bilinear sample of its mask over its box."""
    if pooling != 'sum':
        raise ValueError('Invalid pooling "%s"' % pooling)
    if test_mode:
        raise NotImplementedError("masks_to_layout(test_mode=True) is the inference-time painter's compositing "
                                  "(layout.py:139-151), outside the training hot path")
    W = H if W is None else W
    if W != H:
        raise NotImplementedError("masks_to_layout: non-square layouts are not on the hot path")
    valid = torch.ones((1, vecs.size(0)), dtype=torch.uint8, device=vecs.device)
    (out,) = ops.layout_pyramid(vecs.unsqueeze(0), boxes.unsqueeze(0), valid, H, (H,), masks=masks.unsqueeze(0))
    return out
