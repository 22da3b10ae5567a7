"""Object vectors + boxes -> dense layout (reference: sg2im/layout.py:12-77).

Single-image API kept for drop-in use; the generator and discriminator call the batched
kernels in `ops` directly (one launch per batch instead of the reference's per-sample loop)."""
import torch

from .. import ops


def boxes_to_layout(vecs, boxes, H, W=None, pooling='sum'):
    """vecs (O,D), boxes (O,4) as [x0,y0,w,h] in [0,1] -> (1,D,H,W); differentiable in both."""
    if pooling != 'sum':
        raise ValueError('Invalid pooling "%s"' % pooling)
    W = H if W is None else W
    valid = torch.ones((1, vecs.size(0)), dtype=torch.uint8, device=vecs.device)
    (out,) = ops.layout_pyramid(vecs.unsqueeze(0), boxes.unsqueeze(0), valid, H, ((H, W),), W=W)
    return out


def masks_to_layout(vecs, boxes, masks, H, W=None, pooling='sum', test_mode=False):
    """vecs (O,D), boxes (O,4) xywh, masks (O,M,M) -> (1,D,H,W) (reference sg2im/layout.py:48-77):
    each object's vector is modulated by the bilinear sample of its mask over its box.  `test_mode`
    composites the objects with the painter's algorithm instead of summing them (layout.py:135-151)."""
    if pooling != 'sum':
        raise ValueError('Invalid pooling "%s"' % pooling)
    O, M = vecs.size(0), masks.size(1)
    assert masks.size() == (O, M, M)
    W = H if W is None else W
    valid = torch.ones((1, O), dtype=torch.uint8, device=vecs.device)
    if test_mode:
        (out,) = ops.layout_paint(vecs.unsqueeze(0), boxes.unsqueeze(0), valid, masks.unsqueeze(0), H, ((H, W),), W=W)
        return out
    (out,) = ops.layout_pyramid(vecs.unsqueeze(0), boxes.unsqueeze(0), valid, H, ((H, W),), masks=masks.unsqueeze(0), W=W)
    return out
