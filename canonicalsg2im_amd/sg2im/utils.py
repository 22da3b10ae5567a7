"""Host helpers of the hot path (reference: sg2im/utils.py)."""
import torch

from .. import ops


def int_tuple(s):
    return tuple(int(i) for i in s.split(','))


def bool_flag(s):
    if s == '1':
        return True
    if s == '0':
        return False
    raise ValueError('Invalid value "%s" for bool flag (should be 0 or 1)' % s)


def batch_to(batch, device='cuda'):
    """Move a collate 8-tuple to the device (reference sg2im/utils.py:18-42)."""
    dev = torch.device('cuda') if device == 'cuda' else device
    out = []
    for obj in batch:
        if obj is None or isinstance(obj, list):
            out.append(obj)
        elif torch.is_tensor(obj):
            out.append(obj.to(dev, non_blocking=True))
        else:
            out.append({k: v.to(dev, non_blocking=True) for k, v in obj.items()})
    return out


def remove_dummy_objects(objs, vocab):
    """Real objects of ONE sample, (O,A) -> bool (O,) (reference sg2im/utils.py:56-63).
    Index work is done by the `csg_real_object_mask` kernel and is bit-exact."""
    return real_object_mask(objs.unsqueeze(0), vocab)[0].bool()


def real_object_mask(objs, vocab):
    """Batched form: (B,O,A) int64 -> uint8 (B,O)."""
    return ops.real_object_mask(objs, vocab['object_name_to_idx']['__image__'])
