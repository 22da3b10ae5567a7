"""The reference's "comm backend" (spade/models/networks/sync_batchnorm/), MI355X form.

The reference runs ONE process with `nn.DataParallel` threads and reduces BatchNorm statistics
through Python queues (replicate.py:27-67, comm.py:18-137, batchnorm.py:105-126).  Here there is
one process per GPU: `SynchronizedBatchNorm*` all-reduces its (sum, sum^2) message over the
default `torch.distributed` group (RCCL over xGMI on the GPU box), and `DataParallelWithCallback`
is a transparent wrapper that only keeps the `.module` attribute / `module.` state_dict prefix the
reference's checkpoints and `get_conv_converse` rely on."""
import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from ..... import ops


class _SynchronizedBatchNorm(_BatchNorm):
    """1 rank: F.batch_norm semantics (batchnorm.py:65-68).  N ranks: the N-replica formula with
    clamp(var, eps) (batchnorm.py:128-145).  `num_batches_tracked` is never advanced, as in the
    reference (its forward bypasses `_BatchNorm.forward`)."""

    sync = True

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)

    def _check_input_dim(self, input):
        pass

    def forward(self, input, gb=None, fused_slope=1.0):
        if self.affine:                               # batchnorm.py:51-93 with weight and bias (SPADE itself uses affine=False)
            if gb is not None:
                raise RuntimeError("an affine SynchronizedBatchNorm cannot take a SPADE modulation map as well")
            from .....sg2im.layers import affine_batch_norm
            return affine_batch_norm(input, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                     fused_slope, self.eps, self.momentum, sync=self.sync)
        x = input if input.dim() == 4 else input.reshape(input.size(0), self.num_features, -1, 1)
        y = ops.norm_act(x, gb, self.running_mean, self.running_var, instance=False, training=self.training,
                         slope=fused_slope, eps=self.eps, momentum=self.momentum, sync=self.sync)
        return y if input.dim() == 4 else y.reshape(input.shape)


class SynchronizedBatchNorm1d(_SynchronizedBatchNorm):
    pass


class SynchronizedBatchNorm2d(_SynchronizedBatchNorm):
    pass


class SynchronizedBatchNorm3d(_SynchronizedBatchNorm):
    pass


class LocalBatchNorm2d(_SynchronizedBatchNorm):
    """`spadebatch`: per-replica statistics (nn.BatchNorm2d in the reference, normalization.py:79-80)."""
    sync = False


class DataParallelWithCallback(nn.Module):
    """Same constructor as the reference (`replicate.py:50-67`); no replication happens here —
    the process owns exactly one GPU and data parallelism is across processes (`torchrun`, one rank per
    GPU: canonicalsg2im_amd/dist.py).  The reference's single-process form of `--gpu_ids 0,1,2,3` would
    silently run the whole batch on one device here, so the first forward refuses it: several device ids
    need a process group of that many ranks (scripts/train.py sets `--gpu_ids` from WORLD_SIZE itself)."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids else []
        self._checked = False

    def _check_world(self):
        self._checked = True
        n = len(self.device_ids)
        if n > 1:
            import torch.distributed as dist
            world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
            if world != n:
                raise RuntimeError(
                    "DataParallelWithCallback(device_ids=%r): this implementation is one process per GPU — launch %d ranks "
                    "(python -m torch.distributed.run --nproc-per-node %d ...) instead of one process with %d device ids; "
                    "the current process group has %d rank(s), so the batch would run on ONE device" % (self.device_ids, n, n, n, world))

    def forward(self, *inputs, **kwargs):
        if not self._checked:
            self._check_world()
        return self.module(*inputs, **kwargs)


def patch_replication_callback(data_parallel):
    return data_parallel


def convert_model(module):
    return module


def patch_sync_batchnorm():
    import contextlib
    return contextlib.nullcontext()
