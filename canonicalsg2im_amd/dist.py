"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).

The reference's only parallel strategy is `nn.DataParallel`: parameters are
re-broadcast every forward and gradients reduce-added onto GPU 0.  Here replicas hold identical
parameters and exchange only (1) gradients — bucketed all-reduce, averaged, which equals the
reference because its per-replica losses are means over equal shards averaged with `.mean()`
(scripts/train.py:363,391) — and (2) SyncBN statistics (ops._NormAct)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default group from torchrun's env (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # CSG_DIST_BACKEND=gloo: test hook (two ranks sharing one GPU cannot use RCCL)
            backend = os.environ.get("CSG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def all_reduce_stats(sums):
    """SyncBN exchange (reference sync_batchnorm/batchnorm.py:74-83,105-126): every replica
    contributes its per-channel (sum, sum^2) — one fp64 tensor of 2C values per norm — and gets the
    total back.  One all-reduce replaces the reference's ReduceAddCoalesced + Broadcast pair."""
    if world_size() > 1:
        dist.all_reduce(sums)
    return sums


class GradBuckets:
    """Flat fp32 buckets over the parameters that actually receive gradients.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few large messages beat many small ones,
    so buckets are ~64 MB (the generator's 375 MB of gradients travel in 6 collectives).  Parameters
    without a gradient (the never-used `repr_net` / `image_encoder` of G and D — SURVEY.md §9
    item 11) are skipped at sync time instead of tripping a DDP 'unused parameter' error."""

    def __init__(self, params, bucket_bytes=64 << 20):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        self.bucket_bytes = bucket_bytes

    def _buckets(self, tensors):
        cur, size = [], 0
        for t in tensors:
            n = t.numel() * t.element_size()
            if cur and size + n > self.bucket_bytes:
                yield cur
                cur, size = [], 0
            cur.append(t)
            size += n
        if cur:
            yield cur

    def all_reduce_mean(self):
        """grad <- mean over ranks, in place.  Every rank must hold gradients for the same
        parameters (true here: replicas run the same graph)."""
        n = world_size()
        if n == 1:
            return 0
        grads = [p.grad for p in self.params if p.grad is not None]
        nbytes = 0
        for bucket in self._buckets(grads):
            flat = torch.cat([g.reshape(-1) for g in bucket])
            dist.all_reduce(flat)
            flat.div_(n)
            off = 0
            for g in bucket:
                k = g.numel()
                g.copy_(flat[off:off + k].view_as(g))
                off += k
            nbytes += flat.numel() * 4
        return nbytes


    # ---- split form: start the collectives now, finish them later (overlap with independent work)
    def all_reduce_start(self):
        """Flatten the gradients into buckets and launch one ASYNC all-reduce per bucket.  Returns the
        pending list for `all_reduce_finish` (None on a single rank)."""
        if world_size() == 1:
            return None
        pending = []
        grads = [p.grad for p in self.params if p.grad is not None]
        for bucket in self._buckets(grads):
            flat = torch.cat([g.reshape(-1) for g in bucket])
            pending.append((flat, bucket, dist.all_reduce(flat, async_op=True)))
        return pending

    def all_reduce_finish(self, pending):
        """Wait for the collectives of `all_reduce_start`, average, and scatter back into the .grad tensors."""
        if not pending:
            return 0
        n, nbytes = world_size(), 0
        for flat, bucket, work in pending:
            work.wait()
            flat.div_(n)
            off = 0
            for g in bucket:
                k = g.numel()
                g.copy_(flat[off:off + k].view_as(g))
                off += k
            nbytes += flat.numel() * 4
        return nbytes


def broadcast_module(module, src=0):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
