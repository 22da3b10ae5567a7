"""Synthetic scene-graph batches that follow the reference's collate contract.

The reference's data loaders need COCO / Visual Genome / CLEVR-Dialog on disk,
so every test and benchmark here draws seeded synthetic batches instead.  The
8-tuple this module returns has exactly the layout that
`sg2im/data/packed_coco.py:467-478` (`coco_collate_fn`) hands to the trainer:

    imgs          f32 (B,3,H,W)
    objs          i64 (B,O,A)      padded rows are 0 (== `__image__` id)
    boxes         f32 (B,O,4)      [x0,y0,w,h] in [0,1]; padded rows are -1
    triplets      i64 (B,T,3)      [s,p,o]; padded rows [0,__padding__,0]
    conv_counts   f32 (B,P,P+1)    only read with --learned_converse
    triplet_type  i64 (B,T)        0 original, 1 transitive, 2, 3
    masks         None             (mask_size == 0 in every BASELINE config)
    image_ids     i64 (B,)

Vocabularies mirror `sg2im/data/base_dataset.py:14-15,152-161` (8 packed
predicates) and `sg2im/data/packed_clevr_dialog.py:121-125` (4 attributes).
"""
from dataclasses import dataclass, field
from typing import Dict, Tuple

import numpy as np
import torch

PACKED_PREDICATES = ["__padding__", "__in_image__", "__below__", "__above__",
                     "__left of__", "__right of__", "__inside__", "__surrounding__"]


def make_vocab(kind: str = "coco", num_objects: int = None, num_preds: int = None) -> Dict:
    """Build the vocab dict keys the hot path reads."""
    if kind == "clevr":
        sizes = {"shape": 4, "color": 9, "material": 3, "size": 3}
        attributes = {}
        for name, n in sizes.items():
            d = {"__image__": 0}
            for i in range(1, n):
                d["%s_%d" % (name, i)] = i
            attributes[name] = d
        object_name_to_idx = attributes["shape"]
        preds = list(PACKED_PREDICATES)
    else:
        if num_objects is None:
            num_objects = {"coco": 184, "vg": 179, "tiny": 7}[kind]
        object_name_to_idx = {"__image__": 0}
        for i in range(1, num_objects):
            object_name_to_idx["obj_%d" % i] = i
        attributes = {"objects": object_name_to_idx}
        if num_preds is None:
            num_preds = {"coco": 8, "vg": 46, "tiny": 8}[kind]
        preds = list(PACKED_PREDICATES)
        while len(preds) < num_preds:
            preds.append("rel_%d" % len(preds))
        preds = preds[:max(num_preds, 2)]
    idx_to_name = [None] * (max(object_name_to_idx.values()) + 1)
    for k, v in object_name_to_idx.items():
        idx_to_name[v] = k
    return {
        "object_name_to_idx": object_name_to_idx,
        "object_idx_to_name": idx_to_name,
        "pred_name_to_idx": {p: i for i, p in enumerate(preds)},
        "pred_idx_to_name": preds,
        "attributes": attributes,
    }


@dataclass
class BatchConfig:
    batch_size: int = 4
    image_size: int = 64
    min_objects: int = 3
    max_objects: int = 8
    graph: str = "random"          # random | packed | closure
    pad_objects_to: int = 0        # 0 = pad to the batch max
    pad_triplets_to: int = 0
    extra: dict = field(default_factory=dict)
    mask_size: int = 0             # M > 0: per-object (M,M) int64 segmentation masks (sg2im/data/coco.py:301-346)


# The five BASELINE.json configurations.
BASELINE_CONFIGS = {
    "C1": dict(vocab="coco", cfg=BatchConfig(4, 64, 16, 40, "packed")),
    "C2": dict(vocab="coco", cfg=BatchConfig(16, 128, 3, 8, "random")),
    "C3": dict(vocab="coco", cfg=BatchConfig(16, 256, 1, 30, "random")),
    "C4": dict(vocab="vg", cfg=BatchConfig(32, 256, 3, 30, "random")),
    "C5": dict(vocab="clevr", cfg=BatchConfig(48, 256, 64, 128, "closure")),
}


def _relation(bs, bo, name_to_idx):
    sx0, sy0, sw, sh = bs
    ox0, oy0, ow, oh = bo
    sx1, sy1, ox1, oy1 = sx0 + sw, sy0 + sh, ox0 + ow, oy0 + oh
    if sx0 < ox0 and sx1 > ox1 and sy0 < oy0 and sy1 > oy1:
        return name_to_idx["__surrounding__"]
    if sx0 > ox0 and sx1 < ox1 and sy0 > oy0 and sy1 < oy1:
        return name_to_idx["__inside__"]
    dx = (sx0 + sw / 2) - (ox0 + ow / 2)
    dy = (sy0 + sh / 2) - (oy0 + oh / 2)
    if abs(dx) >= abs(dy):
        return name_to_idx["__left of__"] if dx < 0 else name_to_idx["__right of__"]
    return name_to_idx["__above__"] if dy < 0 else name_to_idx["__below__"]


def _sample_graph(rng, boxes, n, mode, vocab):
    p2i = vocab["pred_name_to_idx"]
    num_preds = len(vocab["pred_idx_to_name"])
    trip, ttype = [], []
    if n <= 1:
        return trip, ttype
    if mode == "random":
        # one random triple per object (sg2im/data/coco.py:372-421)
        for s in range(n):
            o = int(rng.integers(0, n - 1))
            o = o if o < s else o + 1
            trip.append([s, int(rng.integers(2, num_preds)), o])
            ttype.append(0)
    elif mode == "packed":
        # <=4 nearest neighbours per object, relation from box geometry
        cen = boxes[:n, :2] + boxes[:n, 2:] / 2
        for s in range(n):
            d = np.linalg.norm(cen - cen[s], axis=1)
            d[s] = np.inf
            for o in np.argsort(d)[:min(4, n - 1)]:
                trip.append([s, _relation(boxes[s], boxes[int(o)], p2i), int(o)])
                ttype.append(0)
    elif mode == "closure":
        # every ordered pair; a quarter are "original", the rest transitive
        for s in range(n):
            for o in range(n):
                if s == o:
                    continue
                trip.append([s, _relation(boxes[s], boxes[o], p2i), o])
                ttype.append(0 if (s + o) % 4 == 0 else 1)
    else:
        raise ValueError("unknown graph mode %r" % mode)
    return trip, ttype


def make_batch(vocab: Dict, cfg: BatchConfig, seed: int = 0) -> Tuple:
    """Seeded batch in collate layout; CPU tensors."""
    rng = np.random.default_rng(seed)
    B, H = cfg.batch_size, cfg.image_size
    attr_names = list(vocab["attributes"].keys())
    A = len(attr_names)
    sizes = [max(vocab["attributes"][a].values()) + 1 for a in attr_names]
    num_preds = len(vocab["pred_idx_to_name"])
    pad_p = vocab["pred_name_to_idx"]["__padding__"]

    per = []
    for _ in range(B):
        n = int(rng.integers(cfg.min_objects, cfg.max_objects + 1))
        o = np.stack([rng.integers(1, max(s, 2), size=n) for s in sizes], axis=1).astype(np.int64)
        wh = rng.uniform(0.05, 0.45, size=(n, 2))
        xy = rng.uniform(0.0, 1.0, size=(n, 2)) * (1.0 - wh)
        bx = np.concatenate([xy, wh], axis=1).astype(np.float32)
        trip, ttype = _sample_graph(rng, bx, n, cfg.graph, vocab)
        per.append((o, bx, np.asarray(trip, np.int64).reshape(-1, 3), np.asarray(ttype, np.int64)))

    O = max(cfg.pad_objects_to, max(p[0].shape[0] for p in per))
    T = max(cfg.pad_triplets_to, max(p[2].shape[0] for p in per), 1)
    objs = np.zeros((B, O, A), np.int64)
    boxes = -np.ones((B, O, 4), np.float32)
    triplets = np.zeros((B, T, 3), np.int64)
    triplets[:, :, 1] = pad_p
    ttypes = np.zeros((B, T), np.int64)
    for b, (o, bx, tr, tt) in enumerate(per):
        objs[b, :o.shape[0]] = o
        boxes[b, :bx.shape[0]] = bx
        triplets[b, :tr.shape[0]] = tr
        ttypes[b, :tt.shape[0]] = tt
    imgs = rng.uniform(-1.0, 1.0, size=(B, 3, H, H)).astype(np.float32)
    conv_counts = np.zeros((B, num_preds, num_preds + 1), np.float32)
    masks = None
    if cfg.mask_size > 0:
        # an axis-aligned ellipse per real object; padded rows stay 0 (coco.py:512-515)
        M = cfg.mask_size
        masks_np = np.zeros((B, O, M, M), np.int64)
        yy, xx = np.meshgrid((np.arange(M) + 0.5) / M, (np.arange(M) + 0.5) / M, indexing="ij")
        for b, (o, _, _, _) in enumerate(per):
            n = o.shape[0]
            c = rng.uniform(0.35, 0.65, size=(n, 2))
            r = rng.uniform(0.2, 0.5, size=(n, 2))
            for i in range(n):
                masks_np[b, i] = (((xx - c[i, 0]) / r[i, 0]) ** 2 + ((yy - c[i, 1]) / r[i, 1]) ** 2 <= 1.0)
        masks = torch.from_numpy(masks_np)
    return (torch.from_numpy(imgs), torch.from_numpy(objs), torch.from_numpy(boxes),
            torch.from_numpy(triplets), torch.from_numpy(conv_counts), torch.from_numpy(ttypes),
            masks, torch.arange(B, dtype=torch.int64))


def shard_batch(batch: Tuple, rank: int, world_size: int) -> Tuple:
    """Contiguous B/N samples per rank — the partition `nn.DataParallel` scatter
    makes in the reference (`sync_batchnorm/replicate.py:64-67`, `scripts/args.py:234-236`)."""
    B = batch[0].shape[0]
    if B % world_size != 0:
        raise ValueError("Batch size %d is wrong. It must be a multiple of # GPUs %d." % (B, world_size))
    n = B // world_size
    sl = slice(rank * n, (rank + 1) * n)
    return tuple(None if t is None else t[sl] for t in batch)


def deterministic_state(state_dict, seed=0):
    """Seeded stand-in for a checkpoint: every tensor is a pure function of its name and shape, so
    the golden-vector script (reference modules) and the tests (oracle / HIP modules) can rebuild
    identical weights without storing them.  Scales keep activations O(1)."""
    import zlib
    out = {}
    for name in sorted(state_dict.keys()):
        ref = state_dict[name]
        # the reference registers ONE transitive-weight Parameter under six names (model.py:32,45)
        sname = "trans_candidates_weights" if name.endswith("predicates_transitive_weights") else name
        g = torch.Generator().manual_seed((zlib.crc32(sname.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        shape = tuple(ref.shape)
        if not ref.is_floating_point():
            out[name] = torch.zeros(shape, dtype=ref.dtype)
        elif name.endswith("weight_u") or name.endswith("weight_v"):
            v = torch.randn(shape, generator=g)
            out[name] = v / v.norm().clamp_min(1e-12)
        elif name.endswith("running_var"):
            out[name] = 1.0 + 0.2 * torch.rand(shape, generator=g)
        elif name.endswith("running_mean"):
            out[name] = 0.1 * torch.randn(shape, generator=g)
        elif "att_emb" in name or "pred_embeddings" in name:
            out[name] = torch.randn(shape, generator=g)
        elif "candidates_weights" in name or "transitive_weights" in name:
            out[name] = torch.rand(shape, generator=g) * 2 - 1
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            out[name] = torch.randn(shape, generator=g) * (1.5 / max(fan_in, 1) ** 0.5)
        else:
            out[name] = 0.1 * torch.randn(shape, generator=g)
    return out
