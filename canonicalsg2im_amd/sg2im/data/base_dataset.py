"""Canonical scene-graph construction on the GPU (reference: sg2im/data/base_dataset.py).

The reference builds every sample's graph on the host, in python loops over numpy matrices:
`add_location_triplets` (pairwise geometry + per-relation transitive reduction via
scripts/graphs_utils.py `path`/`hsu`), `add_dummy_triplets`, `add_learnt_triplets`
(np.unique + optional transitive closure edges), then the collate function pads the triplets of a
batch.  At O = 128 objects that is ~2.5 s per graph.  `canonical_triplets`
does the same for a whole padded batch with two kernel launches (csrc/canon.hip); the result is
bit-identical to the reference's (tests/golden/canon_graph.npz)."""
import ctypes

import torch

from ..._lib import check, lib, ptr, stream

ORIGINAL_EDGE, TRANSITIVE_EDGE, SYMMETRIC_EDGE, ANTI_SYMMETRIC_EDGE = 0, 1, 2, 3          # base_dataset.py:7-10
meta_relations = ["__padding__", "__in_image__"]                                             # base_dataset.py:14
augmented_relations = ['__below__', '__above__', '__left of__', '__right of__', '__inside__', '__surrounding__']


def register_augmented_relations(vocab):
    """base_dataset.py:153-162: append the meta + location predicates that the vocab lacks."""
    vocab.setdefault("pred_name_to_idx", {})
    vocab.setdefault("pred_idx_to_name", [])
    for p in meta_relations + augmented_relations:
        if p not in vocab["pred_name_to_idx"]:
            vocab["pred_name_to_idx"][p] = max(list(vocab["pred_name_to_idx"].values()) + [-1]) + 1
            vocab["pred_idx_to_name"].append(p)
    return vocab


def canonical_triplets(objs, boxes, obj_centers, n_objs, vocab, learned_transitivity=False, include_dummies=True,
                       learned_converse=False):
    """Batched `add_location_triplets` + `add_dummy_triplets` + `add_learnt_triplets` + collate padding.

    objs (B,O) or (B,O,A) int64 (attribute 0 is used, as `objs['shape']` in packed_clevr_dialog.py:207),
    boxes (B,O,4) xywh, obj_centers (B,O,2), n_objs (B,) = objects per sample incl. its `__image__` row.
    Returns (triplets (B,T,3) int64, conv_counts (B,P,P+1) float32 zeros, triplet_type (B,T) int64) —
    the collate layout of the trainer's batch tuple."""
    if learned_converse:
        raise NotImplementedError("learned_converse samples edges with numpy's global RNG (graphs_utils.py:139-152); "
                                  "it stays on the host")
    objs0 = (objs[..., 0] if objs.dim() == 3 else objs).contiguous()
    first = list(vocab["attributes"].keys())[0]
    image_id = vocab["object_name_to_idx"]["__image__"]
    if vocab["attributes"][first]["__image__"] != image_id:
        raise ValueError("the __image__ id of the first attribute and of object_name_to_idx differ")
    B, O = objs0.shape
    p2i = vocab["pred_name_to_idx"]
    ids = (ctypes.c_int32 * 8)(*[p2i[n] for n in meta_relations + augmented_relations])
    boxes = boxes.to(torch.float32).contiguous()
    obj_centers = obj_centers.to(torch.float32).contiguous()
    n_objs = n_objs.to(device=objs0.device, dtype=torch.int64).contiguous()
    dev = objs0.device
    nbytes = lib.csg_canon_workspace(B)
    ws = torch.empty(nbytes // 8, device=dev, dtype=torch.int64)
    counts = torch.empty((B, 2), device=dev, dtype=torch.int64)
    check(lib.csg_canon_build(ptr(objs0), ptr(boxes), ptr(obj_centers), ptr(n_objs), B, O, ids, image_id,
                              1 if include_dummies else 0, 1 if learned_transitivity else 0, ptr(ws), nbytes,
                              ptr(counts), stream()), "canon_build")
    T = int(counts.sum(dim=1).max().item())          # the collate pads to the longest sample: one 8-byte read-back
    triplets = torch.empty((B, T, 3), device=dev, dtype=torch.int64)
    triplet_type = torch.empty((B, T), device=dev, dtype=torch.int64)
    check(lib.csg_canon_emit(ptr(objs0), ptr(n_objs), B, O, ids, image_id, 1 if include_dummies else 0,
                             1 if learned_transitivity else 0, ptr(ws), ptr(counts), T, ptr(triplets),
                             ptr(triplet_type), stream()), "canon_emit")
    n_rel = len(p2i)
    conv_counts = torch.zeros((B, n_rel, n_rel + 1), device=dev, dtype=torch.float32)     # base_dataset.py:93
    return triplets, conv_counts, triplet_type
