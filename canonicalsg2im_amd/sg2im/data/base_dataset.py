"""Canonical scene-graph construction on the GPU (reference: sg2im/data/base_dataset.py).

The reference builds every sample's graph on the host, in python loops over numpy matrices:
`add_location_triplets` (pairwise geometry + per-relation transitive reduction via
scripts/graphs_utils.py `path`/`hsu`), `add_dummy_triplets`, `add_learnt_triplets`
(np.unique + optional transitive closure edges), then the collate function pads the triplets of a
batch.  At O = 128 objects that is ~2.5 s per graph.  `canonical_triplets`
does the same for a whole padded batch with two kernel launches (csrc/canon.hip); the result is
bit-identical to the reference's (tests/golden/canon_graph.npz)."""
import ctypes

import numpy as np

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


def _choice_cdf(converse_weights, rel, candidates):
    """The cumulative distribution `np.random.choice(dist_vals, p=dist)` searches (scripts/graphs_utils.py:128-140): scipy's
    softmax of the candidate weights and a 0 for "do not sample", numpy's float64 cumsum and normalisation — computed with
    the same library calls, so the device's `u < cdf[j]` comparisons decide exactly as numpy does."""
    from scipy.special import softmax
    dist = [converse_weights[rel, c] for c in candidates]
    dist.append(0)
    cdf = np.array(softmax(dist), dtype=np.double).cumsum()
    cdf /= cdf[-1]
    return cdf


def canonical_triplets(objs, boxes, obj_centers, n_objs, vocab, learned_transitivity=False, include_dummies=True,
                       learned_converse=False, converse_weights=None, uniforms=None):
    """Batched `add_location_triplets` + `add_dummy_triplets` + `add_learnt_triplets` + collate padding.

    objs (B,O) or (B,O,A) int64 (attribute 0 is used, as `objs['shape']` in packed_clevr_dialog.py:207),
    boxes (B,O,4) xywh, obj_centers (B,O,2), n_objs (B,) = objects per sample incl. its `__image__` row.
    Returns (triplets (B,T,3) int64, conv_counts (B,P,P+1) float32, triplet_type (B,T) int64) —
    the collate layout of the trainer's batch tuple.

    `learned_converse=True` (base_dataset.py:104-107): `converse_weights` is the (P,P) array the data loader holds
    (`get_conv_converse(model).detach().cpu().numpy()`, scripts/train.py:276); every original triplet of a location relation
    draws one number from numpy's GLOBAL random stream — the reference's `np.random.choice` — in the reference's order
    (samples one after the other), or from `uniforms` (a float64 sequence) when given.  Costs one more 16-byte-per-sample
    read-back (the number of draws is known only after the graphs are reduced)."""
    objs0 = (objs[..., 0] if objs.dim() == 3 else objs).contiguous()
    first = list(vocab["attributes"].keys())[0]
    image_id = vocab["object_name_to_idx"]["__image__"]
    if vocab["attributes"][first]["__image__"] != image_id:
        raise ValueError("the __image__ id of the first attribute and of object_name_to_idx differ")
    B, O = objs0.shape
    p2i = vocab["pred_name_to_idx"]
    names = meta_relations + augmented_relations
    ids = (ctypes.c_int32 * 8)(*[p2i[n] for n in names])
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
    n_rel = len(p2i)
    conv_counts = torch.zeros((B, n_rel, n_rel + 1), device=dev, dtype=torch.float32)     # base_dataset.py:93
    if learned_converse:
        if converse_weights is None:
            raise ValueError("learned_converse needs the data loader's converse_candidates_weights")
        w = converse_weights.detach().cpu().numpy() if torch.is_tensor(converse_weights) else np.asarray(converse_weights)
        # one draw per original triplet of the six location relations = originals minus the __in_image__ dummies
        in_range = torch.arange(O, device=dev).unsqueeze(0) < n_objs.unsqueeze(1)
        has_img = ((objs0 == image_id) & in_range).any(dim=1)
        dummies = torch.where(has_img, n_objs - 1, torch.zeros_like(n_objs)) if include_dummies else torch.zeros_like(n_objs)
        draws = (counts[:, 0] - dummies).cpu()                       # read-back: the uniforms are the HOST's random stream
        u_off = torch.cumsum(draws, 0) - draws
        total = int(draws.sum())
        if uniforms is None:
            u = np.random.random_sample(total)                       # the reference's np.random.choice draws, in its order
        else:
            u = np.asarray(uniforms, np.float64).reshape(-1)
            if u.shape[0] < total:
                raise ValueError("learned_converse: %d uniform numbers given, %d needed" % (u.shape[0], total))
        loc = [p2i[n] for n in augmented_relations]
        cdf = np.zeros((6, 6), np.float64)
        for r, rel in enumerate(loc):
            cdf[r] = _choice_cdf(w, rel, sorted(c for c in loc if c != rel))
        cdf_d = torch.from_numpy(cdf).to(dev)
        u_d = torch.from_numpy(np.ascontiguousarray(u[:max(total, 1)] if total else np.zeros(1))).to(dev)
        off_d = u_off.to(dev)
        check(lib.csg_canon_converse(ptr(objs0), ptr(n_objs), B, O, ids, image_id, 1 if include_dummies else 0,
                                     1 if learned_transitivity else 0, ptr(ws), ptr(cdf_d), ptr(u_d), ptr(off_d), n_rel,
                                     ptr(conv_counts), ptr(counts), stream()), "canon_converse")
    T = int(counts.sum(dim=1).max().item())          # the collate pads to the longest sample: one 8-byte read-back
    triplets = torch.empty((B, T, 3), device=dev, dtype=torch.int64)
    triplet_type = torch.empty((B, T), device=dev, dtype=torch.int64)
    check(lib.csg_canon_emit(ptr(objs0), ptr(n_objs), B, O, ids, image_id, 1 if include_dummies else 0,
                             1 if learned_transitivity else 0, ptr(ws), ptr(counts), T, ptr(triplets),
                             ptr(triplet_type), stream()), "canon_emit")
    return triplets, conv_counts, triplet_type
