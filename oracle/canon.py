"""CPU oracle (numpy) of the canonical scene-graph construction of the packed datasets.

TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/__init__.py).  Restates, for one sample,
`BaseDataset.add_location_triplets` -> `add_dummy_triplets` -> `add_learnt_triplets` with
learned_converse = 0 (sg2im/data/base_dataset.py:35-151, called in this order by
sg2im/data/packed_clevr_dialog.py:205-209) and the graph helpers they use
(scripts/graphs_utils.py:15-71,96-100), then the triplet padding of the collate function
(sg2im/data/packed_clevr_dialog.py:309-315).  Integer/index work: the HIP path must match
bit-exactly.  Pinned by tests/golden/canon_graph.npz, produced by the reference's own functions.
"""
import numpy as np

AUGMENTED = ("__below__", "__above__", "__left of__", "__right of__", "__inside__", "__surrounding__")  # base_dataset.py:15
ORIGINAL_EDGE, TRANSITIVE_EDGE = 0, 1                                                                    # base_dataset.py:7-8


def location_relations(boxes, centers, real):
    """`add_location_triplets` pair loop (base_dataset.py:42-81) -> bool adjacency (6, N, N) in the
    order of AUGMENTED.  All comparisons in float32, as on the reference's 0-dim FloatTensors.
    Note the reference's `sx1 = sx0 + sw / 2` is the box CENTRE, not its right edge."""
    b = np.asarray(boxes, np.float32)
    c = np.asarray(centers, np.float32)
    N = b.shape[0]
    x0, y0 = b[:, 0], b[:, 1]
    x1 = (x0 + b[:, 2] / np.float32(2)).astype(np.float32)
    y1 = (y0 + b[:, 3] / np.float32(2)).astype(np.float32)
    S, O = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    pair = real[S] & real[O] & (S != O)
    surround = (x0[S] < x0[O]) & (x1[S] > x1[O]) & (y0[S] < y0[O]) & (y1[S] > y1[O])
    inside = (x0[S] > x0[O]) & (x1[S] < x1[O]) & (y0[S] > y0[O]) & (y1[S] < y1[O]) & ~surround
    other = ~surround & ~inside
    dx = (c[S, 0] - c[O, 0]).astype(np.float32)
    dy = (c[S, 1] - c[O, 1]).astype(np.float32)
    adj = np.zeros((6, N, N), bool)
    adj[0] = pair & other & (dy > 0)       # __below__
    adj[1] = pair & other & (dy < 0)       # __above__
    adj[2] = pair & other & (dx < 0)       # __left of__
    adj[3] = pair & other & (dx > 0)       # __right of__
    adj[4] = pair & inside
    adj[5] = pair & surround
    return adj


def path(m):
    """`path` (graphs_utils.py:15-27): in-place Warshall; row j absorbs row i when p[j][i]."""
    p = np.array(m, bool)
    n = p.shape[0]
    for i in range(n):
        rows = p[:, i].copy()
        rows[i] = False
        p[rows] |= p[i]
    return p


def hsu(m):
    """`hsu` (graphs_utils.py:30-38), the same sequential order over j; for fixed j the (i,k) updates
    touch neither column j nor row j unless m[j][j] is set (never for the acyclic location graphs)."""
    m = np.array(m, bool)
    n = m.shape[0]
    for j in range(n):
        if m[j, j]:                                   # cyclic input: replay the scalar loop exactly
            for i in range(n):
                if m[i, j]:
                    for k in range(n):
                        if m[j, k]:
                            m[i, k] = False
            continue
        rows = m[:, j].copy()
        m[np.ix_(rows, m[j])] = False
    return m


def canonical_graph(objs0, boxes, centers, vocab, learned_transitivity=False, include_dummies=True):
    """One sample: (triplets (T,3) int64, triplet_type (T,) int64).

    objs0: (O,) ids of the FIRST attribute (`objs['shape']` / the COCO category), unpadded."""
    objs0 = np.asarray(objs0)
    O = objs0.shape[0]
    p2i = vocab["pred_name_to_idx"]
    image_id = vocab["object_name_to_idx"]["__image__"]
    real = (objs0 != image_id) if O > 1 else np.zeros(O, bool)               # base_dataset.py:39-41
    adj = location_relations(boxes, centers, real)
    closure = np.stack([path(a) for a in adj])                                # triplets_to_minimal: graphs_utils.py:64-71
    minimal = np.stack([hsu(t) for t in closure])
    rows = []
    for r, name in enumerate(AUGMENTED):                                      # base_dataset.py:83-87
        s, o = np.nonzero(minimal[r])
        rows.append(np.stack([s, np.full_like(s, p2i[name]), o], axis=1))
    if include_dummies:                                                       # base_dataset.py:141-151
        first = list(vocab["attributes"].keys())[0]
        img = int(np.nonzero(objs0 == vocab["attributes"][first]["__image__"])[0].squeeze())
        others = np.array([i for i in range(O) if i != img], np.int64)
        rows.append(np.stack([others, np.full_like(others, p2i["__in_image__"]), np.full_like(others, img)], axis=1))
    trip = np.concatenate(rows, axis=0).astype(np.int64) if rows else np.zeros((0, 3), np.int64)
    trip = np.unique(trip, axis=0) if len(trip) else trip                     # base_dataset.py:90,126: sorted by (s,p,o)
    ttype = [ORIGINAL_EDGE] * len(trip)
    if learned_transitivity:                                                  # base_dataset.py:111-120, graphs_utils.py:96-100
        meta = {p2i["__padding__"], p2i["__in_image__"]}
        extra = []
        for rel in sorted(set(p2i.values()) - meta):                          # set of small ints iterates ascending
            names = [n for n in AUGMENTED if p2i[n] == rel]
            if not names:
                continue
            r = AUGMENTED.index(names[0])
            if not minimal[r].any():
                continue
            s, o = np.nonzero(path(minimal[r]) & ~minimal[r])
            extra.append(np.stack([s, np.full_like(s, rel), o], axis=1))
        if extra and sum(len(e) for e in extra):
            extra = np.concatenate(extra, axis=0).astype(np.int64)
            trip = np.concatenate([trip, extra], axis=0)
            ttype = ttype + [TRANSITIVE_EDGE] * len(extra)
    return trip, np.asarray(ttype, np.int64)


def canonical_batch(objs0, boxes, centers, n_objs, vocab, learned_transitivity=False, include_dummies=True):
    """Padded batch in collate layout (packed_clevr_dialog.py:309-315): triplets (B,T,3) padded with
    [0, __padding__, 0], triplet_type (B,T) padded with 0, and the per-sample triplet counts."""
    outs = []
    for b in range(len(n_objs)):
        n = int(n_objs[b])
        outs.append(canonical_graph(objs0[b][:n], boxes[b][:n], centers[b][:n], vocab, learned_transitivity,
                                    include_dummies))
    T = max([len(t) for t, _ in outs] + [0])
    B = len(outs)
    trip = np.zeros((B, T, 3), np.int64)
    trip[:, :, 1] = vocab["pred_name_to_idx"]["__padding__"]
    ttype = np.zeros((B, T), np.int64)
    counts = np.zeros(B, np.int64)
    for b, (t, tt) in enumerate(outs):
        trip[b, :len(t)] = t
        ttype[b, :len(t)] = tt
        counts[b] = len(t)
    return trip, ttype, counts
