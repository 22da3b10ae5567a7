"""CPU oracle (numpy) of the canonical scene-graph construction of the packed datasets.

TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/__init__.py).  Restates, for one sample,
`BaseDataset.add_location_triplets` -> `add_dummy_triplets` -> `add_learnt_triplets`
(sg2im/data/base_dataset.py:35-151, called in this order by sg2im/data/packed_clevr_dialog.py:205-209;
learned_converse = 1 included: `get_edge_converse_triplets`, scripts/graphs_utils.py:126-152, with the uniform
numbers numpy.random.choice would draw passed in explicitly) and the graph helpers they use
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


def choice_cdf(converse_weights, rel, candidates):
    """The cumulative distribution `np.random.choice(dist_vals, p=dist)` searches (graphs_utils.py:128-140): scipy's
    softmax of the candidate weights and a 0 for "do not sample", then numpy's own float64 cumsum / normalisation."""
    from scipy.special import softmax
    dist = [converse_weights[rel, c] for c in candidates]
    dist.append(0)                                                            # e^0 = 1
    cdf = np.array(softmax(dist), dtype=np.double).cumsum()
    cdf /= cdf[-1]
    return cdf


def canonical_graph(objs0, boxes, centers, vocab, learned_transitivity=False, include_dummies=True,
                    learned_converse=False, converse_weights=None, uniforms=None):
    """One sample: (triplets (T,3) int64, triplet_type (T,) int64[, conv_counts (P,P+1) float64 with learned_converse]).

    objs0: (O,) ids of the FIRST attribute (`objs['shape']` / the COCO category), unpadded.
    learned_converse: `converse_weights` (P,P) as the data loader holds them (numpy float32, scripts/train.py:276) and
    `uniforms`, an iterator over the numbers np.random.random_sample() would return — one per original triplet of a
    non-meta relation, in the order of the reference's loops."""
    objs0 = np.asarray(objs0)
    O = objs0.shape[0]
    p2i = vocab["pred_name_to_idx"]
    n_rel = len(p2i)
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
    trip = np.unique(trip, axis=0) if len(trip) else trip                     # base_dataset.py:90: sorted by (s,p,o)
    meta = {p2i["__padding__"], p2i["__in_image__"]}
    non_meta = sorted(set(p2i.values()) - meta)                               # a set of small ints iterates ascending
    conv_counts = np.zeros((n_rel, n_rel + 1))                                # :93
    new = []                                                                  # :99-109
    for rel in non_meta:
        rel_t = trip[trip[:, 1] == rel] if len(trip) else trip
        if len(rel_t) == 0:
            continue
        new.extend(rel_t.tolist())
        if learned_converse:                                                  # graphs_utils.py:126-152
            cands = [c for c in non_meta if c != rel]
            cdf = choice_cdf(converse_weights, rel, cands)
            vals = cands + [n_rel]
            for t in rel_t:
                r = vals[int(np.searchsorted(cdf, next(uniforms), side="right"))]
                conv_counts[rel, r] += 1
                if r != n_rel:
                    new.append([int(t[2]), r, int(t[0])])
    extra = []
    if learned_transitivity and len(new):                                     # :111-120, graphs_utils.py:96-100
        arr = np.asarray(new, np.int64)
        for rel in non_meta:
            rel_t = arr[arr[:, 1] == rel]
            if not len(rel_t):
                continue
            N = int(max(rel_t[:, 0].max(), rel_t[:, 2].max()) + 1)             # triplets_to_adj_matrix
            g = np.zeros((N, N), bool)
            g[rel_t[:, 0], rel_t[:, 2]] = True
            s, o = np.nonzero(path(g) & ~g)
            if len(s):
                extra.append(np.stack([s, np.full_like(s, rel), o], axis=1))
    for rel in sorted(meta):                                                  # :122-124
        new.extend(trip[trip[:, 1] == rel].tolist() if len(trip) else [])
    trip = np.unique(np.asarray(new, np.int64).reshape(-1, 3), axis=0) if len(new) else np.zeros((0, 3), np.int64)   # :127-128
    ttype = [ORIGINAL_EDGE] * len(trip)
    if extra:
        extra = np.concatenate(extra, axis=0).astype(np.int64)
        trip = np.concatenate([trip, extra], axis=0)
        ttype = ttype + [TRANSITIVE_EDGE] * len(extra)
    if learned_converse:
        return trip, np.asarray(ttype, np.int64), conv_counts
    return trip, np.asarray(ttype, np.int64)


def canonical_batch(objs0, boxes, centers, n_objs, vocab, learned_transitivity=False, include_dummies=True,
                    learned_converse=False, converse_weights=None, uniforms=None):
    """Padded batch in collate layout (packed_clevr_dialog.py:309-315): triplets (B,T,3) padded with
    [0, __padding__, 0], triplet_type (B,T) padded with 0, and the per-sample triplet counts (+ conv_counts (B,P,P+1)
    with learned_converse; the samples consume `uniforms` one after the other, as the data loader's loop does)."""
    outs, convs = [], []
    it = iter(uniforms) if uniforms is not None else None
    for b in range(len(n_objs)):
        n = int(n_objs[b])
        r = canonical_graph(objs0[b][:n], boxes[b][:n], centers[b][:n], vocab, learned_transitivity, include_dummies,
                            learned_converse, converse_weights, it)
        outs.append(r[:2])
        if learned_converse:
            convs.append(r[2])
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
    if learned_converse:
        return trip, ttype, counts, np.stack(convs)
    return trip, ttype, counts
