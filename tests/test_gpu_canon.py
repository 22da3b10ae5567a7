"""Device-side canonical graph construction (csrc/canon.hip) on a real MI355X: bit-exact against the
reference's own outputs (golden), against the numpy oracle on seeded random scenes (ties, both
transitivity modes, permuted predicate ids), and size-independent properties at the C5 sizes
(B=48, 129 objects per sample)."""
import copy

import numpy as np
import pytest
import torch

from canonicalsg2im_amd.synth import make_vocab
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _scene(rng, sizes, vocab, ties=True):
    B, O = len(sizes), max(sizes) + 1
    objs = np.zeros((B, O, len(vocab["attributes"])), np.int64)
    boxes = -np.ones((B, O, 4), np.float32)
    cen = np.zeros((B, O, 2), np.float32)
    for b, n in enumerate(sizes):
        wh = rng.uniform(0.05, 0.6, size=(n, 2))
        xy = rng.uniform(0.0, 1.0, size=(n, 2)) * (1.0 - wh)
        if ties and n >= 6:
            xy[1], wh[1] = xy[0], wh[0]
            xy[3, 0] = xy[2, 0]
            wh[5, 1] = wh[4, 1]; xy[5, 1] = xy[4, 1]
        bx = np.concatenate([xy, wh], axis=1)
        boxes[b, :n] = bx.astype(np.float32)
        cen[b, :n] = np.stack([bx[:, 0] + 0.5 * bx[:, 2], bx[:, 1] + 0.5 * bx[:, 3]], axis=1).astype(np.float32)
        for k, a in enumerate(vocab["attributes"]):
            objs[b, :n, k] = rng.integers(1, max(vocab["attributes"][a].values()) + 1, size=n)
        # row n is the __image__ object (id 0), rows > n padding (also 0): n_objs tells them apart
    return objs, boxes, cen, np.asarray([n + 1 for n in sizes], np.int64)


def _run(objs, boxes, cen, n, vocab, **kw):
    from canonicalsg2im_amd.sg2im.data import canonical_triplets
    t, cc, tt = canonical_triplets(torch.from_numpy(objs).cuda(), torch.from_numpy(boxes).cuda(),
                                   torch.from_numpy(cen).cuda(), torch.from_numpy(n).cuda(), vocab, **kw)
    return t.cpu().numpy(), cc, tt.cpu().numpy()


def test_canonical_triplets_vs_reference_golden():
    meta, a = load_golden("canon_graph")
    vocab = make_vocab(meta["vocab"])
    for ci, case in enumerate(meta["cases"]):
        g = {k[len("c%d_" % ci):]: v.numpy() for k, v in a.items() if k.startswith("c%d_" % ci)}
        t, cc, tt = _run(g["objs"], g["boxes"], g["centers"], g["n"], vocab,
                         learned_transitivity=bool(case["learned_transitivity"]))
        assert t.dtype == np.int64 and t.shape == g["triplets"].shape, (t.shape, g["triplets"].shape)
        assert np.array_equal(t, g["triplets"]), ci
        assert np.array_equal(tt, g["tt"]), ci
        assert cc.shape == (len(g["n"]), 8, 9) and float(cc.abs().sum()) == 0.0


@pytest.mark.parametrize("trans", [False, True])
@pytest.mark.parametrize("dummies", [True, False])
def test_canonical_triplets_vs_oracle(trans, dummies):
    from oracle import canon
    rng = np.random.default_rng(11 + trans + 2 * dummies)
    vocab = make_vocab("clevr")
    objs, boxes, cen, n = _scene(rng, (1, 2, 5, 17, 64, 65, 100, 130), vocab)
    t, _, tt = _run(objs, boxes, cen, n, vocab, learned_transitivity=trans, include_dummies=dummies)
    to, tto, _ = canon.canonical_batch(objs[:, :, 0], boxes, cen, n, vocab, trans, dummies)
    assert np.array_equal(t, to) and np.array_equal(tt, tto)


def test_converse_edges_vs_reference_golden():
    """`--learned_converse 1` on the device against the reference's own draws (tests/golden/canon_converse.npz): the
    fixture's seed replayed through numpy's GLOBAL stream — what `canonical_triplets` reads when no numbers are passed —
    triplets, types and conv_counts bit for bit."""
    meta, a = load_golden("canon_converse")
    vocab = make_vocab(meta["vocab"])
    for ci, case in enumerate(meta["cases"]):
        g = {k[len("c%d_" % ci):]: v.numpy() for k, v in a.items() if k.startswith("c%d_" % ci)}
        np.random.seed(case["seed"])
        t, cc, tt = _run(g["objs"], g["boxes"], g["centers"], g["n"], vocab,
                         learned_transitivity=bool(case["learned_transitivity"]), learned_converse=True,
                         converse_weights=g["weights"])
        assert t.shape == g["triplets"].shape, (ci, t.shape, g["triplets"].shape)
        assert np.array_equal(t, g["triplets"]), ci
        assert np.array_equal(tt, g["tt"]), ci
        assert np.array_equal(cc.cpu().numpy(), g["conv"].astype(np.float32)), ci
        # the stream was advanced by exactly the reference's number of draws
        np.random.seed(case["seed"])
        np.random.random_sample(case["draws"])
        expect_next = np.random.random_sample()
        np.random.seed(case["seed"])
        _run(g["objs"], g["boxes"], g["centers"], g["n"], vocab, learned_transitivity=bool(case["learned_transitivity"]),
             learned_converse=True, converse_weights=g["weights"])
        assert np.random.random_sample() == expect_next


@pytest.mark.parametrize("trans", [False, True])
def test_converse_edges_vs_oracle_dense_scenes(trans):
    """Dense scenes (up to 130 objects: thousands of draws per sample, converse edges closing cycles) against the oracle
    with the same explicit uniform numbers."""
    from oracle import canon
    rng = np.random.default_rng(21 + trans)
    vocab = make_vocab("clevr")
    objs, boxes, cen, n = _scene(rng, (1, 2, 7, 40, 64, 65, 130), vocab)
    P = len(vocab["pred_name_to_idx"])
    w = rng.normal(size=(P, P)).astype(np.float32)
    w = np.triu(w) + np.triu(w).T
    u = rng.random(200000)
    t, cc, tt = _run(objs, boxes, cen, n, vocab, learned_transitivity=trans, learned_converse=True, converse_weights=w,
                     uniforms=u)
    to, tto, _, conv = canon.canonical_batch(objs[:, :, 0], boxes, cen, n, vocab, trans, True, True, w, u)
    assert t.shape == to.shape and np.array_equal(t, to) and np.array_equal(tt, tto)
    assert np.array_equal(cc.cpu().numpy(), conv.astype(np.float32)) and conv[:, :, :-1].sum() > 100
    if trans:
        loops = t[(tt == 1) & (t[..., 0] == t[..., 2])]
        assert len(loops) > 0                      # converse edges closed cycles: `path` marks i -> i, as the reference's


def test_permuted_predicate_ids():
    """The (s, p, o) sort and the transitive order follow the NUMERIC predicate ids, whatever they are."""
    from oracle import canon
    rng = np.random.default_rng(3)
    vocab = copy.deepcopy(make_vocab("coco"))
    names = list(vocab["pred_idx_to_name"])
    perm = [3, 7, 1, 0, 6, 2, 5, 4]
    vocab["pred_idx_to_name"] = [names[i] for i in perm]
    vocab["pred_name_to_idx"] = {nm: i for i, nm in enumerate(vocab["pred_idx_to_name"])}
    objs, boxes, cen, n = _scene(rng, (9, 30, 4), vocab)
    t, _, tt = _run(objs, boxes, cen, n, vocab, learned_transitivity=True)
    to, tto, _ = canon.canonical_batch(objs[:, :, 0], boxes, cen, n, vocab, True, True)
    assert np.array_equal(t, to) and np.array_equal(tt, tto)


def test_full_size_properties():
    """C5 sizes: 48 samples x 128 objects (+ __image__).  Sortedness/uniqueness of the original triplets,
    reduction and closure identities per relation (checked with dense boolean algebra on the GPU), and
    two samples against the oracle."""
    from oracle import canon
    rng = np.random.default_rng(99)
    vocab = make_vocab("clevr")
    sizes = [128] * 46 + [97, 64]
    objs, boxes, cen, n = _scene(rng, sizes, vocab, ties=False)
    t, _, tt = _run(objs, boxes, cen, n, vocab, learned_transitivity=True)
    pad = vocab["pred_name_to_idx"]["__padding__"]
    for b in (0, 17, 46, 47):
        nb = int(n[b])
        orig = t[b][(tt[b] == 0) & (t[b][:, 1] != pad)]
        key = (orig[:, 0] * 16 + orig[:, 1]) * 1024 + orig[:, 2]
        assert np.all(np.diff(key) > 0)                                    # np.unique order, no duplicates
        extra = t[b][tt[b] == 1]
        for p in range(2, 8):
            R = torch.zeros(nb, nb, device="cuda")
            e = orig[orig[:, 1] == p]
            R[e[:, 0], e[:, 2]] = 1
            X = torch.zeros(nb, nb, device="cuda")
            x = extra[extra[:, 1] == p]
            X[x[:, 0], x[:, 2]] = 1
            assert float((R * X).sum()) == 0
            Tm = ((R + X) > 0).float()
            C = R.clone()
            for _ in range(8):                                              # closure of R by repeated squaring
                C = ((C + C @ C) > 0).float()
            assert torch.equal(C, Tm)                                       # originals + extras == closure(originals)
            assert float((R * ((Tm @ Tm) > 0).float()).sum()) == 0          # no original edge is implied
    to, tto, _ = canon.canonical_batch(objs[[3, 47], :, 0], boxes[[3, 47]], cen[[3, 47]], n[[3, 47]], vocab, True, True)
    for k, b in enumerate((3, 47)):
        c = int((to[k][:, 1] != pad).sum())
        assert np.array_equal(t[b][:c], to[k][:c]) and np.array_equal(tt[b][:c], tto[k][:c])
        assert np.all(t[b][c:, 1] == pad)


def test_too_many_objects_is_refused():
    vocab = make_vocab("clevr")
    rng = np.random.default_rng(1)
    objs, boxes, cen, n = _scene(rng, (300,), vocab, ties=False)
    with pytest.raises(RuntimeError, match="at most 256 objects"):
        _run(objs, boxes, cen, n, vocab)
