"""The numpy oracle of the canonical graph construction (oracle/canon.py) against the outputs of the
reference's own BaseDataset / graphs_utils functions (tests/golden/canon_graph.npz).  Bit-exact."""
import numpy as np

from canonicalsg2im_amd.synth import make_vocab
from conftest import load_golden
from oracle import canon


def test_canonical_graph_matches_reference():
    meta, a = load_golden("canon_graph")
    vocab = make_vocab(meta["vocab"])
    for ci, case in enumerate(meta["cases"]):
        g = {k[len("c%d_" % ci):]: v.numpy() for k, v in a.items() if k.startswith("c%d_" % ci)}
        trip, tt, counts = canon.canonical_batch(g["objs"][:, :, 0], g["boxes"], g["centers"], g["n"], vocab,
                                                 learned_transitivity=bool(case["learned_transitivity"]))
        assert np.array_equal(counts, g["counts"]), (ci, counts, g["counts"])
        assert np.array_equal(trip, g["triplets"]), ci
        assert np.array_equal(tt, g["tt"]), ci
        if case["learned_transitivity"]:
            assert (tt == 1).any()


def test_reference_known_answer():
    """scripts/graphs_utils.py:166-193 (`test_reduce_transitive_edges`): the reference's own vector."""
    m = np.zeros((4, 4), bool)
    for s, _, o in [[0, 1, 1], [0, 1, 2], [0, 1, 3], [1, 1, 2], [3, 1, 1], [3, 1, 2]]:
        m[s, o] = True
    red = canon.hsu(canon.path(m))
    assert sorted(zip(*np.nonzero(red))) == [(0, 3), (1, 2), (3, 1)]
    assert np.array_equal(canon.path(red), canon.path(m))          # same reachability


def test_closure_and_reduction_properties():
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 33):
        order = rng.permutation(n)
        m = np.zeros((n, n), bool)
        for _ in range(3 * n):
            i, j = rng.integers(0, n, 2)
            if order[i] < order[j]:
                m[i, j] = True                                           # acyclic by construction
        t = canon.path(m)
        r = canon.hsu(t)
        assert not (r & ~t).any() and np.array_equal(canon.path(r), t)
        assert not (r & (t.astype(int) @ t.astype(int) > 0)).any()       # no edge of the reduction is implied


def test_canonical_graph_with_converse_edges_matches_reference():
    """`--learned_converse 1`: the reference drew its converse edges from numpy's global stream (seeded by the fixture
    script); the oracle replays the same stream — np.random.seed(seed); random_sample(draws) — through the cumulative
    distribution np.random.choice would search.  Triplets, types and conv_counts bit for bit."""
    meta, a = load_golden("canon_converse")
    vocab = make_vocab(meta["vocab"])
    for ci, case in enumerate(meta["cases"]):
        g = {k[len("c%d_" % ci):]: v.numpy() for k, v in a.items() if k.startswith("c%d_" % ci)}
        np.random.seed(case["seed"])
        u = np.random.random_sample(case["draws"])
        trip, tt, counts, conv = canon.canonical_batch(g["objs"][:, :, 0], g["boxes"], g["centers"], g["n"], vocab,
                                                       learned_transitivity=bool(case["learned_transitivity"]),
                                                       learned_converse=True, converse_weights=g["weights"], uniforms=u)
        assert np.array_equal(counts, g["counts"]), (ci, counts, g["counts"])
        assert np.array_equal(trip, g["triplets"]), ci
        assert np.array_equal(tt, g["tt"]), ci
        assert np.array_equal(conv, g["conv"]), ci
        assert conv.sum() == case["draws"] and conv[:, :-1].sum() > 0          # some converse edges were drawn
