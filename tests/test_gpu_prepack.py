"""Multi-tensor Winograd weight packs (csg_wino_pack_weights_multi) and the pack-ahead registry of ops.prepack_weights on a
real MI355X: bit-identical operands, bit-identical training steps with the registry on and off, operands actually served
from the registry, nothing stale after an optimiser step or an in-place edit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from canonicalsg2im_amd import ops as O
    return O


def test_multi_pack_is_bit_identical_to_single_packs(ops):
    """30 weights (more than one launch's 24 items), both kernel families, both directions, contiguous and channels-last,
    channel counts that are not multiples of the 32 x 32 pack tile."""
    from canonicalsg2im_amd._lib import WinoPackItem, check, lib, stream
    g = torch.Generator().manual_seed(3)
    shapes = [(128, 32), (256, 128), (64, 128), (36, 20), (100, 72), (512, 256), (8, 8), (128, 128)]
    items, outs, want = [], [], []
    ws = []
    for i in range(30):
        Cout, Cin = shapes[i % len(shapes)]
        w = torch.randn(Cout, Cin, 3, 3, generator=g).cuda()
        if i % 3 == 0:
            w = w.contiguous(memory_format=torch.channels_last)
        bd, var = bool(i & 1), (4 if (i >> 1) & 1 else 2)
        N, K = (Cin, Cout) if bd else (Cout, Cin)
        nbytes = lib.csg_wino4_pack_bytes(N, K) if var == 4 else lib.csg_wino_pack_bytes(N, K)
        out = torch.full((nbytes // 4,), float("nan"), device="cuda")
        it = WinoPackItem()
        st = w.stride()
        it.w, it.s_o, it.s_i, it.s_h, it.s_w = w.data_ptr(), st[0], st[1], st[2], st[3]
        it.Cout, it.Cin, it.backward_data, it.variant, it.packed = Cout, Cin, int(bd), var, out.data_ptr()
        items.append(it)
        outs.append(out)
        ws.append(w)
        saved = ops.PREPACK
        ops.PREPACK = False
        try:
            want.append(ops.wino_pack(w, bd, None, var))
        finally:
            ops.PREPACK = saved
    arr = (WinoPackItem * len(items))(*items)
    check(lib.csg_wino_pack_weights_multi(arr, len(items), stream()), "multi")
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, want)):
        assert torch.equal(o, r), "item %d differs" % i
    bad = WinoPackItem()
    bad.w, bad.Cout, bad.Cin, bad.variant, bad.packed = ws[0].data_ptr(), 8, 8, 34, outs[0].data_ptr()
    assert lib.csg_wino_pack_weights_multi((WinoPackItem * 1)(bad), 1, stream()) != 0     # F(3x3,4x4) operands: not served


def _trainer(prepack):
    from canonicalsg2im_amd import ops, train as T
    from canonicalsg2im_amd.synth import BatchConfig, make_batch, make_vocab
    vocab = make_vocab("tiny")
    opt = T.make_opt(vocab, ["--image_size", "64,64", "--ngf", "8", "--ndf", "8", "--gconv_dim", "32", "--gconv_hidden_dim", "64",
                             "--gconv_num_layers", "2", "--embedding_dim", "8", "--no_vgg_loss", "--batch_size", "4",
                             "--use_img_disc", "1"])
    torch.manual_seed(0)
    tr = T.Trainer(opt, torch.device("cuda:0"))
    tr.graphs = None
    batches = [[None if t is None else t.cuda() for t in make_batch(vocab, BatchConfig(4, 64, 2, 6, "packed"), seed=20 + i)]
               for i in range(2)]
    return tr, batches


def test_training_steps_with_and_without_the_registry_are_bit_identical(ops, monkeypatch):
    """Four eager steps of the tiny recipe (image discriminator only: no atomics anywhere) with pack-ahead on and off:
    same losses, same images, same parameters, bit for bit — and from the second step on the generator's operands come
    out of the registry (the one-weight pack is not called for them)."""
    import copy
    results = {}
    for on in (True, False):
        monkeypatch.setattr(ops, "PREPACK", on)
        ops._PREPACKED.clear()
        ops._PACK_OWNER.clear()
        tr, batches = _trainer(on)
        if on:
            init = (copy.deepcopy(tr.model.state_dict()), copy.deepcopy(tr.discriminator.state_dict()))
        else:
            tr.model.load_state_dict(init[0])
            tr.discriminator.load_state_dict(init[1])
            ops.invalidate_weight_caches()
        singles = []
        real = ops.lib.csg_wino4_pack_weights
        real2 = ops.lib.csg_wino_pack_weights
        calls = {"n": 0}

        def count4(*a):
            calls["n"] += 1
            return real(*a)

        def count2(*a):
            calls["n"] += 1
            return real2(*a)

        monkeypatch.setattr(ops.lib, "csg_wino4_pack_weights", count4)
        monkeypatch.setattr(ops.lib, "csg_wino_pack_weights", count2)
        log = []
        for it in range(4):
            calls["n"] = 0
            G, D = tr.step(batches[it % 2])
            singles.append(calls["n"])
            log.append(({k: v.detach().clone() for k, v in G.items()}, {k: v.detach().clone() for k, v in D.items()},
                        tr.last_model_out[0].detach().clone()))
        monkeypatch.setattr(ops.lib, "csg_wino4_pack_weights", real)
        monkeypatch.setattr(ops.lib, "csg_wino_pack_weights", real2)
        results[on] = (log, [p.detach().clone() for p in tr.model.parameters()], singles)
    (la, pa, sa), (lb, pb, sb) = results[True], results[False]
    for it, ((Ga, Da, ia), (Gb, Db, ib)) in enumerate(zip(la, lb)):
        for k in Ga:
            assert torch.equal(Ga[k], Gb[k]), "step %d G.%s" % (it, k)
        for k in Da:
            assert torch.equal(Da[k], Db[k]), "step %d D.%s" % (it, k)
        assert torch.equal(ia, ib), "step %d image" % it
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
    assert len(ops._PACK_OWNER) < 200 and len(ops._PREPACKED) < 200      # neither map grows with the steps
    # one-weight packs: every step the same number without the registry; with it, the first step asks and is noted, and
    # from the second step on every operand of the generator is found ready
    assert sb[0] == sb[1] == sb[2] == sb[3] and sa[0] == sb[0], (sa, sb)
    assert sa[1] == sa[2] == sa[3] == 0, (sa, sb)


def test_registry_refuses_stale_operands(ops):
    """An operand parked for a weight is not handed out once the weight epoch has moved (any optimiser step,
    invalidate_weight_caches) or for a tensor of another shape at the same address."""
    base = torch.randn(64 * 32 * 9, device="cuda")
    w = base.view(64, 32, 3, 3)
    saved = ops.PREPACK
    ops.PREPACK = False
    try:
        good = ops.wino_pack(w, False, None, 4)
    finally:
        ops.PREPACK = saved
    key = (w.data_ptr(), False, 4)
    marker = torch.zeros_like(good)
    ops._PREPACKED[key] = (marker,) + ops._pack_tag(w)
    assert ops.wino_pack(w, False, None, 4) is marker                       # parked and fitting: handed out, once
    assert key not in ops._PREPACKED
    ops._PREPACKED[key] = (marker,) + ops._pack_tag(w)
    ops.invalidate_weight_caches()
    assert torch.equal(ops.wino_pack(w, False, None, 4), good)               # the epoch moved: packed afresh
    ops._PREPACKED[key] = (marker,) + ops._pack_tag(w)
    v = base[:32 * 16 * 9].view(32, 16, 3, 3)                                # same address, another weight
    got = ops.wino_pack(v, False, None, 4)
    assert got is not marker and got.numel() != marker.numel()
    ops._PREPACKED.clear()
