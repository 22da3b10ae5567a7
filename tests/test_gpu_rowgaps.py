"""Rows a4 / a7 / a8 / a12 of SURVEY.md §8, the parts that were missing after round 1, on a real MI355X against the
reference's own outputs (tests/golden/row_gaps.npz, layout.npz) and the oracle: box gradients of `boxes_to_layout`
and `masks_to_layout`, non-square layouts through the Python surface, `masks_to_layout(test_mode=True)`
(painter's compositing), `build_mlp(batch_norm='batch')`, the affine SynchronizedBatchNorm2d."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, load_golden, state_from_shapes

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def test_boxes_to_layout_box_gradients_and_non_square_vs_reference():
    """sg2im/layout.py:12-45 through the drop-in function: (H, W) = 16x16, 32x32 and 24x40; gradients w.r.t. the
    vectors AND the boxes (the grid of layout.py:98-110 is differentiable in x0, y0, w, h)."""
    from canonicalsg2im_amd.sg2im.layout import boxes_to_layout
    meta, a = load_golden("layout")
    for H, W in meta["sizes"]:
        tag = "%dx%d" % (H, W)
        vecs = a["vecs"].cuda().requires_grad_(True)
        boxes = a["boxes"].cuda().requires_grad_(True)
        out = boxes_to_layout(vecs, boxes, H, W)
        assert out.shape == (1, 8, H, W)
        assert_close(out, a["out_" + tag], RTOL, 2e-6, "layout " + tag)
        gv, gb = torch.autograd.grad((out * a["w_" + tag].cuda()).sum(), [vecs, boxes])
        assert_close(gv, a["gvecs_" + tag], RTOL, 1e-5, "layout dvecs " + tag)
        # a box gradient multiplies O(H*W) border terms by 8/w (up to 8/0.07 here): absolute term scaled to its size
        assert_close(gb, a["gboxes_" + tag], RTOL, 1e-5 * float(a["gboxes_" + tag].abs().max()), "layout dboxes " + tag)


def test_masks_to_layout_box_gradients_vs_reference():
    from canonicalsg2im_amd.sg2im.layout import masks_to_layout
    meta, a = load_golden("row_gaps")
    for (H, W) in ((32, 32), (24, 40)):
        t = "%dx%d" % (H, W)
        vecs = a["p_vecs"].cuda().requires_grad_(True)
        boxes = a["p_boxes"].cuda().requires_grad_(True)
        out = masks_to_layout(vecs, boxes, a["p_soft"].cuda(), H, W)
        assert_close(out, a["m_out_" + t], RTOL, 2e-6, "masks layout " + t)
        gv, gb = torch.autograd.grad((out * a["m_w_" + t].cuda()).sum(), [vecs, boxes])
        assert_close(gv, a["m_gvecs_" + t], RTOL, 1e-5, "masks layout dvecs " + t)
        assert_close(gb, a["m_gboxes_" + t], RTOL, 1e-5 * float(a["m_gboxes_" + t].abs().max()), "masks layout dboxes " + t)


def test_masks_to_layout_test_mode_vs_reference():
    """Painter's compositing (layout.py:135-151): binary and soft masks, square and non-square."""
    from canonicalsg2im_amd.sg2im.layout import masks_to_layout
    meta, a = load_golden("row_gaps")
    for (H, W) in ((32, 32), (24, 40)):
        for tag, key in (("bin", "p_bin"), ("soft", "p_soft")):
            out = masks_to_layout(a["p_vecs"].cuda(), a["p_boxes"].cuda(), a[key].cuda(), H, W, test_mode=True)
            want = a["paint_%s_%dx%d" % (tag, H, W)]
            assert out.shape == want.shape and not out.requires_grad
            assert_close(out, want, RTOL, 2e-6, "paint %s %dx%d" % (tag, H, W))


def test_generator_test_mode_layout_pyramid_vs_oracle():
    """SPADEGenerator.forward(test_mode=True) with masks (generator.py:88-90): every pyramid level the SPADE blocks
    read equals F.interpolate(nearest) of the oracle's full-resolution composited layout; ragged batch (padded rows)."""
    import oracle
    from canonicalsg2im_amd import ops
    from canonicalsg2im_amd.synth import BatchConfig, make_batch, make_vocab
    vocab = make_vocab("coco")
    imgs, objs, boxes, _, _, _, masks, _ = make_batch(vocab, BatchConfig(3, 64, 2, 7, "random", mask_size=16), seed=9)
    g = torch.Generator().manual_seed(3)
    vecs = torch.randn(3, objs.shape[1], 32, generator=g)
    valid = ((objs[..., 0] != 0)).to(torch.uint8)
    levels = (2, 4, 8, 16, 32, 64)
    maps = ops.layout_paint(vecs.cuda(), boxes.cuda(), valid.cuda(), masks.cuda(), 64, levels)
    for b in range(3):
        m = oracle.remove_dummy_objects(objs[b], vocab)
        full = oracle.masks_to_layout(vecs[b][m], boxes[b][m], masks[b][m], 64, 64, test_mode=True)
        for h, got in zip(levels, maps):
            want = F.interpolate(full, size=(h, h), mode="nearest")
            assert_close(got[b:b + 1], want, RTOL, 2e-6, "painted pyramid level %d sample %d" % (h, b))


def test_layout_box_gradients_random_vs_oracle():
    """A denser random case than the fixture (12 objects, S=32, 64x48 and the discriminator's packed input buffer)."""
    import oracle
    from canonicalsg2im_amd import ops
    g = torch.Generator().manual_seed(12)
    O, S, H, W = 12, 32, 64, 48
    vecs = torch.randn(1, O, S, generator=g)
    wh = torch.rand(1, O, 2, generator=g) * 0.5 + 0.08
    boxes = torch.cat([torch.rand(1, O, 2, generator=g) * (1.1 - wh) - 0.05, wh], -1)
    valid = torch.ones(1, O, dtype=torch.uint8)
    valid[0, 5] = 0
    vr, br = vecs[0].clone().requires_grad_(True), boxes[0].clone().requires_grad_(True)
    keep = valid[0].bool()
    ref = oracle.boxes_to_layout(vr[keep], br[keep], H, W)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    vd, bd = vecs.cuda().requires_grad_(True), boxes.cuda().requires_grad_(True)
    (out,) = ops.layout_pyramid(vd, bd, valid.cuda(), H, ((H, W),), W=W)
    assert_close(out, ref, RTOL, 1e-5, "layout")
    (out * w.cuda()).sum().backward()
    assert_close(vd.grad[0], vr.grad, RTOL, 1e-5 * float(vr.grad.abs().max()), "dvecs")
    assert_close(bd.grad[0], br.grad, RTOL, 1e-5 * float(br.grad.abs().max()), "dboxes")
    assert float(bd.grad[0, 5].abs().max()) == 0.0                      # masked-out object: no gradient


def test_build_mlp_batch_norm_vs_reference():
    """build_mlp([12, 24, 8], batch_norm='batch') (sg2im/layers.py:6-25): train step, gradients, running stats, eval."""
    from canonicalsg2im_amd.sg2im.layers import build_mlp
    meta, a = load_golden("row_gaps")
    mlp = build_mlp([12, 24, 8], batch_norm='batch')
    mlp.load_state_dict({k: v.detach().clone() for k, v in state_from_shapes(meta["mlp_shapes"], seed=62, requires_grad=False).items()})
    mlp = mlp.cuda().train()
    x = a["mlp_x"].cuda().requires_grad_(True)
    y = mlp(x)
    assert_close(y, a["mlp_y"], RTOL, 2e-6, "mlp(batch) y")
    (y * a["mlp_w"].cuda()).sum().backward()
    assert_close(x.grad, a["mlp_gx"], RTOL, 1e-5, "mlp(batch) dx")
    for k, p in mlp.named_parameters():
        gref = a["mlp_grad:" + k]
        if float(gref.abs().max()) < 1e-6:             # Linear bias in front of BatchNorm: analytically zero
            assert float(p.grad.abs().max()) < 1e-5, k
        else:
            assert_close(p.grad, gref, RTOL, 1e-5, "mlp(batch) d" + k)
    sd = mlp.state_dict()
    for k in ("1.running_mean", "1.running_var", "1.num_batches_tracked"):
        assert_close(sd[k], a["mlp_after:" + k], RTOL, 1e-6, "mlp(batch) " + k)
    mlp.eval()
    with torch.no_grad():
        assert_close(mlp(a["mlp_x"].cuda()), a["mlp_y_eval"], RTOL, 2e-6, "mlp(batch) eval")
    # the reference's module refuses (B, T, D) inputs whose dim 1 is not the feature count (nn.BatchNorm1d)
    with pytest.raises(RuntimeError, match="running_mean should contain"):
        mlp.train()(torch.randn(2, 5, 12).cuda())


def test_affine_synchronized_batchnorm_vs_reference():
    """SynchronizedBatchNorm2d(8) with weight and bias on one device (sync_batchnorm/batchnorm.py:51-68)."""
    from canonicalsg2im_amd.spade.models.networks.sync_batchnorm import SynchronizedBatchNorm2d
    meta, a = load_golden("row_gaps")
    bn = SynchronizedBatchNorm2d(8)
    with torch.no_grad():
        bn.weight.copy_(a["bn_weight"])
        bn.bias.copy_(a["bn_bias"])
    bn = bn.cuda().train()
    x = a["bn_x"].cuda().requires_grad_(True)
    y = bn(x)
    assert_close(y, a["bn_y"], RTOL, 2e-6, "affine syncbn y")
    (y * a["bn_w"].cuda()).sum().backward()
    assert_close(x.grad, a["bn_gx"], RTOL, 1e-5, "affine syncbn dx")
    assert_close(bn.weight.grad, a["bn_gweight"], RTOL, 1e-5, "affine syncbn dweight")
    assert_close(bn.bias.grad, a["bn_gbias"], RTOL, 1e-5, "affine syncbn dbias")
    assert_close(bn.running_mean, a["bn_running_mean"], RTOL, 1e-6, "running_mean")
    assert_close(bn.running_var, a["bn_running_var"], RTOL, 1e-6, "running_var")
    assert int(bn.num_batches_tracked) == 0               # the reference's forward never advances it
    bn.eval()
    with torch.no_grad():
        assert_close(bn(a["bn_x"].cuda()), a["bn_y_eval"], RTOL, 2e-6, "affine syncbn eval")


# ------------------------------------------------------------------ round 3: the rest of the config space
@pytest.mark.parametrize("shape", [(2, 8, 256, 256, 8, 8), (1, 32, 64, 64, 16, 16), (2, 4, 7, 9, 20, 31), (1, 12, 33, 17, 5, 100),
                                   (1, 4, 96, 40, 35, 13)])
def test_nearest_resize_vs_interpolate(shape):
    """ops.nearest_resize == F.interpolate(mode='nearest') bit for bit (forward), and its backward sums the gradient
    over the output pixels that read an input pixel (reference normalization.py:98, when SPADE is handed a plain
    segmentation tensor instead of the layout pyramid): integer and non-integer ratios, up- and down-sampling."""
    from canonicalsg2im_amd import ops
    B, C, IH, IW, OH, OW = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, C, IH, IW, generator=g)
    w = torch.randn(B, C, OH, OW, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(OH, OW), mode='nearest')
    (yr * w).sum().backward()
    xd = x.clone().cuda().requires_grad_(True)
    yd = ops.nearest_resize(xd, (OH, OW))
    (yd * w.cuda()).sum().backward()
    assert torch.equal(yd.detach().cpu(), yr.detach()), shape
    assert_close(xd.grad, xr.grad, 1e-5, 1e-6, "nearest dx %s" % (shape,))


def test_spade_with_plain_segmap_resamples_on_the_device():
    """SPADE.forward with a (B,S,H,W) tensor as segmap — the reference's calling convention — goes through
    ops.nearest_resize (no ATen interpolate kernel) and equals the pyramid path."""
    from canonicalsg2im_amd import ops
    from canonicalsg2im_amd.spade.models.networks.normalization import SPADE, SegPyramid
    torch.manual_seed(3)
    sp = SPADE('spadesyncbatch3x3', 32, 8).cuda().train()
    g = torch.Generator().manual_seed(4)
    x = ops.nhwc(torch.randn(2, 32, 16, 16, generator=g).cuda())
    seg = ops.nhwc(torch.randn(2, 8, 64, 64, generator=g).cuda())
    y_plain = sp(x, seg)
    seg16 = ops.nhwc(F.interpolate(seg, size=(16, 16), mode='nearest'))
    sp2 = SPADE('spadesyncbatch3x3', 32, 8).cuda().train()
    sp2.load_state_dict(sp.state_dict())
    y_pyr = sp2(x, SegPyramid({16: seg16}))
    assert torch.equal(y_plain, y_pyr)


@pytest.mark.parametrize("norm_D", ["batch", "instance"])
def test_nlayer_discriminator_norm_variants_vs_torch(norm_D):
    """The two non-spectral `norm_D` spellings.  The reference's `get_nonspade_norm_layer` raises UnboundLocalError for
    them (normalization.py:27-31 defines `subnorm_type` only under the `spectral` prefix), so no reference fixture can
    exist; they are held to the construction its code intends (conv -> BatchNorm2d(affine) / InstanceNorm2d -> LeakyReLU)
    built from torch.nn layers on the CPU.  The four spellings the reference CAN build are pinned to its own outputs in
    test_nlayer_discriminator_norm_variants_vs_reference below."""
    import argparse
    import torch.nn as nn
    from canonicalsg2im_amd.spade.models.networks.discriminator import NLayerDiscriminator
    opt = argparse.Namespace(ndf=8, n_layers_D=4, norm_D=norm_D, semantic_nc=5, no_ganFeat_loss=False)
    torch.manual_seed(11)
    D = NLayerDiscriminator(opt).cuda().train()
    # the torch twin
    use_sn = norm_D.startswith("spectral")
    sub = norm_D[len("spectral"):] if use_sn else norm_D
    seq, nf, cin = [], 8, 8
    first = nn.Conv2d(8, nf, 4, 2, 2)
    seq.append([first, nn.LeakyReLU(0.2)])
    for n in range(1, 4):
        nf_prev, nf = nf, min(nf * 2, 512)
        conv = nn.Conv2d(nf_prev, nf, 4, 1 if n == 3 else 2, 2, bias=sub in ("", "none"))
        layers = [torch.nn.utils.spectral_norm(conv) if use_sn else conv]
        if sub in ("batch", "sync_batch"):
            layers.append(nn.BatchNorm2d(nf))
        elif sub == "instance":
            layers.append(nn.InstanceNorm2d(nf))
        seq.append(layers + [nn.LeakyReLU(0.2)])
    seq.append([nn.Conv2d(nf, 1, 4, 1, 2)])
    ref = nn.ModuleList([nn.Sequential(*l) for l in seq]).train()
    sd = {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}
    twin = {}
    for k, v in sd.items():
        m = k.split(".")                       # model<i>.<j>... -> <i>.<j>...
        twin[m[0][len("model"):] + "." + ".".join(m[1:])] = v
    # our Sequential(conv, norm) block sits at index 0 of model<i>: torch twin has conv at <i>.0 and norm at <i>.1
    remap = {}
    for k, v in twin.items():
        parts = k.split(".")
        if len(parts) >= 3 and parts[1] == "0" and parts[2] in ("0", "1"):      # model<i>.0.<0|1>.<name>
            remap[parts[0] + "." + parts[2] + "." + ".".join(parts[3:])] = v
        else:
            remap[k] = v
    missing, unexpected = ref.load_state_dict(remap, strict=False)
    assert not unexpected, unexpected
    assert all("num_batches_tracked" in m for m in missing), missing
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 8, 40, 40, generator=g)
    xr = x.clone().requires_grad_(True)
    feats_r, h = [], xr
    for blk in ref:
        h = blk(h)
        feats_r.append(h)
    from canonicalsg2im_amd import ops
    xd = ops.nhwc(x.clone().cuda()).requires_grad_(True)
    feats = D(xd)
    assert len(feats) == len(feats_r)
    for i, (a, b) in enumerate(zip(feats, feats_r)):
        assert_close(a, b.detach(), RTOL, 1e-5 * float(b.detach().abs().max()) + 1e-6, "%s feature %d" % (norm_D, i))
    wgt = torch.randn(feats_r[-1].shape, generator=g)
    (feats_r[-1] * wgt).sum().backward()
    (feats[-1] * wgt.cuda()).sum().backward()
    assert_close(xd.grad, xr.grad, RTOL, 1e-5 * float(xr.grad.abs().max()) + 1e-7, "%s dx" % norm_D)
    gd = dict(D.named_parameters())
    gr = dict(ref.named_parameters())
    k0 = "model0.0.weight"
    assert_close(gd[k0].grad, gr["0.0.weight"].grad, RTOL, 1e-5 * float(gr["0.0.weight"].grad.abs().max()) + 1e-7, "%s dW0" % norm_D)
    assert_close(gd["model4.0.weight"].grad, gr["4.0.weight"].grad, RTOL, 1e-5 * float(gr["4.0.weight"].grad.abs().max()) + 1e-7,
                 "%s dW4" % norm_D)


def _load_variant(mine, entry, seed_key="seed"):
    """Same Sequential layout and state_dict keys as the reference's builder, then the fixture's deterministic weights."""
    assert list(mine.state_dict().keys()) == entry["keys"], (list(mine.state_dict().keys()), entry["keys"])
    sd = state_from_shapes(entry["shapes"], seed=entry[seed_key], requires_grad=False)
    mine.load_state_dict({k: v.clone() for k, v in sd.items()})
    return mine


def _check_grads_and_buffers(mine, a, tag, what, zero_grad_keys=()):
    """Parameter gradients and buffers against the fixture.  A gradient that is analytically zero (a bias whose output
    reaches a normalisation through nothing but resampling, pooling or a residual skip) holds rounding noise on both sides:
    recognised by its size in the FIXTURE (< 1e-5 of the largest gradient entry of the network) and held to that size."""
    gmax = max(float(a[k].abs().max()) for k in a if k.startswith(tag + "grad:"))
    n = 0
    for k, p in mine.named_parameters():
        key = tag + "grad:" + k
        if key not in a:
            continue
        want = a[key]
        if k in zero_grad_keys or float(want.abs().max()) < 1e-5 * gmax:
            assert float(p.grad.abs().max()) < 1e-4 * gmax + 1e-7, (what, k, float(p.grad.abs().max()), gmax)
            continue
        assert_close(p.grad, want, RTOL, 1e-5 * float(want.abs().max()) + 1e-6, "%s d%s" % (what, k))
        n += 1
    assert n >= 2, (what, n)
    sd = mine.state_dict()
    for key in a:
        if key.startswith(tag + "after:"):
            k = key[len(tag + "after:"):]
            assert_close(sd[k].float(), a[key].float(), RTOL, 1e-6, "%s buffer %s" % (what, k))


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_build_mlp_variants_vs_reference(idx):
    """build_mlp beyond the trainer's settings against the REFERENCE's builder (sg2im/layers.py:6-25, fixture
    tests/golden/variants.npz): LeakyReLU-x / sigmoid activations, a final non-linearity of another kind, BatchNorm1d,
    dropout (eval mode: the masks of two RNGs cannot agree) — Sequential length, state_dict keys, outputs, input and
    parameter gradients, running statistics."""
    from canonicalsg2im_amd.sg2im.layers import build_mlp
    meta, a = load_golden("variants")
    e = meta["mlp"][idx]
    mine = build_mlp(e["dims"], **e["kw"])
    assert len(mine) == e["len"]
    mine = _load_variant(mine, e).cuda()
    mine.train(e["train"])
    tag = "mlp%d_" % idx
    x = a[tag + "x"].cuda().requires_grad_(True)
    y = mine(x)
    (y * a[tag + "w"].cuda()).sum().backward()
    assert_close(y, a[tag + "y"], RTOL, 1e-5, "mlp out %s" % (e["kw"],))
    assert_close(x.grad, a[tag + "gx"], RTOL, 1e-5 * float(a[tag + "gx"].abs().max()) + 1e-6, "mlp dx %s" % (e["kw"],))
    _check_grads_and_buffers(mine, a, tag, "mlp %s" % (e["kw"],))


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 4, 5, 6])
def test_build_cnn_grammar_vs_reference(idx):
    """build_cnn's whole grammar — I / C / R / U / P (max, avg) / FC — on the HIP layers against the REFERENCE's builder
    (sg2im/layers.py:28-112,190-217, fixture tests/golden/variants.npz): same Sequential indices and state_dict keys,
    outputs, gradients, running statistics (a residual block's advance TWICE per call, as the reference's do)."""
    import torch.nn as nn
    from canonicalsg2im_amd import ops
    from canonicalsg2im_amd.sg2im.layers import build_cnn
    meta, a = load_golden("variants")
    e = meta["cnn"][idx]
    mine, cout = build_cnn(e["arch"], **e["kw"])
    assert cout == e["cout"] and len(mine) == e["len"]
    mine = _load_variant(mine, e).cuda().train()
    tag = "cnn%d_" % idx
    x = ops.nhwc(a[tag + "x"].cuda()).requires_grad_(True)
    y = mine(x)
    (y * a[tag + "w"].cuda()).sum().backward()
    want = a[tag + "y"]
    assert y.shape == want.shape, (y.shape, want.shape)
    assert_close(y, want, RTOL, 1e-5 * float(want.abs().max()) + 1e-6, "cnn out %s" % e["arch"])
    assert_close(x.grad, a[tag + "gx"], RTOL, 1e-5 * float(a[tag + "gx"].abs().max()) + 1e-6, "cnn dx %s" % e["arch"])
    # a conv bias whose output reaches a normalisation through nothing but resampling has an analytically zero gradient
    mods = list(mine)

    def feeds_norm(i):
        j = i + 1
        while j < len(mods) and type(mods[j]).__name__ in ("Interpolate", "MaxPool2"):
            j += 1
        return j < len(mods) and isinstance(mods[j], (nn.BatchNorm2d, nn.InstanceNorm2d))
    zero = {"%d.bias" % i for i in range(len(mods)) if isinstance(mods[i], nn.Conv2d) and feeds_norm(i)}
    _check_grads_and_buffers(mine, a, tag, "cnn %s" % e["arch"], zero_grad_keys=zero)


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_nlayer_discriminator_norm_variants_vs_reference(idx):
    """NLayerDiscriminator under the four `norm_D` spellings the reference can build (spectralinstance — the default —,
    spectralbatch, spectralsync_batch, spectralnone; normalization.py:16-50, discriminator.py:163-206) against the
    reference's own module (fixture tests/golden/variants.npz): state_dict keys, the five feature maps of a training-mode
    forward, d input, every parameter gradient, and the u / v vectors and running statistics after the call."""
    import argparse
    from canonicalsg2im_amd import ops
    from canonicalsg2im_amd.spade.models.networks.discriminator import NLayerDiscriminator
    meta, a = load_golden("variants")
    e = meta["nld"][idx]
    opt = argparse.Namespace(ndf=8, n_layers_D=4, norm_D=e["norm_D"], semantic_nc=5, no_ganFeat_loss=False)
    D = _load_variant(NLayerDiscriminator(opt), e).cuda().train()
    tag = "nld%d_" % idx
    x = ops.nhwc(a[tag + "x"].cuda()).requires_grad_(True)
    feats = D(x)
    assert len(feats) == e["n_feats"]
    for j, f in enumerate(feats):
        want = a[tag + "feat%d" % j]
        assert_close(f, want, RTOL, 1e-5 * float(want.abs().max()) + 1e-6, "%s feature %d" % (e["norm_D"], j))
    (feats[-1] * a[tag + "w"].cuda()).sum().backward()
    assert_close(x.grad, a[tag + "gx"], RTOL, 1e-5 * float(a[tag + "gx"].abs().max()) + 1e-7, "%s dx" % e["norm_D"])
    _check_grads_and_buffers(D, a, tag, "NLD " + e["norm_D"])


def test_masks_to_layout_mask_gradients_vs_reference():
    """masks_to_layout (layout.py:48-77) is differentiable in the masks (model.py trains mask_net through it): forward,
    d vecs, d boxes and d masks against the reference's autograd; 16 x 16 and 5 x 5 masks, square and non-square maps;
    the mask gradient is an ordered sum (bit-identical from run to run)."""
    from canonicalsg2im_amd.sg2im.layout import masks_to_layout
    meta, a = load_golden("row_gaps_r3")
    for M in meta["mask_sizes"]:
        for (H, W) in meta["sizes"]:
            t = "%d_%dx%d" % (M, H, W)
            runs = []
            for _ in range(2):
                vecs, boxes, soft = [a[k].cuda().requires_grad_(True) for k in ("vecs", "boxes", "soft_%d" % M)]
                out = masks_to_layout(vecs, boxes, soft, H, W)
                assert_close(out, a["out_" + t], RTOL, 2e-6, "masks layout " + t)
                runs.append(torch.autograd.grad((out * a["w_" + t].cuda()).sum(), [vecs, boxes, soft]))
            gv, gb, gm = runs[0]
            assert_close(gv, a["gvecs_" + t], RTOL, 1e-5, "dvecs " + t)
            assert_close(gb, a["gboxes_" + t], RTOL, 1e-5 * float(a["gboxes_" + t].abs().max()), "dboxes " + t)
            assert_close(gm, a["gmasks_" + t], RTOL, 1e-5, "dmasks " + t)
            assert torch.equal(gm, runs[1][2])


def test_mask_gradients_through_pyramid_and_disc_input_vs_oracle():
    """The batched entry points the generator / discriminator use (several sizes at once, padded object rows,
    the [layout | image] buffer) against the oracle's per-sample masks_to_layout + nearest resize."""
    import oracle
    import torch.nn.functional as F
    from canonicalsg2im_amd import ops
    g = torch.Generator().manual_seed(5)
    B, O, S, M, H = 2, 4, 8, 16, 32
    vecs = torch.randn(B, O, S, generator=g)
    boxes = torch.rand(B, O, 4, generator=g) * 0.5 + 0.05
    masks = torch.rand(B, O, M, M, generator=g)
    valid = torch.tensor([[1, 1, 1, 0], [1, 0, 1, 1]], dtype=torch.uint8)
    img = torch.randn(B, 3, H, H, generator=g)
    sizes = (8, 16, 32)
    ws = [torch.randn(B, S, s, s, generator=g) for s in sizes]
    wd = torch.randn(B, S + 3, H, H, generator=g)
    mo = masks.clone().requires_grad_(True)
    loss = 0
    for b in range(B):
        keep = valid[b].bool()
        full = oracle.masks_to_layout(vecs[b][keep], boxes[b][keep], mo[b][keep], H)
        for s, w in zip(sizes, ws):
            loss = loss + (F.interpolate(full, size=(s, s), mode="nearest") * w[b:b + 1]).sum()
        loss = loss + (full * wd[b:b + 1, :S]).sum()
    (want,) = torch.autograd.grad(loss, [mo])
    md = masks.cuda().requires_grad_(True)
    outs = ops.layout_pyramid(vecs.cuda(), boxes.cuda(), valid.cuda(), H, sizes, masks=md)
    buf = ops.disc_input(img.cuda(), vecs.cuda(), boxes.cuda(), valid.cuda(), H, masks=md)
    wdc = wd.cuda()
    lg = sum((o * w.cuda()).sum() for o, w in zip(outs, ws)) + (buf[:, :S] * wdc[:, :S]).sum()
    (got,) = torch.autograd.grad(lg, [md])
    assert_close(got, want, RTOL, 1e-5, "dmasks through pyramid + disc input")
    assert float(got[0, 3].abs().max()) == 0.0 and float(got[1, 1].abs().max()) == 0.0      # padded rows


def test_object_crop_box_gradients_vs_reference():
    """crop_bbox's sampling grid is differentiable in the boxes (bilinear.py:83-94): d boxes against the reference's
    autograd (a box running off the image included), ordered sums (bit-identical from run to run)."""
    from canonicalsg2im_amd import ops
    from canonicalsg2im_amd.synth import make_vocab
    meta, a = load_golden("row_gaps_r3")
    vocab = make_vocab(meta["vocab"])
    objs = a["c_objs"].cuda()
    valid = ops.real_object_mask(objs, vocab["object_name_to_idx"]["__image__"])
    nz = valid.nonzero()
    runs = []
    for _ in range(2):
        imgs, boxes = a["c_imgs"].cuda().requires_grad_(True), a["c_boxes"].cuda().requires_grad_(True)
        crops = ops.crop_objects(imgs, boxes[nz[:, 0], nz[:, 1]], nz[:, 0].contiguous(), meta["crop_size"])
        assert_close(crops[:, :3], a["c_crops"], RTOL, 2e-6, "crops")
        runs.append(torch.autograd.grad((crops[:, :3] * a["c_w"].cuda()).sum(), [imgs, boxes]))
    gi, gb = runs[0]
    assert_close(gi, a["c_gimgs"], RTOL, 1e-5, "d imgs")
    assert_close(gb, a["c_gboxes"], RTOL, 1e-5 * float(a["c_gboxes"].abs().max()), "d crop boxes")
    assert torch.equal(gb, runs[1][1])
