"""Kernel-level parity on a real MI355X: every C-ABI operator (forward AND backward) against a
plain fp32 PyTorch CPU evaluation of the same op / the CPU oracle.  Tolerance: rtol 1e-4
(BASELINE.json north_star); index outputs bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from canonicalsg2im_amd import ops as o
    return o


def dev(t, grad=False):
    t = t.detach().clone().cuda()
    return t.requires_grad_(True) if grad else t


# ------------------------------------------------------------------------------------ conv / linear
CONV_CASES = [
    # B, Cin, Cout, H,  W,  K, s, p, act,            bias, residual
    (2, 8, 12, 9, 11, 3, 1, 1, "none", True, False),
    (2, 16, 8, 8, 8, 3, 1, 1, "relu", True, False),
    (1, 8, 8, 7, 7, 1, 1, 0, "none", False, False),
    (2, 12, 16, 17, 17, 4, 2, 2, "lrelu", True, False),
    (2, 16, 8, 9, 9, 4, 1, 2, "none", False, False),
    (2, 8, 16, 10, 10, 3, 1, 1, "none", True, True),
    (2, 16, 4, 12, 12, 3, 1, 1, "tanh", True, False),
    (2, 64, 160, 40, 40, 3, 1, 1, "none", True, False),      # multi-tile M and N, split-K wgrad
    (3, 36, 64, 33, 33, 4, 2, 2, "lrelu", True, False),      # D's first conv shape (Cin = 32+3+1)
    (1, 128, 72, 24, 24, 3, 1, 1, "relu", True, False),
    (2, 32, 8, 6, 6, 4, 2, 2, "none", True, False),          # stride-2 backward-data parity classes
    (2, 3, 5, 8, 8, 3, 1, 1, "none", True, False),           # channels not multiples of 4 -> padded by the wrapper
    (4, 64, 32, 130, 130, 3, 1, 1, "lrelu", True, False),    # 529 tiles for 512 block slots: only the last round is split
    (1, 64, 256, 184, 184, 3, 1, 1, "none", True, True),     # 265 x 2 tiles: tail split with two channel tiles, residual
]


def _act(y, act):
    if act == "relu":
        return F.relu(y)
    if act == "lrelu":
        return F.leaky_relu(y, 0.2)
    if act == "tanh":
        return torch.tanh(y)
    return y


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_forward_backward(ops, case):
    B, Cin, Cout, H, W, K, s, p, act, has_bias, has_res = case
    g = torch.Generator().manual_seed(1000 + CONV_CASES.index(case))     # (hash() of a tuple with strings changes per process)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g) if has_bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if has_bias else None
    pre_ref = F.conv2d(xr, wr, br, stride=s, padding=p)
    y_ref = _act(pre_ref, act)
    res = torch.randn(y_ref.shape, generator=g) if has_res else None
    rr = res.clone().requires_grad_(True) if has_res else None
    if has_res:
        y_ref = y_ref + rr
    gy = torch.randn(y_ref.shape, generator=g)
    if act in ("relu", "lrelu"):          # a pre-activation within rounding of the kink may land on either side of it
        gy = gy * (pre_ref.detach().abs() > 1e-5).float()
    y_ref.backward(gy)

    xd, wd = dev(x, True), dev(w, True)
    bd = dev(b, True) if has_bias else None
    rd = dev(res, True) if has_res else None
    code = {"none": (ops.ACT_NONE, 0.0), "relu": (ops.ACT_LEAKY, 0.0), "lrelu": (ops.ACT_LEAKY, 0.2),
            "tanh": (ops.ACT_TANH, 0.0)}[act]
    y = ops.conv2d(xd, wd, bd, s, p, code[0], code[1], rd)
    # absolute term scaled to the tensor (long reductions; Winograd F(4x4,3x3) on maps >= 32 wide has ~3x the rounding
    # error of the direct sum, tests/test_gpu_wino4.py): 1e-5 of the largest entry, as for dw below
    assert_close(y, y_ref, RTOL, 1e-5 * float(y_ref.abs().max()) + 1e-5, "conv y %s" % (case,))
    y.backward(gy.cuda())
    assert_close(xd.grad, xr.grad, RTOL, 1e-5 * float(xr.grad.abs().max()) + 1e-5, "conv dx %s" % (case,))
    # dw sums B*OH*OW products: absolute rounding noise scales with the largest entries
    assert_close(wd.grad, wr.grad, RTOL, 1e-5 * float(wr.grad.abs().max()) + 1e-5, "conv dw %s" % (case,))
    if has_bias:
        assert_close(bd.grad, br.grad, RTOL, 1e-5 * float(br.grad.abs().max()) + 1e-5, "conv db %s" % (case,))
    if has_res:
        assert_close(rd.grad, rr.grad, RTOL, 1e-5, "conv dres %s" % (case,))


def test_conv2d_rejects_activation_with_residual(ops):
    x, w = dev(torch.randn(1, 8, 6, 6)), dev(torch.randn(8, 8, 3, 3))
    with pytest.raises(NotImplementedError, match="residual"):
        ops.conv2d(x, w, None, 1, 1, ops.ACT_LEAKY, 0.2, residual=dev(torch.randn(1, 8, 6, 6)))


@pytest.mark.parametrize("M,K,N,relu", [(37, 24, 64, True), (300, 128, 512, True), (1000, 512, 1152, True),
                                        (64, 128, 4, False), (5, 8, 12, False)])
def test_linear(ops, M, K, N, relu):
    g = torch.Generator().manual_seed(M * 7 + N)
    x, w, b = torch.randn(3, M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    y_ref = F.linear(xr, wr, br)
    y_ref = F.relu(y_ref) if relu else y_ref
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd, wd, bd = dev(x, True), dev(w, True), dev(b, True)
    y = ops.linear(xd, wd, bd, ops.ACT_LEAKY if relu else ops.ACT_NONE, 0.0)
    assert_close(y, y_ref, RTOL, 2e-5, "linear y")
    y.backward(gy.cuda())
    assert_close(xd.grad, xr.grad, RTOL, 2e-5, "linear dx")
    assert_close(wd.grad, wr.grad, RTOL, 1e-5 * float(wr.grad.abs().max()) + 1e-5, "linear dw")
    assert_close(bd.grad, br.grad, RTOL, 1e-5 * float(br.grad.abs().max()) + 1e-5, "linear db")


# ------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("B,C,H,W,instance,mod,slope", [
    (3, 8, 5, 7, False, True, 0.2), (2, 16, 9, 9, False, True, 1.0), (2, 64, 12, 12, False, False, 1.0),
    (3, 12, 6, 5, True, False, 0.2), (2, 1024, 4, 4, False, True, 0.2), (2, 8, 33, 33, True, False, 0.2),
    (4, 128, 40, 40, False, True, 0.2)])
def test_norm_act_train(ops, B, C, H, W, instance, mod, slope):
    g = torch.Generator().manual_seed(C * 3 + H)
    x = torch.randn(B, C, H, W, generator=g) * 1.7 + 0.3
    gb = torch.randn(B, 2 * C, H, W, generator=g) * 0.5 if mod else None
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.clone().requires_grad_(True)
    gbr = gb.clone().requires_grad_(True) if mod else None
    if instance:
        n = F.instance_norm(xr, eps=1e-5)
    else:
        n = F.batch_norm(xr, rm, rv, None, None, True, 0.1, 1e-5)
    y_ref = n * (1 + gbr[:, :C]) + gbr[:, C:] if mod else n
    if slope != 1.0:
        y_ref = F.leaky_relu(y_ref, slope)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd = dev(x, True)
    gbd = dev(gb, True) if mod else None
    rmd, rvd = torch.zeros(C).cuda(), torch.ones(C).cuda()
    y = ops.norm_act(xd, gbd, None if instance else rmd, None if instance else rvd, instance=instance, training=True,
                     slope=slope)
    assert_close(y, y_ref, RTOL, 2e-5, "norm y")
    y.backward(gy.cuda())
    assert_close(xd.grad, xr.grad, 2e-4, 3e-5, "norm dx")
    if mod:
        assert_close(gbd.grad, gbr.grad, RTOL, 2e-5, "norm dgb")
    if not instance:
        assert_close(rmd, rm, RTOL, 1e-6, "running_mean")
        assert_close(rvd, rv, RTOL, 1e-6, "running_var")


def test_norm_act_eval(ops):
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 8, 6, 6
    x = torch.randn(B, C, H, W, generator=g)
    gb = torch.randn(B, 2 * C, H, W, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    xr, gbr = x.clone().requires_grad_(True), gb.clone().requires_grad_(True)
    n = F.batch_norm(xr, rm.clone(), rv.clone(), None, None, False, 0.1, 1e-5)
    y_ref = F.leaky_relu(n * (1 + gbr[:, :C]) + gbr[:, C:], 0.2)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd, gbd = dev(x, True), dev(gb, True)
    rmd, rvd = rm.cuda(), rv.cuda()
    y = ops.norm_act(xd, gbd, rmd, rvd, instance=False, training=False, slope=0.2)
    assert_close(y, y_ref, RTOL, 2e-5, "eval y")
    y.backward(gy.cuda())
    assert_close(xd.grad, xr.grad, RTOL, 2e-5, "eval dx")
    assert_close(gbd.grad, gbr.grad, RTOL, 2e-5, "eval dgb")
    assert_close(rmd, rm, 0, 0, "running_mean untouched")


# ------------------------------------------------------------------------------------ resampling
def test_upsample_and_avgpool(ops):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 5, 7, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.interpolate(xr, scale_factor=2, mode="nearest")
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd = dev(x, True)
    y = ops.upsample2x(xd)
    assert_close(y, y_ref, 0, 0, "upsample (bit exact)")
    y.backward(gy.cuda())
    assert_close(xd.grad, xr.grad, RTOL, 1e-6, "upsample dx")
    for H, W in ((9, 9), (8, 10), (33, 33)):
        x = torch.randn(2, 36, H, W, generator=g)
        xr = x.clone().requires_grad_(True)
        y_ref = F.avg_pool2d(xr, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
        gy = torch.randn(y_ref.shape, generator=g)
        y_ref.backward(gy)
        xd = dev(x, True)
        y = ops.avgpool3s2(xd)
        assert_close(y, y_ref, RTOL, 1e-6, "avgpool %dx%d" % (H, W))
        y.backward(gy.cuda())
        assert_close(xd.grad, xr.grad, RTOL, 1e-6, "avgpool dx %dx%d" % (H, W))


def test_maxpool2_and_l1_mean(ops):
    """MaxPool2d(2,2) incl. odd sizes and tied maxima (gradient goes to the first maximum, as ATen);
    mean |a-b| and its gradient (sign(0) = 0)."""
    g = torch.Generator().manual_seed(10)
    for H, W in ((6, 8), (7, 9), (36, 44)):
        x = torch.round(torch.randn(2, 12, H, W, generator=g) * 2).clamp_min(0) / 2      # many ties, many zeros
        xr = x.clone().requires_grad_(True)
        y_ref = F.max_pool2d(xr, kernel_size=2, stride=2)
        gy = torch.randn(y_ref.shape, generator=g)
        y_ref.backward(gy)
        xd = dev(x, True)
        y = ops.maxpool2(xd)
        assert_close(y, y_ref, 0, 0, "maxpool %dx%d (bit exact)" % (H, W))
        y.backward(gy.cuda())
        assert_close(xd.grad, xr.grad, 0, 0, "maxpool dx %dx%d (bit exact)" % (H, W))
    for shape in ((2, 8, 5, 7), (3, 64, 33, 31), (1, 4, 1, 1)):
        a = torch.randn(shape, generator=g)
        b = torch.randn(shape, generator=g)
        b[0, :2] = a[0, :2]                                                              # exact zeros of a-b
        ar = a.clone().requires_grad_(True)
        l_ref = F.l1_loss(ar, b) * 3.0
        l_ref.backward()
        ad = dev(a, True)
        l = ops.l1_mean(ops.nhwc(ad), ops.nhwc(b.cuda())) * 3.0
        assert_close(l, l_ref, 1e-5, 1e-7, "l1 mean %s" % (shape,))
        l.backward()
        assert_close(ad.grad, ar.grad, 1e-6, 1e-9, "l1 grad %s" % (shape,))


# ------------------------------------------------------------------------------------ layout
def test_layout_golden(ops):
    """boxes_to_layout against the reference's own output (grid_sample + scatter_add)."""
    from canonicalsg2im_amd._lib import check, lib, ptr, stream
    meta, a = load_golden("layout")
    vecs, boxes = a["vecs"], a["boxes"]
    O, S = vecs.shape
    vd, bd = vecs.cuda(), boxes.cuda()                   # keep the device tensors alive across the launches
    valid = torch.ones(1, O, dtype=torch.uint8).cuda()
    for H, W in meta["sizes"]:
        tag = "%dx%d" % (H, W)
        out = torch.empty(1, H, W, S).cuda()
        check(lib.csg_layout_fwd(ptr(vd), ptr(bd), ptr(valid), None, 0, 1, O, S, H, W, H, W, ptr(out), S, 0, stream()))
        assert_close(out.permute(0, 3, 1, 2), a["out_" + tag], RTOL, 2e-6, "layout " + tag)
        dv = torch.empty(1, O, S).cuda()
        gw = a["w_" + tag].permute(0, 2, 3, 1).contiguous().cuda()
        check(lib.csg_layout_bwd(ptr(gw), S, 0, ptr(bd), ptr(valid), None, 0, 1, O, S, H, W, H, W, ptr(dv), 0, None, None,
                                 None, 0, stream()))                       # one block per object
        assert_close(dv[0], a["gvecs_" + tag], RTOL, 1e-5, "layout dvecs " + tag)
        nws = lib.csg_layout_bwd_workspace(1, O, S, H, W, 0, 0)            # the two-pass tiled form where it applies
        if nws > 0:
            ws = torch.full((nws // 4,), float("nan")).cuda()
            dv2 = torch.full((1, O, S), float("nan")).cuda()
            check(lib.csg_layout_bwd(ptr(gw), S, 0, ptr(bd), ptr(valid), None, 0, 1, O, S, H, W, H, W, ptr(dv2), 0, None,
                                     None, ptr(ws), nws, stream()))
            assert_close(dv2[0], a["gvecs_" + tag], RTOL, 1e-5, "layout dvecs (tiled) " + tag)
            dv3 = dv2.clone()
            check(lib.csg_layout_bwd(ptr(gw), S, 0, ptr(bd), ptr(valid), None, 0, 1, O, S, H, W, H, W, ptr(dv3), 1, None,
                                     None, ptr(ws), nws, stream()))           # accumulate
            assert torch.equal(dv3, dv2 + dv2)


def test_masks_to_layout_golden(ops):
    """masks_to_layout (train mode) through the drop-in function against the reference's output."""
    from canonicalsg2im_amd.sg2im.layout import masks_to_layout
    meta, a = load_golden("masks_layout")
    for H in meta["sizes"]:
        for tag, key in (("int", "masks"), ("soft", "soft")):
            vd = dev(a["vecs"], True)
            out = masks_to_layout(vd, a["boxes"].cuda(), a[key].cuda(), H, H)
            assert_close(out, a["out_%s_%d" % (tag, H)], RTOL, 2e-6, "masks layout %s %d" % (tag, H))
            (out * a["w_%s_%d" % (tag, H)].cuda()).sum().backward()
            assert_close(vd.grad, a["gvecs_%s_%d" % (tag, H)], RTOL, 1e-5, "masks layout dvecs %s %d" % (tag, H))


def test_masked_pyramid_and_disc_input_vs_oracle(ops):
    import oracle
    g = torch.Generator().manual_seed(31)
    B, O, S, H, M = 2, 11, 8, 64, 16
    vecs, img = torch.randn(B, O, S, generator=g), torch.randn(B, 3, H, H, generator=g)
    wh = torch.rand(B, O, 2, generator=g) * 0.4 + 0.05
    boxes = torch.cat([torch.rand(B, O, 2, generator=g) * (1 - wh), wh], -1)
    masks = (torch.rand(B, O, M, M, generator=g) > 0.5).long()
    valid = torch.ones(B, O, dtype=torch.bool)
    valid[1, -2:] = False
    vr = vecs.clone().requires_grad_(True)
    full = torch.cat([oracle.masks_to_layout(vr[b][valid[b]], boxes[b][valid[b]], masks[b][valid[b]], H, H)
                      for b in range(B)], 0)
    sizes = (8, 32, 64)
    refs = [F.interpolate(full, size=(h, h), mode="nearest") for h in sizes]
    gys = [torch.randn(r.shape, generator=g) for r in refs]
    sum((r * gy).sum() for r, gy in zip(refs, gys)).backward()
    vd = dev(vecs, True)
    outs = ops.layout_pyramid(vd, boxes.cuda(), valid.to(torch.uint8).cuda(), H, sizes, masks=masks.cuda())
    for h, o, r in zip(sizes, outs, refs):
        assert_close(o, r, RTOL, 5e-6, "masked pyramid level %d" % h)
    sum((o * gy.cuda()).sum() for o, gy in zip(outs, gys)).backward()
    assert_close(vd.grad, vr.grad, RTOL, 2e-5, "masked pyramid dvecs")
    buf = ops.disc_input(img.cuda(), vecs.cuda(), boxes.cuda(), valid.to(torch.uint8).cuda(), H, masks=masks.cuda())
    assert_close(buf[:, :S], full, RTOL, 5e-6, "masked disc input")


def test_layout_pyramid_vs_oracle(ops):
    import oracle
    g = torch.Generator().manual_seed(21)
    B, O, S, H = 3, 37, 32, 64
    vecs = torch.randn(B, O, S, generator=g)
    wh = torch.rand(B, O, 2, generator=g) * 0.4 + 0.05
    xy = torch.rand(B, O, 2, generator=g) * (1 - wh)
    boxes = torch.cat([xy, wh], -1)
    valid = (torch.rand(B, O, generator=g) > 0.3)
    sizes = (2, 4, 8, 16, 32, 64)
    vr = vecs.clone().requires_grad_(True)
    refs = []
    for h in sizes:
        full = torch.cat([oracle.boxes_to_layout(vr[b][valid[b]], boxes[b][valid[b]], H, H) for b in range(B)], 0)
        refs.append(F.interpolate(full, size=(h, h), mode="nearest"))
    gys = [torch.randn(r.shape, generator=g) for r in refs]
    sum((r * gy).sum() for r, gy in zip(refs, gys)).backward()
    vd = dev(vecs, True)
    outs = ops.layout_pyramid(vd, boxes.cuda(), valid.to(torch.uint8).cuda(), H, sizes)
    for h, o, r in zip(sizes, outs, refs):
        assert_close(o, r, RTOL, 5e-6, "pyramid level %d" % h)
    sum((o * gy.cuda()).sum() for o, gy in zip(outs, gys)).backward()
    assert_close(vd.grad, vr.grad, RTOL, 2e-5, "pyramid dvecs")


@pytest.mark.parametrize("S,H,channels_last", [(8, 32, False), (32, 64, True), (12, 16, False)])
def test_disc_input(ops, S, H, channels_last):
    """[layout | img | zero pad] written by ONE kernel (csg_disc_input_fwd): layout channels as csg_layout_fwd, the image
    copied bit for bit from a contiguous or a channels-last tensor, pad channels zero (S = 12: no pad quad beyond the image's)."""
    import oracle
    g = torch.Generator().manual_seed(22)
    B, O = 2, 9
    vecs, img = torch.randn(B, O, S, generator=g), torch.randn(B, 3, H, H, generator=g)
    wh = torch.rand(B, O, 2, generator=g) * 0.4 + 0.05
    boxes = torch.cat([torch.rand(B, O, 2, generator=g) * (1 - wh), wh], -1)
    valid = torch.ones(B, O, dtype=torch.bool)
    valid[0, -3:] = False
    vr, ir = vecs.clone().requires_grad_(True), img.clone().requires_grad_(True)
    seg = torch.cat([oracle.boxes_to_layout(vr[b][valid[b]], boxes[b][valid[b]], H, H) for b in range(B)], 0)
    Ct = (S + 3 + 3) // 4 * 4
    ref = torch.cat([seg, ir, torch.zeros(B, Ct - S - 3, H, H)], 1)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    vd = dev(vecs, True)
    idv = img.cuda()
    if channels_last:
        idv = idv.contiguous(memory_format=torch.channels_last)
    idv.requires_grad_(True)
    buf = ops.disc_input(idv, vd, boxes.cuda(), valid.to(torch.uint8).cuda(), H)
    assert buf.shape[1] == Ct
    assert_close(buf, ref, RTOL, 5e-6, "disc input buffer")
    assert torch.equal(buf[:, S:S + 3].cpu(), img) and float(buf[:, S + 3:].abs().sum()) == 0.0
    buf.backward(gy.cuda())
    assert_close(vd.grad, vr.grad, RTOL, 2e-5, "disc input dvecs")
    assert_close(idv.grad, ir.grad, 0, 0, "disc input dimg")


# ------------------------------------------------------------------------------------ graph
def test_real_object_mask_bit_exact(ops):
    g = torch.Generator().manual_seed(1)
    objs = torch.randint(0, 4, (5, 13, 3), generator=g)
    m = ops.real_object_mask(objs.cuda(), 0).cpu()
    want = ((objs[..., 0] != 0) & (objs[..., 0] != 0)).to(torch.uint8)
    assert torch.equal(m, want)
    m2 = ops.real_object_mask(objs.cuda(), 2).cpu()
    assert torch.equal(m2, ((objs[..., 0] != 0) & (objs[..., 0] != 2)).to(torch.uint8))


def _csr_reference(triplets, O):
    B, T, _ = triplets.shape
    rp = np.zeros((B, O + 1), np.int32)
    col = np.zeros((B, max(2 * T, 1)), np.int32)
    for b in range(B):
        rows = [[] for _ in range(O)]
        for t in range(T):
            rows[int(triplets[b, t, 0])].append(2 * t)
        for t in range(T):
            rows[int(triplets[b, t, 2])].append(2 * t + 1)
        pos = 0
        for i in range(O):
            rp[b, i] = pos
            col[b, pos:pos + len(rows[i])] = rows[i]
            pos += len(rows[i])
        rp[b, O] = pos
    return rp, col


@pytest.mark.parametrize("B,T,O", [(2, 9, 5), (3, 700, 40), (2, 5000, 300), (1, 1, 1), (4, 16256, 129), (2, 40000, 254),
                                   (2, 513, 1)])
def test_graph_csr_bit_exact(ops, B, T, O):
    g = torch.Generator().manual_seed(T)
    tr = torch.stack([torch.randint(0, O, (B, T), generator=g), torch.randint(0, 8, (B, T), generator=g),
                      torch.randint(0, O, (B, T), generator=g)], -1)
    rp, col = ops.graph_csr(tr.cuda(), O)
    rp_ref, col_ref = _csr_reference(tr.numpy(), O)
    assert np.array_equal(rp.cpu().numpy(), rp_ref)
    assert np.array_equal(col.cpu().numpy()[:, :2 * T], col_ref[:, :2 * T])


def test_embed(ops):
    g = torch.Generator().manual_seed(3)
    tabs = [torch.randn(n, 8, generator=g) for n in (4, 9, 3, 3)]
    idx = torch.stack([torch.randint(0, t.shape[0], (3, 11), generator=g) for t in tabs], -1)
    tr = [t.clone().requires_grad_(True) for t in tabs]
    ref = torch.cat([F.embedding(idx[..., k], tr[k]) for k in range(4)], -1)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    td = [dev(t, True) for t in tabs]
    out = ops.embed(idx.cuda(), td)
    assert_close(out, ref, 0, 0, "embedding (bit exact)")
    out.backward(gy.cuda())
    for k in range(4):
        assert_close(td[k].grad, tr[k].grad, RTOL, 1e-5, "dtable %d" % k)
    # dense-graph regime: tens of thousands of rows onto a tiny table (per-block LDS accumulation)
    tab = torch.randn(8, 32, generator=g)
    idx = torch.randint(0, 8, (3, 9000, 1), generator=g)
    t_ref = tab.clone().requires_grad_(True)
    ref = F.embedding(idx[..., 0], t_ref)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    t_dev = dev(tab, True)
    out = ops.embed(idx.cuda(), [t_dev])
    assert_close(out, ref, 0, 0, "embedding, many rows")
    out.backward(gy.cuda())
    assert_close(t_dev.grad, t_ref.grad, RTOL, 2e-4 * float(t_ref.grad.abs().max()), "dtable, many rows")
    # rows are added in row order (chunk partials in chunk order): the same bits on every run, and the bits of a plain
    # sequential fp32 sum per 1 024-row chunk
    first = t_dev.grad.clone()
    for _ in range(3):
        t_dev.grad = None
        ops.embed(idx.cuda(), [t_dev]).backward(gy.cuda())
        assert torch.equal(t_dev.grad, first), "embedding backward is not bit-reproducible"
    flat_idx, flat_gy = idx.reshape(-1), gy.reshape(-1, 32)
    want = torch.zeros(8, 32)
    for c0 in range(0, flat_idx.numel(), 256):
        part = torch.zeros(8, 32)
        ii, gg = flat_idx[c0:c0 + 256], flat_gy[c0:c0 + 256]
        for i in range(8):
            rows = gg[ii == i]
            acc = torch.zeros(32)
            for r in rows:                                   # sequential fp32 adds, row order
                acc = acc + r
            part[i] = acc
        want = want + part
    assert torch.equal(first.cpu(), want), "embedding backward: not the ordered sum"


def test_gather_concat_and_segment_avg_vs_golden(ops):
    """The (s,p,o) gather and the confidence-weighted average of GraphTripleConv against the
    reference's own layer output: net1/net2 evaluated with plain CPU torch around the two kernels."""
    meta, a = load_golden("gconv")
    H, Dp = meta["hidden"], meta["dp_out"]
    obj, pred = dev(a["obj"], True), dev(a["pred"], True)
    edges, p, tt = a["edges"].cuda(), a["p"].cuda(), a["tt"].cuda()
    tr = torch.stack([edges[..., 0], p, edges[..., 1]], -1).contiguous()
    rp, col = ops.graph_csr(tr, obj.shape[1])
    cat = ops.gather_concat(obj, pred, tr, rp, col)
    sd = {k[3:]: dev(v, True) for k, v in a.items() if k.startswith("sd:")}
    w_trans = sd["predicates_transitive_weights"]
    h = F.relu(F.linear(F.relu(F.linear(cat, sd["net1.0.weight"], sd["net1.0.bias"])), sd["net1.2.weight"],
                        sd["net1.2.bias"]))
    conf = (tt == 0).float() + (tt == 1).float() * torch.sigmoid(w_trans)[p]
    pooled, new_p = ops.segment_avg(h, conf, (p != 0).to(torch.uint8), tr, rp, col, H, Dp)
    new_obj = F.relu(F.linear(F.relu(F.linear(pooled, sd["net2.0.weight"], sd["net2.0.bias"])), sd["net2.2.weight"],
                              sd["net2.2.bias"]))
    assert_close(new_obj, a["new_obj"], RTOL, 2e-6, "new_obj")
    assert_close(new_p, a["new_p"], RTOL, 2e-6, "new_p")
    ((new_obj * a["wo"].cuda()).sum() + (new_p * a["wp"].cuda()).sum()).backward()
    assert_close(obj.grad, a["gobj"], RTOL, 1e-5, "dobj")
    assert_close(pred.grad, a["gpred"], RTOL, 1e-5, "dpred")
    assert_close(w_trans.grad, a["grad:predicates_transitive_weights"], RTOL, 1e-5, "dw_trans")
    for k in ("net1.0.weight", "net1.2.bias", "net2.0.weight", "net2.2.weight"):
        assert_close(sd[k].grad, a["grad:" + k], RTOL, 1e-5, "d" + k)


@pytest.mark.parametrize("O,H,Dp,Din", [(70, 512, 128, 128), (130, 64, 32, 32), (9, 2048, 16, 12)])
def test_dense_graph_rows_vs_oracle(ops, O, H, Dp, Din):
    """CLEVR-style closure graphs (every ordered pair: rows of ~2(O-1) edges, split over several workgroups),
    padded triplets, transitive confidences: gather + weighted segment average, forward and backward,
    against the oracle's GraphTripleConv arithmetic (sg2im/graph.py:63-109)."""
    g = torch.Generator().manual_seed(O)
    B = 2
    pairs = [(s, o) for s in range(O) for o in range(O) if s != o]
    T = len(pairs) + 7                                                    # 7 padded triplets at the end
    tr = torch.zeros(B, T, 3, dtype=torch.int64)
    for b in range(B):
        keep = len(pairs) if b == 0 else len(pairs) // 2                 # the second sample is half padding
        tr[b, :keep, 0] = torch.tensor([p[0] for p in pairs[:keep]])
        tr[b, :keep, 2] = torch.tensor([p[1] for p in pairs[:keep]])
        tr[b, :keep, 1] = torch.randint(2, 8, (keep,), generator=g)
    p = tr[..., 1]
    tt = (torch.rand(B, T, generator=g) < 0.7).long() * (p != 0).long()
    obj = torch.randn(B, O, Din, generator=g)
    pred = torch.randn(B, T, Dp if Dp % 4 == 0 else Dp, generator=g)
    h = torch.randn(B, T, 2 * H + Dp, generator=g)
    w_trans = torch.randn(8, generator=g)
    conf = (tt == 0).float() + (tt == 1).float() * torch.sigmoid(w_trans)[p]
    valid = p != 0
    wo, wc = torch.randn(B, O, H, generator=g), torch.randn(B, T, 2 * Din + Dp, generator=g)
    # ---- CPU restatement (graph.py:63-66, 88-106)
    objr, hr, confr = obj.clone().requires_grad_(True), h.clone().requires_grad_(True), conf.clone().requires_grad_(True)
    cat_ref = torch.cat([torch.gather(objr, 1, tr[..., 0:1].expand(-1, -1, Din)), pred,
                         torch.gather(objr, 1, tr[..., 2:3].expand(-1, -1, Din))], dim=-1)
    pooled_ref = torch.zeros(B, O, H)
    cnt_ref = torch.zeros(B, O)
    for b in range(B):
        m = valid[b]
        s_i, o_i = tr[b, m, 0], tr[b, m, 2]
        hs, ho = hr[b, m, :H] * confr[b, m, None], hr[b, m, H + Dp:] * confr[b, m, None]
        pooled_b = torch.zeros(O, H).index_add(0, s_i, hs).index_add(0, o_i, ho)
        cnt_b = torch.zeros(O).index_add(0, s_i, confr[b, m]).index_add(0, o_i, confr[b, m])
        nz = cnt_b > 0
        pooled_b = torch.where(nz[:, None], pooled_b / cnt_b.clamp_min(1e-30)[:, None], pooled_b)
        pooled_ref[b], cnt_ref[b] = pooled_b, cnt_b.detach()
    ((pooled_ref * wo).sum() + (cat_ref * wc).sum()).backward()
    # ---- kernels
    trd = tr.cuda()
    rp, col = ops.graph_csr(trd, O)
    objd, hd, confd = dev(obj, True), dev(h, True), dev(conf, True)
    cat = ops.gather_concat(objd, pred.cuda(), trd, rp, col)
    pooled, new_p = ops.segment_avg(hd, confd, valid.to(torch.uint8).cuda(), trd, rp, col, H, Dp)
    assert_close(cat, cat_ref, 0, 0, "gather (bit exact)")
    assert_close(pooled, pooled_ref, RTOL, 2e-5, "pooled")
    assert_close(new_p, h[..., H:H + Dp] * conf[..., None], RTOL, 1e-6, "new_p")
    ((pooled * wo.cuda()).sum() + (cat * wc.cuda()).sum()).backward()
    assert_close(objd.grad, objr.grad, RTOL, 2e-4 * float(objr.grad.abs().max()), "dobj (row-split sums)")
    assert_close(hd.grad, hr.grad, RTOL, 1e-5, "dh")
    assert_close(confd.grad, confr.grad, 1e-3, 2e-4 * float(confr.grad.abs().max()), "dconf")


@pytest.mark.parametrize("shape", [(64, 32, 3, 3), (128, 64, 4, 4), (1024, 1024, 3, 3), (20, 12, 1, 1)])
def test_spectral_norm_vs_torch(ops, shape):
    """csrc/spectral.hip against torch.nn.utils.spectral_norm (the function the reference calls,
    architecture.py:35-39): three training-mode calls (u/v evolve in place), one eval call, and the
    gradient w.r.t. weight_orig for a cotangent handed over in the weight-gradient kernel's layout."""
    import torch.nn as nn
    from canonicalsg2im_amd.spectral_norm import spectral_norm
    torch.manual_seed(sum(shape))
    Cout, Cin, KH, KW = shape
    ref = nn.Conv2d(Cin, Cout, (KH, KW), bias=False)
    mine = nn.Conv2d(Cin, Cout, (KH, KW), bias=False)
    mine.load_state_dict(ref.state_dict())
    ref = torch.nn.utils.spectral_norm(ref)
    mine = spectral_norm(mine)
    mine.load_state_dict(ref.state_dict())
    assert list(mine.state_dict()) == list(ref.state_dict()) == ["weight_orig", "weight_u", "weight_v"]
    mine = mine.cuda()
    hook_r = next(iter(ref._forward_pre_hooks.values()))
    hook_m = next(iter(mine._forward_pre_hooks.values()))
    for it in range(3):
        hook_r(ref, None)
        hook_m(mine, None)
        assert_close(mine.weight, ref.weight, RTOL, 1e-7, "W_eff call %d" % it)
        assert_close(mine.weight_u, ref.weight_u, RTOL, 1e-6, "u call %d" % it)
        assert_close(mine.weight_v, ref.weight_v, RTOL, 1e-6, "v call %d" % it)
    g = torch.randn(Cout, KH, KW, Cin)                       # [Cout][KH][KW][Cin] memory, as csg_conv_bwd_weight writes
    ref.weight.backward(g.permute(0, 3, 1, 2))
    mine.weight.backward(g.cuda().permute(0, 3, 1, 2))
    scale = float(ref.weight_orig.grad.abs().max())
    assert_close(mine.weight_orig.grad, ref.weight_orig.grad, RTOL, 2e-5 * scale, "dW_orig")
    mine.weight_orig.grad = None
    ref.weight_orig.grad = None
    g2 = torch.randn(Cout, Cin, KH, KW)                       # contiguous cotangent
    hook_r(ref, None); hook_m(mine, None)
    ref.weight.backward(g2); mine.weight.backward(g2.cuda())
    assert_close(mine.weight_orig.grad, ref.weight_orig.grad, RTOL, 2e-5 * float(ref.weight_orig.grad.abs().max()),
                 "dW_orig (contiguous)")
    ref.eval(); mine.eval()
    u_before = mine.weight_u.clone()
    hook_r(ref, None); hook_m(mine, None)
    assert_close(mine.weight, ref.weight, RTOL, 1e-7, "W_eff eval")
    assert torch.equal(mine.weight_u, u_before)               # no power iteration in eval mode


def test_spectral_norm_multi_is_bit_identical_to_single(ops):
    """The multi-tensor spectral normalisation (csg_spectral_norm_*_multi: every weight of a network pass in one launch
    per stage) against the per-weight calls: W / sigma, the updated u / v buffers and d weight_orig bit for bit — 14
    weights (more than one chunk of 12) of mixed shapes, two training-mode calls and an eval call."""
    shapes = [(64, 32, 3, 3), (128, 64, 4, 4), (256, 128, 4, 4), (20, 12, 1, 1), (512, 256, 3, 3), (1024, 1024, 3, 3),
              (64, 64, 3, 3), (32, 16, 1, 1), (128, 128, 3, 3), (512, 256, 4, 4), (96, 48, 3, 3), (256, 256, 1, 1),
              (40, 8, 3, 3), (128, 32, 3, 3)]
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(sh, generator=g).cuda() for sh in shapes]

    def fresh():
        gg = torch.Generator().manual_seed(6)
        us = [torch.nn.functional.normalize(torch.randn(sh[0], generator=gg), dim=0).cuda() for sh in shapes]
        vs = [torch.nn.functional.normalize(torch.randn(sh[1] * sh[2] * sh[3], generator=gg), dim=0).cuda() for sh in shapes]
        return us, vs
    us1, vs1 = fresh()
    us2, vs2 = fresh()
    for call, iterate in enumerate([True, True, False]):
        w1 = [w.clone().requires_grad_(True) for w in ws]
        w2 = [w.clone().requires_grad_(True) for w in ws]
        single = [ops.spectral_weight(w, u, v, iterate) for w, u, v in zip(w1, us1, vs1)]
        multi = ops.spectral_weights(list(zip(w2, us2, vs2)), iterate)
        cots = [torch.randn(sh[0], sh[2], sh[3], sh[1], generator=g).cuda().permute(0, 3, 1, 2) for sh in shapes]
        torch.autograd.backward(single, cots)
        torch.autograd.backward(multi, cots)
        for i, sh in enumerate(shapes):
            assert single[i].stride() == multi[i].stride(), sh
            assert torch.equal(single[i], multi[i]), "W_eff %s call %d" % (sh, call)
            assert torch.equal(us1[i], us2[i]) and torch.equal(vs1[i], vs2[i]), "u / v %s call %d" % (sh, call)
            assert torch.equal(w1[i].grad, w2[i].grad), "dW_orig %s call %d" % (sh, call)


def test_object_crops_golden(ops):
    """crop_bbox_batch of the reference (expand + grid_sample) vs the gather kernel, fwd + d(image)."""
    from canonicalsg2im_amd.synth import make_vocab
    meta, a = load_golden("crops")
    vocab = make_vocab(meta["vocab"])
    objs, boxes = a["objs"].cuda(), a["boxes"].cuda()
    valid = ops.real_object_mask(objs, vocab["object_name_to_idx"]["__image__"])
    nz = valid.nonzero()
    imgs = dev(a["imgs"], True)
    crops = ops.crop_objects(imgs, boxes[nz[:, 0], nz[:, 1]], nz[:, 0].contiguous(), meta["size"])
    assert crops.shape[1] == 4 and float(crops.detach()[:, 3].abs().max()) == 0.0          # padded channel is zero
    assert_close(crops[:, :3], a["crops"], RTOL, 2e-6, "crops")
    (crops[:, :3] * a["w"].cuda()).sum().backward()
    assert_close(imgs.grad, a["gimgs"], RTOL, 1e-5, "d imgs")
    # d(image) is a gather in a fixed order (crops, crop rows, crop columns): identical bits on every run
    first = imgs.grad.clone()
    for _ in range(3):
        imgs.grad = None
        crops = ops.crop_objects(imgs, boxes[nz[:, 0], nz[:, 1]], nz[:, 0].contiguous(), meta["size"])
        (crops[:, :3] * a["w"].cuda()).sum().backward()
        assert torch.equal(imgs.grad, first), "crop backward is not bit-reproducible"


def test_cpu_tensor_is_refused(ops):
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.randn(1, 4, 4, 4), torch.randn(4, 4, 3, 3), None, 1, 1)


@pytest.mark.parametrize("G,P,C,nch", [(1, 4096, 64, 128), (1, 1000, 36, 31), (3, 640, 128, 20), (1, 64, 1024, 2), (5, 37, 20, 1)])
def test_norm_stats_finalize_in_one_launch_matches_the_three_launch_path(ops, G, P, C, nch):
    """csg_norm_stats_finalize (one rank: partial sums, then reduction + finalisation in one kernel) against csg_norm_stats +
    csg_norm_finalize: the same mean / invstd / running statistics bit for bit (the columns are reduced in the same order),
    for batch (G = 1, two modules' running buffers) and instance (G > 1) statistics, channel counts that do not fill the
    16-channel blocks, one chunk."""
    from canonicalsg2im_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(G * 1000 + C)
    x = (torch.randn(G * P, C, generator=g) * 3.0 + 1.5).cuda()
    count, eps, mom = float(P), 1e-5, 0.1
    part = torch.empty(G * nch * 2 * C, device="cuda", dtype=torch.float64)
    sums = torch.empty(G * 2 * C, device="cuda", dtype=torch.float64)
    m0, i0 = torch.empty(G * C, device="cuda"), torch.empty(G * C, device="cuda")
    m1, i1 = torch.empty(G * C, device="cuda"), torch.empty(G * C, device="cuda")
    run = G == 1
    rm = [torch.randn(C, generator=g).cuda() for _ in range(2)]
    rv = [(torch.rand(C, generator=g) + 0.5).cuda() for _ in range(2)]
    rm_a, rv_a = [t.clone() for t in rm], [t.clone() for t in rv]
    rm_b, rv_b = [t.clone() for t in rm], [t.clone() for t in rv]
    check(lib.csg_norm_stats(ptr(x), G, P, C, ptr(sums), ptr(part), nch, stream()), "stats")
    for k in range(2 if run else 1):
        check(lib.csg_norm_finalize(ptr(sums), G, C, count, eps, 0, ptr(m0), ptr(i0), ptr(rm_a[k]) if run else None,
                                    ptr(rv_a[k]) if run else None, mom, stream()), "finalize")
    part.fill_(float("nan"))
    check(lib.csg_norm_stats_finalize(ptr(x), G, P, C, ptr(part), nch, count, eps, ptr(m1), ptr(i1), ptr(rm_b[0]) if run else None,
                                      ptr(rv_b[0]) if run else None, ptr(rm_b[1]) if run else None, ptr(rv_b[1]) if run else None,
                                      mom, stream()), "stats_finalize")
    assert torch.equal(m0, m1) and torch.equal(i0, i1)
    if run:
        for k in range(2):
            assert torch.equal(rm_a[k], rm_b[k]) and torch.equal(rv_a[k], rv_b[k])
            assert not torch.equal(rm_a[k], rm[k])
    xs = x.view(G, P, C).double()
    assert_close(m1.view(G, C), xs.mean(1).float(), 1e-5, 1e-6, "mean")
    assert_close(i1.view(G, C), (1.0 / torch.sqrt(xs.var(1, unbiased=False) + eps)).float(), 1e-5, 1e-6, "invstd")


def test_pool_fanout_sums_both_gradients_in_the_pooling_backward():
    """ops.pool_fanout(x) = (x, avgpool3s2(x)) whose backward forms g_full + avgpool_bwd(g_pooled) in one pass
    (csg_avgpool3s2_bwd_add): outputs and d x bit-identical to the plain pooling + autograd's own addition; one consumer only
    (either gradient missing) still works."""
    from canonicalsg2im_amd import ops
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(3, 36, 33, 47, generator=g)
    w_full, w_pool = torch.randn(3, 36, 33, 47, generator=g), torch.randn(3, 36, 17, 24, generator=g)

    def run(fused, use_full=True, use_pool=True):
        saved = ops.POOL_FANOUT
        ops.POOL_FANOUT = fused
        try:
            x = ops.nhwc(x0.clone().cuda()).requires_grad_(True)
            a, b = ops.pool_fanout(x)
            loss = 0
            if use_full:
                loss = loss + (a * a * w_full.cuda()).sum()
            if use_pool:
                loss = loss + (b * w_pool.cuda()).sum()
            loss.backward()
            return a.detach(), b.detach(), x.grad.detach()
        finally:
            ops.POOL_FANOUT = saved

    for use_full, use_pool in ((True, True), (True, False), (False, True)):
        a1, b1, g1 = run(True, use_full, use_pool)
        a0, b0, g0 = run(False, use_full, use_pool)
        assert torch.equal(a1, a0) and torch.equal(b1, b0)
        assert torch.equal(g1, g0), (use_full, use_pool, float((g1 - g0).abs().max()))
    ref = torch.nn.functional.avg_pool2d(x0, 3, 2, 1, count_include_pad=False)
    assert_close(b1.cpu(), ref, 1e-5, 1e-6, "pooled")


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_hinge_mean_over_scales_vs_torch(kind):
    """ops.hinge_mean: the PatchGAN's GAN term over both scales in one launch (reference loss.py:60-93) against the torch chain
    it replaces — value and d/d(prediction) — on predictions that are the one real channel of padded NHWC buffers."""
    from canonicalsg2im_amd import ops
    from canonicalsg2im_amd.spade.models.networks.loss import GANLoss
    g = torch.Generator().manual_seed(20 + kind)
    bufs = [torch.randn(16, 35, 35, 4, generator=g).cuda(), torch.randn(16, 19, 19, 4, generator=g).cuda() * 2.0]
    crit = GANLoss('hinge')
    real, for_d = {0: (True, False), 1: (True, True), 2: (False, True)}[kind]

    def run(fused):
        saved = ops.HINGE_FUSED
        ops.HINGE_FUSED = fused
        try:
            leaves = [b.clone().requires_grad_(True) for b in bufs]
            preds = [[l.permute(0, 3, 1, 2)[:, :1]] for l in leaves]      # (B,1,h,w) views, element stride 4
            out = crit(preds, real, for_discriminator=for_d)
            (out * 3.0).sum().backward()
            return out.detach(), [l.grad.detach() for l in leaves]
        finally:
            ops.HINGE_FUSED = saved

    o1, g1 = run(True)
    o0, g0 = run(False)
    assert o1.shape == o0.shape == (1,)
    assert abs(float(o1) - float(o0)) <= 2e-6 * abs(float(o0)) + 1e-7, (float(o1), float(o0))
    for a, b in zip(g1, g0):
        assert torch.equal(a[..., 1:], torch.zeros_like(a[..., 1:]))       # the pad channels get no gradient
        assert_close(a, b, 1e-6, 1e-9, "d prediction")
    # non-hinge modes and single tensors keep the torch path
    assert ops.hinge_mean([bufs[0].permute(0, 3, 1, 2)], 0) is None          # 4 channels: not a prediction map
