"""Winograd F(2x2,3x3) convolution (csrc/wino.hip) on a real MI355X against torch's fp32 CPU convolution:
every tile geometry (TW = 4, 8, 16, 32), ragged regions, channel counts that are not multiples of the 16-channel
stage / 64-channel block, the fused epilogue, and the backward-data pass through the autograd op.  Tolerance:
rtol 1e-4 with the absolute term scaled to the output's magnitude (the transform sums 4 inputs per operand and
the result is differenced again: ~2x the direct kernel's rounding error, measured below against fp64)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from canonicalsg2im_amd import ops as O
    return O


def _raw_wino(ops, x, w, bias=None, res=None, act=0, slope=0.0):
    """Direct call of the C ABI: pack + conv (forward operand)."""
    from canonicalsg2im_amd._lib import WinoDesc, check, lib, ptr, stream
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    xd = ops.nhwc(x.cuda())
    up = ops.wino_pack(w.cuda(), False)
    y = ops.empty_nhwc(B, Cout, H, W, xd.device)
    d = WinoDesc()
    d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, W, Cin, Cin, Cout, Cout, act, slope
    bd = bias.cuda() if bias is not None else None
    rd = ops.nhwc(res.cuda()) if res is not None else None
    check(lib.csg_wino_conv(d, ptr(xd), ptr(up), ptr(bd), ptr(rd), None, 0.0, ptr(y), None, 0, stream()), "wino_conv")
    return y


SHAPES = [
    # B, Cin, Cout, H,   W
    (2, 16, 32, 8, 8),          # TW = 4
    (1, 32, 64, 16, 16),        # TW = 8
    (2, 32, 36, 12, 20),        # Cout not x32, ragged region, TW = 8
    (1, 128, 128, 32, 32),      # TW = 16
    (1, 64, 100, 24, 40),       # TW = 16, non-square, ragged in both directions
    (2, 32, 128, 64, 64),       # TW = 32
    (1, 48, 72, 6, 130),        # TW = 32, ragged x, a single tile row
    (1, 1024, 64, 16, 16),      # long K (64 stages)
]


@pytest.mark.parametrize("shape", [(2, 2048, 128, 16, 16), (1, 1168, 64, 32, 16), (1, 272, 32, 8, 8)])
def test_wino_split_over_input_channels(ops, shape):
    """Few tiles, few output channels, many input channels (backward-data of the gamma/beta convolutions): the launch
    is split over the input channels into slabs and summed in a fixed order — same result as the unsplit launch up to
    the association of the sum, bit-identical from run to run."""
    from canonicalsg2im_amd._lib import WinoDesc, check, lib, ptr, stream
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    xd = ops.nhwc(x.cuda())
    up = ops.wino_pack(w.cuda(), False)
    d = WinoDesc()
    d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, W, Cin, Cin, Cout, Cout, 0, 0.0
    nws = lib.csg_wino_conv_workspace(d)
    assert nws > 0 and nws % (B * H * W * Cout * 4) == 0 and nws // (B * H * W * Cout * 4) >= 2
    ws = torch.empty(nws // 4, device="cuda")
    outs = []
    for _ in range(2):
        ws.fill_(float("nan"))
        y = ops.empty_nhwc(B, Cout, H, W, xd.device)
        check(lib.csg_wino_conv(d, ptr(xd), ptr(up), None, None, None, 0.0, ptr(y), ptr(ws), nws, stream()), "wino_conv")
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    y0 = _raw_wino(ops, x, w)                                   # no workspace: unsplit
    scale = float(ref.abs().max())
    assert_close(outs[0], ref.float(), 1e-4, 1e-5 * scale + 1e-5, "wino split %s" % (shape,))
    assert_close(outs[0], y0, 1e-4, 1e-5 * scale + 1e-5, "wino split vs unsplit %s" % (shape,))
    # with an epilogue the plan never splits
    d.act = 1
    assert lib.csg_wino_conv_workspace(d) == 0


@pytest.mark.parametrize("shape", SHAPES)
def test_wino_forward_vs_torch(ops, shape):
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    y = _raw_wino(ops, x, w, b)
    assert_close(y, ref, 1e-4, 1e-5 * float(ref.abs().max()) + 1e-5, "wino fwd %s" % (shape,))


def test_wino_epilogue_and_error_vs_fp64(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 32, 32, generator=g)
    w = torch.randn(96, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(96, generator=g)
    r = torch.randn(2, 96, 32, 32, generator=g)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    assert_close(_raw_wino(ops, x, w, b, None, ops.ACT_LEAKY, 0.2), ref, 1e-4, 2e-5, "leaky epilogue")
    ref = torch.tanh(F.conv2d(x, w, b, padding=1))
    assert_close(_raw_wino(ops, x, w, b, None, ops.ACT_TANH, 0.0), ref, 1e-4, 2e-5, "tanh epilogue")
    ref = F.conv2d(x, w, b, padding=1) + r
    assert_close(_raw_wino(ops, x, w, b, r), ref, 1e-4, 2e-5, "residual epilogue")
    # rounding error against an fp64 convolution, next to the direct MFMA kernel's on the same data
    ref64 = F.conv2d(x.double(), w.double(), None, padding=1)
    e_w = float((_raw_wino(ops, x, w).double().cpu() - ref64).abs().max())
    direct = ops.conv2d(x.cuda(), w.cuda()[:, :, :, :], None, 1, 1) if not ops.wino_eligible(2, 32, 32, 64, 96, 3, 3, 1, 1) else None
    scale = float(ref64.abs().max())
    assert e_w < 1e-5 * scale, (e_w, scale)
    if direct is not None:
        assert e_w < 8 * float((direct.double().cpu() - ref64).abs().max())


@pytest.mark.parametrize("shape", [(4, 32, 64, 64, 64), (1, 64, 128, 128, 128), (16, 128, 32, 32, 32)])
def test_conv2d_autograd_uses_winograd(ops, shape):
    """ops.conv2d on an eligible layer: forward, backward-data (Winograd with the flipped, transposed weights) and the
    weight gradient (direct kernel) against torch."""
    B, Cin, Cout, H, W = shape
    assert ops.wino_eligible(B, H, W, Cin, Cout, 3, 3, 1, 1) and ops.wino_eligible(B, H, W, Cout, Cin, 3, 3, 1, 1)
    g = torch.Generator().manual_seed(B * 7 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    pre = F.conv2d(xr, wr, br, padding=1)
    ref = F.leaky_relu(pre, 0.2)
    gy = torch.randn(ref.shape, generator=g)
    # an output within rounding distance of 0 takes either branch of the LeakyReLU, and ONE flipped gate moves dx by
    # 0.8*gy*w over a 3x3 neighbourhood: keep the upstream gradient away from those few elements
    gy = gy * (pre.detach().abs() > 1e-4)
    ref.backward(gy)
    xd, wd, bd = [t.cuda().requires_grad_(True) for t in (x, w, b)]
    y = ops.conv2d(xd, wd, bd, 1, 1, ops.ACT_LEAKY, 0.2)
    assert_close(y, ref, 1e-4, 2e-5, "y")
    y.backward(gy.cuda())
    assert_close(xd.grad, xr.grad, 1e-4, 1e-5 * float(xr.grad.abs().max()) + 1e-5, "dx (winograd)")
    assert_close(wd.grad, wr.grad, 1e-4, 1e-5 * float(wr.grad.abs().max()) + 1e-5, "dw")
    assert_close(bd.grad, br.grad, 1e-4, 1e-5 * float(br.grad.abs().max()) + 1e-5, "db")


def test_wino_weight_gradient_rejects_images_its_stages_do_not_tile(ops):
    """The F(3x3,2x2) kernel stages 16 tiles (16x1, 8x2 or 4x4) at a time and wants them to tile the image exactly;
    other sizes are refused by the C ABI (the autograd op then keeps the direct kernel)."""
    from canonicalsg2im_amd._lib import WinoDesc, lib
    for (H, W, ok) in ((12, 20, False), (24, 40, False), (6, 130, False), (8, 8, True), (4, 16, True), (2, 64, True)):
        d = WinoDesc()
        d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = 1, H, W, 16, 16, 32, 32, 0, 0.0
        assert (lib.csg_wino_bwd_weight_workspace(d) >= 0) == ok, (H, W)
    x = torch.randn(20, 16, 24, 40, device="cuda", requires_grad=True)       # eligible forward, ragged for the wgrad
    w = (torch.randn(32, 16, 3, 3, device="cuda") / 12.0).requires_grad_(True)
    assert ops.wino_eligible(20, 24, 40, 16, 32, 3, 3, 1, 1)
    y = ops.conv2d(x, w, None, 1, 1)
    gy = torch.randn_like(y)
    y.backward(gy)
    xr, wr = x.detach().cpu().requires_grad_(True), w.detach().cpu().requires_grad_(True)
    ref = F.conv2d(xr, wr, padding=1)
    ref.backward(gy.cpu())
    assert_close(y, ref, 1e-4, 2e-5, "y")
    assert_close(w.grad, wr.grad, 1e-4, 1e-5 * float(wr.grad.abs().max()), "dw (direct kernel fallback)")
    assert_close(x.grad, xr.grad, 1e-4, 1e-5 * float(xr.grad.abs().max()), "dx")


@pytest.mark.parametrize("shape", [(2, 16, 32, 8, 8), (1, 20, 36, 16, 16), (3, 64, 100, 24, 32), (2, 96, 64, 64, 64),
                                   (1, 128, 256, 128, 128), (5, 32, 128, 6, 128), (3, 36, 68, 10, 64)])
def test_wino_weight_gradient_vs_torch(ops, shape):
    """csg_wino_bwd_weight (F(3x3,2x2)) through the C ABI: dW in the direct kernel's [Cout][3][3][Cin] layout and the
    bias gradient; channel tails, every stage geometry (4x4, 8x2, 16x1 tiles), odd region counts, several slices."""
    from canonicalsg2im_amd._lib import WinoDesc, check, lib, ptr, stream
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    gy = torch.randn(B, Cout, H, W, generator=g)
    F.conv2d(x, w, b, padding=1).backward(gy)
    xd, gyd = ops.nhwc(x.cuda()), ops.nhwc(gy.cuda())
    d = WinoDesc()
    d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, W, Cin, Cin, Cout, Cout, 0, 0.0
    nbytes = lib.csg_wino_bwd_weight_workspace(d)
    assert nbytes > 0
    ws = torch.empty(nbytes // 4, device="cuda")
    dwp = torch.empty(Cout, 3, 3, Cin, device="cuda")
    db = torch.empty(Cout, device="cuda")
    check(lib.csg_wino_bwd_weight(d, ptr(xd), ptr(gyd), ptr(dwp), ptr(db), ptr(ws), nbytes, stream()), "wino_bwd_weight")
    assert_close(dwp.permute(0, 3, 1, 2), w.grad, 1e-4, 1e-5 * float(w.grad.abs().max()) + 1e-5, "dW %s" % (shape,))
    assert_close(db, b.grad, 1e-4, 1e-5 * float(b.grad.abs().max()) + 1e-5, "db %s" % (shape,))
    dwp2 = torch.empty_like(dwp)                       # bit-reproducible: fixed slab order, no atomics
    check(lib.csg_wino_bwd_weight(d, ptr(xd), ptr(gyd), ptr(dwp2), None, ptr(ws), nbytes, stream()), "wino_bwd_weight")
    assert torch.equal(dwp, dwp2)


def test_wino_full_size_window(ops):
    """The dominant generator shape (128 -> 256 channels at 256x256, B = 4): four 12x12 output windows against
    F.conv2d of the matching input windows on the CPU."""
    g = torch.Generator().manual_seed(78)
    B, Cin, Cout, H = 4, 128, 256, 256
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 34.0
    b = torch.randn(Cout, generator=g)
    assert ops.wino_eligible(B, H, H, Cin, Cout, 3, 3, 1, 1)
    y = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), 1, 1)
    for (bi, y0, x0) in ((0, 0, 0), (1, 100, 37), (3, 244, 244), (2, 127, 200)):
        ya, xa = max(y0 - 1, 0), max(x0 - 1, 0)
        yb, xb = min(y0 + 13, H), min(x0 + 13, H)
        patch = F.pad(x[bi:bi + 1, :, ya:yb, xa:xb], (1 if x0 == 0 else 0, 1 if x0 + 13 > H else 0,
                                                      1 if y0 == 0 else 0, 1 if y0 + 13 > H else 0))
        ref = F.conv2d(patch, w, b)
        assert_close(y[bi:bi + 1, :, y0:y0 + 12, x0:x0 + 12], ref, 1e-4, 2e-5, "window (%d,%d,%d)" % (bi, y0, x0))


@pytest.mark.parametrize("shape", [(2, 32, 128, 256, 64, 64),     # Winograd, gate folded into the backward-data epilogue
                                   (1, 32, 128, 2048, 16, 16),    # backward-data split over 2048 channels: separate gate pass
                                   (2, 32, 128, 64, 8, 8)])       # direct kernels
def test_producer_activation_folded_into_consumer_backward(ops, shape):
    """SPADE's actv = ReLU(conv(seg)) -> conv(actv): the pair (grad_is_pre, in_act) gives the gradients of the plain
    composition."""
    B, C0, C1, C2, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    seg = torch.randn(B, C0, H, W, generator=g)
    w1 = torch.randn(C1, C0, 3, 3, generator=g) / (3.0 * C0 ** 0.5)
    b1 = torch.randn(C1, generator=g) * 0.1
    w2 = torch.randn(C2, C1, 3, 3, generator=g) / (3.0 * C1 ** 0.5)
    gy = torch.randn(B, C2, H, W, generator=g)
    ref_in = [t.clone().double().requires_grad_(True) for t in (seg, w1, b1, w2)]
    dev = [t.clone().cuda().requires_grad_(True) for t in (seg, w1, b1, w2)]
    actv = ops.conv2d(dev[0], dev[1], dev[2], 1, 1, ops.ACT_LEAKY, 0.0, grad_is_pre=True)
    y = ops.conv2d(actv, dev[3], None, 1, 1, in_act=(ops.ACT_LEAKY, 0.0))
    got = torch.autograd.grad(y, dev, ops.nhwc(gy.cuda()))
    # the reference uses the device's gates: a pre-activation within rounding of zero may land on either side
    pre = F.conv2d(ref_in[0], ref_in[1], ref_in[2], padding=1)
    gates = (actv.detach().cpu() > 0).double()
    assert float(((pre.detach() > 0).double() - gates).abs().mean()) < 1e-4
    out = F.conv2d(pre * gates, ref_in[3], None, padding=1)
    ref = torch.autograd.grad(out, ref_in, gy.double())
    for name, a, b in zip(("dseg", "dw1", "db1", "dw2"), got, ref):
        scale = float(b.abs().max())
        assert_close(a, b.float(), 1e-4, 1e-5 * scale + 1e-6, "%s %s" % (name, shape))
