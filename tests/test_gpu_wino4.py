"""Winograd F(4x4,3x3) convolution (csrc/wino4.hip) on a real MI355X, raw C-ABI calls against an fp64 convolution.

Acceptance gate of the kernel: the error against fp64 stays below 1e-5 of the output scale — on
the shapes the generator runs (Cin = 128 SPADE gamma/beta layers, Cin up to 1024 in conv_0, 2048 input channels in
the gamma/beta backward-data pass, split over the input channels) — next to the error of F(2x2,3x3) and of the direct
MFMA kernel on the same data.  Geometry: ragged regions (W, H not multiples of the 32 x 16 pixel block region),
output-channel counts that are not multiples of 32, the fused epilogue (bias, LeakyReLU, tanh, residual, gate), and
the backward-data pass through the autograd op."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
GATE = 1e-5                       # max |y - fp64| / max |fp64|


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from canonicalsg2im_amd import ops as O
    return O


def _desc(B, H, W, Cin, Cout, act=0, slope=0.0):
    from canonicalsg2im_amd._lib import WinoDesc
    d = WinoDesc()
    d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, W, Cin, Cin, Cout, Cout, act, slope
    return d


def _raw_wino4(ops, x, w, bias=None, res=None, act=0, slope=0.0, gate=None, gate_slope=0.0, workspace=False):
    """Direct call of the C ABI: pack + conv (forward operand)."""
    from canonicalsg2im_amd._lib import check, lib, ptr, stream
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    d = _desc(B, H, W, Cin, Cout, act, slope)
    assert lib.csg_wino4_supported(d) == 1
    xd = ops.nhwc(x.cuda())
    up = ops.wino_pack(w.cuda(), False, None, 4)
    y = ops.empty_nhwc(B, Cout, H, W, xd.device)
    bd = bias.cuda() if bias is not None else None
    rd = ops.nhwc(res.cuda()) if res is not None else None
    gd = ops.nhwc(gate.cuda()) if gate is not None else None
    ws, nws = None, 0
    if workspace:
        nws = lib.csg_wino4_conv_workspace(d)
        assert nws > 0
        ws = torch.full((nws // 4,), float("nan"), device="cuda")
    check(lib.csg_wino4_conv(d, ptr(xd), ptr(up), ptr(bd), ptr(rd), ptr(gd), gate_slope, ptr(y), ptr(ws), nws, stream()),
          "wino4_conv")
    return y


def _data(shape, relu_in=False):
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    if relu_in:
        x = x.clamp_min(0)                       # SPADE: the gamma/beta convolutions read actv = ReLU(mlp_shared(seg))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    return x, w, b


def _err(y, ref64):
    return float((y.detach().cpu().double() - ref64).abs().max() / ref64.abs().max())


SHAPES = [
    # B, Cin, Cout, H,  W
    (2, 128, 256, 64, 64),      # the SPADE gamma/beta shape (Cin = 128), whole regions
    (1, 128, 64, 32, 32),
    (2, 32, 128, 16, 32),       # mlp_shared: 4 stages, one region row
    (1, 64, 36, 20, 40),        # ragged regions in both directions, Cout not a multiple of 32
    (1, 8, 32, 128, 36),        # a single stage, tall map, W = 36
    (1, 1024, 64, 32, 32),      # long reduction: conv_0 of up_0
    (1, 72, 100, 24, 48),       # Cin = 9 stages (odd stage count), Cout = 100
]


@pytest.mark.parametrize("shape", SHAPES)
def test_wino4_forward_vs_fp64(ops, shape):
    x, w, b = _data(shape, relu_in=shape[1] == 128)
    ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    y = _raw_wino4(ops, x, w, b)
    e = _err(y, ref64)
    assert e < GATE, "F(4x4,3x3) %s: error %.2e of the output scale" % (shape, e)
    assert_close(y, ref64.float(), 1e-4, 1e-5 * float(ref64.abs().max()) + 1e-5, "wino4 fwd %s" % (shape,))


def test_wino4_error_next_to_f2_and_direct(ops):
    """The three fp32 paths on the same data against fp64 (printed with -s; the table is quoted in DESIGN.md)."""
    from canonicalsg2im_amd._lib import check, lib, ptr, stream
    rows = []
    for shape in [(2, 128, 256, 64, 64), (1, 512, 256, 64, 64), (1, 1024, 512, 32, 32)]:
        B, Cin, Cout, H, W = shape
        x, w, _ = _data(shape, relu_in=Cin == 128)
        ref64 = F.conv2d(x.double(), w.double(), None, padding=1)
        y4 = _raw_wino4(ops, x, w)
        xd = ops.nhwc(x.cuda())
        y2 = ops.empty_nhwc(B, Cout, H, W, xd.device)
        up2 = ops.wino_pack(w.cuda(), False, None, 2)
        check(lib.csg_wino_conv(_desc(B, H, W, Cin, Cout), ptr(xd), ptr(up2), None, None, None, 0.0, ptr(y2), None, 0,
                                stream()), "wino_conv")
        saved = ops.WINO_ENABLED
        ops.WINO_ENABLED = False                 # wino_eligible() reads it per call: the direct MFMA kernel runs
        try:
            yd = ops.conv2d(xd, w.cuda(), None, 1, 1)
        finally:
            ops.WINO_ENABLED = saved
        rows.append((shape, _err(y4, ref64), _err(y2, ref64), _err(yd, ref64)))
    print("\\nshape (B,Cin,Cout,H,W)            F(4x4,3x3)  F(2x2,3x3)  direct MFMA   (max |y - fp64| / max |fp64|)")
    for shape, e4, e2, ed in rows:
        print("%-32s  %.2e    %.2e    %.2e" % (shape, e4, e2, ed))
        assert e4 < GATE and e2 < GATE and ed < GATE


def test_wino4_epilogue(ops):
    shape = (2, 64, 96, 32, 32)
    x, w, b = _data(shape)
    g = torch.Generator().manual_seed(11)
    r = torch.randn(2, 96, 32, 32, generator=g)
    gt = torch.randn(2, 96, 32, 32, generator=g)
    pre = F.conv2d(x, w, b, padding=1)
    assert_close(_raw_wino4(ops, x, w, b, None, ops.ACT_LEAKY, 0.2), F.leaky_relu(pre, 0.2), 1e-4, 2e-5, "leaky epilogue")
    assert_close(_raw_wino4(ops, x, w, b, None, ops.ACT_TANH, 0.0), torch.tanh(pre), 1e-4, 2e-5, "tanh epilogue")
    assert_close(_raw_wino4(ops, x, w, b, r), pre + r, 1e-4, 2e-5, "residual epilogue")
    want = F.conv2d(x, w, None, padding=1) * torch.where(gt > 0, torch.ones_like(gt), torch.full_like(gt, 0.2))
    assert_close(_raw_wino4(ops, x, w, None, None, gate=gt, gate_slope=0.2), want, 1e-4, 2e-5, "gate epilogue")


@pytest.mark.parametrize("shape", [(1, 2048, 128, 32, 32), (2, 1024, 64, 16, 32)])
def test_wino4_split_over_input_channels(ops, shape):
    """Backward-data of the gamma/beta convolutions at 32 x 32: few blocks, 2048 input channels — slabs over the input
    channels + an ordered sum; bit-identical from run to run, same values as the unsplit launch up to association."""
    from canonicalsg2im_amd._lib import lib
    B, Cin, Cout, H, W = shape
    x, w, _ = _data(shape)
    ref64 = F.conv2d(x.double(), w.double(), None, padding=1)
    d = _desc(B, H, W, Cin, Cout)
    nws = lib.csg_wino4_conv_workspace(d)
    assert nws > 0 and nws % (B * H * W * Cout * 4) == 0 and nws // (B * H * W * Cout * 4) >= 2
    a = _raw_wino4(ops, x, w, workspace=True)
    b2 = _raw_wino4(ops, x, w, workspace=True)
    assert torch.equal(a, b2)
    y0 = _raw_wino4(ops, x, w)                                   # no workspace: unsplit
    # the split launch adds up 128-channel partial sums: it is MORE accurate than one 2048-term fp32 chain per output,
    # which is what the unsplit launch (never chosen for these shapes by the plan) computes
    assert _err(a, ref64) < GATE and _err(y0, ref64) < 3 * GATE
    scale = float(ref64.abs().max())
    assert_close(a, y0, 1e-4, 1e-5 * scale + 1e-5, "wino4 split vs unsplit %s" % (shape,))
    d.act = 1
    assert lib.csg_wino4_conv_workspace(d) == 0                  # with an epilogue the plan never splits


def test_variant_follows_the_grid(ops):
    """F(4x4,3x3) on launches that fill the chip, F(2x2,3x3) where fewer than ~160 (region, channel block) items would leave
    it half idle — unless the launch is plain and long enough to be split over its input channels."""
    assert ops.wino_variant(16, 64, 64, 512, 256) == 4            # 512 items
    assert ops.wino_variant(4, 64, 64, 512, 256) == 2             # 128 items, epilogue: not splittable
    assert ops.wino_variant(4, 64, 64, 512, 256, plain=True) == 4     # split over 512 input channels instead
    assert ops.wino_variant(4, 32, 32, 128, 512, plain=True) == 2     # 64 items, too few channels to split
    assert ops.wino_variant(6, 64, 64, 256, 256) == 4             # 192 items
    assert ops.wino_variant(4, 16, 16, 512, 512) == 2             # below 32 pixels: never F(4x4,3x3)


def _both_forms(fn):
    """fn() with the persistent form of k_wino4_conv_v switched on, then off (csg_wino4_persistent)."""
    from canonicalsg2im_amd._lib import lib
    prev = lib.csg_wino4_persistent(1)
    try:
        a = fn()
        lib.csg_wino4_persistent(0)
        b = fn()
    finally:
        lib.csg_wino4_persistent(prev if prev >= 0 else 1)
    return a, b


# at least two (region, 64-channel block) items per CU (>= 512 on 256 CUs), an even stage count, Cout % 64 == 0
PERSISTENT_SHAPES = [
    (2, 32, 512, 128, 128),     # 4 stages: the shortest pipeline the persistent form takes (mlp_shared-like)
    (3, 128, 256, 112, 96),     # the SPADE gamma/beta shape; 7 x 3 regions per image: 252 items per XCD share, ragged walk
    (1, 64, 64, 272, 512),      # one channel block per region
]


@pytest.mark.parametrize("shape", PERSISTENT_SHAPES)
def test_wino4_persistent_form_is_bit_identical(ops, shape):
    """One block per CU walking its items with the stage pipeline carried across them (wino4.hip, k_wino4_conv_v<4, true>)
    against one block per item: same arithmetic in the same order — equal bits — and both inside the fp64 gate."""
    B, Cin, Cout, H, W = shape
    x, w, b = _data(shape, relu_in=Cin == 128)
    ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    yp, y1 = _both_forms(lambda: _raw_wino4(ops, x, w, b))
    assert torch.equal(yp, y1), "persistent vs one-item %s: max diff %.3e" % (shape, float((yp - y1).abs().max()))
    assert _err(yp, ref64) < GATE
    yp2, _ = _both_forms(lambda: _raw_wino4(ops, x, w, b))
    assert torch.equal(yp, yp2)                                  # and from run to run


def test_wino4_persistent_form_epilogues(ops):
    """Activation / residual / gate epilogues and the SPADE modulation (csg_wino4_conv_part: gamma half, then the beta
    half writing leaky(xhat (1 + gamma) + beta)) through the persistent form, bit-identical to the one-item form."""
    from canonicalsg2im_amd._lib import check, lib, ptr, stream
    shape = (2, 32, 256, 128, 128)
    B, Cin, Cout, H, W = shape
    x, w, b = _data(shape)
    g = torch.Generator().manual_seed(5)
    r = torch.randn(B, Cout, H, W, generator=g)
    gt = torch.randn(B, Cout, H, W, generator=g)
    pre = F.conv2d(x, w, b, padding=1)
    for name, kw, want in (
            ("leaky", dict(bias=b, act=ops.ACT_LEAKY, slope=0.2), F.leaky_relu(pre, 0.2)),
            ("tanh", dict(bias=b, act=ops.ACT_TANH), torch.tanh(pre)),
            ("residual", dict(bias=b, res=r), pre + r),
            ("gate", dict(gate=gt, gate_slope=0.2),
             F.conv2d(x, w, None, padding=1) * torch.where(gt > 0, torch.ones_like(gt), torch.full_like(gt, 0.2)))):
        yp, y1 = _both_forms(lambda: _raw_wino4(ops, x, w, **kw))
        assert torch.equal(yp, y1), name
        assert_close(yp, want, 1e-4, 2e-5, "persistent %s epilogue" % name)

    # the modulation: w holds gamma || beta (2 C output channels), C = 128
    C = Cout // 2
    xm = torch.randn(B, C, H, W, generator=g) * 2.0 + 0.5
    mean = xm.mean(dim=(0, 2, 3))
    invstd = 1.0 / torch.sqrt(xm.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    xd, xmd = ops.nhwc(x.cuda()), ops.nhwc(xm.cuda())
    up = ops.wino_pack(w.cuda(), False, None, 4)
    bd, md, rd = b.cuda(), mean.cuda(), invstd.cuda()

    def modulated():
        gbuf = ops.empty_nhwc(B, C, H, W, xd.device)
        y = ops.empty_nhwc(B, C, H, W, xd.device)
        d = _desc(B, H, W, Cin, C)
        check(lib.csg_wino4_conv_part(d, ptr(xd), ptr(up), 0, 2 * C // 32, ptr(bd), None, None, 0, None, None, 1.0, ptr(gbuf),
                                      stream()), "gamma half")
        check(lib.csg_wino4_conv_part(d, ptr(xd), ptr(up), C // 32, 2 * C // 32, ptr(bd[C:]), ptr(xmd), ptr(gbuf), C, ptr(md),
                                      ptr(rd), 0.2, ptr(y), stream()), "beta half")
        return y

    yp, y1 = _both_forms(modulated)
    assert torch.equal(yp, y1)
    xhat = (xm - mean[None, :, None, None]) * invstd[None, :, None, None]
    want = F.leaky_relu(xhat * (1 + pre[:, :C]) + pre[:, C:], 0.2)
    assert_close(yp, want, 1e-4, 5e-5, "persistent modulation epilogue")


@pytest.mark.parametrize("shape", [(2, 128, 256, 32, 64), (1, 64, 64, 64, 32)])
def test_wino4_backward_data_through_autograd(ops, shape, monkeypatch):
    """conv2d on a >= 32-wide map: forward and backward-data both run F(4x4,3x3) (the weight gradient stays on
    F(3x3,2x2)); gradients against the fp64 reference."""
    B, Cin, Cout, H, W = shape
    monkeypatch.setattr(ops, "WINO4_MIN_ITEMS", 0)               # (the small test shapes would otherwise take F(2x2,3x3))
    assert ops.wino_variant(B, H, W, Cin, Cout) == 4 and ops.wino_variant(B, H, W, Cout, Cin) == 4
    x, w, b = _data(shape)
    gy = torch.randn(B, Cout, H, W, generator=torch.Generator().manual_seed(3))
    xr, wr, br = [t.clone().double().requires_grad_(True) for t in (x, w, b)]
    F.conv2d(xr, wr, br, padding=1).backward(gy.double())
    xd, wd, bd = [t.clone().cuda().requires_grad_(True) for t in (x, w, b)]
    y = ops.conv2d(ops.nhwc(xd), wd, bd, 1, 1)
    y.backward(ops.nhwc(gy.cuda()))
    for name, mine, want in (("dx", xd.grad, xr.grad), ("dw", wd.grad, wr.grad), ("db", bd.grad, br.grad)):
        e = _err(mine, want.detach())
        assert e < GATE, "%s %s: error %.2e of the gradient scale" % (name, shape, e)


# ------------------------------------------------------------------------------------ F(3x3,4x4): 4x4 / stride 1 layers
def _raw_wino34(ops, x, w, pad, bias=None, res=None, act=0, slope=0.0, gate=None, gate_slope=0.0, backward_data=False):
    from canonicalsg2im_amd._lib import check, lib, ptr, stream
    B, Cx, H, W = x.shape
    Cout = w.shape[1] if backward_data else w.shape[0]
    d = _desc(B, H, W, Cx, Cout, act, slope)
    assert lib.csg_wino34_supported(d, pad) == 1
    xd = ops.nhwc(x.cuda())
    up = ops.wino_pack(w.cuda(), backward_data, None, 34)
    y = ops.empty_nhwc(B, Cout, H + 2 * pad - 3, W + 2 * pad - 3, xd.device)
    bd = bias.cuda() if bias is not None else None
    rd = ops.nhwc(res.cuda()) if res is not None else None
    gd = ops.nhwc(gate.cuda()) if gate is not None else None
    check(lib.csg_wino34_conv(d, pad, ptr(xd), ptr(up), ptr(bd), ptr(rd), ptr(gd), gate_slope, ptr(y), None, 0, stream()),
          "wino34_conv")
    return y


def _data44(shape, seed=0):
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + seed)
    x = F.leaky_relu(torch.randn(B, Cin, H, W, generator=g), 0.2)      # what the PatchGAN's fourth layer reads
    w = torch.randn(Cout, Cin, 4, 4, generator=g) / (4.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    return x, w, b


SHAPES44 = [
    # B, Cin, Cout, H,  W,  pad
    (2, 256, 512, 32, 32, 2),     # D0.model3 at 256 x 256 (output 33 x 33 = 11 x 11 tiles)
    (2, 256, 512, 16, 16, 2),     # D1.model3 (output 17 x 17: ragged tiles)
    (1, 64, 36, 20, 29, 2),       # ragged in both directions, Cout not a multiple of 32
    (1, 72, 100, 23, 41, 1),      # padding 1 (the geometry of the backward-data pass), odd stage count
    (1, 512, 64, 33, 33, 1),      # backward-data shape of D0.model3: 512 -> (256), output 32 x 32
]


@pytest.mark.parametrize("shape", SHAPES44)
def test_wino34_forward_vs_fp64(ops, shape):
    x, w, b = _data44(shape[:5])
    pad = shape[5]
    ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=pad)
    y = _raw_wino34(ops, x, w, pad, b)
    assert tuple(y.shape) == tuple(ref64.shape)
    e = _err(y, ref64)
    assert e < GATE, "F(3x3,4x4) %s: error %.2e of the output scale" % (shape, e)
    assert_close(y, ref64.float(), 1e-4, 1e-5 * float(ref64.abs().max()) + 1e-5, "wino34 fwd %s" % (shape,))


def test_wino34_epilogue_and_determinism(ops):
    shape = (2, 64, 96, 17, 22)
    x, w, b = _data44(shape)
    g = torch.Generator().manual_seed(12)
    r = torch.randn(2, 96, 18, 23, generator=g)
    gt = torch.randn(2, 96, 18, 23, generator=g)
    pre = F.conv2d(x, w, b, padding=2)
    tol = 1e-5 * float(pre.abs().max()) + 1e-5
    assert_close(_raw_wino34(ops, x, w, 2, b, None, ops.ACT_LEAKY, 0.2), F.leaky_relu(pre, 0.2), 1e-4, tol, "leaky epilogue")
    assert_close(_raw_wino34(ops, x, w, 2, b, r), pre + r, 1e-4, tol, "residual epilogue")
    want = F.conv2d(x, w, None, padding=2) * torch.where(gt > 0, torch.ones_like(gt), torch.full_like(gt, 0.2))
    assert_close(_raw_wino34(ops, x, w, 2, None, None, gate=gt, gate_slope=0.2), want, 1e-4, tol, "gate epilogue")
    assert torch.equal(_raw_wino34(ops, x, w, 2, b), _raw_wino34(ops, x, w, 2, b))


def test_wino34_error_next_to_direct(ops):
    """F(3x3,4x4) and the direct MFMA kernel on the PatchGAN shapes against fp64 (printed with -s)."""
    rows = []
    for shape in [(2, 256, 512, 32, 32), (2, 256, 512, 16, 16)]:
        x, w, _ = _data44(shape)
        ref64 = F.conv2d(x.double(), w.double(), None, padding=2)
        yw = _raw_wino34(ops, x, w, 2)
        saved = ops.WINO_ENABLED
        ops.WINO_ENABLED = False
        try:
            yd = ops.conv2d(ops.nhwc(x.cuda()), w.cuda(), None, 1, 2)
        finally:
            ops.WINO_ENABLED = saved
        rows.append((shape, _err(yw, ref64), _err(yd, ref64)))
    print("\nshape (B,Cin,Cout,H,W)            F(3x3,4x4)  direct MFMA   (max |y - fp64| / max |fp64|)")
    for shape, ew, ed in rows:
        print("%-32s  %.2e    %.2e" % (shape, ew, ed))
        assert ew < GATE and ed < GATE


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("shape", [(2, 256, 512, 32, 32), (1, 64, 64, 16, 19)])
def test_wino34_through_autograd(ops, shape, mode):
    """conv2d with a 4x4 / stride 1 / pad 2 weight: backward-data on F(3x3,4x4) (default mode) and the forward too
    (mode 3; the weight gradient stays on the direct kernel); against fp64."""
    B, Cin, Cout, H, W = shape
    monkey = ops.WINO34_MODE
    ops.WINO34_MODE = mode
    try:
        _wino34_autograd_case(ops, shape)
    finally:
        ops.WINO34_MODE = monkey


def _wino34_autograd_case(ops, shape):
    B, Cin, Cout, H, W = shape
    assert ops.wino34_eligible(B, H + 1, W + 1, Cout, Cin, 4, 4, 1, 1, backward=True)
    x, w, b = _data44(shape, seed=1)
    gy = torch.randn(B, Cout, H + 1, W + 1, generator=torch.Generator().manual_seed(3))
    xr, wr, br = [t.clone().double().requires_grad_(True) for t in (x, w, b)]
    F.conv2d(xr, wr, br, padding=2).backward(gy.double())
    xd, wd, bd = [t.clone().cuda().requires_grad_(True) for t in (x, w, b)]
    y = ops.conv2d(ops.nhwc(xd), wd, bd, 1, 2)
    y.backward(ops.nhwc(gy.cuda()))
    for name, mine, want in (("dx", xd.grad, xr.grad), ("dw", wd.grad, wr.grad), ("db", bd.grad, br.grad)):
        e = _err(mine, want.detach())
        assert e < GATE, "%s %s: error %.2e of the gradient scale" % (name, shape, e)


def test_wino34_split_over_input_channels(ops):
    """The half-resolution scale's backward-data shape (512 -> 256 channels on a 17 x 17 map, 128 blocks): slabs over the
    input channels + an ordered sum; bit-identical from run to run, the unsplit launch's values up to association."""
    from canonicalsg2im_amd._lib import check, lib, ptr, stream
    shape = (16, 512, 256, 17, 17)
    B, Cin, Cout, H, W = shape
    x, w, _ = _data44(shape)
    ref64 = F.conv2d(x.double(), w.double(), None, padding=1)
    d = _desc(B, H, W, Cin, Cout)
    nws = lib.csg_wino34_conv_workspace(d, 1)
    assert nws > 0 and nws % (B * (H - 1) * (W - 1) * Cout * 4) == 0
    xd, up = ops.nhwc(x.cuda()), ops.wino_pack(w.cuda(), False, None, 34)
    outs = []
    for _ in range(2):
        y = ops.empty_nhwc(B, Cout, H - 1, W - 1, xd.device)
        ws = torch.full((nws // 4,), float("nan"), device="cuda")
        check(lib.csg_wino34_conv(d, 1, ptr(xd), ptr(up), None, None, None, 0.0, ptr(y), ptr(ws), nws, stream()), "wino34 split")
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    y0 = _raw_wino34(ops, x, w, 1)
    assert _err(outs[0], ref64) < GATE and _err(y0, ref64) < GATE
    assert_close(outs[0], y0, 1e-4, 1e-5 * float(ref64.abs().max()) + 1e-5, "wino34 split vs unsplit")


# ------------------------------------------------------------------------------------------------ weight gradient, F(3x3,4x4)
def _raw_wgrad(ops, x, gy, which):
    """csg_wino4_bwd_weight / csg_wino_bwd_weight through the C ABI -> (dW in (Cout, Cin, 3, 3) order, db, workspace bytes)."""
    from canonicalsg2im_amd._lib import check, lib, ptr, stream
    B, Cin, H, W = x.shape
    Cout = gy.shape[1]
    d = _desc(B, H, W, Cin, Cout)
    wsf, fn = ((lib.csg_wino4_bwd_weight_workspace, lib.csg_wino4_bwd_weight) if which == 4
               else (lib.csg_wino_bwd_weight_workspace, lib.csg_wino_bwd_weight))
    nbytes = wsf(d)
    assert nbytes > 0, (which, x.shape)
    xd, gyd = ops.nhwc(x.cuda()), ops.nhwc(gy.cuda())
    ws = torch.empty(nbytes // 4, device="cuda")
    dwp = torch.full((Cout, 3, 3, Cin), float("nan"), device="cuda")
    db = torch.full((Cout,), float("nan"), device="cuda")
    check(fn(d, ptr(xd), ptr(gyd), ptr(dwp), ptr(db), ptr(ws), nbytes, stream()), "bwd_weight")
    dwp2 = torch.empty_like(dwp)
    check(fn(d, ptr(xd), ptr(gyd), ptr(dwp2), None, ptr(ws), nbytes, stream()), "bwd_weight")
    assert torch.equal(dwp, dwp2), "weight gradient is not bit-reproducible"
    return dwp.permute(0, 3, 1, 2).contiguous(), db


WG_SHAPES = [(2, 64, 64, 8, 8),          # one region per image: every halo flag at once
             (1, 64, 64, 16, 24),        # 2 x 3 regions, interior columns
             (3, 128, 64, 32, 32),       # two cin blocks
             (2, 64, 192, 40, 16),       # three cout blocks, H != W
             (2, 72, 100, 16, 16),       # channel tails (72 = 64 + 8, 100 = 64 + 36): zero-filled lanes, masked slab rows
             (5, 128, 256, 64, 64),      # several slices per (cout, cin) block, odd region counts
             (1, 256, 128, 128, 128)]    # a generator shape (up_2.conv_0) at batch 1


@pytest.mark.parametrize("shape", WG_SHAPES)
def test_wino4_weight_gradient_vs_fp64(ops, shape):
    """csg_wino4_bwd_weight (F(3x3,4x4), csrc/wino4w.hip) against the fp64 weight gradient of F.conv2d: dW in the direct
    kernel's [Cout][3][3][Cin] layout and the bias gradient, < 1e-5 of the gradient's scale (the forward kernel's gate)."""
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + 7)
    x = torch.randn(B, Cin, H, W, generator=g)
    gy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, b, padding=1).backward(gy.double())
    dw, db = _raw_wgrad(ops, x, gy, 4)
    err = float((dw.double().cpu() - w.grad).abs().max() / w.grad.abs().max())
    errb = float((db.double().cpu() - b.grad).abs().max() / b.grad.abs().max())
    assert err < GATE and errb < GATE, (shape, err, errb)


def test_wino4_weight_gradient_error_next_to_f2(ops):
    """The error of F(3x3,4x4) next to F(3x3,2x2) on the dominant shape's channel counts (128 -> 256, 64 x 64 map, B = 4:
    16 384 tiles per sum): both far inside the gate; the ratio is recorded by the assertion message of a failure."""
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H = 4, 128, 256, 64
    x = torch.randn(B, Cin, H, H, generator=g)
    gy = torch.randn(B, Cout, H, H, generator=g)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, padding=1).backward(gy.double())
    scale = w.grad.abs().max()
    e4 = float((_raw_wgrad(ops, x, gy, 4)[0].double().cpu() - w.grad).abs().max() / scale)
    e2 = float((_raw_wgrad(ops, x, gy, 2)[0].double().cpu() - w.grad).abs().max() / scale)
    assert e4 < GATE and e2 < GATE and e4 < 8.0 * e2 + 1e-6, (e4, e2)


def test_wino4_weight_gradient_shapes_it_refuses(ops):
    from canonicalsg2im_amd._lib import lib
    for (H, W, ok) in ((8, 8, True), (16, 40, True), (12, 16, False), (16, 20, False), (4, 64, False)):
        assert (lib.csg_wino4_bwd_weight_workspace(_desc(1, H, W, 64, 64)) >= 0) == ok, (H, W)


@pytest.mark.parametrize("shape", [(2, 128, 256, 32, 32), (1, 64, 64, 64, 32)])
def test_wino4_weight_gradient_through_autograd(ops, shape):
    """ops.conv2d's backward takes the F(3x3,4x4) weight gradient on these shapes (>= 64 channels, H and W multiples of 8,
    >= 1 024 pixels): dW and db against fp64 autograd, and the same launch with the kernel turned off for comparison."""
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    gy = torch.randn(B, Cout, H, W, generator=g)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    F.conv2d(x.double(), wr, br, padding=1).backward(gy.double())
    got = {}
    for on in (True, False):
        ops.WINO4_WGRAD = on
        try:
            wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
            ops.conv2d(x.cuda(), wd, bd, 1, 1).backward(gy.cuda())
        finally:
            ops.WINO4_WGRAD = True
        got[on] = wd.grad.double().cpu()
        assert float((got[on] - wr.grad).abs().max() / wr.grad.abs().max()) < GATE, (shape, on)
        assert float((bd.grad.double().cpu() - br.grad).abs().max() / br.grad.abs().max()) < GATE, (shape, on)
    assert not torch.equal(got[True], got[False])      # (two different kernels did run)
