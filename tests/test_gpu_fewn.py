"""csrc/fewn.hip: stride-1 convolutions with at most four output channels (`conv_img`, generator.py:46,120-121; the
PatchGAN prediction heads, discriminator.py:185-187) — forward, backward-data, weight and bias gradient against torch,
through ops.conv2d (the call sites' entry point) and with the MFMA path as a second witness."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__ as ge
    ge.build()
    from canonicalsg2im_amd import ops as o
    return o


CASES = [
    # B, Cin, Cout, k, pad, H,  W,  act
    (2, 64, 3, 3, 1, 40, 72, "tanh"),      # conv_img: ragged 8x32 tiles
    (2, 512, 1, 4, 2, 18, 18, "none"),     # PatchGAN head: output 19x19
    (1, 32, 2, 4, 2, 9, 33, "none"),
    (2, 128, 4, 3, 1, 16, 16, "leaky"),
    (1, 1024, 1, 3, 0, 7, 7, "none"),      # no padding: output 5x5
    (3, 256, 3, 3, 2, 6, 10, "none"),      # padding larger than 'same'
]


@pytest.mark.parametrize("case", CASES)
def test_few_output_conv_vs_torch(ops, case):
    from canonicalsg2im_amd._lib import FewDesc, lib
    B, Cin, Cout, k, pad, H, W, act = case
    d = FewDesc()
    d.B, d.IH, d.IW, d.Cin, d.x_cs, d.KH, d.KW, d.pad, d.cout_real, d.act, d.slope = B, H, W, Cin, Cin, k, k, pad, Cout, 0, 0.0
    assert lib.csg_conv_few_supported(d) == 1
    g = torch.Generator().manual_seed(sum(case[:7]))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (k * Cin ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    code = {"none": ops.ACT_NONE, "tanh": ops.ACT_TANH, "leaky": ops.ACT_LEAKY}[act]
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, w, b)]
    pre = F.conv2d(ref_in[0], ref_in[1], ref_in[2], padding=pad)
    ref = {"none": pre, "tanh": torch.tanh(pre), "leaky": F.leaky_relu(pre, 0.2)}[act]
    gy = torch.randn(ref.shape, generator=g)
    if act == "leaky":
        gy = gy * (pre.detach().abs() > 1e-4).float()             # keep clear of the kink
    ref_g = torch.autograd.grad(ref, ref_in, gy.double())

    outs = {}
    for few in (True, False):                                      # VALU kernels, then the MFMA path
        ops.FEW_ENABLED = few
        try:
            dev = [t.clone().cuda().requires_grad_(True) for t in (x, w, b)]
            y = ops.conv2d(dev[0], dev[1], dev[2], 1, pad, code, 0.2)
            got = torch.autograd.grad(y, dev, gy.cuda())
        finally:
            ops.FEW_ENABLED = True
        outs[few] = (y.detach(), got)
        tag = "%s few=%s" % (case, few)
        assert_close(y, ref.float(), 1e-4, 1e-5 * float(ref.detach().abs().max()) + 1e-6, "y " + tag)
        for name, a, r in zip(("dx", "dw", "db"), got, ref_g):
            assert_close(a, r.float(), 1e-4, 2e-5 * float(r.abs().max()) + 1e-6, name + " " + tag)
    # bit-reproducible weight gradient
    dev = [t.clone().cuda().requires_grad_(True) for t in (x, w, b)]
    y = ops.conv2d(dev[0], dev[1], dev[2], 1, pad, code, 0.2)
    again = torch.autograd.grad(y, dev, gy.cuda())
    assert torch.equal(again[1], outs[True][1][1]) and torch.equal(again[2], outs[True][1][2])


def test_unsupported_shapes_are_declined(ops):
    from canonicalsg2im_amd._lib import FewDesc, lib
    d = FewDesc()
    d.B, d.IH, d.IW, d.Cin, d.x_cs, d.KH, d.KW, d.pad, d.cout_real, d.act, d.slope = 1, 8, 8, 64, 64, 5, 5, 2, 1, 0, 0.0
    assert lib.csg_conv_few_supported(d) == 0                       # 5x5
    d.KH = d.KW = 4
    d.cout_real = 3
    assert lib.csg_conv_few_supported(d) == 0                       # 16 taps x 3 outputs: registers
    d.cout_real, d.Cin, d.x_cs = 1, 48, 48
    assert lib.csg_conv_few_supported(d) == 0                       # Cin not a power-of-two multiple of 32


def test_conv_img_with_the_leaky_relu_in_its_loaders():
    """`conv_img(leaky_relu(x, 0.2))` of the generator (reference generator.py:123-124) with the activation folded into
    the few-output kernels' forward / weight-gradient loaders and its derivative into the backward-data epilogue
    (csg_conv_desc.res_gate): output, d x, d weight, d bias against torch on the CPU, and bit-equal to the unfolded form."""
    import torch.nn.functional as F
    from canonicalsg2im_amd import ops
    torch.manual_seed(3)
    x = torch.randn(2, 64, 40, 48)
    w = torch.randn(3, 64, 3, 3) * 0.05
    b = torch.randn(3) * 0.1
    g = torch.randn(2, 3, 40, 48)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.tanh(F.conv2d(F.leaky_relu(xr, 0.2), wr, br, padding=1))
    yr.backward(g)
    outs = []
    for folded in (True, False):
        xd = ops.nhwc(x.cuda()).requires_grad_(True)
        wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        if folded:
            y = ops.conv2d(xd, wd, bd, 1, 1, act=ops.ACT_TANH, pre_slope=0.2)
        else:
            y = ops.conv2d(F.leaky_relu(xd, 0.2), wd, bd, 1, 1, act=ops.ACT_TANH)
        y.backward(g.cuda())
        outs.append((y.detach(), xd.grad, wd.grad, bd.grad))
        assert torch.allclose(y.cpu(), yr.detach(), rtol=1e-4, atol=1e-5)
        assert torch.allclose(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5 * float(xr.grad.abs().max()))
        assert torch.allclose(wd.grad.cpu(), wr.grad, rtol=1e-4, atol=1e-5 * float(wr.grad.abs().max()))
        assert torch.allclose(bd.grad.cpu(), br.grad, rtol=1e-4, atol=1e-5 * float(br.grad.abs().max()))
    for a, c in zip(*outs):
        assert torch.equal(a, c), "folding the activation into the loaders must not change a bit"
