"""csrc/gemm.hip: the plain fp32 GEMM kernels behind nn.Linear (graph encoder, sg2im/graph.py:63-77) and 1x1 convolutions
(SPADEResnetBlock.conv_s) — through the C ABI against an fp64 product, and through ops.linear / ops.conv2d (the call sites'
entry points) against torch autograd."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__ as ge
    ge.build()
    from canonicalsg2im_amd import ops as o
    old, o.GEMM_MODE = o.GEMM_MODE, "all"              # every shape the kernels serve, not only the ones they are the default for
    yield o
    o.GEMM_MODE = old


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# M, N, K, lda pad, act, gate, bias       (ragged M / N / K tails, K not a multiple of a stage, strided rows)
NT_CASES = [
    (1000, 512, 384, 0, "relu", False, True),
    (257, 130 * 4, 36, 0, "none", True, False),        # N tail, K < one stage, gated
    (128, 128, 32, 0, "none", False, False),           # exactly one tile, one stage
    (4099, 64, 1152, 8, "leaky", False, True),         # N < tile, padded rows
    (33, 1152, 516, 4, "none", True, True),            # M < tile, K tail of 4
    (70000, 128, 128, 0, "relu", False, True),         # many row tiles
    (5000, 36, 64, 0, "none", True, True),             # 128 x 64 tiles with a column tail, two stages
]


@pytest.mark.parametrize("case", NT_CASES)
def test_gemm_nt_vs_fp64(ops, case):
    from canonicalsg2im_amd._lib import GemmDesc, lib, last_error
    M, N, K, pad, act, gated, biased = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    a = torch.randn(M, K + pad, generator=g).to(dev)
    w = torch.randn(N, K + pad, generator=g).to(dev) / K ** 0.5
    b = torch.randn(N, generator=g).to(dev) if biased else None
    gt = torch.randn(M, N + 4, generator=g).to(dev) if gated else None
    y = torch.full((M, N + 4), 7.0, device=dev)
    d = GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldy, d.ldg = M, N, K, K + pad, K + pad, N + 4, N + 4
    d.act, d.slope, d.gate_slope = {"none": ops.ACT_NONE, "relu": ops.ACT_LEAKY, "leaky": ops.ACT_LEAKY}[act], \
        (0.2 if act == "leaky" else 0.0), 0.1
    assert lib.csg_gemm_supported(d) == 1
    rc = lib.csg_gemm_nt(d, _ptr(a), _ptr(w), _ptr(b), _ptr(gt), _ptr(y), _stream())
    assert rc == 0, last_error()
    ref = a[:, :K].double() @ w[:, :K].double().t()
    if biased:
        ref = ref + b.double()
    if act != "none":
        ref = F.leaky_relu(ref, 0.2 if act == "leaky" else 0.0)
    if gated:
        ref = ref * torch.where(gt[:, :N] > 0, 1.0, 0.1).double()
    err = (y[:, :N].double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-6 * scale + 1e-6, "max error %g at scale %g" % (err, scale)
    assert bool((y[:, N:] == 7.0).all()), "wrote outside the N columns"


TN_CASES = [
    (96000, 512, 384, True),
    (1000, 132, 36, True),            # ragged tiles, one slice
    (5000, 1152, 512, False),
    (257, 64, 1024, True),
    (31, 128, 128, True),             # less than one stage of rows
]


@pytest.mark.parametrize("case", TN_CASES)
def test_gemm_tn_vs_fp64(ops, case):
    from canonicalsg2im_amd._lib import lib, last_error
    M, N, K, with_db = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    dy = torch.randn(M, N + 8, generator=g).to(dev)
    x = torch.randn(M, K + 4, generator=g).to(dev)
    nbytes = lib.csg_gemm_tn_workspace(M, N, K)
    assert nbytes >= 0
    ws = torch.empty(max(nbytes // 4, 4), device=dev)
    dw = torch.full((N, K), 7.0, device=dev)
    db = torch.full((N,), 7.0, device=dev) if with_db else None
    runs = []
    for _ in range(2):                                     # twice: bit-reproducible
        rc = lib.csg_gemm_tn(M, N, K, _ptr(dy), N + 8, _ptr(x), K + 4, _ptr(dw), _ptr(db), _ptr(ws), nbytes, _stream())
        assert rc == 0, last_error()
        runs.append(dw.clone())
    assert torch.equal(runs[0], runs[1])
    ref = dy[:, :N].double().t() @ x[:, :K].double()
    err = (dw.double() - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item() + 1e-5, "dw: max error %g (scale %g)" % (err, ref.abs().max().item())
    if with_db:
        rb = dy[:, :N].double().sum(0)
        assert (db.double() - rb).abs().max().item() <= 1e-5 * rb.abs().max().item() + 1e-4


@pytest.mark.parametrize("shape", [(40000, 384, 512), (70000, 128, 128)])
def test_linear_chain_on_the_gemm_kernels_vs_torch(ops, shape):
    """ops.linear (what sg2im.layers.Linear calls): a Linear -> ReLU -> Linear chain with the activation derivative folded
    into the second layer's backward-data epilogue, on the GEMM kernels, against torch autograd in fp64.  Among 10^7 hidden
    units a few sit within fp32 rounding of the ReLU's kink and would be gated differently in fp64 (an O(1) difference in
    that row's gradient): the fp64 reference takes the kernels' gate decisions (tests/fp64_band.py does the same for the
    full-width steps)."""
    M, K, N = shape
    assert ops.gemm_eligible(M, N, K) and ops.gemm_eligible(M, K, N), "the shapes should take the GEMM path"
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, K, generator=g)
    w1 = torch.randn(N, K, generator=g) / K ** 0.5
    b1 = torch.randn(N, generator=g) * 0.1
    w2 = torch.randn(K, N, generator=g) / N ** 0.5
    b2 = torch.randn(K, generator=g) * 0.1
    gy = torch.randn(M, K, generator=g)
    dev = [t.clone().cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    before = ops.gemm_calls()
    hh = ops.linear(dev[0], dev[1], dev[2], ops.ACT_LEAKY, 0.0, grad_is_pre=True)
    y = ops.linear(hh, dev[3], dev[4], ops.ACT_LEAKY, 0.0, in_act=(ops.ACT_LEAKY, 0.0))
    got = torch.autograd.grad(y, dev, gy.cuda())
    assert ops.gemm_calls() - before == 6, "forward, backward-data and weight gradient of both layers on gemm.hip"
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    a1 = F.linear(ref_in[0], ref_in[1], ref_in[2])
    gate1 = (hh.detach() > 0).double().cpu()
    assert float((gate1 != (a1.detach() > 0).double()).sum()) <= 1e-5 * gate1.numel()       # the decisions differ only at the kink
    h = a1 * gate1
    a2 = F.linear(h, ref_in[3], ref_in[4])
    ref = a2 * (y.detach() > 0).double().cpu()
    ref_g = torch.autograd.grad(ref, ref_in, gy.double())
    assert (y.double().cpu() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    for name, a, b in zip(("dx", "dw1", "db1", "dw2", "db2"), got, ref_g):
        err = (a.double().cpu() - b).abs().max().item()
        assert err <= 2e-5 * b.abs().max().item() + 1e-6, "%s: %g at scale %g" % (name, err, b.abs().max().item())


def test_conv1x1_on_the_gemm_kernels_vs_torch(ops):
    """A 1x1 convolution on an NHWC map (conv_s, architecture.py:37-39: no bias) is the same product with rows = pixels."""
    B, Cin, Cout, H, W = 2, 256, 128, 128, 128
    assert ops.gemm_eligible(B * H * W, Cout, Cin)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    gy = torch.randn(B, Cout, H, W, generator=g)
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, w)]
    ref = F.conv2d(ref_in[0], ref_in[1])
    ref_g = torch.autograd.grad(ref, ref_in, gy.double())
    dev = [t.clone().cuda().requires_grad_(True) for t in (x, w)]
    before = ops.gemm_calls()
    y = ops.conv2d(dev[0], dev[1], None, 1, 0)
    got = torch.autograd.grad(y, dev, gy.cuda())
    assert ops.gemm_calls() - before == 3
    assert (y.double().cpu() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    for name, a, b in zip(("dx", "dw"), got, ref_g):
        err = (a.double().cpu() - b).abs().max().item()
        assert err <= 2e-5 * b.abs().max().item() + 1e-6, "%s: %g at scale %g" % (name, err, b.abs().max().item())


def test_default_rule_sends_the_narrow_shortcuts_to_the_gemm_kernels(ops):
    """Mode "auto" (the default): N <= 256 <= K with at least 256 tiles — conv_s of the 64 x 64 and 128 x 128 residual blocks
    at batch 16; the graph encoder's wide linears and everything small stay on the implicit-GEMM kernel."""
    old, ops.GEMM_MODE = ops.GEMM_MODE, "auto"
    try:
        assert ops.gemm_eligible(16 * 128 * 128, 128, 256) and ops.gemm_eligible(16 * 64 * 64, 256, 512)
        assert not ops.gemm_eligible(96000, 512, 384) and not ops.gemm_eligible(96000, 1152, 512)
        assert not ops.gemm_eligible(768, 128, 512) and not ops.gemm_eligible(16 * 256 * 256, 64, 128)
    finally:
        ops.GEMM_MODE = old
