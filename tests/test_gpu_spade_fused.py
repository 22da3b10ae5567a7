"""SPADE with the modulation in the gamma || beta convolution's epilogue (ops._SpadeFused, csg_wino4_conv_part) against the
unfused path (joined gamma || beta convolution + the apply pass) on the same module, weights and inputs, and against an
fp32 torch restatement of reference normalization.py:96-110 / architecture.py:50-68."""
import copy

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _block(fin, fout, S):
    from canonicalsg2im_amd.scripts.args import make_opt
    from canonicalsg2im_amd.spade.models.networks.architecture import SPADEResnetBlock
    from canonicalsg2im_amd.synth import make_vocab
    opt = make_opt(make_vocab("tiny"), ["--image_size", "64,64"], embedding_dim=S)
    torch.manual_seed(5)
    return SPADEResnetBlock(fin, fout, opt).cuda().train()


def _run(blk, x, seg, w, fused):
    from canonicalsg2im_amd import ops
    saved = ops.SPADE_FUSED
    ops.SPADE_FUSED = fused
    try:
        xd, sd = x.clone().cuda().requires_grad_(True), seg.clone().cuda().requires_grad_(True)
        y = blk(xd, sd)
        (y * w.cuda()).sum().backward()
    finally:
        ops.SPADE_FUSED = saved
    grads = {k: p.grad.detach().clone() for k, p in blk.named_parameters() if p.grad is not None}
    state = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    return y.detach(), xd.grad, sd.grad, grads, state


@pytest.mark.parametrize("shape", [(2, 64, 32, 32, 32, 8), (1, 64, 64, 16, 32, 8)])
def test_fused_block_equals_unfused(shape):
    """A residual block with (fin != fout) and without a learned shortcut: the norm_s/norm_0 pair and the single
    modulations; outputs, input gradients, every parameter gradient and the running statistics."""
    from canonicalsg2im_amd import ops
    B, fin, fout, H, W, S = shape
    blk_a = _block(fin, fout, S)
    blk_b = copy.deepcopy(blk_a)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, fin, H, W, generator=g)
    seg = torch.randn(B, S, H, W, generator=g)
    w = torch.randn(B, fout, H, W, generator=g)
    assert blk_a.norm_1.fusable(ops.nhwc(torch.empty(B, min(fin, fout), H, W, device="cuda")))
    ya, gxa, gsa, ga, sa = _run(blk_a, x, seg, w, True)
    yb, gxb, gsb, gb, sb = _run(blk_b, x, seg, w, False)
    sc = float(yb.abs().max())
    assert_close(ya, yb, RTOL, 2e-5 * sc, "block out")
    assert_close(gxa, gxb, RTOL, 2e-5 * float(gxb.abs().max()), "dx")
    assert_close(gsa, gsb, RTOL, 2e-5 * float(gsb.abs().max()), "dseg")
    assert set(ga) == set(gb)
    for k in gb:
        if k in ("conv_0.bias", "conv_1.bias"):
            # a bias in front of a BatchNorm (conv_0 feeds norm_1; conv_1 the next block's norms, here the loss): the
            # gradient of conv_0.bias is analytically zero and both sides hold rounding noise of the summed terms
            scale = float(gb[k.replace("bias", "weight_orig")].abs().max())
            if k == "conv_0.bias":
                assert float(ga[k].abs().max()) < 1e-4 * scale + 1e-4 and float(gb[k].abs().max()) < 1e-4 * scale + 1e-4
                continue
        assert_close(ga[k], gb[k], RTOL, 2e-5 * float(gb[k].abs().max()) + 1e-5, "d" + k)
    for k in sb:
        assert_close(sa[k], sb[k], RTOL, 1e-6, "state " + k)


def test_fused_spade_vs_torch():
    """One SPADE layer (param-free BatchNorm, training) with the fused path against plain torch: forward, d x, d seg and
    the gradients of the three convolutions."""
    from canonicalsg2im_amd import ops
    from canonicalsg2im_amd.spade.models.networks.normalization import SPADE
    torch.manual_seed(2)
    sp = SPADE("spadesyncbatch3x3", 64, 8).cuda().train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 32, 32, generator=g) * 2 + 0.5
    seg = torch.randn(2, 8, 32, 32, generator=g)
    w = torch.randn(2, 64, 32, 32, generator=g)
    assert sp.fusable(ops.nhwc(x.cuda()))
    xd, sd = x.clone().cuda().requires_grad_(True), seg.clone().cuda().requires_grad_(True)
    y = sp(xd, sd, fused_slope=0.2)
    (y * w.cuda()).sum().backward()
    P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in sp.named_parameters()}
    xr, sr = x.clone().double().requires_grad_(True), seg.clone().double().requires_grad_(True)
    xh = F.batch_norm(xr, None, None, None, None, True, 0.1, 1e-5)
    actv = F.relu(F.conv2d(sr, P["mlp_shared.0.weight"], P["mlp_shared.0.bias"], padding=1))
    ga = F.conv2d(actv, P["mlp_gamma.weight"], P["mlp_gamma.bias"], padding=1)
    be = F.conv2d(actv, P["mlp_beta.weight"], P["mlp_beta.bias"], padding=1)
    yr = F.leaky_relu(xh * (1 + ga) + be, 0.2)
    (yr * w.double()).sum().backward()
    assert_close(y, yr.detach().float(), RTOL, 2e-5 * float(yr.detach().abs().max()), "spade out")
    assert_close(xd.grad, xr.grad.float(), RTOL, 2e-5 * float(xr.grad.abs().max()), "dx")
    assert_close(sd.grad, sr.grad.float(), RTOL, 2e-5 * float(sr.grad.abs().max()), "dseg")
    for k, p in sp.named_parameters():
        assert_close(p.grad, P[k].grad.float(), RTOL, 2e-5 * float(P[k].grad.abs().max()) + 1e-5, "d" + k)


@pytest.mark.parametrize("shape", [(4, 256, 16, 16, 8), (2, 64, 8, 8, 8)])
def test_joined_spade_is_the_plain_path_bit_for_bit(shape):
    """ops._SpadeJoined (maps below 32 pixels: head_0, G_middle_*) only ORDERS the plain path's kernels inside one autograd
    Function (so that the SyncBN messages of N > 1 ranks travel under the gamma || beta convolution): on one rank output,
    every gradient and the running statistics are those of conv2d + norm_act, bit for bit."""
    from canonicalsg2im_amd import ops
    from canonicalsg2im_amd.spade.models.networks.normalization import SPADE
    B, C, H, W, S = shape
    torch.manual_seed(4)
    sp_a = SPADE("spadesyncbatch3x3", C, S).cuda().train()
    sp_b = copy.deepcopy(sp_a)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
    seg = torch.randn(B, S, H, W, generator=g)
    w = torch.randn(B, C, H, W, generator=g)
    assert not sp_a.fusable(ops.nhwc(x.cuda())) and sp_a.joinable(ops.nhwc(x.cuda()))
    out = []
    for sp, joined in ((sp_a, True), (sp_b, False)):
        saved, ops.SPADE_JOINED = ops.SPADE_JOINED, joined
        try:
            xd, sd = x.clone().cuda().requires_grad_(True), seg.clone().cuda().requires_grad_(True)
            y = sp(xd, sd, fused_slope=0.2)
            assert (y.grad_fn.__class__.__name__ == "_SpadeJoinedBackward") == joined, y.grad_fn
            (y * w.cuda()).sum().backward()
        finally:
            ops.SPADE_JOINED = saved
        out.append((y.detach(), xd.grad, sd.grad, {k: p.grad.clone() for k, p in sp.named_parameters()},
                    {k: v.clone() for k, v in sp.state_dict().items()}))
    (ya, gxa, gsa, ga, sa), (yb, gxb, gsb, gb, sb) = out
    assert torch.equal(ya, yb) and torch.equal(gxa, gxb) and torch.equal(gsa, gsb)
    assert set(ga) == set(gb) and all(torch.equal(ga[k], gb[k]) for k in gb), [k for k in gb if not torch.equal(ga[k], gb[k])]
    assert all(torch.equal(sa[k], sb[k]) for k in sb)
