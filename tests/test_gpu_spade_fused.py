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


# ------------------------------------------------------------------ round 6: ONE launch per modulation (csg_wino4_conv_spade)
@pytest.mark.parametrize("shape", [(8, 128, 64, 128, 128, 0.2), (16, 128, 128, 64, 64, 1.0), (6, 32, 96, 128, 128, 0.2)])
def test_joint_launch_is_bit_identical_to_the_launch_pair(shape):
    """Raw C ABI.  csg_wino4_conv_spade (blocks own a gamma tile and the beta tile of the same 32 channels; gamma written once,
    never read back) against csg_wino4_conv_part(gamma) + csg_wino4_conv_part(beta): y AND the gamma map, bit for bit; with
    gamma_out = NULL (inference) y is unchanged; a launch too small for the persistent form is refused by the query."""
    from canonicalsg2im_amd import ops
    from canonicalsg2im_amd._lib import check, lib, ptr, stream
    B, nh, C, H, W, slope = shape
    g = torch.Generator().manual_seed(3)
    actv = ops.nhwc(torch.randn(B, nh, H, W, generator=g).clamp_min(0).cuda())
    x = ops.nhwc(torch.randn(B, C, H, W, generator=g).cuda())
    w = (torch.randn(2 * C, nh, 3, 3, generator=g) / (3 * nh ** 0.5)).cuda()
    b = (0.1 * torch.randn(2 * C, generator=g)).cuda()
    mean = (0.1 * torch.randn(C, generator=g)).cuda()
    invstd = (1.0 + 0.1 * torch.rand(C, generator=g)).cuda()
    up = ops.wino_pack(w, False, None, 4)
    d = ops._wino_desc(B, H, W, nh, C)
    d.y_cs = C
    assert lib.csg_wino4_conv_spade_supported(d) == 1
    gam_a, y_a = ops.empty_nhwc(B, C, H, W, x.device), torch.empty_like(x)
    check(lib.csg_wino4_conv_part(d, ptr(actv), ptr(up), 0, 2 * C // 32, ptr(b), None, None, 0, None, None, 1.0, ptr(gam_a),
                                  stream()), "gamma")
    check(lib.csg_wino4_conv_part(d, ptr(actv), ptr(up), C // 32, 2 * C // 32, ptr(b[C:]), ptr(x), ptr(gam_a), C, ptr(mean),
                                  ptr(invstd), slope, ptr(y_a), stream()), "beta")
    gam_b, y_b = torch.full_like(gam_a, float("nan")), torch.full_like(y_a, float("nan"))
    check(lib.csg_wino4_conv_spade(d, ptr(actv), ptr(up), ptr(b), ptr(x), ptr(gam_b), C, ptr(mean), ptr(invstd), slope,
                                   ptr(y_b), stream()), "joint")
    torch.cuda.synchronize()
    assert torch.equal(gam_a, gam_b), float((gam_a - gam_b).abs().max())
    assert torch.equal(y_a, y_b), float((y_a - y_b).abs().max())
    y_c = torch.full_like(y_a, float("nan"))
    check(lib.csg_wino4_conv_spade(d, ptr(actv), ptr(up), ptr(b), ptr(x), None, 0, ptr(mean), ptr(invstd), slope, ptr(y_c),
                                   stream()), "joint, no gamma")
    torch.cuda.synchronize()
    assert torch.equal(y_a, y_c)
    # against fp64: the modulation of a direct convolution
    gb = F.conv2d(actv.double(), w.double(), b.double(), padding=1)
    xh = (x.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    ref = F.leaky_relu(xh * (1 + gb[:, :C]) + gb[:, C:], slope) if slope != 1.0 else xh * (1 + gb[:, :C]) + gb[:, C:]
    assert float((y_b.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    small = ops._wino_desc(1, 32, 32, nh, C)
    small.y_cs = C
    assert lib.csg_wino4_conv_spade_supported(small) == 0


def test_joint_and_pair_training_blocks_agree_bit_for_bit():
    """A residual block at a size where the joint launch is taken (ops.SPADE_JOINT) against the same block on the launch pair:
    outputs, every gradient and the running statistics are identical bits (the backward reads the same gamma map)."""
    from canonicalsg2im_amd import ops
    B, fin, fout, H, W, S = 8, 128, 64, 128, 128, 8
    blk_a = _block(fin, fout, S)
    blk_b = copy.deepcopy(blk_a)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, fin, H, W, generator=g)
    seg = torch.randn(B, S, H, W, generator=g)
    w = torch.randn(B, fout, H, W, generator=g)
    saved = ops.SPADE_JOINT
    try:
        ops.SPADE_JOINT = True
        ya, gxa, gsa, ga, sa = _run(blk_a, x, seg, w, True)
        ops.SPADE_JOINT = False
        yb, gxb, gsb, gb, sb = _run(blk_b, x, seg, w, True)
    finally:
        ops.SPADE_JOINT = saved
    assert torch.equal(ya, yb) and torch.equal(gxa, gxb) and torch.equal(gsa, gsb)
    assert set(ga) == set(gb) and all(torch.equal(ga[k], gb[k]) for k in gb)
    assert all(torch.equal(sa[k], sb[k]) for k in sb)
