"""Autograd operators over the C ABI of libcsg_hip.so.

Each `torch.autograd.Function` below calls hand-written gfx950 kernels for forward AND backward;
PyTorch supplies device memory, the current HIP stream and the autograd graph only.  Image-like
activations are logical (B,C,H,W) tensors whose MEMORY is NHWC (channel stride 1) — build them
with `empty_nhwc` / `nhwc`.  Nothing here runs on CPU tensors.
"""
import ctypes
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch.optim.optimizer import register_optimizer_step_post_hook as _register_optimizer_step_post_hook

from . import _lib
from . import dist as csg_dist
from ._lib import ACT_LEAKY, ACT_NONE, ACT_TANH, ConvDesc, FewDesc, GemmDesc, HingeItem, WinoDesc, WinoPackItem, check, lib, ptr, stream

__all__ = [
    "nhwc", "empty_nhwc", "conv2d", "linear", "norm_act", "upsample2x", "nearest_resize", "avgpool3s2", "embed", "real_object_mask",
    "norm_act_pair", "graph_csr", "gather_concat", "segment_avg", "layout_pyramid", "layout_paint", "disc_input", "crop_objects", "maxpool2", "avgpool2", "l1_mean",
    "invalidate_weight_caches", "pack_conv_weight", "wino_pack", "prepack_weights", "wino_eligible", "wino_variant", "spectral_weight", "spectral_weights", "ACT_NONE", "ACT_LEAKY", "ACT_TANH",
]


# ------------------------------------------------------------------------------------ layout helpers
def nhwc(t):
    """Return `t` (logical B,C,H,W) with NHWC memory, copying only if needed."""
    if t.permute(0, 2, 3, 1).is_contiguous():
        return t
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def empty_nhwc(B, C, H, W, device, zero=False):
    f = torch.zeros if zero else torch.empty
    return f((B, H, W, C), device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def _f32(t):
    if t.dtype != torch.float32:
        raise RuntimeError("canonicalsg2im_amd kernels compute in fp32; got %s" % t.dtype)
    return t


_NORM_BLOCKS = int(os.environ.get("CSG_NORM_BLOCKS", "1024"))


def _chunks(P, G=1):
    """Pixel chunks of a two-stage reduction: >= 32 pixels each, up to ~1024 blocks in flight
    (low-resolution layers have few pixels but up to 1024 channels: they need many small chunks)."""
    c = max(1, min(P // 32, max(1, _NORM_BLOCKS // max(G, 1))))
    return int(c)


# ------------------------------------------------------------------------------------ convolution
def _desc_forward(B, IH, IW, Cin, Cout, KH, KW, stride, pad, act=ACT_NONE, slope=0.0, x_cs=None, y_cs=None):
    d = ConvDesc()
    d.B, d.IHp, d.IWp, d.Cin, d.x_cs = B, IH, IW, Cin, (Cin if x_cs is None else x_cs)
    d.IHv, d.IWv, d.in_up = IH, IW, 0
    OH = (IH + 2 * pad - KH) // stride + 1
    OW = (IW + 2 * pad - KW) // stride + 1
    d.OHg, d.OWg, d.OHf, d.OWf, d.os, d.ooy, d.oox = OH, OW, OH, OW, 1, 0, 0
    d.Cout, d.y_cs = Cout, (Cout if y_cs is None else y_cs)
    d.istride, d.ntaps, d.wtaps = stride, KH * KW, KH * KW
    for ky in range(KH):
        for kx in range(KW):
            s = ky * KW + kx
            d.tap_dy[s], d.tap_dx[s], d.tap_w[s] = ky - pad, kx - pad, s
    d.act, d.slope, d.accumulate = act, slope, 0
    return d, OH, OW


def _descs_backward_data(B, IH, IW, Cin, Cout, KH, KW, stride, pad, OH, OW):
    """Descriptors that compute dX (B,IH,IW,Cin) from dY (B,OH,OW,Cout) with weights packed
    [Cin][KH*KW][Cout]: one per parity class of the input grid (a single one for stride 1)."""
    out = []
    for py in range(stride):
        for px in range(stride):
            gh = (IH - py + stride - 1) // stride
            gw = (IW - px + stride - 1) // stride
            if gh <= 0 or gw <= 0:
                continue
            d = ConvDesc()
            d.B, d.IHp, d.IWp, d.Cin, d.x_cs = B, OH, OW, Cout, Cout
            d.IHv, d.IWv, d.in_up = OH, OW, 0
            d.OHg, d.OWg, d.OHf, d.OWf, d.os, d.ooy, d.oox = gh, gw, IH, IW, stride, py, px
            d.Cout, d.y_cs = Cin, Cin
            d.istride, d.wtaps = 1, KH * KW
            n = 0
            for ky in range(KH):
                if (py + pad - ky) % stride:
                    continue
                for kx in range(KW):
                    if (px + pad - kx) % stride:
                        continue
                    d.tap_dy[n], d.tap_dx[n], d.tap_w[n] = (py + pad - ky) // stride, (px + pad - kx) // stride, ky * KW + kx
                    n += 1
            d.ntaps = n
            d.act, d.slope, d.accumulate = ACT_NONE, 0.0, 0
            out.append(d)
    return out


def _conv_launch(d, x, w, bias, res, y, what):
    """csg_conv_fwd with the split-K slabs it asks for."""
    nbytes = lib.csg_conv_fwd_workspace(d)
    if nbytes < 0:
        raise RuntimeError("conv_fwd_workspace: " + _lib.last_error())
    ws = torch.empty(nbytes // 4, device=y.device, dtype=torch.float32) if nbytes > 0 else None
    check(lib.csg_conv_fwd(d, ptr(x), ptr(w), ptr(bias), ptr(res), ptr(y), ptr(ws), nbytes, stream()), what)


def _conv_launch_classes(descs, x, w, y_ptr, dev, what):
    """The parity-class descriptors of a strided backward-data pass (same weights, input and output tensor): one
    launch for all of them (csg_conv_fwd_multi) — each class alone is a quarter of the pixels and cannot fill the chip."""
    if len(descs) == 1:
        d = descs[0]
        nbytes = lib.csg_conv_fwd_workspace(d)
        if nbytes < 0:
            raise RuntimeError("conv_fwd_workspace: " + _lib.last_error())
        ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32) if nbytes > 0 else None
        check(lib.csg_conv_fwd(d, ptr(x), ptr(w), None, None, y_ptr, ptr(ws), nbytes, stream()), what)
        return
    arr = (ConvDesc * len(descs))(*descs)
    nbytes = lib.csg_conv_fwd_multi_workspace(arr, len(descs))
    if nbytes < 0:
        raise RuntimeError("conv_fwd_multi_workspace: " + _lib.last_error())
    ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32) if nbytes > 0 else None
    check(lib.csg_conv_fwd_multi(arr, len(descs), ptr(x), ptr(w), None, None, y_ptr, ptr(ws), nbytes, stream()), what)


# ---- Winograd F(2x2,3x3) path (csrc/wino.hip): 3x3 / stride 1 / pad 1 layers with enough tiles to fill the chip
FEW_ENABLED = os.environ.get("CSG_FEW_OUTPUT_KERNELS", "1") != "0"   # csrc/fewn.hip for convolutions with <= 4 outputs
WINO_MIN_PIXELS = int(os.environ.get("CSG_WINO_MIN_PIXELS", "1024"))     # B*H*W below which the direct kernel stays
WINO_ENABLED = os.environ.get("CSG_WINOGRAD", "1") != "0"
WINO_WGRAD = os.environ.get("CSG_WINOGRAD_WGRAD", "1") != "0"
# F(3x3,4x4) weight gradient (csrc/wino4w.hip) where it pays: H, W multiples of 8 (8 x 8 maps up: 1.17-1.40x over F(3x3,2x2) on
# every generator shape at batch 4 and 16, tools/wgrad_bench.py) and at least 64 input and output channels (a block owns
# 64 x 64 channels: the 32-channel mlp_shared layers stay on F(3x3,2x2)); CSG_WINO4_WGRAD=0 turns it off
WINO4_WGRAD = os.environ.get("CSG_WINO4_WGRAD", "1") != "0"
WINO4_WGRAD_MIN_PIXELS = int(os.environ.get("CSG_WINO4_WGRAD_MIN_PIXELS", "64"))

# ---- plain GEMM kernels (csrc/gemm.hip) for the 1x1 convolutions / linears they are measured faster on
GEMM_MODE = os.environ.get("CSG_GEMM", "auto")        # "auto": the measured rule below; "all": every shape the kernels fill the
#                                                        chip with (tests, tools/gemm_bench.py); "off" / "0": never
GEMM_MIN_TILES = int(os.environ.get("CSG_GEMM_MIN_TILES", "256"))       # 128 x 128 output tiles of the forward product
_GEMM_CALLS = [0]


def gemm_calls():
    """Launches of csrc/gemm.hip so far (tests: which path served a layer)."""
    return _GEMM_CALLS[0]


def gemm_eligible(M, N, K):
    """A (M rows) x (K -> N) linear map goes to csrc/gemm.hip when its forward product has at least one 128 x 128 tile per CU
    and — mode "auto" — N <= 256 <= K: the shapes on which the dedicated kernels beat the implicit-GEMM kernel in all three
    directions (tools/gemm_bench.py on MI355X, forward / backward pair, TFLOP/s: M 96 000, 512 -> 128: 115 / 47 against
    93 / 39; M 262 144, 256 -> 128: 96 / 43 against 89 / 29; M 65 536, 512 -> 256: 116 / 51 against 109 / 51).  On the graph
    encoder's wide layers (384 -> 512 -> 1152 at 96 000 rows) the two tie — both run the same 128 x 128 x 32 MFMA tile loop at
    ~0.8 of the matrix pipe — and the implicit-GEMM kernel keeps them."""
    if GEMM_MODE in ("off", "0") or K % 4 or N % 4 or K < 32:
        return False
    if ((M + 127) // 128) * ((N + 127) // 128) < GEMM_MIN_TILES:
        return False
    return GEMM_MODE == "all" or (N <= 256 <= K)


def _gemm_nt(a, M, K, w2, N, bias, act, slope, gate, gate_slope, y):
    """y (M, N) = epi(a (M, K) . w2 (N, K)^T + bias), all dense row-major (csg_gemm_nt)."""
    d = GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldy, d.ldg = M, N, K, K, K, N, N
    d.act, d.slope, d.gate_slope = act, slope, gate_slope
    check(lib.csg_gemm_nt(d, ptr(a), ptr(w2), ptr(bias), ptr(gate), ptr(y), stream()), "gemm_nt")
    _GEMM_CALLS[0] += 1



def wino_eligible(B, H, W, Cin, Cout, KH, KW, stride, pad):
    """At least 1 024 output pixels (round 4; 4 096 before).  Launches that small are split over the input channels into
    slabs and still beat the direct kernel's split-K launch on K = 9 216: the 16 x 16 maps at batch 4 gain 1.0 ms/step
    (config C4), the 8 x 8 maps at batch 16 0.8 ms (C2) and 0.2 ms (C3), C5 1.5 ms (tools/sweep_knobs.sh).  Round 2's
    measurement that 8 x 8 maps lose 1.4 ms on Winograd predates the split of under-filled grids."""
    enough = B * H * W >= WINO_MIN_PIXELS
    return (WINO_ENABLED and KH == 3 and KW == 3 and stride == 1 and pad == 1 and H % 2 == 0 and W % 2 == 0 and W >= 8
            and Cin % 16 == 0 and Cout % 4 == 0 and Cout >= 32 and enough)


def _wino_desc(B, H, W, Cin, Cout, act=ACT_NONE, slope=0.0):
    d = WinoDesc()
    d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, W, Cin, Cin, Cout, Cout, act, slope
    return d


WINO4_MIN_ITEMS = int(os.environ.get("CSG_WINO4_MIN_ITEMS", "160"))


def wino_variant(B, H, W, Cin, Cout, plain=False):
    """4: the layer runs Winograd F(4x4,3x3) (csrc/wino4.hip: maps >= 32 wide); 2: F(2x2,3x3) (csrc/wino.hip).
    F(4x4,3x3) works on (32 x 16 pixel region, 64 output channels) items, one per CU at a time: a launch with fewer than
    ~160 of them (small batches on the 32 x 32 / 64 x 64 maps: config C4's shard of 4 images) leaves half the chip idle and
    F(2x2,3x3) — 1.78x the multiplications in 4x the blocks — is 1.2-1.9x faster (profiles/archive/r05t_variant_ab.txt: 64-128 items;
    at 192 the larger tile wins again).  `plain` (no bias / activation / residual / gate): such a launch is split over >= 256
    input channels instead and stays on F(4x4,3x3)."""
    if lib.csg_wino4_supported(_wino_desc(B, H, W, Cin, Cout)) != 1:
        return 2
    items = B * ((H + 15) // 16) * ((W + 31) // 32) * ((Cout + 63) // 64)
    if items < WINO4_MIN_ITEMS and not (plain and Cin >= 256):
        return 2
    return 4


WINO34_MODE = int(os.environ.get("CSG_WINO34_MODE", "2"))      # bit 0: forward passes, bit 1: backward-data passes (see below)


def wino34_eligible(B, H, W, Cin, Cout, KH, KW, stride, pad, backward=False):
    """4x4 / stride 1 / pad 1 or 2 layers served by Winograd F(3x3,4x4) (csrc/wino4.hip; the PatchGAN's fourth layer and
    its backward-data pass).  Default: the backward-data passes only.  The forward pass is 1.9x faster too, but its error
    (3e-6 of the output scale, the direct kernel 5e-7) sits in front of the discriminator's LeakyReLU(0.2) gates: on the
    C3 full-width step one more gate flips than in the fp32 reference and seven D gradient tensors leave the fp64 noise
    band (tests/test_gpu_fullwidth.py); a backward-data error passes no gate.  CSG_WINO34_MODE=3 turns both on."""
    return (WINO_ENABLED and (WINO34_MODE & (2 if backward else 1)) and KH == 4 and KW == 4 and stride == 1 and pad in (1, 2)
            and lib.csg_wino34_supported(_wino_desc(B, H, W, Cin, Cout), pad) == 1)


def wino_pack(weight, backward_data, sigma=None, variant=2):
    """Transformed weights U = G g G^T of a (Cout,Cin,3,3) weight in the MFMA operand order of k_wino_conv
    (variant 2: 16 positions) or k_wino4_conv (variant 4: 36 positions); variant 34: a (Cout,Cin,4,4) weight for
    F(3x3,4x4), 36 positions.  A network that called `prepack_weights` at the top of its forward finds its operands
    ready (one multi-tensor launch) — this call then hands that buffer out."""
    w = _f32(weight.detach())                    # any strides: contiguous and channels-last parameters alike
    if sigma is None and variant in (2, 4):
        ready = take_prepacked(w, backward_data, variant)
        if ready is not None:
            return ready
        _note_pack_request(w, backward_data, variant)
    Cout, Cin = w.shape[0], w.shape[1]
    N, K = (Cin, Cout) if backward_data else (Cout, Cin)
    nbytes = lib.csg_wino4_pack_bytes(N, K) if variant in (4, 34) else lib.csg_wino_pack_bytes(N, K)
    packed = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    st = w.stride()
    fn = {4: lib.csg_wino4_pack_weights, 34: lib.csg_wino34_pack_weights}.get(variant, lib.csg_wino_pack_weights)
    check(fn(ptr(w), st[0], st[1], st[2], st[3], Cout, Cin, 1 if backward_data else 0, ptr(sigma), ptr(packed), stream()),
          "wino_pack_weights")
    return packed


# ---- multi-tensor packs.  One pack costs ~7 us of fixed latency whatever the weight's size (tools/pack_bench.py) and a
# generator step needs about a hundred (forward and backward-data operand of every 3x3 convolution).  A network calls
# `prepack_weights(root)` at the top of its forward (after spectral_norm.prepare): every operand its previous passes asked
# for is produced by ONE launch per 24 weights and parked in _PREPACKED under (address, direction, variant), tagged with the
# weight's shape, strides and the weight epoch; `wino_pack` hands a parked buffer out when the tag fits the weight it is
# given.  The convolution Functions take the backward-data operand at FORWARD time and keep it in their ctx — whatever
# happens to the registry between a forward and its backward (another network's optimiser step bumps the epoch), the
# backward uses the operand of the weights its forward saw.  What a pass asks for that was not parked is packed on the spot
# (as before) and noted on the owning module for the next pass.
PREPACK = os.environ.get("CSG_PREPACK", "1") != "0"
_PREPACKED = {}          # (data_ptr, backward_data, variant) -> (packed, shape, strides, epoch)
_PACK_OWNER = {}         # data_ptr -> (weakref to module, attribute): who to note a request on


def _pack_tag(w):
    # `_version`: detach() shares the version counter with its source, so an in-place edit outside an optimiser (copy_,
    # load_state_dict, an EMA swap that forgot invalidate_weight_caches) between prepack_weights and the take shows here
    return (tuple(w.shape), tuple(w.stride()), weight_epoch(), int(w._version))


def take_prepacked(w, backward_data, variant):
    """The parked operand of `w` (a detached fp32 weight), or None.  Taken once: the registry forgets it."""
    if not _PREPACKED:
        return None
    hit = _PREPACKED.pop((w.data_ptr(), bool(backward_data), int(variant)), None)
    if hit is None:
        return None
    packed, tag = hit[0], tuple(hit[1:])
    if tag != _pack_tag(w) or packed.device != w.device:
        return None
    return packed


def _note_pack_request(w, backward_data, variant):
    owner = _PACK_OWNER.get(w.data_ptr())
    if owner is None:
        return
    mod = owner[0]()
    if mod is not None:
        mod.__dict__.setdefault("_pp_plan", {}).setdefault(owner[1], set()).add((bool(backward_data), int(variant)))


def _prepack_sources(root):
    """(module, attribute) of every 3x3 weight under `root` a convolution may be handed: `weight` of a Conv2d (the
    spectrally normalised ones through the weight spectral_norm.prepare parked), `_joined_w` of a SPADE layer."""
    cached = root.__dict__.get("_pp_srcs")
    if cached is not None and cached[1]:
        return cached[0]
    out, complete, halves = [], True, set()
    for m in root.modules():
        if hasattr(m, "mlp_gamma") and hasattr(m, "mlp_beta"):
            halves.update((id(m.mlp_gamma), id(m.mlp_beta)))      # convolved as ONE weight, never on their own
            if m.__dict__.get("_joined_w") is None and hasattr(m, "joined_weight") and m.mlp_gamma.weight.is_cuda:
                m.joined_weight()
            if m.__dict__.get("_joined_w") is not None:
                out.append((m, "_joined_w"))
            else:
                complete = False                      # the halves are joined at the layer's first call
        if isinstance(m, torch.nn.Conv2d) and tuple(m.kernel_size) == (3, 3) and id(m) not in halves:
            out.append((m, "weight"))
    root.__dict__["_pp_srcs"] = (out, complete)
    return out


def _prepack_weight_of(m, attr):
    if attr == "_joined_w":
        j = m.__dict__.get("_joined_w")
        if j is None or j.data_ptr() != m.mlp_gamma.weight.data_ptr():
            return None
        return j
    if hasattr(m, "weight_orig"):                     # spectrally normalised: W / sigma of this call, if prepared
        ready = m.__dict__.get("_sn_prepared")
        return ready[1] if (ready is not None and ready[0] == "weight") else None
    return m.weight


def prepack_weights(root):
    """Pack, in one launch per 24 weights, every Winograd operand the previous passes through `root` asked for."""
    if not PREPACK:
        return
    import weakref
    mine = root.__dict__.setdefault("_pp_keys", [])
    for k in mine:                                    # operands of the previous pass nobody took
        _PREPACKED.pop(k, None)
    del mine[:]
    owned = root.__dict__.setdefault("_pp_owned", [])
    for k in owned:                                   # addresses of the previous pass's weights (W / sigma tensors are
        _PACK_OWNER.pop(k, None)                      # new allocations every pass: the map must not grow with the steps)
    del owned[:]
    grad = torch.is_grad_enabled()
    items, keep = [], []
    for m, attr in _prepack_sources(root):
        wt = _prepack_weight_of(m, attr)
        if wt is None or not wt.is_cuda or wt.dim() != 4 or wt.dtype != torch.float32:
            continue
        w = wt.detach()
        _PACK_OWNER[w.data_ptr()] = (weakref.ref(m), attr)
        owned.append(w.data_ptr())
        plan = m.__dict__.get("_pp_plan", {}).get(attr)
        if not plan:
            continue
        Cout, Cin = w.shape[0], w.shape[1]
        st = w.stride()
        tag = _pack_tag(w)
        for backward_data, variant in sorted(plan):
            if backward_data and not grad:
                continue
            N, K = (Cin, Cout) if backward_data else (Cout, Cin)
            nbytes = lib.csg_wino4_pack_bytes(N, K) if variant == 4 else lib.csg_wino_pack_bytes(N, K)
            packed = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
            it = WinoPackItem()
            it.w, it.s_o, it.s_i, it.s_h, it.s_w = w.data_ptr(), st[0], st[1], st[2], st[3]
            it.Cout, it.Cin, it.backward_data, it.variant = Cout, Cin, 1 if backward_data else 0, variant
            it.packed = packed.data_ptr()
            items.append(it)
            key = (w.data_ptr(), backward_data, variant)
            keep.append((key, (packed,) + tag))
    if not items:
        return
    arr = (WinoPackItem * len(items))(*items)
    check(lib.csg_wino_pack_weights_multi(arr, len(items), stream()), "wino_pack_weights_multi")
    for key, val in keep:
        _PREPACKED[key] = val
        mine.append(key)


def _take_bwd_operand(weight, B, IH, IW, Cin, Cout, KH, KW, stride, pad):
    """At FORWARD time: (variant, parked backward-data operand or None) of a convolution whose backward-data pass will run
    Winograd — kept in the Function's ctx, so that the backward uses the operand of the weights its forward saw.  (That
    pass counts as plain for `wino_variant`: its only epilogue is the producer's activation derivative, which a launch split
    over the input channels leaves to a separate pass — _wino_launch.)"""
    if not _PREPACKED or not wino_eligible(B, IH, IW, Cout, Cin, KH, KW, stride, pad):
        return None
    var = wino_variant(B, IH, IW, Cout, Cin, plain=True)
    return (var, take_prepacked(_f32(weight.detach()), True, var))


WINO4_AUDIT = None     # developer audit (tools/wino4_traffic_model.py): a dict collects the ALGORITHMIC HBM bytes of the
#                        F(4x4,3x3) convolution launches (every operand once, the packed weights included) and their count


def _wino4_audit(kind, pixels, floats_per_pixel, packed):
    if WINO4_AUDIT is not None:
        a = WINO4_AUDIT.setdefault(kind, [0, 0.0])
        a[0] += 1
        a[1] += 4.0 * pixels * floats_per_pixel + 4.0 * packed.numel()


def _wino_launch(x, packed, bias, res, y, B, H, W, Cin, Cout, act, slope, what, gate=None, gate_slope=0.0, variant=2):
    """Returns True when `gate` was folded into the launch (a launch split over the input channels has no epilogue:
    the caller applies the gate in a separate pass then)."""
    d = _wino_desc(B, H, W, Cin, Cout, act, slope)
    if variant == 4:
        _wino4_audit(what, B * H * W, Cin + Cout + (Cout if res is not None else 0) + (Cout if gate is not None else 0), packed)
    ws_fn, conv_fn = (lib.csg_wino4_conv_workspace, lib.csg_wino4_conv) if variant == 4 else \
        (lib.csg_wino_conv_workspace, lib.csg_wino_conv)
    ws, nws = None, 0
    if bias is None and res is None and act == ACT_NONE:
        nws = ws_fn(d)
        if nws > 0:
            ws = torch.empty(nws // 4, device=x.device, dtype=torch.float32)
            gate = None
    check(conv_fn(d, ptr(x), ptr(packed), ptr(bias), ptr(res), ptr(gate), gate_slope, ptr(y), ptr(ws), nws, stream()), what)
    return gate is not None


# Caches derived from trainable weights are keyed on (weight_epoch(), tensor._version): the fused multi-tensor Adam updates
# parameters WITHOUT bumping `_version` (checked on torch 2.10), so every optimiser step — of any optimiser, the
# reference trainer's own included — advances a global counter through torch's optimiser post-step hook; in-place edits
# outside an optimiser (load_state_dict, copy_) still show in `_version`.
_WEIGHT_EPOCH = [0]


def _bump_weight_epoch(*_args, **_kwargs):
    _WEIGHT_EPOCH[0] += 1


_register_optimizer_step_post_hook(_bump_weight_epoch)


def weight_epoch():
    return _WEIGHT_EPOCH[0]


def invalidate_weight_caches():
    """Call after editing parameters in place OUTSIDE an optimiser step through `.data` (a broadcast, an EMA swap,
    weight clipping): such edits bump neither `_version` nor the optimiser hook, so tensors derived from the weights
    (Winograd operands, the PatchGAN's permuted first-layer weight) would otherwise stay stale until the next step.
    `dist.broadcast_module` and `Trainer.load_checkpoint` call it.  Parked Winograd operands are dropped as well (they
    would be refused by their epoch tag anyway; dropping them frees their memory now)."""
    _bump_weight_epoch()
    _PREPACKED.clear()


# ---- gradient destinations (N > 1 ranks): `dist.GradBuckets` registers, for the backward that is about to run, where each
# parameter's gradient has to end up — its slot in a flat all-reduce bucket, laid out like the parameter's own memory.
# The weight-gradient producers below (convolution / linear, spectral normalisation, the joined gamma || beta convolution)
# then write there directly and hand autograd an alias of the slot, so `.grad` IS the bucket slot without a copy.  Only the
# FIRST contribution of a backward takes the slot (a discriminator weight used by two passes accumulates its second
# contribution into it through autograd's in-place add).  Empty on a single rank.
_GRAD_DEST = {}


def set_grad_destinations(mapping):
    """mapping: (data_ptr, numel) of a weight tensor as its producer sees it -> 1-D slot of that many floats."""
    global _GRAD_DEST
    _GRAD_DEST = {k: [v, False] for k, v in mapping.items()}


def clear_grad_destinations():
    global _GRAD_DEST
    _GRAD_DEST = {}


def _grad_dest(weight, shape):
    """A fresh tensor of `shape` (row-major) aliasing the registered slot of `weight`, once per registration; else None."""
    if not _GRAD_DEST:
        return None
    e = _GRAD_DEST.get((weight.data_ptr(), weight.numel()))
    if e is None or e[1]:
        return None
    e[1] = True
    return e[0].view(shape)


def _ohwi_dense(w):
    """True if the memory of the (Cout, Cin, KH, KW) tensor `w` is [Cout][KH][KW][Cin] dense — what the weight-gradient
    kernels write (channels-last parameters, 1 x 1 kernels, linears)."""
    Cout, Cin, KH, KW = w.shape
    if KH * KW == 1:
        return w.stride(0) == Cin and (Cin == 1 or w.stride(1) == 1)
    return w.stride() == (KH * KW * Cin, 1, KW * Cin, Cin)


def _few_desc(B, IH, IW, Cin, KH, KW, stride, pad, cout_real, act, slope):
    """Descriptor of csrc/fewn.hip if this convolution is one it serves (<= 4 outputs, stride 1, 3x3 or 4x4), else None."""
    if not FEW_ENABLED or cout_real is None or stride != 1 or KH != KW:
        return None
    d = FewDesc()
    d.B, d.IH, d.IW, d.Cin, d.x_cs, d.KH, d.KW, d.pad, d.cout_real, d.act, d.slope = \
        B, IH, IW, Cin, Cin, KH, KW, pad, cout_real, act, slope
    return d if lib.csg_conv_few_supported(d) == 1 else None


class _Conv2d(torch.autograd.Function):
    """y = act(conv2d(x, w) + b) [+ residual] — reference nn.Conv2d call sites listed in
    include/csg_hip.h (K8/K11)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, pad, act, slope, packs=None, dx_range=None, in_act=None,
                grad_is_pre=False, cout_real=None, pre_slope=None):
        """`in_act=(act, slope)`: x is the output of that activation and THIS call is its only consumer — the
        backward returns dx already multiplied by act'(x), i.e. the gradient of the producer's pre-activation (folded
        into the Winograd backward-data epilogue).  `grad_is_pre=True` is the producer's half of the pair: its incoming
        gradient needs no activation derivative any more."""
        x = nhwc(_f32(x))
        B, Cin, IH, IW = x.shape
        Cout, Cin_w, KH, KW = weight.shape
        if packs is not None:
            Cin_w = packs[0].shape[3]              # packed weights carry their own channel padding
        if Cin_w != Cin:
            raise RuntimeError("conv2d: weight expects %d input channels, x has %d" % (Cin_w, Cin))
        ctx.packs, ctx.dx_range, ctx.in_act, ctx.grad_is_pre = packs, dx_range, in_act, grad_is_pre
        res = nhwc(residual) if residual is not None else None
        ctx.few = None
        ctx.gemm = False
        ctx.cout_w = Cout                       # rows of the weight as given (1..3 when the few-output path takes it raw)
        if (Cout == 4 or (cout_real is not None and Cout == cout_real and Cout < 4)) and res is None and dx_range is None \
                and packs is None and in_act is None:
            ctx.few = _few_desc(B, IH, IW, Cin, KH, KW, stride, pad, cout_real, act, slope)
        ctx.pre_slope = pre_slope
        if pre_slope is not None:
            # `pre_slope`: the convolution sees leaky(x, pre_slope) — applied in the few-output kernels' loaders (conv_img)
            if ctx.few is None:
                raise RuntimeError("conv2d: pre_slope is served by the few-output kernels only (the caller applies the "
                                   "activation itself otherwise)")
            ctx.few.in_act, ctx.few.in_slope = 1, float(pre_slope)
        if ctx.few is None and Cout % 4:
            raise RuntimeError("conv2d: an output-channel count that is not a multiple of 4 reached the kernels")
        if ctx.few is not None:
            Cout = 4                            # the kernels' padded output width
            OH, OW = IH + 2 * pad - KH + 1, IW + 2 * pad - KW + 1
            y = empty_nhwc(B, Cout, OH, OW, x.device)
            wp = weight.detach().permute(0, 2, 3, 1).contiguous()
            nws = lib.csg_conv_few_fwd_workspace(ctx.few)
            ws = torch.empty(nws // 4, device=x.device, dtype=torch.float32) if nws > 0 else None
            check(lib.csg_conv_few_fwd(ctx.few, ptr(x), ptr(wp), ptr(bias.detach() if bias is not None else None), ptr(y),
                                       ptr(ws), nws, stream()), "conv_few_fwd")
        elif dx_range is None and wino_eligible(B, IH, IW, Cin, Cout, KH, KW, stride, pad):
            OH, OW = IH, IW
            y = empty_nhwc(B, Cout, OH, OW, x.device)
            var = wino_variant(B, IH, IW, Cin, Cout, plain=bias is None and res is None and act == ACT_NONE)
            up = _frozen_pack(packs, False, var) if (packs is not None and len(packs) > 2) else wino_pack(weight, False, None, var)
            if packs is None and ctx.needs_input_grad[0]:
                ctx.ut_pre = _take_bwd_operand(weight, B, IH, IW, Cin, Cout, KH, KW, stride, pad)
            _wino_launch(x, up, bias.detach() if bias is not None else None, res, y, B, IH, IW, Cin, Cout, act, slope,
                         "wino_conv_fwd", variant=var)
        elif dx_range is None and packs is None and wino34_eligible(
                B, IH, IW, Cin, Cout, KH, KW, stride, pad, backward=not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])):
            # (a forward pass nothing is differentiated through — the discriminator on the real images in the generator
            # step, inference — counts as a backward-class pass: its error meets no gate of a backward pass)
            OH, OW = IH + 2 * pad - 3, IW + 2 * pad - 3
            y = empty_nhwc(B, Cout, OH, OW, x.device)
            check(lib.csg_wino34_conv(_wino_desc(B, IH, IW, Cin, Cout, act, slope), pad, ptr(x),
                                      ptr(wino_pack(weight, False, None, 34)), ptr(bias.detach() if bias is not None else None),
                                      ptr(res), None, 0.0, ptr(y), None, 0, stream()), "wino34_conv_fwd")
        elif (KH == 1 and KW == 1 and stride == 1 and pad == 0 and res is None and packs is None and dx_range is None
              and act in (ACT_NONE, ACT_LEAKY) and gemm_eligible(B * IH * IW, Cout, Cin)):
            # a matrix product: rows = pixels (or the rows of a Linear), csrc/gemm.hip
            OH, OW = IH, IW
            y = empty_nhwc(B, Cout, OH, OW, x.device)
            w2 = weight.detach().reshape(Cout, Cin)
            _gemm_nt(x, B * IH * IW, Cin, w2 if w2.is_contiguous() else w2.contiguous(), Cout,
                     bias.detach() if bias is not None else None, act, slope, None, 0.0, y)
            ctx.gemm = True
        else:
            # [Cout][KH][KW][Cin]: free for channels-last parameters (sg2im.layers.Conv2d keeps them that way)
            wp = packs[0] if packs is not None else weight.detach().permute(0, 2, 3, 1).contiguous()
            d, OH, OW = _desc_forward(B, IH, IW, Cin, Cout, KH, KW, stride, pad, act, slope)
            y = empty_nhwc(B, Cout, OH, OW, x.device)
            _conv_launch(d, x, wp, bias.detach() if bias is not None else None, res, y, "conv_fwd")
        ctx.geom = (B, IH, IW, Cin, Cout, KH, KW, stride, pad, OH, OW, act, slope)
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        B, IH, IW, Cin, Cout, KH, KW, stride, pad, OH, OW, act, slope = ctx.geom
        dy = nhwc(dy)
        gated = ctx.in_act is None
        if act != ACT_NONE and not ctx.grad_is_pre:
            dpre = torch.empty_like(dy)
            check(lib.csg_act_bwd(ptr(dy), ptr(y), dy.numel(), act, slope, ptr(dpre), stream()), "act_bwd")
        else:
            dpre = dy
        dx = dw = db = dres = None
        if ctx.few is not None:
            if ctx.needs_input_grad[0] and Cin >= 256:
                if ctx.pre_slope is not None:
                    raise RuntimeError("conv2d: pre_slope with >= 256 input channels is not served")
                wp = weight.detach().permute(0, 2, 3, 1).contiguous()
                dx = empty_nhwc(B, Cin, IH, IW, dy.device)
                check(lib.csg_conv_few_bwd_data(ctx.few, ptr(dpre), ptr(wp), ptr(dx), stream()), "conv_few_bwd_data")
            elif ctx.needs_input_grad[0]:
                # few input channels as well (conv_img: 64): a (pixels x 36) x (36 x 64) product, fine on the matrix cores
                w4 = weight.detach() if ctx.cout_w == 4 else F.pad(weight.detach(), (0, 0, 0, 0, 0, 0, 0, 4 - ctx.cout_w))
                wt = w4.permute(1, 2, 3, 0).contiguous()                    # [Cin][KH][KW][Cout]
                dx = empty_nhwc(B, Cin, IH, IW, dy.device)
                for d in _descs_backward_data(B, IH, IW, Cin, Cout, KH, KW, stride, pad, OH, OW):
                    if ctx.pre_slope is not None:            # d leaky(x) / dx in the epilogue: x gates through the residual slot
                        d.res_gate, d.slope = 1, ctx.pre_slope
                        _conv_launch(d, dpre, wt, None, x, dx, "conv_bwd_data")
                    else:
                        _conv_launch(d, dpre, wt, None, None, dx, "conv_bwd_data")
            if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
                nbytes = lib.csg_conv_few_bwd_weight_workspace(ctx.few)
                ws = torch.empty(nbytes // 4, device=dy.device, dtype=torch.float32)
                dwp = torch.empty((Cout, KH, KW, Cin), device=dy.device, dtype=torch.float32)
                db = torch.empty(Cout, device=dy.device, dtype=torch.float32) if ctx.has_bias else None
                check(lib.csg_conv_few_bwd_weight(ctx.few, ptr(x), ptr(dpre), ptr(dwp), ptr(db), ptr(ws), nbytes, stream()),
                      "conv_few_bwd_weight")
                dw = dwp[:ctx.cout_w].permute(0, 3, 1, 2)
                db = db[:ctx.cout_w] if db is not None else None
            return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None
        if ctx.needs_input_grad[0] and ctx.dx_range is not None:
            # only input channels [lo, hi) are wanted by the consumer of dx (the discriminator's packed
            # [layout | img | pad] input: the image part in the generator pass, the layout part in the
            # discriminator passes): the transposed convolution runs over that slice of the weight only
            lo, hi = ctx.dx_range
            wt = weight.detach()[:, lo:hi].permute(1, 2, 3, 0).contiguous()      # [hi-lo][KH][KW][Cout]
            dx = empty_nhwc(B, Cin, IH, IW, dy.device, zero=True)
            descs = _descs_backward_data(B, IH, IW, hi - lo, Cout, KH, KW, stride, pad, OH, OW)
            for d in descs:
                d.y_cs = Cin
            _conv_launch_classes(descs, dpre, wt, ctypes_ptr_off(dx, lo), dy.device, "conv_bwd_data")
        elif ctx.needs_input_grad[0] and wino_eligible(B, IH, IW, Cout, Cin, KH, KW, stride, pad):
            # dX = conv3x3(dY, flipped W^T): the same Winograd kernel with the roles of the channel counts swapped
            var = wino_variant(B, IH, IW, Cout, Cin, plain=True)
            pre = getattr(ctx, "ut_pre", None)           # parked at forward time (prepack_weights)
            if ctx.packs is not None and len(ctx.packs) > 3:
                ut = _frozen_pack(ctx.packs, True, var)
            elif pre is not None and pre[1] is not None and pre[0] == var:
                ut = pre[1]
            else:
                ut = wino_pack(weight, True, None, var)
            dx = empty_nhwc(B, Cin, IH, IW, dy.device)
            gated = _wino_launch(dpre, ut, None, None, dx, B, IH, IW, Cout, Cin, ACT_NONE, 0.0, "wino_conv_bwd_data",
                                 gate=x if ctx.in_act is not None else None,
                                 gate_slope=ctx.in_act[1] if ctx.in_act is not None else 0.0, variant=var) or ctx.in_act is None
        elif ctx.needs_input_grad[0] and ctx.packs is None and wino34_eligible(B, OH, OW, Cout, Cin, KH, KW, stride, 3 - pad,
                                                                                    backward=True):
            # dX = conv4x4(dY, flipped W^T) with padding 3 - pad: the same F(3x3,4x4) kernel, channel counts swapped
            dx = empty_nhwc(B, Cin, IH, IW, dy.device)
            has_gate = ctx.in_act is not None
            d34 = _wino_desc(B, OH, OW, Cout, Cin)
            ws, nws = None, 0
            if not has_gate:                     # a small tile grid is split over the input channels (slabs + ordered sum)
                nws = lib.csg_wino34_conv_workspace(d34, 3 - pad)
                if nws > 0:
                    ws = torch.empty(nws // 4, device=dy.device, dtype=torch.float32)
            check(lib.csg_wino34_conv(d34, 3 - pad, ptr(dpre), ptr(wino_pack(weight, True, None, 34)), None, None,
                                      ptr(x) if has_gate else None, ctx.in_act[1] if has_gate else 0.0, ptr(dx), ptr(ws), nws,
                                      stream()), "wino34_conv_bwd_data")
            gated = True
        elif ctx.needs_input_grad[0] and getattr(ctx, "gemm", False):
            # dX (M, Cin) = dY (M, Cout) . W^T stored [Cin][Cout] (the reduction index contiguous), the producer's
            # activation derivative as the epilogue's gate
            wt = weight.detach().reshape(Cout, Cin).t().contiguous()
            dx = empty_nhwc(B, Cin, IH, IW, dy.device)
            has_gate = ctx.in_act is not None
            _gemm_nt(dpre, B * IH * IW, Cout, wt, Cin, None, ACT_NONE, 0.0, x if has_gate else None,
                     ctx.in_act[1] if has_gate else 0.0, dx)
            gated = True
        elif ctx.needs_input_grad[0]:
            wt = ctx.packs[1] if ctx.packs is not None else \
                weight.detach().permute(1, 2, 3, 0).contiguous()       # [Cin][KH][KW][Cout]
            dx = empty_nhwc(B, Cin, IH, IW, dy.device)
            descs = _descs_backward_data(B, IH, IW, Cin, Cout, KH, KW, stride, pad, OH, OW)
            if ctx.in_act is not None and len(descs) == 1:
                # the producer's (Leaky)ReLU derivative in the epilogue: x (its output) rides in the residual slot as a gate
                d = descs[0]
                d.res_gate, d.slope = 1, ctx.in_act[1]
                _conv_launch(d, dpre, wt, None, x, dx, "conv_bwd_data")
                gated = True
            else:
                _conv_launch_classes(descs, dpre, wt, ptr(dx), dy.device, "conv_bwd_data")
        if dx is not None and not gated:              # the producer's activation derivative as a separate pass
            check(lib.csg_act_bwd(ptr(dx), ptr(x), dx.numel(), ctx.in_act[0], ctx.in_act[1], ptr(dx), stream()), "act_bwd")
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        wino_wg = wino4_wg = -1
        if ctx.needs_input_grad[1] and WINO_WGRAD and wino_eligible(B, IH, IW, Cin, Cout, KH, KW, stride, pad):
            d = WinoDesc()
            d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, IH, IW, Cin, Cin, Cout, Cout, ACT_NONE, 0.0
            wino_wg = lib.csg_wino_bwd_weight_workspace(d)      # < 0: the 16-tile stages do not tile this image exactly
            if (WINO4_WGRAD and Cin >= 64 and Cout >= 64 and IH % 8 == 0 and IW % 8 == 0
                    and IH * IW >= WINO4_WGRAD_MIN_PIXELS):
                wino4_wg = lib.csg_wino4_bwd_weight_workspace(d)
        if ctx.needs_input_grad[1] and getattr(ctx, "gemm", False):
            # dW (Cout, Cin) = dY^T . X over the rows, the bias gradient from the same staged tiles (csg_gemm_tn)
            M = B * IH * IW
            nbytes = lib.csg_gemm_tn_workspace(M, Cout, Cin)
            if nbytes < 0:
                raise RuntimeError("gemm_tn_workspace: (%d, %d, %d) not served" % (M, Cout, Cin))
            ws = torch.empty(max(nbytes // 4, 4), device=dy.device, dtype=torch.float32)
            dwp = _grad_dest(weight, (Cout, KH, KW, Cin)) if _ohwi_dense(weight) else None
            if dwp is None:
                dwp = torch.empty((Cout, KH, KW, Cin), device=dy.device, dtype=torch.float32)
            if want_db:
                db = torch.empty(Cout, device=dy.device, dtype=torch.float32)
            check(lib.csg_gemm_tn(M, Cout, Cin, ptr(dpre), Cout, ptr(x), Cin, ptr(dwp), ptr(db), ptr(ws), nbytes, stream()),
                  "gemm_tn")
            _GEMM_CALLS[0] += 1
            dw = dwp.permute(0, 3, 1, 2)
        elif wino4_wg >= 0:
            # Winograd F(3x3,4x4) weight gradient (csrc/wino4w.hip), same output layout as the direct kernel
            nbytes = wino4_wg
            ws = torch.empty(nbytes // 4, device=dy.device, dtype=torch.float32)
            dwp = _grad_dest(weight, (Cout, KH, KW, Cin)) if _ohwi_dense(weight) else None
            if dwp is None:
                dwp = torch.empty((Cout, KH, KW, Cin), device=dy.device, dtype=torch.float32)
            if want_db:
                db = torch.empty(Cout, device=dy.device, dtype=torch.float32)
            check(lib.csg_wino4_bwd_weight(d, ptr(x), ptr(dpre), ptr(dwp), ptr(db), ptr(ws), nbytes, stream()),
                  "wino4_bwd_weight")
            dw = dwp.permute(0, 3, 1, 2)
        elif wino_wg >= 0:
            # Winograd F(3x3,2x2) weight gradient (csrc/wino.hip), same output layout as the direct kernel
            nbytes = wino_wg
            ws = torch.empty(nbytes // 4, device=dy.device, dtype=torch.float32)
            dwp = _grad_dest(weight, (Cout, KH, KW, Cin)) if _ohwi_dense(weight) else None
            if dwp is None:
                dwp = torch.empty((Cout, KH, KW, Cin), device=dy.device, dtype=torch.float32)
            if want_db:
                db = torch.empty(Cout, device=dy.device, dtype=torch.float32)
            check(lib.csg_wino_bwd_weight(d, ptr(x), ptr(dpre), ptr(dwp), ptr(db), ptr(ws), nbytes, stream()),
                  "wino_bwd_weight")
            dw = dwp.permute(0, 3, 1, 2)
        elif ctx.needs_input_grad[1]:
            d, _, _ = _desc_forward(B, IH, IW, Cin, Cout, KH, KW, stride, pad)
            nbytes = lib.csg_conv_bwd_weight_workspace(d)
            if nbytes < 0:
                raise RuntimeError("conv_bwd_weight_workspace: " + _lib.last_error())
            ws = torch.empty(max(nbytes // 4, 4), device=dy.device, dtype=torch.float32)
            dwp = _grad_dest(weight, (Cout, KH, KW, Cin)) if _ohwi_dense(weight) else None
            if dwp is None:
                dwp = torch.empty((Cout, KH, KW, Cin), device=dy.device, dtype=torch.float32)
            if want_db:                                   # column sums of dY ride along in the same kernel
                db = torch.empty(Cout, device=dy.device, dtype=torch.float32)
            check(lib.csg_conv_bwd_weight(d, ptr(x), ptr(dpre), ptr(dwp), ptr(db), ptr(ws), nbytes, stream()),
                  "conv_bwd_weight")
            dw = dwp.permute(0, 3, 1, 2)
        elif want_db:
            rows = B * OH * OW
            nch = _chunks(rows)
            part = torch.empty(nch * 2 * Cout, device=dy.device, dtype=torch.float64)
            db = torch.empty(Cout, device=dy.device, dtype=torch.float32)
            check(lib.csg_colsum(ptr(dpre), rows, Cout, Cout, ptr(db), ptr(part), nch, stream()), "colsum")
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy                            # the residual is added AFTER the activation (igemm.hip epilogue)
        return dx, dw, db, dres, None, None, None, None, None, None, None, None, None, None


def pack_conv_weight(weight):
    """The kernel-side layouts of a FROZEN (Cout,Cin,KH,KW) weight, for `conv2d(..., packs=)`:
    [Cout][KH][KW][Cin] for the forward and [Cin][KH][KW][Cout] for backward-data, input channels
    zero-padded to a multiple of 4 — plus, for 3x3 weights, the two Winograd operands (csrc/wino.hip).
    Trainable weights are repacked per call instead."""
    pc = (-weight.shape[1]) % 4
    w = F.pad(weight.detach(), (0, 0, 0, 0, 0, pc)) if pc else weight.detach()
    out = (w.permute(0, 2, 3, 1).contiguous(), w.permute(1, 2, 3, 0).contiguous())
    if WINO_ENABLED and w.shape[2] == 3 and w.shape[3] == 3 and w.shape[0] % 4 == 0:
        # F(2x2,3x3) operands now; the F(4x4,3x3) ones (maps >= 32 wide) are added on first use (_frozen_pack)
        out = out + (wino_pack(w, False), wino_pack(w, True), {"w": w})
    return out


def _frozen_pack(packs, backward_data, variant):
    """The Winograd operand of a frozen weight for the kernel variant a call site runs (packs from pack_conv_weight)."""
    if variant == 2:
        return packs[3 if backward_data else 2]
    cache = packs[4]
    key = ("bwd" if backward_data else "fwd", variant)
    if key not in cache:
        cache[key] = wino_pack(cache["w"], backward_data, None, variant)
    return cache[key]


def conv2d(x, weight, bias=None, stride=1, padding=0, act=ACT_NONE, slope=0.0, residual=None, packs=None,
           dx_range=None, in_act=None, grad_is_pre=False, pre_slope=None):
    """Channel counts that are not multiples of 4 (conv_img: 3 outputs, the PatchGAN head: 1) are
    zero-padded to 16-byte pixel rows; the result is a channel-slice view of the padded output."""
    Cout, Cin = weight.shape[0], weight.shape[1]
    pc, po = (-Cin) % 4, (-Cout) % 4
    if residual is not None and act != ACT_NONE:
        # the epilogue adds the residual AFTER the activation and the backward recovers the activation mask from the
        # saved output, which would then include the residual; the only user (SPADEResnetBlock.conv_1) has no activation
        raise NotImplementedError("conv2d: a fused residual needs act == ACT_NONE")
    if packs is not None and po:
        raise RuntimeError("conv2d: packed weights need an output-channel count that is a multiple of 4")
    if pc:
        if x.shape[1] == Cin:                  # an already padded input (e.g. the 4-channel object crops) is used as is
            x = F.pad(x, (0, 0, 0, 0, 0, pc))
        if packs is None:
            weight = F.pad(weight, (0, 0, 0, 0, 0, pc))
    few_raw = False
    if (po and Cout + po == 4 and not pc and residual is None and packs is None and dx_range is None and in_act is None
            and x.dim() == 4 and x.is_cuda):
        # csrc/fewn.hip takes the 1..3 real output channels as they are (no zero-padded copies of weight and bias)
        few_raw = _few_desc(x.shape[0], x.shape[2], x.shape[3], Cin, weight.shape[2], weight.shape[3], int(stride),
                            int(padding), Cout, int(act), float(slope)) is not None
    if pre_slope is not None and not (few_raw and x.shape[1] < 256):
        x, pre_slope = F.leaky_relu(x, float(pre_slope)), None      # no loader to fold it into: a pass of its own
    if po and not few_raw:
        weight = F.pad(weight, (0, 0, 0, 0, 0, 0, 0, po))
        bias = F.pad(bias, (0, po)) if bias is not None else None
        residual = F.pad(residual, (0, 0, 0, 0, 0, po)) if residual is not None else None
    if dx_range is not None and (dx_range[0] % 4 or dx_range[1] % 4 or pc):
        raise RuntimeError("conv2d: dx_range must be 4-aligned channel bounds of an unpadded input")
    if in_act is not None and (pc or dx_range is not None or in_act[0] != ACT_LEAKY):
        raise RuntimeError("conv2d: in_act needs an unpadded input, no dx_range and a (Leaky)ReLU producer")
    if grad_is_pre and po:
        raise RuntimeError("conv2d: grad_is_pre needs an unpadded output")
    y = _Conv2d.apply(x, weight, bias, residual, int(stride), int(padding), int(act), float(slope), packs, dx_range,
                      in_act, bool(grad_is_pre), Cout if Cout + po == 4 else None,
                      None if pre_slope is None else float(pre_slope))
    return y[:, :Cout] if po else y


def linear(x, weight, bias=None, act=ACT_NONE, slope=0.0, in_act=None, grad_is_pre=False):
    """F.linear (+ReLU) as a 1x1 implicit GEMM: rows of x are 'pixels'.  `in_act` / `grad_is_pre`: as conv2d's (a Linear ->
    ReLU -> Linear chain: the second Linear's backward-data pass applies the ReLU derivative in its epilogue and the first
    receives the gradient of its pre-activation directly)."""
    lead = x.shape[:-1]
    K = x.shape[-1]
    N = weight.shape[0]
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = conv2d(x2.view(-1, K, 1, 1), weight.view(N, K, 1, 1), bias, 1, 0, act, slope, in_act=in_act, grad_is_pre=grad_is_pre)
    return y.reshape(*lead, N)


class _SpectralWeight(torch.autograd.Function):
    """W_eff = W / sigma(W, u, v) with torch.nn.utils.spectral_norm's power iteration (one step, in place on
    the u / v buffers) — csrc/spectral.hip.  u and v are constants of the graph, as in PyTorch."""

    @staticmethod
    def forward(ctx, w, u, v, iterate, eps):
        w = _f32(w).contiguous()
        Cout = w.shape[0]
        K = w.numel() // Cout
        nbytes = lib.csg_spectral_norm_workspace(Cout, K)
        if nbytes < 0:
            raise RuntimeError("spectral_weight: weight (%d x %d) needs K %% 4 == 0" % (Cout, K))
        dev = w.device
        ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        # 4-D weights come out in channels-last memory: the convolution kernels' forward operand, no repack per call
        cl = w.shape[1] if (w.dim() == 4 and w.shape[1] % 4 == 0 and w.shape[2] * w.shape[3] > 1
                           and w.shape[2] * w.shape[3] * (w.shape[1] + 4) * 4 <= 65536) else 0
        w_eff = torch.empty_like(w, memory_format=torch.channels_last) if cl else torch.empty_like(w)
        small = torch.empty(1 + Cout + K, device=dev, dtype=torch.float32)      # sigma | u_used | v_used
        sigma, u_used, v_used = small[:1], small[1:1 + Cout], small[1 + Cout:]
        check(lib.csg_spectral_norm_fwd(ptr(w), ptr(u), ptr(v), Cout, K, 1 if iterate else 0, eps, ptr(w_eff), cl,
                                        ptr(sigma), ptr(u_used), ptr(v_used), ptr(ws), nbytes, stream()),
              "spectral_norm_fwd")
        ctx.save_for_backward(w, small)                 # u, v are buffers (no grad): updated in place by the kernel
        return w_eff

    @staticmethod
    def backward(ctx, dweff):
        w, small = ctx.saved_tensors
        Cout = w.shape[0]
        K = w.numel() // Cout
        Cin, KH, KW = (w.shape[1], w.shape[2], w.shape[3]) if w.dim() == 4 else (K, 1, 1)
        dweff = _f32(dweff)
        st = dweff.stride() if dweff.dim() == 4 else (K, 1, 1, 1)
        if not _rows_dense(st, (Cin, KH, KW), K):       # e.g. a slice of a channel-padded gradient
            dweff = dweff.contiguous()
            st = dweff.stride() if dweff.dim() == 4 else (K, 1, 1, 1)
        sigma, u_used, v_used = small[:1], small[1:1 + Cout], small[1 + Cout:]
        nbytes = lib.csg_spectral_norm_workspace(Cout, K)
        ws = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
        dw = _grad_dest(w, tuple(w.shape))           # w is contiguous (rows of W.view(Cout, -1)): the slot has its order
        if dw is None:
            dw = torch.empty_like(w)
        check(lib.csg_spectral_norm_bwd(ptr(dweff), Cout, Cin, KH, KW, st[0], st[1], st[2], st[3], ptr(w), ptr(u_used),
                                        ptr(v_used), ptr(sigma), ptr(dw), ptr(ws), nbytes, stream()),
              "spectral_norm_bwd")
        return dw, None, None, None, None


class _SpectralWeightMulti(torch.autograd.Function):
    """`_SpectralWeight` for every spectrally normalised weight of a network pass at once (csg_spectral_norm_*_multi): one
    launch per stage instead of one per weight and stage — a generator forward has 18 such weights, a PatchGAN pass 3 per
    scale, ~200 launches of a few microseconds each per training step.  Same arithmetic, bit-identical per weight."""

    @staticmethod
    def forward(ctx, iterate, eps, *wuv):
        n = len(wuv) // 3
        ws = [_f32(wuv[3 * i]).contiguous() for i in range(n)]
        items = (_lib.SnFwdItem * n)()
        outs, smalls, keep = [], [], []
        for i, w in enumerate(ws):
            u, v = wuv[3 * i + 1], wuv[3 * i + 2]
            Cout = w.shape[0]
            K = w.numel() // Cout
            nbytes = lib.csg_spectral_norm_workspace(Cout, K)
            if nbytes < 0:
                raise RuntimeError("spectral_weight: weight (%d x %d) needs K %% 4 == 0" % (Cout, K))
            wsb = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
            cl = w.shape[1] if (w.dim() == 4 and w.shape[1] % 4 == 0 and w.shape[2] * w.shape[3] > 1
                               and w.shape[2] * w.shape[3] * (w.shape[1] + 4) * 4 <= 65536) else 0
            w_eff = torch.empty_like(w, memory_format=torch.channels_last) if cl else torch.empty_like(w)
            small = torch.empty(1 + Cout + K, device=w.device, dtype=torch.float32)      # sigma | u_used | v_used
            it = items[i]
            it.w, it.u, it.v, it.Cout, it.K = w.data_ptr(), u.data_ptr(), v.data_ptr(), Cout, K
            it.w_eff, it.cl_Cin = w_eff.data_ptr(), cl
            it.sigma, it.u_used, it.v_used = small.data_ptr(), small.data_ptr() + 4, small.data_ptr() + 4 * (1 + Cout)
            it.workspace, it.workspace_bytes = wsb.data_ptr(), nbytes
            outs.append(w_eff)
            smalls.append(small)
            keep.append(wsb)
        check(lib.csg_spectral_norm_fwd_multi(items, n, 1 if iterate else 0, eps, stream()), "spectral_norm_fwd_multi")
        ctx.n = n
        ctx.save_for_backward(*ws, *smalls)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dweffs):
        n = ctx.n
        ws, smalls = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        idx = [i for i in range(n) if dweffs[i] is not None and ctx.needs_input_grad[2 + 3 * i]]
        grads = [None] * (2 + 3 * n)
        if not idx:
            return tuple(grads)
        items = (_lib.SnBwdItem * len(idx))()
        keep = []
        for j, i in enumerate(idx):
            w, small = ws[i], smalls[i]
            Cout = w.shape[0]
            K = w.numel() // Cout
            Cin, KH, KW = (w.shape[1], w.shape[2], w.shape[3]) if w.dim() == 4 else (K, 1, 1)
            dweff = _f32(dweffs[i])
            st = dweff.stride() if dweff.dim() == 4 else (K, 1, 1, 1)
            if not _rows_dense(st, (Cin, KH, KW), K):
                dweff = dweff.contiguous()
                st = dweff.stride() if dweff.dim() == 4 else (K, 1, 1, 1)
            nbytes = lib.csg_spectral_norm_workspace(Cout, K)
            wsb = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
            dw = _grad_dest(w, tuple(w.shape))
            if dw is None:
                dw = torch.empty_like(w)
            it = items[j]
            it.dweff, it.Cout, it.Cin, it.KH, it.KW = dweff.data_ptr(), Cout, Cin, KH, KW
            it.s0, it.s1, it.s2, it.s3 = st[0], st[1], st[2], st[3]
            it.w, it.sigma = w.data_ptr(), small.data_ptr()
            it.u_used, it.v_used = small.data_ptr() + 4, small.data_ptr() + 4 * (1 + Cout)
            it.dw, it.workspace, it.workspace_bytes = dw.data_ptr(), wsb.data_ptr(), nbytes
            keep += [dweff, wsb]
            grads[2 + 3 * i] = dw
        check(lib.csg_spectral_norm_bwd_multi(items, len(idx), stream()), "spectral_norm_bwd_multi")
        return tuple(grads)


def spectral_weights(wuv, iterate, eps=1e-12):
    """[(weight_orig, u, v), ...] -> [W / sigma, ...] (see _SpectralWeightMulti)."""
    flat = [t for triple in wuv for t in triple]
    for t in flat:
        if not t.is_cuda:
            raise RuntimeError("canonicalsg2im_amd ops need HIP (cuda) tensors; there is no CPU path")
    return list(_SpectralWeightMulti.apply(bool(iterate), float(eps), *flat))


def _rows_dense(st, dims, K):
    """True if a (Cout, Cin, KH, KW) tensor with element strides `st` stores every Cout-row as a dense
    permutation of its K elements (contiguous, or the weight-gradient kernel's [Cout][KH][KW][Cin])."""
    run = 1
    for i in sorted(range(3), key=lambda i: st[1 + i]):
        if dims[i] > 1 and st[1 + i] != run:
            return False
        run *= dims[i]
    return st[0] == K


def spectral_weight(w, u, v, iterate, eps=1e-12):
    return _SpectralWeight.apply(w, u, v, bool(iterate), float(eps))


# ------------------------------------------------------------------------------------ normalisation
def _sync_world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _multi(world, syncs):
    """Whether the N-replica SyncBN path runs (the statistics message and `clamp(var, eps)^-1/2`) for a layer that `syncs`
    (a synchronised BatchNorm in training mode — never an InstanceNorm or an eval-mode norm): several ranks, or one rank
    under CSG_DIST_FORCE=1 (csg_dist.active(): RCCL bring-up on a 1-GPU box)."""
    return bool(syncs) and (world > 1 or csg_dist.active())


class _NormAct(torch.autograd.Function):
    """BatchNorm (G=1) or InstanceNorm (G=B) statistics + optional SPADE modulation + LeakyReLU.

    Batch mode, training: running stats updated as F.batch_norm does; with an initialised process
    group of N > 1 ranks the (sum, sum^2) message is all-reduced and inv_std uses the reference's
    N-replica formula clamp(var, eps)^-1/2 (sync_batchnorm/batchnorm.py:128-145)."""

    @staticmethod
    def forward(ctx, x, gb, running_mean, running_var, instance, training, slope, eps, momentum, sync):
        x = nhwc(_f32(x))
        B, C, H, W = x.shape
        G = B if instance else 1
        P = (B * H * W) // G
        dev = x.device
        syncs = bool(sync and not instance and training)
        world = _sync_world() if syncs else 1
        multi = _multi(world, syncs)
        use_batch_stats = training or instance
        mean = torch.empty(G * C, device=dev, dtype=torch.float32)
        invstd = torch.empty(G * C, device=dev, dtype=torch.float32)
        count = float(P * world)
        if use_batch_stats:
            nch = _chunks(P, G)
            part = torch.empty(G * nch * 2 * C, device=dev, dtype=torch.float64)
            rm = running_mean if (training and not instance and running_mean is not None) else None
            rv = running_var if rm is not None else None
            if not multi:                            # one rank: second reduction stage and finalisation in one launch
                check(lib.csg_norm_stats_finalize(ptr(x), G, P, C, ptr(part), nch, count, eps, ptr(mean), ptr(invstd), ptr(rm),
                                                  ptr(rv), None, None, momentum, stream()), "norm_stats_finalize")
            else:
                sums = torch.empty(G * 2 * C, device=dev, dtype=torch.float64)
                check(lib.csg_norm_stats(ptr(x), G, P, C, ptr(sums), ptr(part), nch, stream()), "norm_stats")
                csg_dist.all_reduce_stats(sums)
                check(lib.csg_norm_finalize(ptr(sums), G, C, count, eps, 1, ptr(mean), ptr(invstd), ptr(rm), ptr(rv), momentum,
                                            stream()), "norm_finalize")
        else:
            mean.copy_(running_mean)
            invstd.copy_(torch.rsqrt(running_var + eps))
        gbn = nhwc(gb) if gb is not None else None
        y = torch.empty_like(x)
        check(lib.csg_norm_apply_fwd(ptr(x), ptr(mean), ptr(invstd), ptr(gbn), slope, G, P, C, ptr(y), None, 1.0, None,
                                     stream()),
              "norm_apply_fwd")
        ctx.save_for_backward(x, gbn, mean, invstd)
        ctx.cfg = (G, P, C, slope, use_batch_stats, multi, count)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gb, mean, invstd = ctx.saved_tensors
        G, P, C, slope, use_batch_stats, multi, count = ctx.cfg
        dy = nhwc(dy)
        dev = dy.device
        dgb = torch.empty_like(gb) if gb is not None else None
        nch = _chunks(P, G)
        part = torch.empty(G * nch * 2 * C, device=dev, dtype=torch.float64)
        dsums = torch.empty(G * 2 * C, device=dev, dtype=torch.float64)
        check(lib.csg_norm_apply_bwd_reduce(ptr(dy), ptr(x), ptr(mean), ptr(invstd), ptr(gb), None, slope, G, P, C, ptr(dgb),
                                            ptr(dsums), ptr(part), nch, 2 * C, stream()), "norm_bwd_reduce")
        dx = None
        if ctx.needs_input_grad[0]:
            if not use_batch_stats:
                dsums.zero_()                       # eval mode: statistics are constants
            elif multi:
                csg_dist.all_reduce_stats(dsums)
            dx = torch.empty_like(x)
            check(lib.csg_norm_apply_bwd_dx(ptr(dy), ptr(x), ptr(mean), ptr(invstd), ptr(gb), slope, ptr(dsums), count,
                                            G, P, C, ptr(dx), None, None, 1.0, ptr(dgb), None, 2 * C, stream()), "norm_bwd_dx")
        return dx, dgb, None, None, None, None, None, None, None, None


class _NormActPair(torch.autograd.Function):
    """Two SPADE modulations of ONE normalised x — norm_0 and norm_s of a residual block with a learned shortcut
    (reference architecture.py:37-47): in training both param-free BatchNorms see the same batch, so the statistics
    are computed once (each module's running statistics are updated from them), and the backward makes one pass over x
    that folds both gradients — instead of two statistics passes, two dx passes and autograd's addition."""

    @staticmethod
    def forward(ctx, x, gb0, gb1, rm0, rv0, rm1, rv1, slope0, slope1, eps, momentum, sync):
        x = nhwc(_f32(x))
        B, C, H, W = x.shape
        P = B * H * W
        dev = x.device
        world = _sync_world() if sync else 1
        multi = _multi(world, sync)
        count = float(P * world)
        mean = torch.empty(C, device=dev, dtype=torch.float32)
        invstd = torch.empty(C, device=dev, dtype=torch.float32)
        nch = _chunks(P, 1)
        part = torch.empty(nch * 2 * C, device=dev, dtype=torch.float64)
        if not multi:                                # same batch statistics, each module's own running buffers: one launch
            check(lib.csg_norm_stats_finalize(ptr(x), 1, P, C, ptr(part), nch, count, eps, ptr(mean), ptr(invstd), ptr(rm0),
                                              ptr(rv0 if rm0 is not None else None), ptr(rm1),
                                              ptr(rv1 if rm1 is not None else None), momentum, stream()), "norm_stats_finalize")
        else:
            sums = torch.empty(2 * C, device=dev, dtype=torch.float64)
            check(lib.csg_norm_stats(ptr(x), 1, P, C, ptr(sums), ptr(part), nch, stream()), "norm_stats")
            csg_dist.all_reduce_stats(sums)
            for rm, rv in ((rm0, rv0), (rm1, rv1)):
                check(lib.csg_norm_finalize(ptr(sums), 1, C, count, eps, 1, ptr(mean), ptr(invstd), ptr(rm),
                                            ptr(rv if rm is not None else None), momentum, stream()), "norm_finalize")
        gb0, gb1 = nhwc(gb0), nhwc(gb1)
        y0, y1 = torch.empty_like(x), torch.empty_like(x)
        check(lib.csg_norm_apply_fwd(ptr(x), ptr(mean), ptr(invstd), ptr(gb0), slope0, 1, P, C, ptr(y0), ptr(gb1), slope1,
                                     ptr(y1), stream()), "norm_apply_fwd")
        ctx.save_for_backward(x, gb0, gb1, mean, invstd)
        ctx.cfg = (P, C, slope0, slope1, multi, count)
        return y0, y1

    @staticmethod
    def backward(ctx, dy0, dy1):
        x, gb0, gb1, mean, invstd = ctx.saved_tensors
        P, C, slope0, slope1, multi, count = ctx.cfg
        dy0, dy1 = nhwc(dy0), nhwc(dy1)
        dev = x.device
        nch = _chunks(P, 1)
        part = torch.empty(nch * 2 * C, device=dev, dtype=torch.float64)
        dsums = torch.empty(2, 2 * C, device=dev, dtype=torch.float64)
        dgb0, dgb1 = torch.empty_like(gb0), torch.empty_like(gb1)
        check(lib.csg_norm_apply_bwd_reduce(ptr(dy0), ptr(x), ptr(mean), ptr(invstd), ptr(gb0), None, slope0, 1, P, C, ptr(dgb0),
                                            ptr(dsums[0]), ptr(part), nch, 2 * C, stream()), "norm_bwd_reduce")
        check(lib.csg_norm_apply_bwd_reduce(ptr(dy1), ptr(x), ptr(mean), ptr(invstd), ptr(gb1), None, slope1, 1, P, C, ptr(dgb1),
                                            ptr(dsums[1]), ptr(part), nch, 2 * C, stream()), "norm_bwd_reduce")
        dx = None
        if ctx.needs_input_grad[0]:
            both = dsums[0] + dsums[1]                    # the reductions are linear in dn: 4C doubles
            if multi:
                csg_dist.all_reduce_stats(both)
            dx = torch.empty_like(x)
            check(lib.csg_norm_apply_bwd_dx(ptr(dy0), ptr(x), ptr(mean), ptr(invstd), ptr(gb0), slope0, ptr(both), count,
                                            1, P, C, ptr(dx), ptr(dy1), ptr(gb1), slope1, ptr(dgb0), ptr(dgb1), 2 * C, stream()),
                  "norm_bwd_dx")
        return dx, dgb0, dgb1, None, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------ SPADE with the modulation in the
# gamma || beta convolution's epilogue
SPADE_FUSED = os.environ.get("CSG_SPADE_FUSED", "1") != "0"
SPADE_JOINED = os.environ.get("CSG_SPADE_JOINED", "1") != "0"     # _SpadeJoined for the maps the fused epilogue does not serve


def spade_fused_eligible(x, nhidden, C, ks, training):
    """Training-mode SPADE layers whose gamma || beta convolution runs Winograd F(4x4,3x3) (maps >= 32 wide): the beta half
    of the convolution writes leaky(xhat (1 + gamma) + beta) itself (csg_wino4_conv_part)."""
    if not (SPADE_FUSED and WINO_ENABLED and training and ks == 3 and x.dim() == 4 and C % 32 == 0):
        return False
    B, _, H, W = x.shape
    return lib.csg_wino4_supported(_wino_desc(B, H, W, nhidden, C)) == 1


SPADE_JOINT = os.environ.get("CSG_SPADE_JOINT", "1") != "0"      # 0: the gamma / beta launch pair also on one rank (A/B)


class _SpadeFused(torch.autograd.Function):
    """K = 1 or 2 SPADE modulations of ONE batch-normalised x (reference normalization.py:96-110; K = 2: norm_s and norm_0
    of a residual block, architecture.py:37-47), each `leaky(xhat (1 + gamma_k) + beta_k, slope_k)` with gamma_k || beta_k =
    conv3x3(actv_k, w_k) + b_k.  Forward per modulation: the gamma half of the convolution into a (B,H,W,C) buffer, then
    the beta half with the modulation as its epilogue (beta itself is never materialised).  Backward: the two norm passes of
    _NormAct / _NormActPair (the LeakyReLU gate read off y's sign), then the joined convolution's backward-data and
    weight-gradient passes on d(gamma || beta)."""

    NARG = 7          # per modulation: actv, w, b, running_mean, running_var, slope, in_slope

    @staticmethod
    def forward(ctx, x, eps, momentum, sync, *mods):
        K = len(mods) // _SpadeFused.NARG
        x = nhwc(_f32(x))
        B, C, H, W = x.shape
        P = B * H * W
        dev = x.device
        world = _sync_world() if sync else 1
        multi = _multi(world, sync)
        count = float(P * world)
        mean = torch.empty(C, device=dev, dtype=torch.float32)
        invstd = torch.empty(C, device=dev, dtype=torch.float32)
        nch = _chunks(P, 1)
        part = torch.empty(nch * 2 * C, device=dev, dtype=torch.float64)
        pending = sums = None
        if not multi:                                # one rank: statistics finalised right away (two launches in all)
            r0, r1 = (mods[3], mods[4]), ((mods[10], mods[11]) if K == 2 else (None, None))
            check(lib.csg_norm_stats_finalize(ptr(x), 1, P, C, ptr(part), nch, count, eps, ptr(mean), ptr(invstd), ptr(r0[0]),
                                              ptr(r0[1] if r0[0] is not None else None), ptr(r1[0]),
                                              ptr(r1[1] if r1[0] is not None else None), momentum, stream()),
                  "norm_stats_finalize")
        else:
            sums = torch.empty(2 * C, device=dev, dtype=torch.float64)
            check(lib.csg_norm_stats(ptr(x), 1, P, C, ptr(sums), ptr(part), nch, stream()), "norm_stats")
            # N > 1: the statistics travel while the gamma halves (which do not need them) are computed
            pending = csg_dist.all_reduce_stats_async(sums)
        saved, outs, cfg, launches, pres = [x, mean, invstd], [], [], [], []
        # One rank, C a multiple of 32: ONE launch per modulation (csg_wino4_conv_spade: blocks own a gamma tile and its beta
        # tile; gamma is written for the backward but never read back, the 128-channel input is staged once).  N > 1 ranks keep
        # the launch pair: the gamma halves, which need no statistics, run while the SyncBN message travels.
        joint = SPADE_JOINT and not multi and C % 32 == 0
        for k in range(K):
            actv, w, b, rm, rv, slope, in_slope = mods[k * 7:(k + 1) * 7]
            actv = nhwc(_f32(actv))
            nh = actv.shape[1]
            if tuple(w.shape) != (2 * C, nh, 3, 3) or tuple(actv.shape) != (B, nh, H, W):
                raise RuntimeError("spade_fused: weight %s / actv %s do not fit x %s" % (tuple(w.shape), tuple(actv.shape),
                                                                                         tuple(x.shape)))
            up = wino_pack(w, False, None, 4)
            pres.append(_take_bwd_operand(w, B, H, W, nh, 2 * C, 3, 3, 1, 1) if ctx.needs_input_grad[4 + k * 7] else None)
            bd = b.detach().contiguous()
            gbuf = empty_nhwc(B, C, H, W, dev)                # gamma only: beta is consumed in the epilogue that forms it
            d = _wino_desc(B, H, W, nh, C)
            d.y_cs = C
            if joint and lib.csg_wino4_conv_spade_supported(d):
                y = torch.empty_like(x)
                _wino4_audit("spade_joint", B * H * W, nh + 3 * C, up)          # actv, x read; gamma, y written
                check(lib.csg_wino4_conv_spade(d, ptr(actv), ptr(up), ptr(bd), ptr(x), ptr(gbuf), C, ptr(mean), ptr(invstd),
                                               slope, ptr(y), stream()), "wino4_conv_spade")
                saved += [actv, w, gbuf, y]
                outs.append(y)
                cfg.append((slope, in_slope, nh))
                continue
            _wino4_audit("spade_gamma", B * H * W, nh + C, up)
            _wino4_audit("spade_beta", B * H * W, nh + 3 * C, up)
            check(lib.csg_wino4_conv_part(d, ptr(actv), ptr(up), 0, 2 * C // 32, ptr(bd), None, None, 0, None, None, 1.0,
                                          ptr(gbuf), stream()), "wino4_conv_part(gamma)")
            launches.append((actv, w, up, bd, gbuf, nh, slope, in_slope))
        if pending is not None:
            pending.wait()
        for k in range(K if multi else 0):
            rm, rv = mods[k * 7 + 3], mods[k * 7 + 4]
            check(lib.csg_norm_finalize(ptr(sums), 1, C, count, eps, 1, ptr(mean), ptr(invstd), ptr(rm),
                                        ptr(rv if rm is not None else None), momentum, stream()), "norm_finalize")
        for (actv, w, up, bd, gbuf, nh, slope, in_slope) in launches:
            y = torch.empty_like(x)
            d = _wino_desc(B, H, W, nh, C)
            d.y_cs = C
            check(lib.csg_wino4_conv_part(d, ptr(actv), ptr(up), C // 32, 2 * C // 32, ptr(bd[C:]), ptr(x), ptr(gbuf), C,
                                          ptr(mean), ptr(invstd), slope, ptr(y), stream()), "wino4_conv_part(beta)")
            saved += [actv, w, gbuf, y]
            outs.append(y)
            cfg.append((slope, in_slope, nh))
        ctx.save_for_backward(*saved)
        ctx.cfg = (K, P, C, B, H, W, multi, count, tuple(cfg))
        ctx.ut_pres = pres
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        import types
        K, P, C, B, H, W, multi, count, cfg = ctx.cfg
        sv = ctx.saved_tensors
        x, mean, invstd = sv[0], sv[1], sv[2]
        dev = x.device
        nch = _chunks(P, 1)
        part = torch.empty(nch * 2 * C, device=dev, dtype=torch.float64)
        dsums = torch.empty(K, 2 * C, device=dev, dtype=torch.float64)
        dys = [nhwc(dy) for dy in dys]
        dgbs = []
        for k in range(K):
            actv, w, gbuf, y = sv[3 + 4 * k:7 + 4 * k]
            dgb = empty_nhwc(B, 2 * C, H, W, dev)             # d(gamma || beta): the joined convolution's incoming gradient
            check(lib.csg_norm_apply_bwd_reduce(ptr(dys[k]), ptr(x), ptr(mean), ptr(invstd), ptr(gbuf), ptr(y), cfg[k][0], 1, P,
                                                C, ptr(dgb), ptr(dsums[k]), ptr(part), nch, C, stream()), "norm_bwd_reduce")
            dgbs.append(dgb)
        dx, both, pending = None, None, None
        if ctx.needs_input_grad[0]:
            both = dsums[0] + dsums[1] if K == 2 else dsums[0]
            # N > 1: the reductions travel while the convolution's backward passes (which need only d(gamma || beta)) run
            pending = csg_dist.all_reduce_stats_async(both) if multi else None
        grads = [None, None, None, None]
        for k in range(K):
            actv, w, gbuf, y = sv[3 + 4 * k:7 + 4 * k]
            slope, in_slope, nh = cfg[k]
            base = 4 + 7 * k
            need = ctx.needs_input_grad[base:base + 3]
            # the joined gamma || beta convolution's backward on d(gamma || beta): _Conv2d.backward on a stand-in context
            fake = types.SimpleNamespace(
                saved_tensors=(actv, w, None), geom=(B, H, W, nh, 2 * C, 3, 3, 1, 1, H, W, ACT_NONE, 0.0),
                in_act=(ACT_LEAKY, in_slope) if in_slope is not None else None, grad_is_pre=False, few=None, dx_range=None,
                packs=None, needs_input_grad=(need[0], need[1], need[2], False), has_bias=True, has_res=False, cout_w=2 * C,
                pre_slope=None, ut_pre=ctx.ut_pres[k])
            r = _Conv2d.backward(fake, dgbs[k])
            grads += [r[0], r[1], r[2], None, None, None, None]
        if ctx.needs_input_grad[0]:
            if pending is not None:
                pending.wait()
            dx = torch.empty_like(x)
            two = K == 2
            check(lib.csg_norm_apply_bwd_dx(ptr(dys[0]), ptr(x), ptr(mean), ptr(invstd), ptr(sv[5]), cfg[0][0], ptr(both), count,
                                            1, P, C, ptr(dx), ptr(dys[1]) if two else None, ptr(sv[9]) if two else None,
                                            cfg[1][0] if two else 1.0, ptr(dgbs[0]), ptr(dgbs[1]) if two else None, C, stream()),
                  "norm_bwd_dx")
            grads[0] = dx
        return tuple(grads)


class _SpadeJoined(torch.autograd.Function):
    """One SPADE modulation of a batch-normalised x whose gamma || beta convolution is NOT one the fused epilogue serves (the
    8 x 8 and 16 x 16 maps of `head_0` / `G_middle_*`): the same kernels as conv2d + norm_act in the same arithmetic, but
    ordered inside one Function so that on N > 1 ranks both SyncBN messages travel under a convolution, as in _SpadeFused —
    forward: statistics, asynchronous all-reduce, gamma || beta convolution (needs no statistics), wait, finalise, modulate;
    backward: pass 1 (d gamma || beta and the two reductions), asynchronous all-reduce, the convolution's backward passes
    (they need only d gamma || beta), wait, pass 2 (dx).  On one rank the launches are those of the unfused path."""

    @staticmethod
    def forward(ctx, x, actv, w, b, running_mean, running_var, pad, slope, in_slope, eps, momentum, sync):
        import types
        x = nhwc(_f32(x))
        B, C, H, W = x.shape
        P = B * H * W
        dev = x.device
        world = _sync_world() if sync else 1
        multi = _multi(world, sync)
        count = float(P * world)
        mean = torch.empty(C, device=dev, dtype=torch.float32)
        invstd = torch.empty(C, device=dev, dtype=torch.float32)
        nch = _chunks(P, 1)
        part = torch.empty(nch * 2 * C, device=dev, dtype=torch.float64)
        pending = sums = None
        if not multi:
            check(lib.csg_norm_stats_finalize(ptr(x), 1, P, C, ptr(part), nch, count, eps, ptr(mean), ptr(invstd),
                                              ptr(running_mean), ptr(running_var if running_mean is not None else None), None,
                                              None, momentum, stream()), "norm_stats_finalize")
        else:
            sums = torch.empty(2 * C, device=dev, dtype=torch.float64)
            check(lib.csg_norm_stats(ptr(x), 1, P, C, ptr(sums), ptr(part), nch, stream()), "norm_stats")
            pending = csg_dist.all_reduce_stats_async(sums)
        # the joined convolution through _Conv2d's own forward on a stand-in context (its saved tensors are kept for the
        # backward below)
        fake = types.SimpleNamespace(needs_input_grad=(True, True, True, False))
        fake.save_for_backward = lambda *t: setattr(fake, "saved_tensors", t)
        in_act = (ACT_LEAKY, float(in_slope)) if in_slope is not None else None
        gb = _Conv2d.forward(fake, actv, w, b, None, 1, int(pad), ACT_NONE, 0.0, None, None, in_act, False, None, None)
        if pending is not None:
            pending.wait()
        if multi:
            check(lib.csg_norm_finalize(ptr(sums), 1, C, count, eps, 1, ptr(mean), ptr(invstd),
                                        ptr(running_mean), ptr(running_var if running_mean is not None else None), momentum,
                                        stream()), "norm_finalize")
        y = torch.empty_like(x)
        check(lib.csg_norm_apply_fwd(ptr(x), ptr(mean), ptr(invstd), ptr(gb), slope, 1, P, C, ptr(y), None, 1.0, None, stream()),
              "norm_apply_fwd")
        cx, cw, cy = fake.saved_tensors
        ctx.save_for_backward(x, gb, mean, invstd, cx, cw)
        fake.save_for_backward = None
        fake.saved_tensors = None
        ctx.conv = fake
        ctx.cfg = (P, C, slope, multi, count)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gb, mean, invstd, cx, cw = ctx.saved_tensors
        P, C, slope, multi, count = ctx.cfg
        dy = nhwc(dy)
        dev = x.device
        nch = _chunks(P, 1)
        part = torch.empty(nch * 2 * C, device=dev, dtype=torch.float64)
        dsums = torch.empty(2 * C, device=dev, dtype=torch.float64)
        dgb = torch.empty_like(gb)
        check(lib.csg_norm_apply_bwd_reduce(ptr(dy), ptr(x), ptr(mean), ptr(invstd), ptr(gb), None, slope, 1, P, C, ptr(dgb),
                                            ptr(dsums), ptr(part), nch, 2 * C, stream()), "norm_bwd_reduce")
        pending = csg_dist.all_reduce_stats_async(dsums) if (multi and ctx.needs_input_grad[0]) else None
        fake = ctx.conv
        fake.saved_tensors = (cx, cw, None)
        fake.needs_input_grad = (ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3], False)
        r = _Conv2d.backward(fake, dgb)
        dx = None
        if ctx.needs_input_grad[0]:
            if pending is not None:
                pending.wait()
            dx = torch.empty_like(x)
            check(lib.csg_norm_apply_bwd_dx(ptr(dy), ptr(x), ptr(mean), ptr(invstd), ptr(gb), slope, ptr(dsums), count,
                                            1, P, C, ptr(dx), None, None, 1.0, ptr(dgb), None, 2 * C, stream()), "norm_bwd_dx")
        return dx, r[0], r[1], r[2], None, None, None, None, None, None, None, None


def spade_joined(x, actv, w, b, running_mean, running_var, pad, slope, in_slope, eps=1e-5, momentum=0.1, sync=True):
    """leaky(batchnorm(x) (1 + gamma) + beta, slope) with gamma || beta = conv(actv, w) + b — see _SpadeJoined."""
    return _SpadeJoined.apply(x, actv, w, b, running_mean, running_var, int(pad), float(slope),
                              None if in_slope is None else float(in_slope), float(eps), float(momentum), bool(sync))


def spade_fused(x, mods, eps=1e-5, momentum=0.1, sync=True):
    """mods: one or two tuples (actv, w, b, running_mean, running_var, slope, in_slope) — see _SpadeFused.  Returns the
    list of modulated maps."""
    flat = []
    for (actv, w, b, rm, rv, slope, in_slope) in mods:
        flat += [actv, w, b, rm, rv, float(slope), None if in_slope is None else float(in_slope)]
    return list(_SpadeFused.apply(x, float(eps), float(momentum), bool(sync), *flat))


def norm_act_pair(x, gb0, gb1, rm0, rv0, rm1, rv1, slope0, slope1, eps=1e-5, momentum=0.1, sync=True):
    """Training-mode BatchNorm statistics of x, then the SPADE modulations (gb0, slope0) and (gb1, slope1) of it."""
    return _NormActPair.apply(x, gb0, gb1, rm0, rv0, rm1, rv1, float(slope0), float(slope1), float(eps), float(momentum),
                              bool(sync))


def norm_act(x, gb=None, running_mean=None, running_var=None, instance=False, training=True, slope=1.0, eps=1e-5,
             momentum=0.1, sync=True):
    return _NormAct.apply(x, gb, running_mean, running_var, bool(instance), bool(training), float(slope), float(eps),
                          float(momentum), bool(sync))


# ------------------------------------------------------------------------------------ resampling
class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = nhwc(_f32(x))
        B, C, H, W = x.shape
        y = empty_nhwc(B, C, 2 * H, 2 * W, x.device)
        check(lib.csg_upsample2x_fwd(ptr(x), B, H, W, C, ptr(y), stream()), "upsample2x_fwd")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = nhwc(dy)
        dx = empty_nhwc(B, C, H, W, dy.device)
        check(lib.csg_upsample2x_bwd(ptr(dy), B, H, W, C, ptr(dx), stream()), "upsample2x_bwd")
        return dx


def upsample2x(x):
    return _Upsample2x.apply(x)


class _NearestResize(torch.autograd.Function):
    """F.interpolate(x, size=(OH, OW), mode='nearest') (reference normalization.py:98)."""

    @staticmethod
    def forward(ctx, x, OH, OW):
        x = nhwc(_f32(x))
        B, C, IH, IW = x.shape
        if C % 4:
            raise RuntimeError("nearest_resize: the channel count must be a multiple of 4")
        y = empty_nhwc(B, C, OH, OW, x.device)
        check(lib.csg_nearest_resize_fwd(ptr(x), B, IH, IW, C, OH, OW, ptr(y), stream()), "nearest_resize_fwd")
        ctx.shape = (B, C, IH, IW, OH, OW)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, IH, IW, OH, OW = ctx.shape
        dy = nhwc(dy)
        dx = empty_nhwc(B, C, IH, IW, dy.device)
        check(lib.csg_nearest_resize_bwd(ptr(dy), B, IH, IW, C, OH, OW, ptr(dx), stream()), "nearest_resize_bwd")
        return dx, None, None


def nearest_resize(x, size):
    return _NearestResize.apply(x, int(size[0]), int(size[1]))


class _AvgPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = nhwc(_f32(x))
        B, C, H, W = x.shape
        y = empty_nhwc(B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, x.device)
        check(lib.csg_avgpool3s2_fwd(ptr(x), B, H, W, C, ptr(y), stream()), "avgpool_fwd")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = nhwc(dy)
        dx = empty_nhwc(B, C, H, W, dy.device)
        check(lib.csg_avgpool3s2_bwd(ptr(dy), B, H, W, C, ptr(dx), stream()), "avgpool_bwd")
        return dx


def avgpool3s2(x):
    return _AvgPool3s2.apply(x)


class _PoolFanout(torch.autograd.Function):
    """x -> (x, avgpool3s2(x)): a map that feeds one consumer at full resolution and another through its pooled copy (the two
    scales of MultiscaleDiscriminator, reference discriminator.py:120-131).  The backward receives both gradients and forms
    d x = g_full + avgpool_bwd(g_pooled) in ONE pass (csg_avgpool3s2_bwd_add) — where autograd would have run the pooling
    backward into a map of its own and then added the two (600 MB of traffic at 256 x 256 x 36 channels where this pass moves
    340, three times per step, and one launch less).  Out of place: an incoming gradient may be shared with another node.
    Bit-identical: a two-term fp32 sum does not depend on the order of its terms."""

    @staticmethod
    def forward(ctx, x):
        xc = nhwc(_f32(x))
        B, C, H, W = xc.shape
        y = empty_nhwc(B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, xc.device)
        check(lib.csg_avgpool3s2_fwd(ptr(xc), B, H, W, C, ptr(y), stream()), "avgpool_fwd")
        ctx.shape = (B, C, H, W)
        ctx.set_materialize_grads(False)          # an unused output arrives as None, not as a map of zeros
        return xc.view_as(xc), y

    @staticmethod
    def backward(ctx, g_full, g_pool):
        B, C, H, W = ctx.shape
        if g_pool is None:
            return g_full
        g_pool = nhwc(g_pool)
        if g_full is None:
            dx = empty_nhwc(B, C, H, W, g_pool.device)
            check(lib.csg_avgpool3s2_bwd(ptr(g_pool), B, H, W, C, ptr(dx), stream()), "avgpool_bwd")
            return dx
        g_full = nhwc(g_full)
        dx = empty_nhwc(B, C, H, W, g_pool.device)
        check(lib.csg_avgpool3s2_bwd_add(ptr(g_pool), B, H, W, C, ptr(g_full), ptr(dx), stream()), "avgpool_bwd_add")
        return dx


POOL_FANOUT = os.environ.get("CSG_POOL_FANOUT", "1") != "0"       # 0: plain avgpool3s2 + autograd's own addition (A/B)


def pool_fanout(x):
    """(x, avgpool3s2(x)) with the two gradients of x summed inside the pooling backward (ops._PoolFanout)."""
    if not POOL_FANOUT:
        return x, avgpool3s2(x)
    return _PoolFanout.apply(x)


HINGE_FUSED = os.environ.get("CSG_HINGE_FUSED", "1") != "0"       # 0: the GAN terms as torch's sub / clamp / mean / neg chain (A/B)


class _HingeMean(torch.autograd.Function):
    """(1/n) sum_i -mean(term(pred_i)) over the PatchGAN's scales in ONE launch (reference loss.py:60-93: hinge and `w` modes;
    kind 0: -mean(x), 1: -mean(min(x - 1, 0)), 2: -mean(min(-x - 1, 0))) and one launch in the backward, where torch ran four
    small kernels per scale and direction plus the sum over scales.  A prediction is (B, 1, h, w), any strides."""

    @staticmethod
    def _items(preds, dxs=None):
        arr = (HingeItem * len(preds))()
        for i, p in enumerate(preds):
            B, _, H, W = p.shape
            st = p.stride()
            arr[i].x, arr[i].dx = p.data_ptr(), (dxs[i].data_ptr() if dxs is not None else None)
            arr[i].sb, arr[i].sh, arr[i].sw = st[0], st[2], st[3]
            arr[i].B, arr[i].H, arr[i].W = B, H, W
        return arr

    @staticmethod
    def forward(ctx, kind, *preds):
        preds = [_f32(p.detach()) for p in preds]
        out = torch.empty(1, device=preds[0].device, dtype=torch.float32)
        check(lib.csg_hinge_mean_fwd(_HingeMean._items(preds), len(preds), int(kind), ptr(out), stream()), "hinge_mean_fwd")
        ctx.kind = int(kind)
        ctx.save_for_backward(*preds)
        return out

    @staticmethod
    def backward(ctx, g):
        preds = ctx.saved_tensors
        dxs = [torch.empty((p.shape[0], 1, p.shape[2], p.shape[3]), device=p.device, dtype=torch.float32) for p in preds]
        g = _f32(g).contiguous()
        check(lib.csg_hinge_mean_bwd(_HingeMean._items(preds, dxs), len(preds), ctx.kind, ptr(g), stream()), "hinge_mean_bwd")
        return (None,) + tuple(dxs)


def hinge_mean(preds, kind):
    """None when the fused form does not apply (CPU tensors, more than four scales, maps with more than one channel)."""
    if not HINGE_FUSED or not (1 <= len(preds) <= 4):
        return None
    if any((not p.is_cuda) or p.dim() != 4 or p.shape[1] != 1 or p.dtype != torch.float32 for p in preds):
        return None
    return _HingeMean.apply(kind, *preds)


class _MaxPool2(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) (torchvision vgg19().features, reference architecture.py:96-110)."""

    @staticmethod
    def forward(ctx, x):
        x = nhwc(_f32(x))
        B, C, H, W = x.shape
        y = empty_nhwc(B, C, H // 2, W // 2, x.device)
        check(lib.csg_maxpool2_fwd(ptr(x), B, H, W, C, ptr(y), stream()), "maxpool2_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        B, C, H, W = x.shape
        dy = nhwc(dy)
        dx = torch.empty_like(x)
        check(lib.csg_maxpool2_bwd(ptr(dy), ptr(x), B, H, W, C, ptr(dx), stream()), "maxpool2_bwd")
        return dx


def maxpool2(x):
    return _MaxPool2.apply(x)


class _AvgPool2(torch.autograd.Function):
    """nn.AvgPool2d(2, 2) (reference sg2im/layers.py:88-90, `build_cnn(pooling='avg')`)."""

    @staticmethod
    def forward(ctx, x):
        x = nhwc(_f32(x))
        B, C, H, W = x.shape
        y = empty_nhwc(B, C, H // 2, W // 2, x.device)
        check(lib.csg_avgpool2_fwd(ptr(x), B, H, W, C, ptr(y), stream()), "avgpool2_fwd")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = nhwc(dy)
        dx = empty_nhwc(B, C, H, W, dy.device)
        check(lib.csg_avgpool2_bwd(ptr(dy), B, H, W, C, ptr(dx), stream()), "avgpool2_bwd")
        return dx


def avgpool2(x):
    return _AvgPool2.apply(x)


class _L1Mean(torch.autograd.Function):
    """nn.L1Loss()(a, b) with b a constant (reference loss.py:115: the target is detached)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a), _f32(b)
        if a.shape != b.shape or a.stride() != b.stride():
            raise RuntimeError("l1_mean: operands must share shape and memory layout")
        if not a.permute(0, 2, 3, 1).is_contiguous() and not a.is_contiguous():
            raise RuntimeError("l1_mean: operands must be dense")
        n = a.numel()
        nbytes = lib.csg_l1_mean_workspace(n)
        if nbytes < 0:
            raise RuntimeError("l1_mean: element count %d is not a multiple of 4" % n)
        ws = torch.empty(nbytes // 8, device=a.device, dtype=torch.float64)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        check(lib.csg_l1_mean_fwd(ptr(a), ptr(b), n, ptr(out), ptr(ws), nbytes, stream()), "l1_mean_fwd")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a)
        g = g.contiguous()
        check(lib.csg_l1_mean_bwd(ptr(a), ptr(b), ptr(g), a.numel(), ptr(da), stream()), "l1_mean_bwd")
        return da, None


def l1_mean(a, b):
    return _L1Mean.apply(a, b.detach())


# ------------------------------------------------------------------------------------ graph encoder
class _Embed(torch.autograd.Function):
    """Concatenated embedding lookups: out[..., k*E:(k+1)*E] = table_k[idx[..., k]]."""

    @staticmethod
    def forward(ctx, idx, *tables):
        lead = idx.shape[:-1]
        A = idx.shape[-1]
        assert A == len(tables)
        idx2 = idx.reshape(-1, A).contiguous()
        rows = idx2.shape[0]
        dims = [t.shape[1] for t in tables]
        D = sum(dims)
        if any(ctx.needs_input_grad[1:]) and max(dims) > 256:
            # the limit of csg_embed_bwd (include/csg_hip.h): refused HERE, before a step is half executed
            raise RuntimeError("embedding lookup: the backward supports embedding_dim <= 256 (got %d)" % max(dims))
        out = torch.empty((rows, D), device=idx.device, dtype=torch.float32)
        off = 0
        for k, t in enumerate(tables):
            tc = _f32(t.detach()).contiguous()
            check(lib.csg_embed_fwd(ctypes_ptr_off(idx2, k), rows, A, ptr(tc), tc.shape[0], dims[k], ptr(out), D, off,
                                    stream()), "embed_fwd")
            off += dims[k]
        ctx.save_for_backward(idx2)
        ctx.meta = (A, dims, [t.shape[0] for t in tables], D)
        return out.reshape(*lead, D)

    @staticmethod
    def backward(ctx, dout):
        (idx2,) = ctx.saved_tensors
        A, dims, sizes, D = ctx.meta
        dout = dout.reshape(-1, D).contiguous()
        rows = idx2.shape[0]
        grads, off = [], 0
        for k in range(A):
            g = torch.zeros((sizes[k], dims[k]), device=dout.device, dtype=torch.float32)
            wb = lib.csg_embed_bwd_workspace(rows, sizes[k], dims[k])
            ws = torch.empty((wb // 4,), device=dout.device, dtype=torch.float32) if wb > 0 else None
            check(lib.csg_embed_bwd(ctypes_ptr_off(idx2, k), rows, A, ptr(dout), D, off, sizes[k], dims[k], ptr(g),
                                    ptr(ws) if ws is not None else None, wb, stream()), "embed_bwd")
            grads.append(g)
            off += dims[k]
        return (None, *grads)


def ctypes_ptr_off(t, elem_off):
    if not t.is_cuda:
        raise RuntimeError("canonicalsg2im_amd ops need HIP tensors; there is no CPU path")
    return ctypes.c_void_p(t.data_ptr() + elem_off * t.element_size())


def embed(idx, tables):
    if idx.dtype != torch.int64:
        raise RuntimeError("embed: indices must be int64 (collate contract)")
    return _Embed.apply(idx, *tables)


def real_object_mask(objs, image_id):
    """uint8 (B,O): objs[...,0] != 0 and != __image__ (sg2im/utils.py:56-63)."""
    B, O, A = objs.shape
    o = objs.contiguous()
    m = torch.empty((B, O), device=objs.device, dtype=torch.uint8)
    check(lib.csg_real_object_mask(ptr(o), B, O, A, int(image_id), ptr(m), stream()), "real_object_mask")
    return m


def graph_csr(triplets, O):
    """(row_ptr (B,O+1) int32, col (B,2T) int32) — incident triplets per object, reference order."""
    B, T, _ = triplets.shape
    tr = triplets.contiguous()
    row_ptr = torch.empty((B, O + 1), device=tr.device, dtype=torch.int32)
    col = torch.empty((B, max(2 * T, 1)), device=tr.device, dtype=torch.int32)
    check(lib.csg_graph_csr_build(ptr(tr), B, T, O, ptr(row_ptr), ptr(col), stream()), "graph_csr_build")
    return row_ptr, col


class _GatherConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obj, pred, triplets, row_ptr, col):
        obj, pred = _f32(obj).contiguous(), _f32(pred).contiguous()
        B, O, Din = obj.shape
        T, Dp = pred.shape[1], pred.shape[2]
        out = torch.empty((B, T, 2 * Din + Dp), device=obj.device, dtype=torch.float32)
        check(lib.csg_gather_concat_fwd(ptr(obj), ptr(pred), ptr(triplets), B, O, T, Din, Dp, ptr(out), stream()),
              "gather_concat_fwd")
        ctx.save_for_backward(row_ptr, col)
        ctx.dims = (B, O, T, Din, Dp)
        return out

    @staticmethod
    def backward(ctx, dcat):
        row_ptr, col = ctx.saved_tensors
        B, O, T, Din, Dp = ctx.dims
        dcat = dcat.contiguous()
        dobj = torch.empty((B, O, Din), device=dcat.device, dtype=torch.float32)
        dpred = torch.empty((B, T, Dp), device=dcat.device, dtype=torch.float32)
        nbytes = lib.csg_gather_concat_bwd_workspace(B, O, T, Din)
        ws = torch.empty(nbytes // 4, device=dcat.device, dtype=torch.float32) if nbytes > 0 else None
        check(lib.csg_gather_concat_bwd(ptr(dcat), ptr(row_ptr), ptr(col), B, O, T, Din, Dp, ptr(dobj), ptr(dpred),
                                        ptr(ws), nbytes, stream()), "gather_concat_bwd")
        return dobj, dpred, None, None, None


def gather_concat(obj, pred, triplets, row_ptr, col):
    return _GatherConcat.apply(obj, pred, triplets, row_ptr, col)


class _SegmentAvg(torch.autograd.Function):
    """(pooled (B,O,H), new_p (B,T,Dp)) from net1's output h (B,T,2H+Dp) and confidences (B,T)."""

    @staticmethod
    def forward(ctx, h, conf, valid, triplets, row_ptr, col, H, Dp, h_is_relu=False):
        """`h_is_relu`: h is the output of a ReLU whose only consumer is this call; the backward then returns the gradient of
        the ReLU's pre-activation (the producer runs with grad_is_pre) — no activation-derivative pass over (B,T,2H+Dp)."""
        h, conf = _f32(h).contiguous(), _f32(conf).contiguous()
        B, T, _ = h.shape
        O = row_ptr.shape[1] - 1
        pooled = torch.empty((B, O, H), device=h.device, dtype=torch.float32)
        cnt = torch.empty((B, O), device=h.device, dtype=torch.float32)
        new_p = torch.empty((B, T, Dp), device=h.device, dtype=torch.float32)
        nbytes = lib.csg_segment_avg_fwd_workspace(B, O, T, H)
        ws = torch.empty(nbytes // 4, device=h.device, dtype=torch.float32) if nbytes > 0 else None
        check(lib.csg_segment_avg_fwd(ptr(h), ptr(conf), ptr(valid), ptr(row_ptr), ptr(col), B, O, T, H, Dp,
                                      ptr(pooled), ptr(cnt), ptr(new_p), ptr(ws), nbytes, stream()), "segment_avg_fwd")
        ctx.save_for_backward(h, conf, valid, triplets, pooled, cnt)
        ctx.dims = (B, O, T, H, Dp)
        ctx.h_is_relu = bool(h_is_relu)
        return pooled, new_p

    @staticmethod
    def backward(ctx, dpooled, dnew_p):
        h, conf, valid, triplets, pooled, cnt = ctx.saved_tensors
        B, O, T, H, Dp = ctx.dims
        dpooled = dpooled.contiguous()
        dnew_p = dnew_p.contiguous() if dnew_p is not None else None
        dh = torch.empty_like(h)
        dconf = torch.empty_like(conf)
        scratch = torch.empty((B, O), device=h.device, dtype=torch.float32)
        check(lib.csg_segment_avg_bwd(ptr(dpooled), ptr(dnew_p), ptr(h), ptr(conf), ptr(valid), ptr(triplets),
                                      ptr(pooled), ptr(cnt), B, O, T, H, Dp, 1 if ctx.h_is_relu else 0, ptr(dh), ptr(dconf),
                                      ptr(scratch), stream()), "segment_avg_bwd")
        return dh, dconf, None, None, None, None, None, None, None


def segment_avg(h, conf, valid, triplets, row_ptr, col, H, Dp, h_is_relu=False):
    return _SegmentAvg.apply(h, conf, valid, triplets, row_ptr, col, int(H), int(Dp), bool(h_is_relu))


# ------------------------------------------------------------------------------------ layout
def _prep_masks(masks):
    """(B,O,M,M) int64/float masks -> contiguous fp32 (what the reference's `.float()` does), or None."""
    if masks is None:
        return None, 0
    m = masks.detach().to(torch.float32).contiguous()
    return m, int(m.shape[-1])


def _layout_bwd_masks(g, cs, boxes, valid, vecs, dims, hw, dmasks, accumulate):
    """Accumulates d loss / d masks of one layout output (layout.py:48-77 is differentiable in the masks)."""
    B, O, S, H, W, M = dims
    check(lib.csg_layout_bwd_masks(ptr(g), cs, 0, ptr(boxes), ptr(valid), M, B, O, S, H, W, hw[0], hw[1], ptr(vecs),
                                   ptr(dmasks), 1 if accumulate else 0, stream()), "layout_bwd_masks")


def _hw(size):
    return (int(size[0]), int(size[1])) if isinstance(size, (tuple, list)) else (int(size), int(size))


class _LayoutPyramid(torch.autograd.Function):
    """boxes_to_layout / masks_to_layout for a whole batch at several output sizes at once: size (h,w)
    samples the full-resolution (H,W) layout at rows floor(y*H/h), columns floor(x*W/w)
    (= F.interpolate(seg, (h,w), 'nearest')).  Differentiable in vecs and in the boxes (layout.py:98-110)."""

    @staticmethod
    def forward(ctx, vecs, boxes, valid, masks, H, W, sizes):
        vecs = _f32(vecs).contiguous()
        boxes = _f32(boxes).contiguous()
        masks, M = _prep_masks(masks)
        B, O, S = vecs.shape
        outs = []
        for (h, w) in sizes:
            seg = empty_nhwc(B, S, h, w, vecs.device)
            check(lib.csg_layout_fwd(ptr(vecs), ptr(boxes), ptr(valid), ptr(masks), M, B, O, S, H, W, h, w, ptr(seg), S,
                                     0, stream()), "layout_fwd")
            outs.append(seg)
        ctx.save_for_backward(boxes, valid, masks, vecs if ctx.needs_input_grad[1] or ctx.needs_input_grad[3] else None)
        ctx.meta = (B, O, S, H, W, M, tuple(sizes))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        boxes, valid, masks, vecs = ctx.saved_tensors
        B, O, S, H, W, M, sizes = ctx.meta
        dvecs = torch.zeros((B, O, S), device=boxes.device, dtype=torch.float32)
        dboxes = torch.zeros((B, O, 4), device=boxes.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        dmasks = None
        if masks is not None and ctx.needs_input_grad[3]:
            dmasks = torch.zeros((B, O, M, M), device=boxes.device, dtype=torch.float32)
        for (h, w), g in zip(sizes, douts):
            if g is None:
                continue
            g = nhwc(g)
            if dmasks is not None:
                _layout_bwd_masks(g, S, boxes, valid, vecs, (B, O, S, H, W, M), (h, w), dmasks, True)
            nws = lib.csg_layout_bwd_workspace(B, O, S, h, w, 0 if masks is None else 1, 0 if dboxes is None else 1)
            ws = torch.empty(nws // 4, device=boxes.device, dtype=torch.float32) if nws > 0 else None
            check(lib.csg_layout_bwd(ptr(g), S, 0, ptr(boxes), ptr(valid), ptr(masks), M, B, O, S, H, W, h, w,
                                     ptr(dvecs), 1, ptr(vecs), ptr(dboxes), ptr(ws), nws, stream()), "layout_bwd")
        return dvecs, dboxes, None, dmasks, None, None, None


def layout_pyramid(vecs, boxes, valid, H, sizes, masks=None, W=None):
    """sizes: ints (square) or (h, w) pairs."""
    W = int(H) if W is None else int(W)
    return _LayoutPyramid.apply(vecs, boxes, valid, masks, int(H), W, tuple(_hw(s) for s in sizes))


def layout_paint(vecs, boxes, valid, masks, H, sizes, W=None):
    """masks_to_layout(..., test_mode=True) for a batch (reference layout.py:71-74,135-151): painter's-algorithm
    compositing in ascending order of each object's mass; inference only, so no autograd graph is recorded."""
    W = int(H) if W is None else int(W)
    with torch.no_grad():
        vecs = _f32(vecs.detach()).contiguous()
        boxes = _f32(boxes.detach()).contiguous()
        masks, M = _prep_masks(masks.detach() if masks is not None else None)
        if masks is None:
            raise RuntimeError("layout_paint needs masks")
        B, O, S = vecs.shape
        mass = torch.empty((B, O), device=vecs.device, dtype=torch.float32)
        check(lib.csg_layout_mass(ptr(vecs), ptr(boxes), ptr(valid), ptr(masks), M, B, O, S, H, W, ptr(mass), stream()),
              "layout_mass")
        srt, idx = torch.sort(mass, dim=1, stable=True)               # np.argsort(mass) of layout.py:141
        order = torch.where(torch.isinf(srt), torch.full_like(idx, -1), idx).to(torch.int32).contiguous()
        outs = []
        for (h, w) in (_hw(s) for s in sizes):
            seg = empty_nhwc(B, S, h, w, vecs.device)
            check(lib.csg_layout_paint(ptr(vecs), ptr(boxes), ptr(masks), M, ptr(order), B, O, S, H, W, h, w, ptr(seg), S,
                                       0, stream()), "layout_paint")
            outs.append(seg)
    return tuple(outs)


class _DiscInput(torch.autograd.Function):
    """cat([img, layout], dim=1) of discriminator.py:120 built in place: one NHWC buffer with
    channels [layout(S) | img(3) | zero pad] so that every pixel row is 16-byte aligned; the
    first conv's weight is permuted to match by the caller."""

    @staticmethod
    def forward(ctx, img, vecs, boxes, valid, masks, H):
        vecs = _f32(vecs).contiguous()
        boxes = _f32(boxes).contiguous()
        masks, M = _prep_masks(masks)
        B, O, S = vecs.shape
        Ct = (S + 3 + 3) // 4 * 4
        # one pass writes every channel of every pixel: the layout, the image behind it, the zero pad
        buf = torch.empty((B, H, H, Ct), device=vecs.device, dtype=torch.float32)
        if tuple(img.shape) != (B, 3, H, H):
            raise RuntimeError("disc_input: image %s does not fit (B=%d, 3, %d, %d)" % (tuple(img.shape), B, H, H))
        sb, sc, sh, sw = img.stride()
        check(lib.csg_disc_input_fwd(ptr(vecs), ptr(boxes), ptr(valid), ptr(masks), M, B, O, S, H, H, ptr(img), sb, sc, sh, sw,
                                     ptr(buf), Ct, stream()), "disc_input_fwd")
        ctx.save_for_backward(boxes, valid, masks, vecs if ctx.needs_input_grad[2] or ctx.needs_input_grad[4] else None)
        ctx.meta = (B, O, S, H, Ct, M)
        return buf.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dbuf):
        boxes, valid, masks, vecs = ctx.saved_tensors
        B, O, S, H, Ct, M = ctx.meta
        dbuf = nhwc(dbuf)
        dimg = dvecs = dboxes = dmasks = None
        if ctx.needs_input_grad[0]:
            dimg = dbuf[:, S:S + 3].contiguous()
        if masks is not None and ctx.needs_input_grad[4]:
            dmasks = torch.empty((B, O, M, M), device=dbuf.device, dtype=torch.float32)
            _layout_bwd_masks(dbuf, Ct, boxes, valid, vecs, (B, O, S, H, H, M), (H, H), dmasks, False)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dvecs = torch.empty((B, O, S), device=dbuf.device, dtype=torch.float32)
            if ctx.needs_input_grad[2]:
                dboxes = torch.empty((B, O, 4), device=dbuf.device, dtype=torch.float32)
            nws = lib.csg_layout_bwd_workspace(B, O, S, H, H, 0 if masks is None else 1, 0 if dboxes is None else 1)
            ws = torch.empty(nws // 4, device=dbuf.device, dtype=torch.float32) if nws > 0 else None
            check(lib.csg_layout_bwd(ptr(dbuf), Ct, 0, ptr(boxes), ptr(valid), ptr(masks), M, B, O, S, H, H, H, H,
                                     ptr(dvecs), 0, ptr(vecs), ptr(dboxes), ptr(ws), nws, stream()), "layout_bwd")
        return dimg, dvecs, dboxes, None, dmasks, None


def disc_input(img, vecs, boxes, valid, H, masks=None):
    return _DiscInput.apply(_f32(img), vecs, boxes, valid, masks, int(H))


# ------------------------------------------------------------------------------------ object crops
class _CropObjects(torch.autograd.Function):
    """Bilinear crops of every real object's box from its own image (sg2im/bilinear.py:44-94).
    Output (N, Cp, HH, HH) NHWC with Cp = channels padded to a multiple of 4 (extra channels 0)."""

    @staticmethod
    def forward(ctx, img, boxes, img_idx, HH):
        img = nhwc(_f32(img))
        B, C, H, W = img.shape
        boxes = _f32(boxes).contiguous()
        N = boxes.shape[0]
        Cp = (C + 3) // 4 * 4
        if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and (C > 4 or HH > 64):
            # the limits of csg_crop_bwd (include/csg_hip.h): refused HERE, before a step is half executed
            raise RuntimeError("crop_objects: the backward supports at most 4 image channels and 64 x 64 crops "
                               "(got C = %d, crop_size = %d)" % (C, HH))
        out = empty_nhwc(N, Cp, HH, HH, img.device)
        check(lib.csg_crop_fwd(ptr(img), B, H, W, C, C, ptr(boxes), ptr(img_idx), N, HH, HH, ptr(out), Cp, stream()),
              "crop_fwd")
        ctx.save_for_backward(boxes, img_idx, img if ctx.needs_input_grad[1] else None)
        ctx.meta = (B, C, H, W, N, HH, Cp)
        return out

    @staticmethod
    def backward(ctx, dout):
        boxes, img_idx, img = ctx.saved_tensors
        B, C, H, W, N, HH, Cp = ctx.meta
        dimg = dboxes = None
        dout = nhwc(dout)
        if ctx.needs_input_grad[0]:
            dimg = empty_nhwc(B, C, H, W, dout.device, zero=True)
            check(lib.csg_crop_bwd(ptr(dout), B, H, W, C, C, ptr(boxes), ptr(img_idx), N, HH, HH, Cp, ptr(dimg),
                                   stream()), "crop_bwd")
        if ctx.needs_input_grad[1]:                      # the sampling grid is differentiable in the boxes (bilinear.py:83-94)
            dboxes = torch.zeros((N, 4), device=dout.device, dtype=torch.float32)
            check(lib.csg_crop_bwd_boxes(ptr(dout), ptr(img), B, H, W, C, C, ptr(boxes), ptr(img_idx), N, HH, HH, Cp,
                                         ptr(dboxes), stream()), "crop_bwd_boxes")
        return dimg, dboxes, None, None


def crop_objects(img, boxes, img_idx, HH):
    """img (B,C,H,W); boxes (N,4) xywh of the real objects in (image, object) order; img_idx (N,) int64."""
    return _CropObjects.apply(img, boxes, img_idx.contiguous(), int(HH))
