import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from canonicalsg2im_amd import ops
from canonicalsg2im_amd._lib import WinoDesc, check, lib, ptr, stream

def raw(x, up, Cout):
    B, Cin, H, W = x.shape
    xd = ops.nhwc(x.cuda())
    y = ops.empty_nhwc(B, Cout, H, W, xd.device)
    d = WinoDesc()
    d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, W, Cin, Cin, Cout, Cout, 0, 0.0
    check(lib.csg_wino_conv(d, ptr(xd), ptr(up), None, None, None, 0.0, ptr(y), None, 0, stream()), "wino")
    torch.cuda.synchronize()
    return y.cpu()

for (B, Cin, Cout, H, W) in [(1, 128, 64, 128, 128), (1, 64, 128, 128, 128), (2, 128, 64, 128, 128), (1, 128, 64, 64, 64), (1, 128, 32, 128, 128)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    ref = F.conv2d(x, w, padding=1)
    up = ops.wino_pack(w.cuda(), False)
    for rep in range(3):
        y = raw(x, up, Cout)
        bad = ((y - ref).abs() > 1e-3).nonzero()
        print((B, Cin, Cout, H, W), "rep", rep, "bad", bad.shape[0], "of", ref.numel(), bad[:6].tolist(), flush=True)
    # dgrad-style operand of the transposed problem
    wt = torch.randn(Cin, Cout, 3, 3, generator=g) / (3 * Cin ** 0.5)       # (Cout'=Cin, Cin'=Cout)
    xr = torch.randn(B, Cout, H, W, generator=g).requires_grad_(True)
    gy = torch.randn(B, Cin, H, W, generator=g)
    F.conv2d(xr, wt, padding=1).backward(gy)
    ut = ops.wino_pack(wt.cuda(), True)
    dx = raw(gy, ut, Cout)
    bad = ((dx - xr.grad).abs() > 1e-3).nonzero()
    print("   dgrad operand: bad", bad.shape[0], bad[:6].tolist(), flush=True)
