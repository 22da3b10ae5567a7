import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from canonicalsg2im_amd import ops

def run(shape, wgrad, act):
    B, Cin, Cout, H, W = shape
    ops.WINO_WGRAD = wgrad
    g = torch.Generator().manual_seed(B * 7 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    ref = F.conv2d(xr, wr, br, padding=1)
    if act: ref = F.leaky_relu(ref, 0.2)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    xd, wd, bd = [t.cuda().requires_grad_(True) for t in (x, w, b)]
    y = ops.conv2d(xd, wd, bd, 1, 1, ops.ACT_LEAKY if act else ops.ACT_NONE, 0.2)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    ex = (xd.grad.cpu() - xr.grad).abs()
    bad = (ex > 1e-3).nonzero()
    print(shape, "wino_wgrad", wgrad, "act", act, "y", float((y.cpu() - ref).abs().max()), "dx", float(ex.max()), "nbad", bad.shape[0],
          bad[:5].tolist(), "dw", float((wd.grad.cpu() - wr.grad).abs().max()), "db", float((bd.grad.cpu() - br.grad).abs().max()), flush=True)

for shape in [(1, 64, 128, 128, 128), (4, 32, 64, 64, 64)]:
    for wgrad in (True, False):
        for act in (True, False):
            run(shape, wgrad, act)
