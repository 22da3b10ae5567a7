"""Ablations of k_wino4_wgrad (csrc/wino4w.hip) on one shape: what the loop costs without its MFMAs, without its transforms,
without its DMAs (CSG_WW_DBG bit mask, read per call).  Results of the ablated runs are garbage by construction."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from canonicalsg2im_amd import ops  # noqa: E402
from canonicalsg2im_amd._lib import WinoDesc, check, lib, ptr, stream  # noqa: E402

B, Cin, Cout, H = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (8, 128, 256, 256)
x = ops.nhwc(torch.randn(B, Cin, H, H, device="cuda"))
gy = ops.nhwc(torch.randn(B, Cout, H, H, device="cuda"))
d = WinoDesc()
d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, H, Cin, Cin, Cout, Cout, 0, 0.0
nbytes = lib.csg_wino4_bwd_weight_workspace(d)
ws = torch.empty(nbytes // 4, device="cuda")
dw = torch.empty(Cout, 3, 3, Cin, device="cuda")
flop = 2.0 * B * H * H * 9 * Cin * Cout


def bench(n=10):
    def fn():
        check(lib.csg_wino4_bwd_weight(d, ptr(x), ptr(gy), ptr(dw), None, ptr(ws), nbytes, stream()), "w4")
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


mfma_ms = flop * 0.25 / 157.3e12 * 1e3
print("B%d %d->%d %dx%d: MFMA bound at 2.4 GHz %.3f ms" % (B, Cin, Cout, H, H, mfma_ms))
for dbg, name in ((0, "everything"), (1, "no MFMAs"), (2, "no transform"), (4, "no DMA"), (3, "no MFMAs, no transform (DMA + barriers)"),
                  (6, "MFMAs + operand reads only"), (5, "transform only"), (7, "barriers only"), (8, "everything, staggered"), (0, "everything (again)"), (8, "everything, staggered (again)")):
    os.environ["CSG_WW_DBG"] = str(dbg)
    print("  %-44s %.3f ms" % (name, bench()), flush=True)
os.environ["CSG_WW_DBG"] = "0"
