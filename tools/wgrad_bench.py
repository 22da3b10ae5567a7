"""Weight gradient of the generator's 3x3 layers: Winograd F(3x3,4x4) (csrc/wino4w.hip) against F(3x3,2x2) (csrc/wino.hip),
raw C-ABI calls at the bench's shapes (B = 16): ms, algorithmic TFLOP/s (2*M*9*Cin*Cout / time), executed fraction of the
fp32 MFMA peak (36/144 resp. 64/144 of the algorithmic FLOPs), error of both against each other.

    python tools/wgrad_bench.py [shape indices...] [--blocks N]      (CSG_WINO4_WGRAD_BLOCKS=N: target blocks per launch)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from canonicalsg2im_amd import ops  # noqa: E402
from canonicalsg2im_amd._lib import WinoDesc, check, lib, ptr, stream  # noqa: E402

PEAK = 157.3
# (B, Cin, Cout, H): gamma||beta, conv_0, conv_1 of up_0 .. up_3 (profiles/*_conv_shapes.txt)
SHAPES = [(16, 128, 256, 256), (16, 128, 128, 256), (16, 128, 64, 256), (16, 64, 64, 256),
          (16, 128, 512, 128), (16, 128, 256, 128), (16, 256, 128, 128), (16, 128, 128, 128),
          (16, 128, 1024, 64), (16, 128, 512, 64), (16, 512, 256, 64), (16, 256, 256, 64),
          (16, 128, 2048, 32), (16, 128, 1024, 32), (16, 1024, 512, 32), (16, 512, 512, 32),
          (4, 128, 256, 256), (4, 128, 512, 128), (4, 512, 256, 64),
          (16, 1024, 1024, 16), (16, 128, 2048, 16), (16, 1024, 1024, 8), (4, 1024, 1024, 16), (4, 128, 2048, 32)]
ONLY = [int(a) for a in sys.argv[1:] if a.isdigit()]
if ONLY:
    SHAPES = [SHAPES[i] for i in ONLY]


def bench(fn, n=8):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("%-28s %10s %8s %6s | %10s %8s %6s | %6s %9s" % ("shape", "F(3,4) ms", "TF alg", "exec", "F(3,2) ms", "TF alg", "exec",
                                                        "x", "max diff"))
tot4 = tot2 = 0.0
for (B, Cin, Cout, H) in SHAPES:
    x = ops.nhwc(torch.randn(B, Cin, H, H, device="cuda"))
    gy = ops.nhwc(torch.randn(B, Cout, H, H, device="cuda"))
    d = WinoDesc()
    d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, H, Cin, Cin, Cout, Cout, 0, 0.0
    flop = 2.0 * B * H * H * 9 * Cin * Cout
    res = {}
    for name, wsf, fn in (("w4", lib.csg_wino4_bwd_weight_workspace, lib.csg_wino4_bwd_weight),
                          ("w2", lib.csg_wino_bwd_weight_workspace, lib.csg_wino_bwd_weight)):
        nbytes = wsf(d)
        ws = torch.empty(nbytes // 4, device="cuda")
        dw = torch.empty(Cout, 3, 3, Cin, device="cuda")
        db = torch.empty(Cout, device="cuda")
        ms = bench(lambda: check(fn(d, ptr(x), ptr(gy), ptr(dw), ptr(db), ptr(ws), nbytes, stream()), name))
        res[name] = (ms, dw, db)
    ms4, ms2 = res["w4"][0], res["w2"][0]
    tot4 += ms4
    tot2 += ms2
    diff = float((res["w4"][1] - res["w2"][1]).abs().max() / res["w2"][1].abs().max())
    print("B%-2d %4d->%-4d %3dx%-3d          %10.3f %8.1f %6.3f | %10.3f %8.1f %6.3f | %6.2f %9.2e" % (
        B, Cin, Cout, H, H, ms4, flop / ms4 / 1e9, flop / ms4 / 1e9 * 0.25 / PEAK, ms2, flop / ms2 / 1e9,
        flop / ms2 / 1e9 * 4.0 / 9.0 / PEAK, ms2 / ms4, diff), flush=True)
print("total %.3f ms vs %.3f ms" % (tot4, tot2))
