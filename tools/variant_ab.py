"""F(4x4,3x3) against F(2x2,3x3) where the larger tile leaves the chip under-filled: small batches / small maps, launches
WITH an epilogue (bias: not splittable over the input channels).  Raw C-ABI calls, same process, alternating.
    python tools/variant_ab.py [B ...]      (default 4 6 16)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from canonicalsg2im_amd import ops  # noqa: E402
from canonicalsg2im_amd._lib import WinoDesc, check, lib, ptr, stream  # noqa: E402

SHAPES = [(512, 256, 64), (256, 256, 64), (128, 512, 64), (128, 256, 64), (1024, 512, 32), (512, 512, 32), (128, 1024, 32),
          (128, 512, 32), (256, 128, 128), (128, 128, 128), (128, 256, 128), (128, 64, 256), (64, 64, 256), (32, 128, 64), (32, 128, 32),
          (2048, 128, 32), (1024, 128, 32), (1024, 128, 64), (512, 128, 64)]   # backward-data of the gamma || beta convolutions


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("%-26s %8s | %10s %10s | %6s" % ("shape", "F4 items", "F(4x4) ms", "F(2x2) ms", "F2/F4"))
for B in ([int(a) for a in sys.argv[1:]] or [4, 6, 16]):
    for (Cin, Cout, H) in SHAPES:
        x = ops.nhwc(torch.randn(B, Cin, H, H, device="cuda").clamp_min(0))
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)
        bias = torch.randn(Cout, device="cuda")
        y = ops.empty_nhwc(B, Cout, H, H, x.device)
        d = WinoDesc()
        d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, H, Cin, Cin, Cout, Cout, 0, 0.0
        ms = {}
        for var, fn in ((4, lib.csg_wino4_conv), (2, lib.csg_wino_conv)):
            up = ops.wino_pack(w, False, None, var)
            call = lambda: check(fn(d, ptr(x), ptr(up), ptr(bias), None, None, 0.0, ptr(y), None, 0, stream()), "conv")
            ms[var] = min(bench(call), bench(call))
        items = B * ((H + 15) // 16) * ((H + 31) // 32) * ((Cout + 63) // 64)
        print("B%-2d %4d->%-4d %3dx%-3d        %8d | %10.3f %10.3f | %6.2f" % (B, Cin, Cout, H, H, items, ms[4], ms[2], ms[2] / ms[4]),
              flush=True)
