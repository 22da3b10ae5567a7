"""k_wino4_conv_v: one block per item against the persistent form (one block per CU walking its items, the stage pipeline
carried across them), raw C-ABI calls at the generator's shapes (B = 16), same process, alternating.

    python tools/wino4_ab.py            ms per launch, executed fraction of the fp32 MFMA peak (1/4 of 2*M*9*Cin*Cout), ratio
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
SHAPES = [(16, 128, 256, 256), (16, 128, 128, 256), (16, 128, 64, 256), (16, 64, 64, 256), (16, 32, 128, 256),
          (16, 128, 512, 128), (16, 128, 256, 128), (16, 256, 128, 128), (16, 128, 128, 128),
          (16, 128, 1024, 64), (16, 128, 512, 64), (16, 512, 256, 64), (16, 256, 256, 64),
          (16, 128, 2048, 32), (16, 128, 1024, 32), (16, 1024, 512, 32), (16, 512, 512, 32),
          (4, 128, 256, 256), (4, 128, 512, 128), (6, 128, 256, 256)]


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


print("%-26s %10s %6s | %10s %6s | %6s" % ("shape", "1/item ms", "exec", "persist ms", "exec", "x"))
t0 = t1 = 0.0
for (B, Cin, Cout, H) in SHAPES:
    x = ops.nhwc(torch.randn(B, Cin, H, H, device="cuda").clamp_min(0))
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)
    up = ops.wino_pack(w, False, None, 4)
    y = ops.empty_nhwc(B, Cout, H, H, x.device)
    d = WinoDesc()
    d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, H, Cin, Cin, Cout, Cout, 0, 0.0
    nws = lib.csg_wino4_conv_workspace(d)
    ws = torch.empty(max(nws, 4) // 4, device="cuda")
    call = lambda: check(lib.csg_wino4_conv(d, ptr(x), ptr(up), None, None, None, 0.0, ptr(y), ptr(ws), nws, stream()), "conv")
    ms = []
    for rep in range(2):
        for on in (0, 1):
            lib.csg_wino4_persistent(on)
            ms.append(bench(call))
    a, b = min(ms[0], ms[2]), min(ms[1], ms[3])
    t0 += a
    t1 += b
    ex = 2.0 * B * H * H * 9 * Cin * Cout / 4 / 1e9
    print("B%-2d %4d->%-4d %3dx%-3d        %10.3f %6.3f | %10.3f %6.3f | %6.3f" % (B, Cin, Cout, H, H, a, ex / a / PEAK, b,
                                                                                 ex / b / PEAK, a / b), flush=True)
print("total %.3f ms vs %.3f ms" % (t0, t1))
lib.csg_wino4_persistent(1)
