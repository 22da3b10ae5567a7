#!/usr/bin/env python3
"""Stand-alone launches of the Winograd F(4x4,3x3) / F(2x2,3x3) kernels on one shape (raw C ABI, weights packed once),
for rocprofv3 --pmc passes and quick timing:  python tools/wino4_probe.py B Cin Cout H W [variant] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalsg2im_amd import ops  # noqa: E402
from canonicalsg2im_amd._lib import WinoDesc, check, lib, ptr, stream  # noqa: E402

B, Cin, Cout, H, W = [int(a) for a in sys.argv[1:6]]
variant = int(sys.argv[6]) if len(sys.argv) > 6 else 4
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
KS = 4 if variant in (34, 44) else 3               # 34: F(3x3,4x4) on a 4x4 / pad 2 layer; 44: the direct kernel on it
x = ops.nhwc(torch.randn(B, Cin, H, W, device="cuda").clamp_min(0))
w = torch.randn(Cout, Cin, KS, KS, device="cuda") / (KS * Cin ** 0.5)
OH, OW = (H + 1, W + 1) if KS == 4 else (H, W)
y = ops.empty_nhwc(B, Cout, OH, OW, x.device)
d = WinoDesc()
d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, W, Cin, Cin, Cout, Cout, 0, 0.0
if variant == 44:
    ops.WINO_ENABLED = False

    def run():
        global y
        y = ops.conv2d(x, w, None, 1, 2)
elif variant == 34:
    up = ops.wino_pack(w, False, None, 34)

    def run():
        check(lib.csg_wino34_conv(d, 2, ptr(x), ptr(up), None, None, None, 0.0, ptr(y), None, 0, stream()), "conv")
else:
    up = ops.wino_pack(w, False, None, variant)
    fn = lib.csg_wino4_conv if variant == 4 else lib.csg_wino_conv

    def run():
        check(fn(d, ptr(x), ptr(up), None, None, None, 0.0, ptr(y), None, 0, stream()), "conv")


for _ in range(3):
    run()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for e0, e1 in ev:
    e0.record()
    run()
    e1.record()
torch.cuda.synchronize()
ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
ms = ts[len(ts) // 2]
flop = 2.0 * B * OH * OW * KS * KS * Cin * Cout
print("checksum %.6e  " % float(y.double().abs().sum()), end="")
name = {2: "F(2x2,3x3)", 4: "F(4x4,3x3)", 34: "F(3x3,4x4)", 44: "direct 4x4"}[variant]
exe = {2: 4.0 / 9.0, 4: 0.25, 34: 0.25, 44: 1.0}[variant]
print("%s B %d Cin %d Cout %d %dx%d: %.3f ms, %.1f TFLOP/s algorithmic, %.1f executed" %
      (name, B, Cin, Cout, H, W, ms, flop / ms / 1e9, flop / ms / 1e9 * exe))
