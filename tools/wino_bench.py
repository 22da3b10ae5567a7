"""Winograd vs direct implicit GEMM on the generator's 3x3 shapes (B = 16): ms and algorithmic TFLOP/s
(2*M*9*Cin*Cout / time, the direct convolution's FLOP count for both)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from canonicalsg2im_amd import ops  # noqa: E402

SHAPES = [(16, 128, 256, 256), (16, 128, 128, 256), (16, 64, 64, 256), (16, 128, 64, 256), (16, 32, 128, 256),
          (16, 128, 512, 128), (16, 256, 128, 128), (16, 128, 1024, 64), (16, 512, 256, 64), (16, 256, 256, 64),
          (16, 128, 2048, 32), (16, 1024, 512, 32), (16, 512, 512, 32), (16, 1024, 1024, 16), (16, 128, 2048, 16),
          (16, 256, 128, 256)]


ONLY = [int(a) for a in sys.argv[1:] if a.isdigit()]
if ONLY:
    SHAPES = [SHAPES[i] for i in ONLY]
FWD_ONLY = "--fwd" in sys.argv


def bench(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (B, Cin, Cout, H) in SHAPES:
    x = ops.nhwc(torch.randn(B, Cin, H, H, device="cuda"))
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)
    b = torch.randn(Cout, device="cuda")
    flop = 2.0 * B * H * H * 9 * Cin * Cout
    out = {}
    for mode in ("wino", "direct"):
        ops.WINO_ENABLED = mode == "wino"
        with torch.no_grad():
            ms = bench(lambda: ops.conv2d(x, w, b, 1, 1))
        out[mode] = ms
    if FWD_ONLY:
        print("Cin %4d Cout %4d %3dx%-3d  fwd: wino %7.3f ms %6.1f TF | direct %7.3f ms %6.1f TF" %
              (Cin, Cout, H, H, out["wino"], flop / out["wino"] / 1e9, out["direct"], flop / out["direct"] / 1e9), flush=True)
        continue
    # weight gradient: time backward with only the weight requiring grad
    wg = {}
    for mode in ("wino", "direct"):
        ops.WINO_ENABLED = True
        ops.WINO_WGRAD = mode == "wino"
        wr = w.clone().requires_grad_(True)
        y = ops.conv2d(x, wr, None, 1, 1)
        gy = torch.randn_like(y)

        def run():
            wr.grad = None
            y.backward(gy, retain_graph=True)
        wg[mode] = bench(run, 5)
    ops.WINO_WGRAD = True
    print("Cin %4d Cout %4d %3dx%-3d  fwd: wino %7.3f ms %6.1f TF | direct %7.3f ms %6.1f TF | x%.2f   wgrad: wino %7.3f ms "
          "%6.1f TF | direct %7.3f ms %6.1f TF | x%.2f" %
          (Cin, Cout, H, H, out["wino"], flop / out["wino"] / 1e9, out["direct"], flop / out["direct"] / 1e9,
           out["direct"] / out["wino"], wg["wino"], flop / wg["wino"] / 1e9, wg["direct"], flop / wg["direct"] / 1e9,
           wg["direct"] / wg["wino"]), flush=True)
