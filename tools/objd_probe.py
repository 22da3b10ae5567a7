#!/usr/bin/env python3
"""Forward / backward-data / weight-gradient time of the object discriminator's three 4x4 / stride 2 convolutions
(discriminator.py:253-260 crops -> build_cnn 'C4-64-2,C4-128-2,C4-256-2') at N crops of 64 x 64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalsg2im_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 270


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for (cin, cout, h) in ((4, 64, 64), (64, 128, 32), (128, 256, 16)):
    x = ops.nhwc(torch.randn(N, cin, h, h, device="cuda")).requires_grad_(True)
    w = (torch.randn(cout, cin, 4, 4, device="cuda") / (4 * cin ** 0.5)).requires_grad_(True)
    b = torch.zeros(cout, device="cuda", requires_grad=True)
    y = ops.conv2d(x, w, b, 2, 1)
    gy = torch.randn_like(y)
    flop = 2.0 * y.numel() * cin * 16
    tf = timeit(lambda: ops.conv2d(x, w, b, 2, 1))

    def fb():
        x.grad = w.grad = b.grad = None
        ops.conv2d(x, w, b, 2, 1).backward(gy)
    tfb = timeit(fb)
    print("%4d -> %4d at %2d x %2d, N = %d: fwd %.3f ms (%.1f TFLOP/s), fwd + bwd-data + wgrad %.3f ms (%.1f TFLOP/s)" %
          (cin, cout, h, h, N, tf, flop / tf / 1e9, tfb, 3 * flop / tfb / 1e9))
