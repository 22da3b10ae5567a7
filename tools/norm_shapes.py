#!/usr/bin/env python3
"""Per-shape timing of the normalisation / reduction kernels on the 256x256 step's shapes (B=16).
Development tool: python tools/norm_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalsg2im_amd import _lib, ops  # noqa: E402


def main():
    B = 16
    shapes = [("up_3 C=128", 128, 256), ("up_3 C=64", 64, 256), ("up_2 C=256", 256, 128), ("up_2 C=128", 128, 128),
              ("up_1 C=512", 512, 64), ("up_0 C=1024", 1024, 32), ("G_mid C=1024", 1024, 16)]
    print("%-14s %9s | %s" % ("shape", "MB(x)", "kernel: ms (GB/s algorithmic)"))
    for name, C, h in shapes:
        x = ops.nhwc(torch.randn(B, C, h, h, device="cuda")).requires_grad_(True)
        gb = ops.nhwc(torch.randn(B, 2 * C, h, h, device="cuda")).requires_grad_(True)
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        y = ops.norm_act(x, gb, rm, rv, instance=False, training=True, slope=0.2)
        gy = torch.randn_like(y)
        _lib.prof_reset()
        _lib.prof_enable(True)
        for _ in range(3):
            y = ops.norm_act(x, gb, rm, rv, instance=False, training=True, slope=0.2)
            torch.autograd.grad(y, [x, gb], gy)
            g2 = torch.randn(B, 2 * C, h, h, device="cuda").permute(0, 2, 3, 1).contiguous()
        prof = _lib.prof_read()
        _lib.prof_enable(False)
        mb = x.numel() * 4 / 1e6
        line = []
        for k in ("norm_stats", "norm_apply_fwd", "norm_bwd_reduce", "norm_bwd_dx"):
            ms, n, work = prof[k]
            line.append("%s %.3f (%.0f)" % (k[5:], ms / n, work / n / (ms / n) / 1e6))
        print("%-14s %9.1f | %s" % (name, mb, "  ".join(line)))
        del x, gb, y, gy


if __name__ == "__main__":
    main()
