#!/usr/bin/env python3
"""Per-shape timing of the implicit-GEMM conv kernels on the shapes of the 256x256 AttSPADE step
(B=16): forward, backward-data and weight-gradient TFLOP/s per layer shape.  Development tool:
    python tools/conv_shapes.py [--batch 16] [--size 256]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalsg2im_amd import ops  # noqa: E402


def shapes(H, S=32, ngf=64, ndf=64):
    out = []
    nf = ngf
    res = H // 32
    out.append(("fc", S, 16 * nf, res, 3, 1, 1, 1))
    plan = [("head_0", 16, 16, res), ("G_middle_0", 16, 16, res * 2), ("G_middle_1", 16, 16, res * 2),
            ("up_0", 16, 8, res * 4), ("up_1", 8, 4, res * 8), ("up_2", 4, 2, res * 16), ("up_3", 2, 1, res * 32)]
    for name, fi, fo, r in plan:
        fin, fout = fi * nf, fo * nf
        fmid = min(fin, fout)
        norms = [fin, fmid] + ([fin] if fin != fout else [])
        out.append((name + ".mlp_shared", S, 128, r, 3, 1, 1, len(norms)))
        for c in sorted(set(norms)):
            out.append((name + ".gamma_beta[%d]" % c, 128, 2 * c, r, 3, 1, 1, norms.count(c)))
        out.append((name + ".conv_0", fin, fmid, r, 3, 1, 1, 1))
        out.append((name + ".conv_1", fmid, fout, r, 3, 1, 1, 1))
        if fin != fout:
            out.append((name + ".conv_s", fin, fout, r, 1, 1, 0, 1))
    out.append(("conv_img", nf, 3, H, 3, 1, 1, 1))
    for sc, h in (("D0", H), ("D1", (H - 1) // 2 + 1)):
        c_in = (S + 3 + 3) // 4 * 4
        out.append((sc + ".model0", c_in, ndf, h, 4, 2, 2, 4))
        h1 = h // 2 + 1
        out.append((sc + ".model1", ndf, 2 * ndf, h1, 4, 2, 2, 4))
        h2 = h1 // 2 + 1
        out.append((sc + ".model2", 2 * ndf, 4 * ndf, h2, 4, 2, 2, 4))
        h3 = h2 // 2 + 1
        out.append((sc + ".model3", 4 * ndf, 8 * ndf, h3, 4, 1, 2, 4))
        out.append((sc + ".model4", 8 * ndf, 1, h3 + 1, 4, 1, 2, 4))
    return out


def timeit(fn, iters=12, warmup=3):
    """Median of `iters` individually timed calls after `warmup` untimed ones.  All calls are enqueued back to back
    with their own event pair and resolved after ONE synchronisation, so the host runs ahead of the GPU (no launch
    latency inside a pair) and the caching allocator is in steady state (every call re-uses the blocks the warm-ups
    left).  The median drops the occasional outlier (an allocator miss, a clock ramp) that a 3-call mean let through."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    B = a.batch
    tot = {"fwd": [0.0, 0.0], "bwd_data": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    print("%-26s %5s %5s %4s k s |  M      K     N   | %8s %8s %8s   (TFLOP/s; ms)" %
          ("layer", "Cin", "Cout", "HxW", "fwd", "bwd_data", "wgrad"))
    for name, cin, cout, h, k, s, p, mult in shapes(a.size):
        if a.filter and a.filter not in name:
            continue
        x = ops.nhwc(torch.randn(B, cin, h, h, device="cuda")).requires_grad_(True)
        w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).requires_grad_(True)
        y = ops.conv2d(x, w, None, s, p)
        gy = torch.randn_like(y)
        flops = 2.0 * y.numel() / cout * cout * cin * k * k
        t_f = timeit(lambda: ops.conv2d(x.detach(), w.detach(), None, s, p))
        xd = x.detach().requires_grad_(True)
        yd = ops.conv2d(xd, w.detach(), None, s, p)
        t_d = timeit(lambda: torch.autograd.grad(yd, xd, gy, retain_graph=True))
        wd = w.detach().requires_grad_(True)
        yw = ops.conv2d(x.detach(), wd, None, s, p)
        t_w = timeit(lambda: torch.autograd.grad(yw, wd, gy, retain_graph=True))
        M = y.numel() // cout
        print("%-26s %5d %5d %4d %d %d | %7d %5d %5d | %5.1f %5.2f  %5.1f %5.2f  %5.1f %5.2f  x%d" %
              (name, cin, cout, h, k, s, M, cin * k * k, cout, flops / t_f / 1e9, t_f, flops / t_d / 1e9, t_d,
               flops / t_w / 1e9, t_w, mult))
        for key, t in (("fwd", t_f), ("bwd_data", t_d), ("wgrad", t_w)):
            tot[key][0] += flops * mult
            tot[key][1] += t * mult
        del x, w, y, gy, xd, yd, wd, yw
    for key, (f, t) in tot.items():
        print("TOTAL %-9s %.1f GFLOP  %.2f ms  %.1f TFLOP/s" % (key, f / 1e9, t, f / t / 1e9 if t else 0))


if __name__ == "__main__":
    main()
